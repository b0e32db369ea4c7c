"""GPU parity of the device-pointer layer (torch-owned HBM) used by bench.py and the multi-GPU path."""
import pytest

from oracle import py_oracle as O
from tests import golden_util as GU
from tests import gpu_util as DU

pytestmark = pytest.mark.gpu


def test_dev_layer_matches_oracle():
    import torch
    from besst_amd import pipeline, workload
    wl = workload.make('C2', 0, pairs=300000, nc=1500)
    batch, table, lib = wl['batch'], wl['table'], wl['lib']
    tab = dict(cls=table['cls'].tolist(), scaf=table['scaf_id'].tolist(), slen=table['scaf_len'].tolist(),
               cpos=table['ctg_pos'].tolist(), clen=table['ctg_len'].tolist(),
               cdir=[bool(x) for x in table['direction'].tolist()])
    p = O.LibParams(read_len=lib['read_len'], ins_size_threshold=lib['ins_size_threshold'])
    loop = O.record_loop(GU.rec_lists(batch), tab, p)
    dev = torch.device('cuda', 0)
    rec = pipeline.DeviceRecords(batch, dev)
    gb = pipeline.DeviceGraphBuilder(dev, wl['asm'].nc, wl['node_bits'], lib, rec.n, rec.n)
    gb.set_contigs(**table)
    for _ in range(2):          # a second step must start from clean state
        gb.step(rec)
    torch.cuda.synchronize()
    edges = gb.fetch_table()
    DU.assert_matches_oracle(edges, gb.aligned.cpu().numpy(), gb.read_counters(), loop, wl['asm'].nc)
