"""GPU: the mate-elsewhere bit column of the resident record layout (besst_dev_mate_bits, besst_lib_params.mate_bits).

The reference's loop body asks `contig1 != contig2` before everything but the coverage sum (BESST/CreateGraph.py:141-169);
the resident layout keeps that predicate as one bit per record, made once when the records arrive, and the record loop
reads `mtid` only for the lanes that hold such a record.  Checked here: the bits themselves on ragged lengths, the
context's incremental bits over pushes of odd sizes, and both forms of both record loops - with the bit column and with
BESST_MATE_BITS=0 (the loop compares the columns itself) - against the oracle."""
import ctypes as C

import numpy as np
import pytest

from besst_amd import synth
from oracle import c_oracle as CO
from oracle import py_oracle as O
from tests import golden_util as GU
from tests import gpu_util as DU
from tests.test_gpu_graph_build import _oracle_inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n', [1, 7, 8, 9, 4095, 16384 + 3, 100_001])
def test_bits_are_tid_ne_mtid_packed(n):
    import torch
    from besst_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(n)
    tid = rng.integers(-1, 50, n).astype(np.int32)
    mtid = np.where(rng.random(n) < 0.7, tid, rng.integers(-1, 50, n)).astype(np.int32)
    dev = torch.device('cuda', 0)
    d_tid, d_mtid = torch.from_numpy(tid).to(dev), torch.from_numpy(mtid).to(dev)
    nbytes = int(lib.besst_dev_mate_bits_bytes(n))
    assert nbytes >= (n + 7) // 8
    bits = torch.full((nbytes,), 0xAA, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.besst_dev_mate_bits(C.c_void_p(stream), n, C.c_void_p(d_tid.data_ptr()), C.c_void_p(d_mtid.data_ptr()),
                                       C.c_void_p(bits.data_ptr())), 'dev_mate_bits')
    got = bits.cpu().numpy()[:(n + 7) // 8]
    want = np.packbits((tid != mtid).astype(np.uint8), bitorder='little')
    assert np.array_equal(got, want)                         # (bits behind the last record: 0)


@pytest.mark.parametrize('bits', ['1', '0'])
@pytest.mark.parametrize('path', ['0', '1'])
@pytest.mark.parametrize('name', ['fr_given', 'rf_contam', 'fr_edgecases', 'rf_second_lib'])
def test_both_loops_with_and_without_the_bit_column(name, path, bits, monkeypatch):
    monkeypatch.setenv('BESST_RECORD_PATH', path)
    monkeypatch.setenv('BESST_MATE_BITS', bits)
    doc, batch = GU.load(name)
    p, rec, tab = _oracle_inputs(doc, batch)
    loop = O.record_loop(rec, tab, p)
    # several pushes of odd sizes: the context extends its bits from where it stopped, inside a byte
    table, aligned, ctr = DU.device_build(batch, tab, p, chunks=7)
    DU.assert_matches_oracle(table, aligned, ctr, loop, len(batch.references))


def test_context_extends_its_bits_over_pushes_and_builds(monkeypatch):
    """push - build - push - build on one context: the second build's bits begin inside the byte the first one ended in."""
    from besst_amd import device
    monkeypatch.delenv('BESST_MATE_BITS', raising=False)
    doc, batch = GU.load('rf_contam')
    p, rec, tab = _oracle_inputs(doc, batch)
    cut = len(batch) // 2 + 3                                # (not a multiple of 8)
    with device.GraphContext(0) as ctx:
        ctx.set_contigs(**DU.table_columns(tab))
        ctx.set_library(p.read_len, p.ins_size_threshold, p.min_mapq, p.orientation, p.detect_duplicate, p.extend_paths, p.no_score)
        ctx.push_records(batch.slice(0, cut))
        first = ctx.build_graph()
        part = {c: getattr(batch, c)[:cut].tolist() for c in GU.COLS}
        DU.assert_matches_oracle(first[0], first[1], first[2], O.record_loop(part, tab, p), len(batch.references))
        ctx.push_records(batch.slice(cut, len(batch)))
        table, aligned, ctr = ctx.build_graph()
    DU.assert_matches_oracle(table, aligned, ctr, O.record_loop(rec, tab, p), len(batch.references))


@pytest.mark.parametrize('config,pairs,nc', [('C2', 1_500_000, 3000), ('C3', 2_000_000, 4000)])
def test_resident_builder_same_table_with_and_without_bits(config, pairs, nc, monkeypatch):
    """DeviceGraphBuilder.step() - what bench.py times - on a sparse (C2: two-pass loop) and a dense (C3: fused loop, runs
    named in the loop) library: the edge table with the bit column equals the table without it and the C oracle's."""
    import torch
    from besst_amd import pipeline, workload
    from tests.test_gpu_fullsize import assert_table_equals_c_oracle
    dev = torch.device('cuda', 0)
    wl = workload.make_device(dev, config, 0, pairs=pairs, nc=nc)
    tables = []
    for bits in ('1', '0'):
        monkeypatch.setenv('BESST_MATE_BITS', bits)
        rec = pipeline.DeviceRecords.from_columns(wl['cols'])
        assert (rec.mate_bits is not None) == (bits == '1')
        gb = pipeline.DeviceGraphBuilder(dev, wl['asm'].nc, wl['node_bits'], wl['lib'], rec.n, rec.n)
        gb.set_contigs(**wl['table'])
        for _ in range(2):
            gb.step(rec)
        table, ctr = gb.fetch_table(), gb.read_counters()
        assert_table_equals_c_oracle(table, gb.aligned.cpu().numpy(), ctr, wl['batch'], wl)
        tables.append((table.key.copy(), table.n.copy(), table.sum_obs.copy(), table.obs_lo.copy(), gb.aligned.cpu().numpy().copy()))
    for a, b in zip(*tables):
        assert np.array_equal(a, b)
