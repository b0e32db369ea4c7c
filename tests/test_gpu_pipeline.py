"""GPU parity of the device-pointer layer (torch-owned HBM) used by bench.py and the multi-GPU path."""
import pytest

from oracle import py_oracle as O
from tests import golden_util as GU
from tests import gpu_util as DU

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('path', ['0', '1', None])
def test_dev_layer_matches_oracle(path, monkeypatch):
    import torch
    if path is None:
        monkeypatch.delenv('BESST_RECORD_PATH', raising=False)
    else:
        monkeypatch.setenv('BESST_RECORD_PATH', path)
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
    if path is None:
        assert gb.params.record_path == (1 if gb.candidate_share >= 0.05 else 0)


def test_record_path_follows_candidate_density(monkeypatch):
    """Sampled share of tid != mtid records: a paired-end library on long contigs stays on the two-pass form, a
    mate-pair library goes to the fused pass."""
    import torch
    from besst_amd import pipeline, workload
    monkeypatch.delenv('BESST_RECORD_PATH', raising=False)
    dev = torch.device('cuda', 0)
    for config, want in (('C2', 0), ('C3', 1)):
        wl = workload.make(config, 0, pairs=200000, nc=400 if config == 'C2' else 300)
        rec = pipeline.DeviceRecords(wl['batch'], dev)
        gb = pipeline.DeviceGraphBuilder(dev, wl['asm'].nc, wl['node_bits'], wl['lib'], rec.n, rec.n)
        assert gb.record_path(rec) == want, (config, gb.candidate_share)
        share = float((wl['batch'].tid != wl['batch'].mtid).mean())
        assert abs(gb.candidate_share - share) < 0.02


@pytest.mark.parametrize('n,key_bits,hub', [(50_000, 31, 20_000), (3_000, 9, 0), (200_000, 41, 700), (1, 31, 0),
                                            (600_000, 33, 100_000), (700_000, 41, 0), (1_000_000, 41, 3_000),
                                            (1_000_000, 45, 0), (1_500_000, 41, 0), (200_000, 57, 500),
                                            (700_000, 55, 0), (1_000_000, 59, 0), (5_000_000, 41, 0),
                                            (6_000_000, 37, 150_000), (5_000_000, 45, 0), (4_300_000, 3, 0),
                                            (8192 * 600 + 1, 37, 0), (4096 * 1100, 5, 0), (4_500_000, 39, 3_000),
                                            (4_400_000, 33, 1_800)])
def test_sort_reduce_on_synthetic_tuples(n, key_bits, hub):
    """The sort/reduce stage alone, on skewed keys: a hub bucket larger than the LDS sort capacity, fewer key bits
    than one digit, the small-stream MSD path (scan-free table, rank sort), the mid-size MSD path (row-scanned table,
    rank sort / LDS bitonic / global bitonic buckets), 57- and 55-bit keys that only pack because the MSD digit is implied by the bucket, a
    stream whose words do not fit 64 bits even so (59-bit keys: LSD passes with index arrays), and streams beyond 4 M
    tuples (chained-scan radix passes + atomic-free row reduction, csrc/onesweep.hip): packed words, a hub row of
    150 000 tuples, 45-bit keys that travel with a separate index array, 3- and 5-bit keys whose few rows run
    across hundreds of reduce tiles, and streams that end one tuple into / exactly at a tile.  Packed keys of four digits
    or more take two chained-scan passes on the top 16 bits and finish bucket by bucket: random keys overflow the
    distinct-key limit of the wave-per-bucket kernel and go through the LDS digit passes, the 3 000- and 1 800-tuple
    hubs of 40 keys through the LDS passes / the wave kernel's largest size class, the 150 000-tuple hub through the
    global-memory passes."""
    import ctypes as C
    import numpy as np
    import torch
    from besst_amd import pipeline
    from oracle import c_oracle as CO
    rng = np.random.default_rng(n + key_bits)
    node_bits = (key_bits - 1) // 2
    pair = rng.integers(0, 1 << (2 * node_bits), n, dtype=np.int64)
    if hub:
        pair[rng.choice(n, hub, replace=False)] = (5 << node_bits) | rng.integers(0, 40, hub)   # one crowded top-bits bucket
    fishy = (rng.random(n) < 0.01).astype(np.int64)
    keys = ((pair << 1) | fishy).astype(np.uint64)
    lo = rng.integers(26, 5000, n).astype(np.uint64)
    hi = rng.integers(26, 5000, n).astype(np.uint64) | (np.uint64(3) << np.uint64(30))
    lo[fishy == 1] = 0
    hi[fishy == 1] = 0
    payload = lo | (hi << np.uint64(32))
    dev = torch.device('cuda', 0)
    lib = dict(read_len=100.0, ins_size_threshold=800.0, min_mapq=11, orientation='fr', detect_duplicate=True,
               extend_paths=True, no_score=False)
    gb = pipeline.DeviceGraphBuilder(dev, 4, node_bits, lib, n, n)
    dk = torch.from_numpy(keys.view(np.int64)).to(dev)
    dp = torch.from_numpy(payload.view(np.int64)).to(dev)
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    for _ in range(2):
        gb.reduce(keys=dk, payload=dp, n_tuples_ptr=C.c_void_p(cnt.data_ptr()), capacity=n)
    torch.cuda.synchronize()
    want = CO.edge_rows(keys, payload)
    r = len(want['key'])
    assert gb.read_sizes()[1] == r        # (random keys beyond 4 M tuples: the run-grouped form overflows, the call is repeated)
    get = lambda t, m, dt: t[:m].cpu().numpy().view(dt)
    assert np.array_equal(get(gb.row_key, r, np.uint64), want['key'])
    assert np.array_equal(get(gb.row_n, r, np.uint32).astype(np.int64), want['n'])
    assert np.array_equal(get(gb.row_sum, r, np.int64), want['sum_obs'])
    assert np.array_equal(get(gb.row_sum_sq, r, np.int64), want['sum_obs_sq'])
    assert np.array_equal(get(gb.row_first, r, np.uint32).astype(np.int64), want['first_idx'])
    assert np.array_equal(get(gb.obs_lo, n, np.int32).astype(np.int64), want['obs_lo'])
    assert np.array_equal(get(gb.obs_hi, n, np.int32).astype(np.int64), want['obs_hi'])


def test_pass_pool_overlapped_passes_are_independent():
    """Three different record sets in flight on three streams, twice over: every slot ends with exactly the edge
    table of the record set it processed last."""
    import numpy as np
    import torch
    from besst_amd import pipeline, workload
    from oracle import c_oracle as CO
    dev = torch.device('cuda', 0)
    wls = [workload.make('C2', 0, pairs=300000, nc=700, seed_offset=k) for k in range(3)]
    base = wls[0]
    recs = [pipeline.DeviceRecords(w['batch'], dev) for w in wls]
    cap = max(r.n for r in recs)
    pool = pipeline.PassPool(dev, base['asm'].nc, base['node_bits'], base['lib'], cap, cap, in_flight=3)
    pool.set_contigs(**base['table'])
    last = {}
    for i in range(6):
        k = (i * 2 + 1) % 3                      # slots see different record sets in the two rounds
        gb = pool.submit(recs[k])
        last[id(gb)] = (gb, k)
    pool.synchronize()
    assert len(last) == 3
    for gb, k in last.values():
        table = gb.fetch_table()
        keys, payload, aligned, ctr = CO.record_loop(wls[k]['batch'], base['table'], base['lib'], base['node_bits'])
        rows = CO.edge_rows(keys, payload)
        assert np.array_equal(table.key, rows['key']) and np.array_equal(table.n.astype(np.int64), rows['n'])
        assert np.array_equal(table.obs_lo.astype(np.int64), rows['obs_lo'])
        assert gb.aligned.cpu().numpy().tolist() == aligned.tolist()


@pytest.mark.parametrize('n', [300_000, 2_000_000, 6_000_000, 24_000_000])
def test_sort_reduce_with_offset_scaffold_ids(n):
    """Scaffold ids of a later library start far above 1 (param.scaffold_indexer keeps growing, MakeScaffolds.py:276):
    all keys share a long prefix.  With key_base / key_bits describing the occupied range the MSD buckets stay balanced
    (without it 2 M such tuples fell into ~160 of 2048 buckets and took the global-memory fallback); results must be
    the same as ever on every sort path.  (The two largest sizes are the wave-per-bucket kernel's own case: a dozen
    links per edge, a few to a few dozen distinct keys per top-16-bit bucket.)"""
    import ctypes as C
    import numpy as np
    import torch
    from besst_amd import pipeline
    from oracle import c_oracle as CO
    rng = np.random.default_rng(n)
    node_bits, lo_id, hi_id = 21, 600_001, 760_000
    rows = max(1000, n // 12)
    a = rng.integers(lo_id * 2, hi_id * 2 + 2, rows, dtype=np.int64)
    b = rng.integers(lo_id * 2, hi_id * 2 + 2, rows, dtype=np.int64)
    pair = ((np.minimum(a, b) << node_bits) | np.maximum(a, b))[rng.integers(0, rows, n)]
    fishy = (rng.random(n) < 0.01).astype(np.int64)
    keys = ((pair << 1) | fishy).astype(np.uint64)
    lo = rng.integers(26, 5000, n).astype(np.uint64)
    hi = rng.integers(26, 5000, n).astype(np.uint64) | (np.uint64(3) << np.uint64(30))
    lo[fishy == 1] = 0
    hi[fishy == 1] = 0
    payload = lo | (hi << np.uint64(32))
    dev = torch.device('cuda', 0)
    lib = dict(read_len=100.0, ins_size_threshold=800.0, min_mapq=11, orientation='fr', detect_duplicate=True,
               extend_paths=True, no_score=False)
    gb = pipeline.DeviceGraphBuilder(dev, 4, node_bits, lib, n, n)
    top = hi_id * 2 + 1
    gb.key_base = ((lo_id * 2) << node_bits) << 1
    gb.key_bits = int(((((top << node_bits) | top) << 1) | 1) - gb.key_base).bit_length()
    assert gb.key_bits < 2 * node_bits + 1
    dk = torch.from_numpy(keys.view(np.int64)).to(dev)
    dp = torch.from_numpy(payload.view(np.int64)).to(dev)
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    gb.reduce(keys=dk, payload=dp, n_tuples_ptr=C.c_void_p(cnt.data_ptr()), capacity=n)
    torch.cuda.synchronize()
    want = CO.edge_rows(keys, payload)
    r = len(want['key'])
    assert gb.read_sizes()[1] == r
    get = lambda t, m, dt: t[:m].cpu().numpy().view(dt)
    assert np.array_equal(get(gb.row_key, r, np.uint64), want['key'])
    assert np.array_equal(get(gb.row_n, r, np.uint32).astype(np.int64), want['n'])
    assert np.array_equal(get(gb.row_sum, r, np.int64), want['sum_obs'])
    assert np.array_equal(get(gb.row_first, r, np.uint32).astype(np.int64), want['first_idx'])
    assert np.array_equal(get(gb.obs_lo, n, np.int32).astype(np.int64), want['obs_lo'])


@pytest.mark.gpu
@pytest.mark.parametrize('n', [0, 1, 100, 70_000])
def test_large_capacity_with_few_tuples(n):
    """The sort path is chosen by the CAPACITY of the tuple buffers (a resident builder is sized for the largest pass):
    the chained-scan passes + wave-per-bucket kernels with a stream of 0, 1 and a few tuples (every bucket but a handful
    empty, n_rows from the last workgroup of the row mover), and 70 000 tuples that still fit one scatter tile each."""
    import ctypes as C
    import numpy as np
    import torch
    from besst_amd import pipeline
    from oracle import c_oracle as CO
    cap, key_bits = 5_000_000, 37
    rng = np.random.default_rng(n + 5)
    node_bits = (key_bits - 1) // 2
    rows = max(1, n // 9)
    pair = rng.integers(0, 1 << (2 * node_bits), rows, dtype=np.int64)[rng.integers(0, rows, n)] if n else np.zeros(0, np.int64)
    keys = (pair << 1).astype(np.uint64)
    lo = rng.integers(26, 5000, n).astype(np.uint64)
    hi = rng.integers(26, 5000, n).astype(np.uint64) | (np.uint64(1) << np.uint64(30))
    payload = lo | (hi << np.uint64(32))
    dev = torch.device('cuda', 0)
    lib = dict(read_len=100.0, ins_size_threshold=800.0, min_mapq=11, orientation='fr', detect_duplicate=True,
               extend_paths=True, no_score=False)
    gb = pipeline.DeviceGraphBuilder(dev, 4, node_bits, lib, cap, cap)
    dk = torch.zeros(cap, dtype=torch.int64, device=dev)
    dp = torch.zeros(cap, dtype=torch.int64, device=dev)
    dk[:n] = torch.from_numpy(keys.view(np.int64)).to(dev)
    dp[:n] = torch.from_numpy(payload.view(np.int64)).to(dev)
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    for _ in range(2):
        gb.reduce(keys=dk, payload=dp, n_tuples_ptr=C.c_void_p(cnt.data_ptr()), capacity=cap)
    torch.cuda.synchronize()
    want = CO.edge_rows(keys, payload)
    r = len(want['key'])
    assert gb.read_sizes()[1] == r
    get = lambda t, m, dt: t[:m].cpu().numpy().view(dt)
    assert np.array_equal(get(gb.row_key, r, np.uint64), want['key'])
    assert np.array_equal(get(gb.row_n, r, np.uint32).astype(np.int64), want['n'])
    assert np.array_equal(get(gb.row_sum, r, np.int64), want['sum_obs'])
    assert np.array_equal(get(gb.row_sum_sq, r, np.int64), want['sum_obs_sq'])
    assert np.array_equal(get(gb.row_first, r, np.uint32).astype(np.int64), want['first_idx'])
    assert np.array_equal(get(gb.row_mask, r, np.uint32).astype(np.int64), np.ones(r, np.int64))
    assert np.array_equal(get(gb.obs_lo, n, np.int32).astype(np.int64), want['obs_lo'])
    assert np.array_equal(get(gb.obs_hi, n, np.int32).astype(np.int64), want['obs_hi'])


@pytest.mark.gpu
@pytest.mark.parametrize('links_per_edge', [1, 40])
@pytest.mark.parametrize('key_bits', [25, 26, 31, 33, 40, 41])
def test_bucket_form_over_the_key_widths(key_bits, links_per_edge):
    """The two-pass + buckets form of the large-stream sort serves packed keys of 25 to 41 significant bits (with 4.3 M
    tuples): 9 to 25 key bits are left to the buckets.  One link per edge sends every bucket through the digit-pass
    kernels (one or more 7-bit passes), forty links per edge through the wave kernel's smallest-key peel."""
    import ctypes as C
    import numpy as np
    import torch
    from besst_amd import pipeline
    from oracle import c_oracle as CO
    n = 4_300_000
    rng = np.random.default_rng(key_bits * 100 + links_per_edge)
    node_bits = (key_bits - 1) // 2
    rows = max(1, n // links_per_edge)
    pair = rng.integers(0, 1 << (2 * node_bits), rows, dtype=np.int64)[rng.integers(0, rows, n)]
    fishy = (rng.random(n) < 0.01).astype(np.int64)
    keys = ((pair << 1) | fishy).astype(np.uint64)
    lo = rng.integers(26, 5000, n).astype(np.uint64)
    hi = rng.integers(26, 5000, n).astype(np.uint64) | (np.uint64(2) << np.uint64(30))
    lo[fishy == 1] = 0
    hi[fishy == 1] = 0
    payload = lo | (hi << np.uint64(32))
    dev = torch.device('cuda', 0)
    lib = dict(read_len=100.0, ins_size_threshold=800.0, min_mapq=11, orientation='fr', detect_duplicate=True,
               extend_paths=True, no_score=False)
    gb = pipeline.DeviceGraphBuilder(dev, 4, node_bits, lib, n, n)
    gb.key_bits = key_bits
    dk = torch.from_numpy(keys.view(np.int64)).to(dev)
    dp = torch.from_numpy(payload.view(np.int64)).to(dev)
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    gb.reduce(keys=dk, payload=dp, n_tuples_ptr=C.c_void_p(cnt.data_ptr()), capacity=n)
    torch.cuda.synchronize()
    want = CO.edge_rows(keys, payload)
    r = len(want['key'])
    assert gb.read_sizes()[1] == r
    get = lambda t, m, dt: t[:m].cpu().numpy().view(dt)
    assert np.array_equal(get(gb.row_key, r, np.uint64), want['key'])
    assert np.array_equal(get(gb.row_n, r, np.uint32).astype(np.int64), want['n'])
    assert np.array_equal(get(gb.row_sum, r, np.int64), want['sum_obs'])
    assert np.array_equal(get(gb.row_sum_sq, r, np.int64), want['sum_obs_sq'])
    assert np.array_equal(get(gb.row_first, r, np.uint32).astype(np.int64), want['first_idx'])
    assert np.array_equal(get(gb.obs_lo, n, np.int32).astype(np.int64), want['obs_lo'])
    assert np.array_equal(get(gb.obs_hi, n, np.int32).astype(np.int64), want['obs_hi'])
