"""The run-grouped form of stage 2 (csrc/runs.hip) on tuple streams beyond 4 M tuples: runs of equal keys inside chunks
of 1024 consecutive tuples are what gets sorted.  Checked against the numpy aggregation of the same stream
(oracle.c_oracle.edge_rows: stable sort by key, exact sums) - CreateGraph.py:842-862."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LIB = dict(read_len=100.0, ins_size_threshold=800.0, min_mapq=11, orientation='fr', detect_duplicate=True,
           extend_paths=True, no_score=False)


def clustered_stream(n, node_bits, noise, seed, hub=0, links=40, spread=24):
    """A tuple stream shaped like a coordinate-sorted library's: the stream walks over the scaffolds in order, every
    stretch of ~`links` * 8 tuples belongs to one scaffold end and is spread over a handful of partner ends (the smaller
    node is the own end for about half of them); a share `noise` of the tuples are chimeric (any partner), and `hub`
    tuples anywhere in the stream hit one edge."""
    rng = np.random.default_rng(seed)
    n_nodes = 1 << node_bits
    stretch = max(1, links * 8)
    own = (np.arange(n, dtype=np.int64) // stretch) % (n_nodes - 64) + 32
    partner = own + rng.integers(-spread, spread + 1, n)
    partner[partner == own] += 1
    chim = rng.random(n) < noise
    partner[chim] = rng.integers(0, n_nodes, int(chim.sum()))
    partner[partner == own] = (own[partner == own] + 7) % n_nodes
    a, b = np.minimum(own, partner), np.maximum(own, partner)
    pair = (a << node_bits) | b
    if hub:
        pair[rng.choice(n, hub, replace=False)] = (5 << node_bits) | 9
    fishy = (rng.random(n) < 0.01).astype(np.int64)
    keys = ((pair << 1) | fishy).astype(np.uint64)
    lo = rng.integers(26, 5000, n).astype(np.uint64)
    hi = rng.integers(26, 5000, n).astype(np.uint64) | (rng.integers(1, 4, n).astype(np.uint64) << np.uint64(30))
    lo[fishy == 1] = 0
    hi[fishy == 1] = 0
    # the graph mask is a property of the edge in real streams (both contigs' classes): take it from the key
    hi = (hi & np.uint64(0x3fffffff)) | (((keys >> np.uint64(1)) % np.uint64(3) + np.uint64(1)) << np.uint64(30))
    hi[fishy == 1] = 0
    return keys, lo | (hi << np.uint64(32))


def run_reduce(keys, payload, node_bits, cap=None, first_map=None, flags=0):
    import torch
    from besst_amd import pipeline
    n = len(keys)
    cap = cap or n
    dev = torch.device('cuda', 0)
    gb = pipeline.DeviceGraphBuilder(dev, 4, node_bits, LIB, 1, cap)
    gb.sort_flags = flags
    dk = torch.zeros(cap, dtype=torch.int64, device=dev)
    dp = torch.zeros(cap, dtype=torch.int64, device=dev)
    dk[:n] = torch.from_numpy(keys.view(np.int64)).to(dev)
    dp[:n] = torch.from_numpy(payload.view(np.int64)).to(dev)
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    fm = torch.from_numpy(first_map.view(np.int32)).to(dev) if first_map is not None else None
    for _ in range(2):                                       # the second call starts from the first one's leftovers
        gb.reduce(keys=dk, payload=dp, n_tuples_ptr=C.c_void_p(cnt.data_ptr()), capacity=cap, first_map=fm)
    n_rows = gb.read_sizes()[1]
    return gb, n_rows


def assert_rows(gb, n_rows, keys, payload, first_map=None):
    from oracle import c_oracle as CO
    want = CO.edge_rows(keys, payload)
    n, r = len(keys), len(want['key'])
    assert n_rows == r
    get = lambda t, m, dt: t[:m].cpu().numpy().view(dt)
    assert np.array_equal(get(gb.row_key, r, np.uint64), want['key'])
    assert np.array_equal(get(gb.row_n, r, np.uint32).astype(np.int64), want['n'])
    assert np.array_equal(get(gb.row_sum, r, np.int64), want['sum_obs'])
    assert np.array_equal(get(gb.row_sum_sq, r, np.int64), want['sum_obs_sq'])
    first = want['first_idx'] if first_map is None else first_map[want['first_idx']].astype(np.int64)
    assert np.array_equal(get(gb.row_first, r, np.uint32).astype(np.int64), first)
    assert np.array_equal(get(gb.row_offset, r, np.uint32).astype(np.int64), want['offset'])
    assert np.array_equal(get(gb.row_mask, r, np.uint32).astype(np.int64), want['mask'])
    assert np.array_equal(get(gb.obs_lo, n, np.int32).astype(np.int64), want['obs_lo'])
    assert np.array_equal(get(gb.obs_hi, n, np.int32).astype(np.int64), want['obs_hi'])


@pytest.mark.parametrize('n,node_bits,noise,hub,links,spread', [
    (4_300_000, 18, 0.0, 0, 40, 2),         # the clean case: 17 keys per chunk
    (6_000_001, 18, 0.03, 0, 40, 2),        # 3 % chimeric tuples (a run of one tuple each), a last chunk of one tuple
    (4_400_000, 22, 0.25, 50_000, 40, 4),   # ~290 distinct keys per chunk: after 63 the rest travels as single tuples; a hub edge
    (4_500_000, 12, 0.0, 0, 2, 1),          # a small genome walked over many times, 74 keys of ~14 tuples per chunk
    (5_000_000, 28, 0.01, 0, 300, 3),       # 57-bit keys, long runs
])
def test_clustered_streams(n, node_bits, noise, hub, links, spread):
    from besst_amd import pipeline
    keys, payload = clustered_stream(n, node_bits, noise, seed=n + node_bits, hub=hub, links=links, spread=spread)
    gb, n_rows = run_reduce(keys, payload, node_bits)
    assert gb.sort_flags == 0, 'the run-grouped form should have served this stream'
    assert_rows(gb, n_rows, keys, payload)


def test_first_map_and_large_capacity():
    """first_map (the sharded build's global emit index) goes through the runs' first indexes; the capacity, not the
    stream, picks the form: 5 M slots holding 0, 1 and 70 000 tuples."""
    rng = np.random.default_rng(3)
    for n in (0, 1, 70_000):
        keys, payload = clustered_stream(n, 18, 0.02, seed=n) if n else (np.zeros(0, np.uint64), np.zeros(0, np.uint64))
        first_map = (np.arange(n, dtype=np.uint32) * 3 + 7 + rng.integers(0, 2, n).astype(np.uint32).cumsum().astype(np.uint32))
        gb, n_rows = run_reduce(keys, payload, 18, cap=5_000_000, first_map=first_map)
        assert gb.sort_flags == 0
        assert_rows(gb, n_rows, keys, payload, first_map=first_map)


def test_unclustered_stream_falls_back():
    """Random keys: as many runs as tuples.  *n_rows reads BESST_ROWS_RUN_OVERFLOW, DeviceGraphBuilder.read_sizes()
    repeats the call with BESST_REDUCE_NO_RUNS (chained-scan passes over the tuples) and keeps the flag."""
    from besst_amd import pipeline
    n, node_bits = 4_700_000, 18
    rng = np.random.default_rng(11)
    pair = rng.integers(0, 1 << (2 * node_bits), n, dtype=np.int64)
    keys = (pair << 1).astype(np.uint64)
    payload = rng.integers(26, 5000, n).astype(np.uint64) | (rng.integers(26, 5000, n).astype(np.uint64) << np.uint64(32))
    gb, n_rows = run_reduce(keys, payload, node_bits)
    assert gb.sort_flags == pipeline.REDUCE_NO_RUNS
    assert_rows(gb, n_rows, keys, payload)
    # and the raw word, as a caller of the C ABI sees it
    import torch
    gb.sort_flags = 0
    gb._redo()
    torch.cuda.synchronize()
    raw = gb.small.cpu().numpy()
    word = int(np.frombuffer(raw[pipeline.COUNTER_BYTES + 12:pipeline.COUNTER_BYTES + 16].tobytes(), np.uint32)[0])
    assert word == pipeline.ROWS_RUN_OVERFLOW


_SPIN_SCRIPT = r'''
import sys
sys.path.insert(0, %(repo)r)
import numpy as np
sys.path.insert(0, %(tests)r)
import test_gpu_runs as T
from besst_amd import pipeline, _lib
rng = np.random.default_rng(5)
n, node_bits = 6_000_000, 18
keys = (rng.integers(0, 1 << 36, n, dtype=np.int64) << 1).astype(np.uint64)
payload = rng.integers(26, 5000, n).astype(np.uint64) | (rng.integers(26, 5000, n).astype(np.uint64) << np.uint64(32))
try:
    T.run_reduce(keys, payload, node_bits, flags=pipeline.REDUCE_NO_RUNS)
except _lib.BesstDeviceError as e:
    print('RAISED', e)
else:
    print('NO ERROR')
'''


def test_look_back_that_gives_up_is_reported():
    """BESST_OS_SPIN_LIMIT=0: the first unanswered poll of a chained-scan look-back gives up.  The partition it leaves is
    wrong, so the call must not look like a success: *n_rows carries BESST_ROWS_SORT_FAILED and the Python layer raises
    (the knob is read once per process, hence the subprocess)."""
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, BESST_OS_SPIN_LIMIT='0')
    out = subprocess.run([sys.executable, '-c', _SPIN_SCRIPT % dict(repo=os.path.dirname(here), tests=here)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert 'RAISED' in out.stdout and 'look-back gave up' in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])


_MSD_SPIN_SCRIPT = r"""
import sys
sys.path.insert(0, %(repo)r)
import numpy as np
sys.path.insert(0, %(tests)r)
import test_gpu_runs as T
from besst_amd import pipeline, _lib
rng = np.random.default_rng(6)
n, node_bits = %(n)d, 18
pair = rng.integers(0, %(pairs)d, n, dtype=np.int64)
keys = (np.sort(pair) << 1).astype(np.uint64) if %(clustered)d else (pair << 1).astype(np.uint64)
payload = rng.integers(26, 5000, n).astype(np.uint64) | (rng.integers(26, 5000, n).astype(np.uint64) << np.uint64(32))
try:
    T.run_reduce(keys, payload, node_bits)
except _lib.BesstDeviceError as e:
    print('RAISED', e)
else:
    print('NO ERROR')
"""


@pytest.mark.parametrize('n,pairs,clustered', [(300_000, 1 << 30, 0), (6_000_000, 40_000, 1)])
def test_partition_tile_that_gives_up_is_reported(n, pairs, clustered):
    """BESST_MSD_SPIN_LIMIT=0: a tile of the one-launch partition (small streams, and the runs of a large one) that
    finds the other tiles' counts missing gives up at once.  The bucket kernels behind it must not touch the stale
    bucket table, the run-grouped form's copy must not move anything, and the call reports BESST_ROWS_SORT_FAILED."""
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, BESST_MSD_SPIN_LIMIT='0')
    script = _MSD_SPIN_SCRIPT % dict(repo=os.path.dirname(here), tests=here, n=n, pairs=pairs, clustered=clustered)
    out = subprocess.run([sys.executable, '-c', script], env=env, capture_output=True, text=True, timeout=600)
    assert 'RAISED' in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])


@pytest.mark.parametrize('one_launch', ['0', '1'])
def test_partition_forms_agree(one_launch):
    """The one-launch partition (tiles wait for each other's counts) and the histogram / scan / scatter launches it
    replaces leave the same edge table (BESST_MSD_ONE_LAUNCH is read once per process, hence the subprocess)."""
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, BESST_MSD_ONE_LAUNCH=one_launch)
    script = r"""
import sys
sys.path.insert(0, %(repo)r)
import numpy as np
sys.path.insert(0, %(tests)r)
import test_gpu_runs as T
rng = np.random.default_rng(8)
for n, pairs in ((1, 5), (4097, 300), (140_000, 9_000), (700_000, 1 << 34), (3_000_000, 200_000)):
    node_bits = 18
    keys = (rng.integers(0, pairs, n, dtype=np.int64) << 1).astype(np.uint64)
    payload = rng.integers(26, 5000, n).astype(np.uint64) | (rng.integers(26, 5000, n).astype(np.uint64) << np.uint64(32))
    for rep in range(2):
        gb, n_rows = T.run_reduce(keys, payload, node_bits)
        T.assert_rows(gb, n_rows, keys, payload)
print('ALL EQUAL')
""" % dict(repo=os.path.dirname(here), tests=here)
    out = subprocess.run([sys.executable, '-c', script], env=env, capture_output=True, text=True, timeout=900)
    assert 'ALL EQUAL' in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])


def test_partitions_on_several_streams_do_not_wait_for_each_other():
    """Four streams partition ~1.9 M tuples each at the same time: ~930 tiles per launch, more workgroups together
    than the chip holds.  A form in which the tiles of a launch wait for EACH OTHER could then hold the slots another
    launch's missing tiles need; the split form (scattering workgroups wait only for counting workgroups started before
    them) cannot.  Every stream's table is checked."""
    import torch
    from besst_amd import pipeline
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(21)
    node_bits, streams, jobs = 18, [torch.cuda.Stream(device=dev) for _ in range(4)], []
    for j in range(4):
        n = 1_900_000 + 1000 * j
        keys = (rng.integers(0, 60_000 + 7 * j, n, dtype=np.int64) << 1).astype(np.uint64)
        payload = rng.integers(26, 5000, n).astype(np.uint64) | (rng.integers(26, 5000, n).astype(np.uint64) << np.uint64(32))
        gb = pipeline.DeviceGraphBuilder(dev, 4, node_bits, LIB, 1, n)
        dk = torch.from_numpy(keys.view(np.int64)).to(dev)
        dp = torch.from_numpy(payload.view(np.int64)).to(dev)
        cnt = torch.tensor([n], dtype=torch.int32, device=dev)
        jobs.append((gb, keys, payload, dk, dp, cnt, n))
    torch.cuda.synchronize()
    for st in streams:                                       # hold the four queues back so that they start together
        with torch.cuda.stream(st):
            torch.cuda._sleep(200_000_000)
    for rep in range(3):
        for st, (gb, keys, payload, dk, dp, cnt, n) in zip(streams, jobs):
            with torch.cuda.stream(st):
                gb.reduce(keys=dk, payload=dp, n_tuples_ptr=C.c_void_p(cnt.data_ptr()), capacity=n)
    torch.cuda.synchronize()
    for st, (gb, keys, payload, dk, dp, cnt, n) in zip(streams, jobs):
        with torch.cuda.stream(st):
            n_rows = gb.read_sizes()[1]
        assert gb.sort_flags == 0
        assert_rows(gb, n_rows, keys, payload)


def test_sparse_segments_chunks_spanning_many_blocks(monkeypatch):
    """The fused record loop hands its block segments over; a capacity beyond 4 M tuples picks the run-grouped form
    whatever the stream holds.  Long contigs and short inserts leave a dozen tuples per 16 384-record block, so a
    chunk of 1024 tuples spans more than the 64 blocks a wave keeps in its lanes and looks its blocks up in memory."""
    import torch
    from besst_amd import pipeline, synth, workload
    from oracle import c_oracle as CO
    monkeypatch.setenv('BESST_RECORD_PATH', '1')
    asm = synth.make_assembly(1100, 8_000, 5, sigma_log=0.3, min_len=3_000)     # (>= 1024 contigs: 25-bit keys, the hand-over is on)
    spec = synth.LibrarySpec('fr', 300.0, 20.0)
    batch = synth.simulate_library(asm, spec, 2_500_000, 6)
    lib = workload.library_constants(spec)
    table = workload.first_library_table(asm.lengths, spec.mean + 4 * spec.sd)
    node_bits = workload.node_bits_for(table)
    dev = torch.device('cuda', 0)
    rec = pipeline.DeviceRecords(batch, dev)
    gb = pipeline.DeviceGraphBuilder(dev, asm.nc, node_bits, lib, rec.n, 4_300_000)
    gb.set_contigs(**table)
    for _ in range(2):
        gb.step(rec)
    got = gb.fetch_table()
    spec_h = gb._args['presort'][0]
    assert gb._args['presort'][1] and spec_h.segmented == 1 and gb.sort_flags == 0
    keys, payload, aligned, ctr = CO.record_loop(batch, table, lib, node_bits)
    n_blocks = (rec.n + 16383) // 16384
    assert 0 < len(keys) < 14 * n_blocks, 'the stream is meant to be sparse: %d tuples in %d blocks' % (len(keys), n_blocks)
    rows = CO.edge_rows(keys, payload)
    assert np.array_equal(got.key, rows['key']) and np.array_equal(got.n.astype(np.int64), rows['n'])
    assert np.array_equal(got.first_idx.astype(np.int64), rows['first_idx'])
    assert np.array_equal(got.offset.astype(np.int64), rows['offset'])
    assert np.array_equal(got.sum_obs, rows['sum_obs']) and np.array_equal(got.sum_obs_sq, rows['sum_obs_sq'])
    assert np.array_equal(got.obs_lo.astype(np.int64), rows['obs_lo'])
    assert np.array_equal(got.obs_hi.astype(np.int64), rows['obs_hi'])
    assert gb.aligned.cpu().numpy().tolist() == aligned.tolist()
