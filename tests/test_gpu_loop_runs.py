"""Runs of equal keys grouped by the record loop itself (fused_wave_kernel<true>, RunLayout in csrc/common.h): stage 2 then
lists the blocks' run tables, sorts the list and places the observations (rl_list / rl_place / rl_rows in csrc/runs.hip)
instead of reading the tuples back (rg_group_kernel).  Everything is compared with the C oracle's record loop and its
aggregation of the tuples (oracle/besst_oracle.c; CreateGraph.py:118-206, 812-871 and 842-862): row keys, link counts,
sums, first-occurrence indexes, offsets, graph masks, and every observation in BAM order inside its row."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CAP = 4_300_000          # a tuple capacity beyond 4 M picks the run-grouped form of stage 2 whatever the stream holds


def _variant(batch, kind, seed=5):
    """The same library seen through inputs the fast path was not tuned on."""
    from besst_amd.records import RecordBatch
    rng = np.random.default_rng(seed)
    cols = {k: getattr(batch, k).copy() for k in ('tid', 'mtid', 'pos', 'mpos', 'tlen', 'flag', 'mapq', 'qlen')}
    n = len(cols['tid'])
    if kind == 'triplicated':
        # every record three times in a row: two of three reaching records are duplicates of their predecessor
        # (CreateGraph.py:835-838), also where the predecessor is the last record of the block before - the heads the
        # stitch drops, whose run of one must go with them
        cols = {k: np.repeat(v, 3) for k, v in cols.items()}
    elif kind == 'chimeric':
        # 40 % of the records get a random mate contig: dozens of distinct keys per evaluation round, chunks that close
        # on their run count, not on their tuples
        hit = rng.random(n) < 0.4
        cols['mtid'][hit] = rng.integers(0, len(batch.references), int(hit.sum()))
    elif kind == 'shuffled':
        # no order at all (a name-sorted file): a key per tuple, the run tables overflow and the pass is repeated
        perm = rng.permutation(n)
        cols = {k: v[perm] for k, v in cols.items()}
    elif kind != 'plain':
        raise ValueError(kind)
    return RecordBatch(batch.references, batch.lengths, **cols)


def _build(wl, batch, steps=2):
    import torch
    from besst_amd import pipeline
    dev = torch.device('cuda', 0)
    rec = pipeline.DeviceRecords(batch, dev)
    gb = pipeline.DeviceGraphBuilder(dev, wl['asm'].nc, wl['node_bits'], wl['lib'], rec.n, CAP)
    gb.set_contigs(**wl['table'])
    for _ in range(steps):                                   # the second pass starts from the first one's leftovers
        gb.step(rec)
    table = gb.fetch_table()
    return gb, table, gb.read_counters()


def _check(gb, table, ctr, wl, batch):
    from oracle import c_oracle as CO
    keys, payload, c_aligned, c_ctr = CO.record_loop(batch, wl['table'], wl['lib'], wl['node_bits'])
    rows = CO.edge_rows(keys, payload)
    assert [ctr.count, ctr.non_unique, ctr.non_unique_for_scaf, ctr.nr_of_duplicates, ctr.reads_with_too_long_insert,
            ctr.fishy_reads, ctr.n_tuples, ctr.n_reach, ctr.prev_obs1, ctr.prev_obs2] == c_ctr.tolist()
    assert gb.aligned.cpu().numpy().tolist() == c_aligned.tolist()
    assert np.array_equal(table.key, rows['key']) and np.array_equal(table.n.astype(np.int64), rows['n'])
    assert np.array_equal(table.first_idx.astype(np.int64), rows['first_idx'])
    assert np.array_equal(table.offset.astype(np.int64), rows['offset'])
    assert np.array_equal(table.sum_obs, rows['sum_obs']) and np.array_equal(table.sum_obs_sq, rows['sum_obs_sq'])
    link = ~table.is_fishy
    assert np.array_equal(table.mask[link].astype(np.int64), rows['mask'][link])
    assert np.array_equal(table.obs_lo.astype(np.int64), rows['obs_lo'])
    assert np.array_equal(table.obs_hi.astype(np.int64), rows['obs_hi'])
    return len(keys), len(rows['key'])


@pytest.fixture(autouse=True)
def fused_loop(monkeypatch):
    monkeypatch.setenv('BESST_RECORD_PATH', '1')


@pytest.mark.parametrize('config,pairs,nc,kind', [
    ('C3', 1_200_000, 3000, 'plain'),          # mate pairs with PE contamination: ~100 tuples per 1000 records
    ('C3', 400_000, 1500, 'triplicated'),      # duplicates everywhere, block heads that the stitch drops
    ('C3', 700_000, 2500, 'chimeric'),         # chunks closed by their run count
    ('C2', 900_000, 2000, 'plain'),            # a paired-end library forced through the fused loop: a few tuples per block
    ('C3', 5_000, 1200, 'plain'),              # one block, a handful of tuples (>= 1024 contigs: 25-bit keys, the hand-over is on)
])
def test_record_loop_groups_runs(config, pairs, nc, kind):
    from besst_amd import workload
    wl = workload.make(config, 0, pairs=pairs, nc=nc)
    batch = _variant(wl['batch'], kind)
    gb, table, ctr = _build(wl, batch)
    spec = gb._args['presort'][0]
    assert gb._args['presort'][1] and spec.segmented == 1 and spec.in_record_loop == 3, 'the record loop should have grouped the runs'
    assert gb.sort_flags == 0
    n_tuples, n_rows = _check(gb, table, ctr, wl, batch)
    assert n_tuples > 0 and n_rows > 0


def test_shuffled_stream_is_served_with_a_run_per_tuple():
    """A shuffled library (a name-sorted file) at this density: a key per tuple, every evaluation round closes a chunk,
    but a block holds fewer rounds than it has room for chunks and the runs fit their list - slow, exact, no repeat."""
    from besst_amd import workload
    wl = workload.make('C3', 0, pairs=600_000, nc=2000)
    batch = _variant(wl['batch'], 'shuffled')
    gb, table, ctr = _build(wl, batch, steps=1)
    assert gb.sort_flags == 0 and gb._args['presort'][0].in_record_loop == 3
    _check(gb, table, ctr, wl, batch)


def test_scattered_stream_runs_out_of_chunks_and_falls_back():
    """Every record a read 2 with its mate on a random contig and an insert-size threshold that accepts them all: most
    records emit a tuple and no two of a round share a key, so a block closes more chunks than kRlMaxChunks.  The record loop says so
    (run_status), stage 2 reports BESST_ROWS_RUN_OVERFLOW, read_sizes() repeats the pass with BESST_REDUCE_NO_RUNS -
    record loop included, which this time writes keys and counts the sort's digits."""
    from besst_amd import pipeline, workload
    from besst_amd.records import RecordBatch
    wl = workload.make('C3', 0, pairs=500_000, nc=2000)
    wl['lib'] = dict(wl['lib'], ins_size_threshold=5.0e8)
    b = wl['batch']
    rng = np.random.default_rng(17)
    cols = {k: getattr(b, k).copy() for k in ('tid', 'mtid', 'pos', 'mpos', 'tlen', 'flag', 'mapq', 'qlen')}
    cols['mtid'] = rng.integers(0, len(b.references), len(cols['tid'])).astype(np.int32)
    cols['flag'] = ((cols['flag'] & ~np.uint16(0x4 | 0x8 | 0x40)) | np.uint16(0x80)).astype(np.uint16)   # every record a mapped read 2
    cols['mapq'][:] = 60
    batch = RecordBatch(b.references, b.lengths, **cols)
    gb, table, ctr = _build(wl, batch, steps=1)
    assert gb.sort_flags == pipeline.REDUCE_NO_RUNS
    assert gb._args['presort'][0].in_record_loop == 1
    n_tuples, _ = _check(gb, table, ctr, wl, batch)
    assert n_tuples > 700_000                                # (dense enough: > 11 000 tuples per 16 384-record block)
    gb.step(pipeline.DeviceRecords(batch, gb.device))        # the flag stays: no failed attempt on the passes that follow
    assert gb.fetch_table().key.tolist() == table.key.tolist()


def test_empty_and_linkless_streams():
    """No tuple at all (every mate on the read's own contig): no chunk, no run, an empty table - and the coverage sums."""
    from besst_amd import workload
    from besst_amd.records import RecordBatch
    wl = workload.make('C3', 0, pairs=200_000, nc=800)
    b = wl['batch']
    cols = {k: getattr(b, k).copy() for k in ('tid', 'mtid', 'pos', 'mpos', 'tlen', 'flag', 'mapq', 'qlen')}
    cols['mtid'] = cols['tid'].copy()
    batch = RecordBatch(b.references, b.lengths, **cols)
    gb, table, ctr = _build(wl, batch)
    assert len(table.key) == 0 and ctr.n_tuples == 0
    _check(gb, table, ctr, wl, batch)


_AB_SCRIPT = r'''
import sys
sys.path.insert(0, %(repo)r)
sys.path.insert(0, %(tests)r)
import test_gpu_loop_runs as T
from besst_amd import workload
wl = workload.make('C3', 0, pairs=900_000, nc=2500)
for kind in ('plain', 'triplicated'):
    batch = T._variant(wl['batch'], kind)
    gb, table, ctr = T._build(wl, batch)
    assert gb._args['presort'][0].in_record_loop == 2 and gb.sort_flags == 0, gb._args['presort'][0].in_record_loop
    T._check(gb, table, ctr, wl, batch)
print('ALL EQUAL')
'''


def test_grouping_kernel_still_serves_key_segments():
    """BESST_LOOP_RUNS=0 (read once per process, hence the subprocess): the record loop writes keys and rg_group_kernel
    finds the runs in the block segments, as it does behind the two-pass record loop - same tables."""
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, BESST_LOOP_RUNS='0', BESST_RECORD_PATH='1')
    out = subprocess.run([sys.executable, '-c', _AB_SCRIPT % dict(repo=os.path.dirname(here), tests=here)], env=env,
                         capture_output=True, text=True, timeout=900)
    assert 'ALL EQUAL' in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])
