"""world_size-2 gloo run of the multi-GPU orchestration (besst_amd.distributed) on CPU.

The per-rank kernel stages are replaced by the oracle-backed stand-in of tests/dist_util.py; everything else
- tail all-gather, carry resolution across the rank boundary, owner partition, the equal-split all-to-all of
fixed-capacity regions, source-ordered unpack, coverage/counter all-reduce - is the product's code path.
"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from besst_amd import distributed, workload
from tests import dist_util as DU

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        wl = workload.make('C2', 0, pairs=60000, nc=400)
        # many duplicates so that the chain really crosses the rank boundary
        parts = DU.split_batch(wl['batch'], WORLD)
        backend = DU.OracleBackend(parts[rank], wl['table'], wl['lib'], wl['node_bits'], rank, WORLD, 4096)
        job = distributed.ShardedGraphBuild(torch.device('cpu'), wl, rank, WORLD, backend=backend)
        for _ in range(2):
            job.step()
        job.check_capacity()
        want_rows, want = DU.expected_rows(wl['batch'], wl['table'], wl['lib'], wl['node_bits'])
        # every rank ends with the global coverage and counters
        assert backend.aligned.tolist() == want.aligned
        assert backend.counter_words.tolist() == [want.count, want.non_unique, want.non_unique_for_scaf,
                                                  want.nr_of_duplicates, want.too_long, want.fishy_reads,
                                                  len(want.tuples), want.n_reach]
        assert job.final_prev_obs() == want.prev
        n_tuples, n_rows = job.sizes()
        assert (n_tuples, n_rows) == (len(want.tuples), len(want_rows))
        # this rank's rows are exactly the keys it owns, with identical sums, order and global first index
        lib = backend.lib
        mine = {k: r for k, r in want_rows.items()
                if lib.besst_owner_of_scaffold(k >> (2 + wl['node_bits']), WORLD) == rank}
        assert backend.rows == mine
        # final gather of the owned rows to rank 0: the union is the whole edge table, keys disjoint
        tables = job.gather_edges(0)
        if rank == 0:
            union = {}
            for t in tables:
                assert not set(t) & set(union)
                union.update(t)
            assert union == want_rows
        else:
            assert tables is None
        out.put((rank, len(mine), want.nr_of_duplicates))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process_oracle():
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, out)) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = sorted(out.get(timeout=5) for _ in range(WORLD))
    assert [g[0] for g in got] == [0, 1]
    assert got[0][1] > 0 and got[1][1] > 0 and got[0][2] > 0


def _metrics_worker(rank, port, out):
    import numpy as np
    from oracle import c_oracle as CO
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        for cfg in ('C2', 'C3'):
            wl = workload.make(cfg, 0, pairs=60000, nc=400)
            lib, asm, batch = wl['lib'], wl['asm'], wl['batch']
            top = np.zeros(asm.nc, np.uint8)
            top[np.lexsort((np.arange(asm.nc), -asm.lengths))[:100]] = 1
            parts = DU.split_batch(batch, WORLD)
            job = distributed.ShardedMetricsSample(DU.OracleMetricsBackend(parts[rank], top), rank, WORLD)
            isize, contam, counts = job.sample(lib['orientation'], lib['min_mapq'], lib['read_len'])
            w_isize, w_contam, w_counts = CO.metrics_sample(batch, top, lib['orientation'], lib['min_mapq'], lib['read_len'])
            assert np.array_equal(isize, w_isize) and np.array_equal(contam, w_contam)
            assert [counts['n_isize'], counts['n_contam'], counts['counter_total'], counts['sample_counter']] == \
                   w_counts.tolist()
            assert len(isize) > 1000
        out.put((rank, len(isize), len(contam)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_metrics_sample_matches_single_process_oracle():
    """libmetrics' scans over two slices: counts all-gathered, samples placed at global positions, one all-reduce."""
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_metrics_worker, args=(r, port, out)) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = sorted(out.get(timeout=5) for _ in range(WORLD))
    assert got[0][1:] == got[1][1:] and got[0][2] > 0


def test_memory_budget_of_the_eight_gpu_configs():
    """distributed.memory_budget (what bench.py --gpus N checks before allocating): C4 and C5 on
    eight GPUs fit 288 GB of HBM with every library resident at once, and the budget grows with its inputs."""
    from besst_amd import distributed, synth
    for config, limit in (('C4', 40e9), ('C5', 150e9)):
        cfg = synth.CONFIGS[config]
        per_lib = cfg['pairs'] // len(cfg['libs']) // 8
        total = 0
        for spec in cfg['libs']:
            tuples = int(2 * per_lib * (0.214 if spec.orientation == 'rf' else 0.014))
            b = distributed.memory_budget(2 * per_lib, cfg['nc'], 8, int(tuples * 1.5 / 8) + 4096, int(tuples * 1.25) + 4096)
            assert b['total'] == sum(b['items'].values()) and b['received_capacity'] == 8 * b['pair_capacity']
            total += b['total']
        assert total < limit < 288e9
    small = distributed.memory_budget(1_000_000, 1000, 2, 10_000)['total']
    assert distributed.memory_budget(2_000_000, 1000, 2, 10_000)['total'] > small
    assert distributed.memory_budget(1_000_000, 1000, 2, 20_000)['total'] > small
    assert distributed.memory_budget(1_000_000, 1000, 4, 10_000)['total'] > small
