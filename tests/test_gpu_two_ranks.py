"""Two real processes, real HIP kernel stages, real torch.distributed groups - on the ONE GPU of the test box.

RCCL refuses two ranks on one device, so the process group is gloo and besst_amd.distributed stages its
collectives through host copies (distributed._host_staged).  Everything else is what a multi-GPU node runs:
HipBackend on each rank's slice, tail gather, carry across the rank boundary, owner partition, equal-split
all-to-all, unpack, reduce, coverage/counter sums (as riders of the exchange, all-reduced in order, and
asynchronously on the side group),
capacity growth, the final gather of
the edge rows.  The union must equal the single-process C oracle on the whole stream.
"""
import os
import socket

import pytest

pytestmark = pytest.mark.gpu

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, port, config, tail_mode, pair_cap, coverage, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['BESST_TAIL_MODE'] = tail_mode
    os.environ['BESST_ALLREDUCE_ASYNC'] = '1' if coverage == 'allreduce' and tail_mode == 'late' else '0'
    os.environ['BESST_COVERAGE_EXCHANGE'] = coverage
    import torch
    import torch.distributed as dist
    from besst_amd import distributed, workload
    from tests import dist_util as DU
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        dev = torch.device('cuda', 0)
        torch.cuda.set_device(dev)
        wl = workload.make(config, 0, pairs=200000, nc=500)
        sub = dict(wl)
        sub['batch'] = DU.split_batch(wl['batch'], WORLD)[rank]
        job = distributed.ShardedGraphBuild(dev, sub, rank, WORLD, pair_capacity=pair_cap)
        # the side communicator exists only where something runs beside the main one
        assert (job.side_group is not job.group) == (coverage == 'allreduce' and tail_mode == 'late')
        assert job.backend.sums_ride_exchange == (coverage != 'allreduce')
        for _ in range(2):
            job.step()
        torch.cuda.synchronize()
        job.check_capacity()                              # small pair_cap: grows the regions and repeats the step
        b = job.backend
        want_rows, want = DU.expected_rows(wl['batch'], wl['table'], wl['lib'], wl['node_bits'])
        assert b.aligned.cpu().tolist() == want.aligned
        assert b.counter_words.cpu().tolist() == [want.count, want.non_unique, want.non_unique_for_scaf,
                                                  want.nr_of_duplicates, want.too_long, want.fishy_reads,
                                                  len(want.tuples), want.n_reach]
        assert job.final_prev_obs() == want.prev
        assert job.sizes() == (len(want.tuples), len(want_rows))
        tables = job.gather_edges(0)
        if rank == 0:
            union = {}
            for t in tables:
                rows = DU.rows_from_table(t)
                assert not set(rows) & set(union)
                union.update(rows)
            for k in union:
                if k & 1:                                 # fishy rows carry no observations
                    union[k]['lo'] = [0] * union[k]['n']
                    union[k]['hi'] = [0] * union[k]['n']
            for k, r in want_rows.items():
                if k & 1:
                    r['s'] = r['s2'] = 0
            assert union == want_rows
        out.put((rank, b.pair_cap, want.nr_of_duplicates))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('config,tail_mode,pair_cap,coverage', [('C2', 'late', 16384, 'auto'),
                                                                ('C3', 'late', 65536, 'rider'),
                                                                ('C2', 'late', 512, 'allreduce'),
                                                                ('C2', 'late', 512, 'rider'),
                                                                ('C2', 'exchange', 16384, 'auto'),
                                                                ('C3', 'exchange', 65536, 'allreduce'),
                                                                ('C2', 'exchange', 512, 'auto')])
def test_two_processes_one_gpu(config, tail_mode, pair_cap, coverage):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, config, tail_mode, pair_cap, coverage, out)) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0
    got = sorted(out.get(timeout=5) for _ in range(WORLD))
    assert [g[0] for g in got] == [0, 1]
    assert got[0][2] > 0
    if pair_cap == 512:
        assert got[0][1] > 512                            # the regions really grew


def _slice_worker(rank, port, path, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import numpy as np
    import torch
    import torch.distributed as dist
    from besst_amd import distributed
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        torch.cuda.set_device(0)
        bam, cols = distributed.ingest_slice(path, rank, WORLD, device_index=0, threads=2, chunk_blocks=64)
        try:
            rec = bam.ctx.fetch_records()
            out.put((rank, tuple(bam.boundary), {k: np.asarray(v) for k, v in rec.items()}))
        finally:
            bam.close()
    finally:
        dist.destroy_process_group()


def test_two_processes_ingest_slices_of_a_straddling_file(tmp_path):
    """distributed.ingest_slice over a real process group: a BAM whose records straddle BGZF blocks and whose qualities hold
    bytes that pass for a record header; the two ranks' slices - guessed, exchanged (all_gather_object), read again where the
    guess was wrong - are contiguous pieces of the stream and together the whole of it."""
    import numpy as np
    import torch.multiprocessing as mp
    from tests import bam_writer, test_gpu_ingest as T
    batch = T._library(5000)
    path = str(tmp_path / 'x.bam')
    bam_writer.write_bam(path, batch, block_bytes=5000, align_records=False, decoys=True)
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_slice_worker, args=(r, port, path, out)) for r in range(WORLD)]
    for p in procs:
        p.start()
    got = sorted((out.get(timeout=300) for _ in range(WORLD)), key=lambda g: g[0])
    for p in procs:
        p.join(120)
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0
    assert got[1][1][0] == got[0][1][1] and got[1][1][1] == 0
    for c in T.COLS:
        assert np.array_equal(np.concatenate([got[0][2][c], got[1][2][c]]), getattr(batch, c)), c


def test_memory_budget_covers_what_a_rank_allocates():
    """distributed.memory_budget against the allocator: records + HipBackend of one rank of a two-rank build."""
    import torch
    from besst_amd import distributed, workload
    dev = torch.device('cuda', 0)
    wl = workload.make('C3', 0, pairs=1_000_000, nc=2000)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    before = torch.cuda.memory_allocated(dev)
    backend = distributed.HipBackend(dev, wl, 0, 2, 150_000, 500_000)
    recv = torch.empty_like(backend.send)                # ShardedGraphBuild's receive buffer
    used = torch.cuda.memory_allocated(dev) - before
    budget = distributed.memory_budget(len(wl['batch']), wl['asm'].nc, 2, 150_000, 500_000)
    assert 0.9 * used <= budget['total'] <= 1.15 * used, (used, budget)
    del backend, recv
