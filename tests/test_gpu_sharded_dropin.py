"""The sharded drop-in on the real kernels: W processes, one process group (gloo: RCCL refuses two ranks on one device, so
besst_amd.distributed stages its collectives through host copies), all on the ONE GPU of the test box.

Every rank makes the single-GPU drop-in's calls - bamio.open_bam / libmetrics.get_metrics / CreateGraph.PE - and the process
group makes them sharded (besst_amd.sharded): slices of the stream (or of the BAM file: distributed.ingest_slice on the
GPU), ShardedMetricsSample, ShardedGraphBuild, owners scoring their rows, rank 0 assembling the graphs.  Rank 0's graphs,
objects and `param` must equal (1) the reference goldens and (2) what the single-GPU drop-in returns in the same process,
`==` on every field including gap and score.
"""
import os

import pytest

from tests import golden_util as GU
from tests.test_sharded_dropin_cpu import _free_port, check_against_golden, follower_checks, run_sharded

pytestmark = pytest.mark.gpu


def _snapshot(res):
    from tests.test_gpu_dropin import edge_rows
    param, G, G_prime, Contigs, Scaffolds, small_contigs, small_scaffolds = res
    fields = ('read_len', 'mean_ins_size', 'std_dev_ins_size', 'ins_size_threshold', 'contig_threshold', 'contamination_ratio',
              'contamination_mean', 'contamination_stddev', 'mean_coverage', 'std_dev_coverage', 'edgesupport',
              'expected_links_over_mean_plus_stddev', 'scaffold_indexer', 'tot_assembly_length', 'current_N50', 'current_L50',
              'lognormal')
    obs = {}
    for name, graph in (('G', G), ('G_prime', G_prime)):
        for u, v in graph.edges():
            d = graph[u][v]
            if d['nr_links'] is not None:
                obs[(name, u, v)] = list(d['observations'])
    return dict(G=edge_rows(G, True), G_prime=edge_rows(G_prime, True), G_nodes=list(G.nodes()),
                Gp_nodes=list(G_prime.nodes()), obs=obs,
                contigs=[[c.name, c.scaffold, c.coverage, c.position, c.direction] for c in Contigs.values()],
                small=[[c.name, c.scaffold, c.coverage] for c in small_contigs.values()],
                scaffolds=list(Scaffolds), small_scaffolds=list(small_scaffolds),
                param={k: getattr(param, k, None) for k in fields})


def _run_from_file(doc, path, batch):
    from besst_amd import CreateGraph, bamio, libmetrics, session
    from tests.test_gpu_dropin import make_param
    records = bamio.open_bam(path, threads=2, chunk_blocks=64)
    assert len(records) == len(batch) and list(records.references) == list(batch.references)
    param = make_param(doc['overrides'])
    info = param.information_file
    libmetrics.get_metrics(records, param, info)
    lens = dict(zip(batch.references, batch.lengths))
    C_dict = {n: 'A' * int(lens.get(n, 10)) for n in doc['fasta_names']}
    objs = ({}, {}, {}, {})
    G, G_prime = CreateGraph.PE(objs[0], objs[1], info, C_dict, param, objs[2], objs[3], records)
    session.close_session(records)
    records.close()
    return (param, G, G_prime) + objs, records


def _worker(rank, world, port, names, bam_cases, tmp, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch
    import torch.distributed as dist
    from besst_amd import bamio, sharded
    from tests import bam_writer
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from besst_amd import sharded as _sh
    _sh.enable()
    try:
        n_scored = 0
        for name in names:
            doc, batch = GU.load(name)
            single = None
            if rank == 0:
                os.environ['BESST_SHARDED'] = '0'
                single = _snapshot(run_sharded(doc, batch))
                os.environ['BESST_SHARDED'] = '1'
            assert sharded.active_group() == (rank, world)
            res = run_sharded(doc, batch)
            if rank == 0:
                check_against_golden(name, doc, *res, exact_scores=False)
                got = _snapshot(res)
                for k in single:
                    assert got[k] == single[k], (name, k)
                n_scored += sum(1 for e in got['G'] if 'score' in e)
            else:
                follower_checks(name, doc, res[0], res[1], res[2])
        for name, layout in bam_cases:
            doc, batch = GU.load(name)
            path = os.path.join(tmp, '%s_%s.bam' % (name, layout))
            if rank == 0:
                if layout == 'htslib':
                    bamio.write_bam(path, batch, threads=2)
                else:
                    bam_writer.write_bam(path, batch, block_bytes=5000, align_records=False, decoys=True)
            dist.barrier()
            single = None
            if rank == 0:
                os.environ['BESST_SHARDED'] = '0'
                res1, rec1 = _run_from_file(doc, path, batch)
                assert type(rec1).__name__ == 'ResidentBam'
                single = _snapshot(res1)
                os.environ['BESST_SHARDED'] = '1'
            res, rec = _run_from_file(doc, path, batch)
            assert type(rec).__name__ == 'ShardedBam' and rec.ingest.on_device == 1
            assert sum(rec.head.slice_records) == len(batch)
            if rank == 0:
                got = _snapshot(res)
                for k in single:
                    assert got[k] == single[k], (name, layout, k)
                strip = lambda rows: [{k: e[k] for k in ('u', 'v', 'nr_links', 'obs', 'obs_sq')} for e in rows]
                assert strip(got['G']) == strip(doc['final']['G']) and strip(got['G_prime']) == strip(doc['final']['G_prime'])
                n_scored += len(got['G'])
        out.put((rank, n_scored))
    finally:
        dist.destroy_process_group()


def _launch(world, names, bam_cases, tmp_path, timeout=1500):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, names, bam_cases, str(tmp_path), out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout)
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0
    got = sorted(out.get(timeout=5) for _ in range(world))
    assert [g[0] for g in got] == list(range(world)) and got[0][1] > 0


def test_two_ranks_every_golden_and_the_bam_files(tmp_path):
    _launch(2, GU.scenario_names(), [('fr_infer', 'htslib'), ('rf_contam', 'straddling')], tmp_path)


@pytest.mark.parametrize('world', [3, 8])
def test_more_ranks_second_library_and_a_straddling_file(world, tmp_path):
    """rf_second_lib: a non-trivial contig table (scaffolds of several contigs, directions, positions) that only rank 0
    holds as objects - the followers get it with the build command; fr_infer from a file whose records straddle blocks."""
    _launch(world, ['rf_second_lib', 'fr_edgecases'], [('fr_infer', 'straddling')], tmp_path)


def test_cli_under_torchrun(tmp_path):
    """python -m torch.distributed.run ... -m besst_amd.cli: two ranks (gloo, one GPU), each ingesting its slice of the BAM;
    rank 0 writes the same scored edge table the reference golden holds."""
    import subprocess
    import sys
    from tests import bam_writer
    doc, batch = GU.load('fr_infer')
    bam = str(tmp_path / 'lib.bam')
    bam_writer.write_bam(bam, batch, block_bytes=20000, align_records=False)
    fasta = str(tmp_path / 'contigs.fa')
    lens = dict(zip(batch.references, batch.lengths))
    with open(fasta, 'w') as fh:
        for n in doc['fasta_names']:
            fh.write('>%s\n%s\n' % (n, 'A' * lens[n]))
    env = dict(os.environ, BESST_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), '-m', 'besst_amd.cli', '-c', fasta, '-f', bam, '-orientation', 'fr', '-o',
           str(tmp_path)]
    done = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert done.returncode == 0, done.stdout.decode()[-3000:]
    rows = [l.rstrip('\n').split('\t') for l in open(str(tmp_path / 'BESST_output' / 'pass1' / 'edges_G.tsv'))][1:]
    got = [(int(r[0]), r[1], int(r[2]), r[3], int(r[4]), int(r[5]), int(r[6])) for r in rows]
    want = [(e['u'][0], e['u'][1], e['v'][0], e['v'][1], e['nr_links'], e['obs'], e['obs_sq']) for e in doc['final']['G']]
    assert got == want and len(got) > 10
    stats = open(str(tmp_path / 'BESST_output' / 'Statistics.txt')).read()
    assert 'LIBRARY STATISTICS' in stats and 'Number of edges in G (after repeat removal)' in stats


def test_cli_two_libraries_under_torchrun_equals_one_process(tmp_path):
    """Two libraries through the CLI (fr, then rf on the objects the first pass leaves: CleanObjects, a contig table that only
    rank 0 holds as objects): three ranks under torchrun write the same edge tables for both passes as one process does."""
    import subprocess
    import sys
    from tests import bam_writer
    doc, batch = GU.load('fr_infer')
    doc2, batch2 = GU.load('rf_contam')
    assert list(batch2.references) == list(batch.references)[:len(batch2.references)]    # (the second library's header names a part of the contigs)
    bams = [str(tmp_path / 'lib1.bam'), str(tmp_path / 'lib2.bam')]
    bam_writer.write_bam(bams[0], batch, block_bytes=20000, align_records=False)
    bam_writer.write_bam(bams[1], batch2, block_bytes=30000, align_records=True)
    fasta = str(tmp_path / 'contigs.fa')
    lens = dict(zip(batch.references, batch.lengths))
    with open(fasta, 'w') as fh:
        for n in doc['fasta_names']:
            fh.write('>%s\n%s\n' % (n, 'A' * lens[n]))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ['-m', 'besst_amd.cli', '-c', fasta, '-f'] + bams + ['-orientation', 'fr', 'rf']
    one, many = str(tmp_path / 'one'), str(tmp_path / 'many')
    done = subprocess.run([sys.executable] + args + ['-o', one], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert done.returncode == 0, done.stdout.decode()[-3000:]
    env = dict(os.environ, BESST_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '3', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port())] + args + ['-o', many]
    done = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert done.returncode == 0, done.stdout.decode()[-3000:]
    for p in ('pass1', 'pass2'):
        for name in ('edges_G.tsv', 'edges_Gprime.tsv'):
            a = open(os.path.join(one, 'BESST_output', p, name)).read()
            b = open(os.path.join(many, 'BESST_output', p, name)).read()
            assert a == b and a.count('\n') > 1, (p, name)


def _tiny_worker(rank, world, port, tmp, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch
    import torch.distributed as dist
    from tests import bam_writer
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from besst_amd import sharded as _sh
    _sh.enable()
    try:
        doc, batch = GU.load('fr_given')
        batch = batch.slice(0, 1500)                          # ~5 BGZF blocks of 64 KiB: fewer blocks than ranks
        path = os.path.join(tmp, 'tiny.bam')
        if rank == 0:
            bam_writer.write_bam(path, batch, block_bytes=65000, align_records=False)
        dist.barrier()
        single = None
        if rank == 0:
            os.environ['BESST_SHARDED'] = '0'
            res1, _ = _run_from_file(doc, path, batch)
            single = _snapshot(res1)
            os.environ['BESST_SHARDED'] = '1'
        res, rec = _run_from_file(doc, path, batch)
        empty = sum(1 for n in rec.head.slice_records if n == 0)
        if rank == 0:
            got = _snapshot(res)
            for k in single:
                assert got[k] == single[k], k
            assert len(got['G_prime']) > 0
        out.put((rank, empty))
    finally:
        dist.destroy_process_group()


def test_more_ranks_than_blocks(tmp_path):
    """A file of a handful of BGZF blocks over eight ranks: several slices hold no block and no record - those ranks still
    take part in the scans, the exchange and the scoring, and rank 0's result equals the single-GPU one."""
    import torch.multiprocessing as mp
    world = 8
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tiny_worker, args=(r, world, port, str(tmp_path), out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0
    got = sorted(out.get(timeout=5) for _ in range(world))
    assert [g[0] for g in got] == list(range(world)) and got[0][1] >= 2      # at least two ranks held nothing


def _shaped_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch
    import torch.distributed as dist
    from besst_amd import CreateGraph, libmetrics, session, workload
    from tests.test_gpu_dropin import make_param
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from besst_amd import sharded as _sh
    _sh.enable()
    try:
        snaps = {}
        for config, pairs, nc in (('C3', 3_000_000, 6000), ('C2', 1_500_000, 3000)):
            wl = workload.make(config, 0, pairs=pairs, nc=nc)
            batch = wl['batch']
            for mode in (('0', '1') if rank == 0 else ('1',)):
                os.environ['BESST_SHARDED'] = mode
                param = make_param(dict(orientation=wl['lib']['orientation']))
                info = param.information_file
                libmetrics.get_metrics(batch, param, info)
                objs = ({}, {}, {}, {})
                C_dict = {name: 'A' * int(n) for name, n in zip(batch.references, batch.lengths)} if rank == 0 else {}
                G, Gp = CreateGraph.PE(objs[0], objs[1], info, C_dict, param, objs[2], objs[3], batch)
                session.close_session(batch)
                if rank == 0:
                    snaps[(config, mode)] = _snapshot((param, G, Gp) + objs)
            os.environ['BESST_SHARDED'] = '1'
            if rank == 0:
                one, many = snaps[(config, '0')], snaps[(config, '1')]
                for k in one:
                    assert many[k] == one[k], (config, k)
                assert len(one['G']) > 500 and len(one['G_prime']) > len(one['G'])
        out.put((rank, True))
    finally:
        dist.destroy_process_group()


def test_three_ranks_on_config_shaped_libraries():
    """The sharded drop-in at a size where the regions, the gather and the owners' scoring carry weight: a mate-pair library
    with PE contamination (C3's shape, 6 M records, 6000 contigs: the fused record loop, ~600 k link tuples, inferred
    insert-size statistics through the sharded sampler) and a paired-end one (C2's shape) over three ranks `==` one GPU."""
    import torch.multiprocessing as mp
    world = 3
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shaped_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(1500)
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0
    assert sorted(out.get(timeout=5) for _ in range(world)) == [(r, True) for r in range(world)]


def _rccl_worker(port, tmp, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['BESST_SHARDED'] = 'force'
    import torch
    import torch.distributed as dist
    from besst_amd import sharded
    from tests import bam_writer
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    from besst_amd import sharded as _sh
    _sh.enable()
    try:
        assert sharded.active_group() == (0, 1) and dist.get_backend() == 'nccl'
        n = 0
        for name in ('fr_infer', 'rf_contam', 'rf_second_lib', 'fr_edgecases', 'fr_dense'):
            doc, batch = GU.load(name)
            res = run_sharded(doc, batch)
            check_against_golden(name, doc, *res, exact_scores=False)
            n += len(res[1].edges())
        doc, batch = GU.load('fr_infer')
        path = os.path.join(tmp, 'x.bam')
        bam_writer.write_bam(path, batch, block_bytes=5000, align_records=False, decoys=True)
        res, rec = _run_from_file(doc, path, batch)
        assert type(rec).__name__ == 'ShardedBam'
        strip = lambda rows: [{k: e[k] for k in ('u', 'v', 'nr_links', 'obs', 'obs_sq')} for e in rows]
        got = _snapshot(res)
        assert strip(got['G']) == strip(doc['final']['G']) and strip(got['G_prime']) == strip(doc['final']['G_prime'])
        out.put(n)
    finally:
        dist.destroy_process_group()


def test_one_rank_over_rccl_through_the_sharded_dropin(tmp_path):
    """BESST_SHARDED=force with ONE rank over RCCL (backend nccl): the whole sharded orchestration - object collectives on the
    device, the all-to-all and the all-reduces on device tensors without host staging, owners' scoring, the gather - on the
    transport a multi-GPU node uses, as far as one GPU can show it; results equal the reference goldens."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), str(tmp_path), out))
    p.start()
    p.join(900)
    if p.is_alive():
        p.kill()
    assert p.exitcode == 0
    assert out.get(timeout=5) > 100
