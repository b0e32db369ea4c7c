"""The sharded drop-in (besst_amd.sharded: get_metrics + CreateGraph.PE under a process group) on CPU: two gloo ranks.

Every rank makes the calls the single-GPU drop-in makes; rank 0 leads PE, rank 1 follows.  The per-rank kernel stages are
answered by the oracle (tests/fake_device.OracleRankEngine, tests/dist_util.OracleBackend); everything else - slicing the
stream, the sharded library scans with their global cut-offs, the collective build (probe, capacity agreement, exchange,
gather of the owners' rows), the owners scoring the rows rank 0 has left after its filters, `param` travelling to the
followers - is the product's code.  Rank 0's result must equal the reference goldens, like the single-process host test.
"""
import os
import socket

import pytest

from tests import golden_util as GU

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def check_against_golden(name, doc, param, G, G_prime, Contigs, Scaffolds, small_contigs, small_scaffolds, exact_scores=True):
    from tests.test_gpu_dropin import edge_rows
    for k, want in doc['metrics'].items():
        if k == 'empirical_distribution':
            ed = getattr(param, 'empirical_distribution', None)
            got = None if ed is None else [ed[i] for i in range(len(ed))]
        else:
            got = getattr(param, k, None)
        assert got == want, (name, k)
    fin = doc['final']
    if exact_scores:
        GU.assert_scored_rows(edge_rows(G, True), fin['G'], doc, name)
        GU.assert_scored_rows(edge_rows(G_prime, True), fin['G_prime'], doc, name)
    else:
        strip = lambda rows: [{k: e[k] for k in ('u', 'v', 'nr_links', 'obs', 'obs_sq')} for e in rows]
        assert edge_rows(G, False) == strip(fin['G'])
        assert edge_rows(G_prime, False) == strip(fin['G_prime'])
    assert [list(n) for n in G.nodes()] == fin['G_nodes']
    assert [list(n) for n in G_prime.nodes()] == fin['G_prime_nodes']
    assert [[c.name, c.scaffold, c.coverage] for c in Contigs.values()] == fin['contigs']
    assert [[c.name, c.scaffold, c.coverage] for c in small_contigs.values()] == fin['small_contigs']
    assert list(Scaffolds) == fin['scaffolds'] and list(small_scaffolds) == fin['small_scaffolds']
    for k in ('mean_coverage', 'std_dev_coverage', 'edgesupport', 'expected_links_over_mean_plus_stddev',
              'scaffold_indexer', 'tot_assembly_length', 'current_N50', 'current_L50'):
        assert getattr(param, k) == fin['param'][k], (name, k)
    want_obs = {frozenset((tuple(e['u']), tuple(e['v']))): e['observations'] for e in doc['after_loop']['G_prime']}
    for u, v in G_prime.edges():
        d = G_prime[u][v]
        if d['nr_links'] is not None:
            assert d['observations'] == want_obs[frozenset((u, v))]


def follower_checks(name, doc, param, G, G_prime):
    """What a follower holds after the two calls: the same `param` as rank 0 (library metrics by its own host finishing on
    the identical sample, PE's fields from rank 0) and empty graphs."""
    for k, want in doc['metrics'].items():
        if k != 'empirical_distribution':
            assert getattr(param, k, None) == want, (name, k)
    for k in ('mean_coverage', 'std_dev_coverage', 'edgesupport', 'expected_links_over_mean_plus_stddev',
              'scaffold_indexer', 'tot_assembly_length', 'current_N50', 'current_L50'):
        assert getattr(param, k) == doc['final']['param'][k], (name, k)
    assert len(G.nodes()) == 0 and len(G_prime.nodes()) == 0


def run_sharded(doc, batch):
    """The calls of tests/test_host_dropin_cpu.run_dropin, unchanged - the process group makes them sharded."""
    from besst_amd import CreateGraph, libmetrics, session
    from tests.test_gpu_dropin import make_param, state_from_layout
    param = make_param(doc['overrides'])
    info = param.information_file
    libmetrics.get_metrics(batch, param, info)
    if doc['layout'] is not None:
        objs = state_from_layout(doc, batch, doc['layout_threshold'])
        param.scaffold_indexer = doc['layout']['next_scaffold_id']
        param.tot_assembly_length = sum(batch.lengths)
    else:
        objs = ({}, {}, {}, {})
    Contigs, Scaffolds, small_contigs, small_scaffolds = objs
    lens = dict(zip(batch.references, batch.lengths))
    C_dict = {n: 'A' * int(lens.get(n, 10)) for n in doc['fasta_names']}
    G, G_prime = CreateGraph.PE(Contigs, Scaffolds, info, C_dict, param, small_contigs, small_scaffolds, batch)
    session.close_session(batch)
    return param, G, G_prime, Contigs, Scaffolds, small_contigs, small_scaffolds


def _worker(rank, port, names, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    from besst_amd import sharded
    from tests import fake_device
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        sharded.RankEngine = fake_device.OracleRankEngine
        sharded.enable()
        assert sharded.active_group() == (rank, WORLD)
        for name in names:
            doc, batch = GU.load(name)
            res = run_sharded(doc, batch)
            if rank == 0:
                check_against_golden(name, doc, *res)
            else:
                follower_checks(name, doc, res[0], res[1], res[2])
        # the input guard across the slice boundary: both halves sorted, the second half's first record in front of the first's
        # last one (libmetrics.py:237-241; every rank warns, the index is the stream's)
        import numpy
        from besst_amd import libmetrics, session
        from tests.test_gpu_dropin import make_param
        doc, batch = GU.load('fr_given')
        half = len(batch) * 1 // WORLD
        swapped = batch.take(numpy.concatenate([numpy.arange(half, len(batch)), numpy.arange(half)]))
        param = make_param(doc['overrides'])
        libmetrics.get_metrics(swapped, param, param.information_file)
        session.close_session(swapped)
        assert param.stream_unsorted_at == len(batch) - half
        out.put((rank, len(names)))
    finally:
        dist.destroy_process_group()


def _run(target, args, timeout=900, world=WORLD):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, port) + args + (out,)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout)
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0
    return sorted(out.get(timeout=5) for _ in range(world))


def test_two_rank_sharded_dropin_reproduces_reference_goldens():
    names = GU.scenario_names()
    got = _run(_worker, (names,))
    assert got == [(0, len(names)), (1, len(names))]


def _abort_worker(rank, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    from besst_amd import CreateGraph, libmetrics, session, sharded
    from tests import fake_device
    from tests.test_gpu_dropin import make_param
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        sharded.RankEngine = fake_device.OracleRankEngine
        sharded.enable()
        doc, batch = GU.load('fr_infer')
        param = make_param(doc['overrides'])
        libmetrics.get_metrics(batch, param, param.information_file)
        # rank 0 alone finds too few contigs to compute the coverage on (CreateGraph.py:912: sys.exit(str)): the follower
        # must leave PE too instead of waiting for the next command
        lens = dict(zip(batch.references, batch.lengths))
        longest = max(doc['fasta_names'], key=lambda n: lens.get(n, 0))
        C_dict = {longest: 'A' * int(lens[longest])}
        try:
            CreateGraph.PE({}, {}, param.information_file, C_dict, param, {}, {}, batch)
            left = 'returned'
        except SystemExit as e:
            left = 'exit:%s' % ('message' if isinstance(e.code, str) else e.code)
        session.close_session(batch)
        # ... and the group is still usable: the next library runs
        doc, batch = GU.load('fr_given')
        from tests.test_sharded_dropin_cpu import run_sharded
        res = run_sharded(doc, batch)
        if rank == 0:
            check_against_golden('fr_given', doc, *res)
        out.put((rank, left))
    finally:
        dist.destroy_process_group()


def test_followers_leave_pe_when_rank_0_exits():
    got = _run(_abort_worker, ())
    assert got == [(0, 'exit:message'), (1, 'exit:0')]


def _failing_score_worker(rank, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    from besst_amd import sharded
    from tests import fake_device
    dist.init_process_group('gloo', rank=rank, world_size=3)
    try:
        class Engine(fake_device.OracleRankEngine):
            def score(self, *a, **kw):
                if rank == 1:
                    raise RuntimeError('injected: rank 1 cannot score')
                return fake_device.OracleRankEngine.score(self, *a, **kw)
        sharded.RankEngine = Engine
        sharded.enable()
        doc, batch = GU.load('rf_second_lib')
        try:
            run_sharded(doc, batch)
            left = 'returned'
        except sharded.RankFailure as e:
            left = 'failure' if 'rank 1: RuntimeError: injected' in str(e) else 'other: %s' % e
        from besst_amd import session
        session.close_session(batch)
        dist.barrier()                                       # nobody is stuck in a collective of the failed stage
        out.put((rank, left))
    finally:
        dist.destroy_process_group()


def test_a_stage_that_fails_on_one_rank_fails_on_all_three():
    """Rank 1's scoring raises; rank 0 AND rank 2 - whose own rows were fine - must leave PE with the same error instead of
    waiting for each other (three ranks: with two there is no bystander)."""
    got = _run(_failing_score_worker, (), world=3)
    assert got == [(0, 'failure'), (1, 'failure'), (2, 'failure')]
