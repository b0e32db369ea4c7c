"""CPU check of the drop-in's HOST side and of the hand-over to the reference's unchanged downstream stage.

The device stages are answered by the oracle (tests/fake_device.py); everything else is the product's
besst_amd.libmetrics.get_metrics / besst_amd.CreateGraph.PE.  The result must (1) equal the reference goldens and
(2) when the reference is present here, drive the reference's own MakeScaffolds.Algorithm to the same scaffolds
as the all-reference pipeline.
"""
import importlib
import io

import numpy
import pytest

from besst_amd import CreateGraph, libmetrics, session
from tests import fake_device
from tests import golden_util as GU
from tests.refharness import loader
from tests.test_gpu_dropin import edge_rows, make_param, state_from_layout


@pytest.fixture
def fake_gpu(monkeypatch):
    monkeypatch.setattr(session.device, 'GraphContext', fake_device.FakeGraphContext)
    yield


def run_dropin(doc, batch, **extra):
    param = make_param(dict(doc['overrides'], **extra))
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


@pytest.mark.parametrize('name', GU.scenario_names())
def test_host_side_reproduces_reference_goldens(fake_gpu, name):
    doc, batch = GU.load(name)
    param, G, G_prime, Contigs, Scaffolds, small_contigs, small_scaffolds = run_dropin(doc, batch)
    for k, want in doc['metrics'].items():
        if k == 'empirical_distribution':
            ed = getattr(param, 'empirical_distribution', None)
            got = None if ed is None else [ed[i] for i in range(len(ed))]
        else:
            got = getattr(param, k, None)
        assert got == want, (name, k)
    fin = doc['final']
    GU.assert_scored_rows(edge_rows(G, True), fin['G'], doc, name)          # incl. gap and score: same float expressions
    GU.assert_scored_rows(edge_rows(G_prime, True), fin['G_prime'], doc, name)
    assert [list(n) for n in G.nodes()] == fin['G_nodes']
    assert [list(n) for n in G_prime.nodes()] == fin['G_prime_nodes']
    assert [[c.name, c.scaffold, c.coverage] for c in Contigs.values()] == fin['contigs']
    assert [[c.name, c.scaffold, c.coverage] for c in small_contigs.values()] == fin['small_contigs']
    assert list(Scaffolds) == fin['scaffolds'] and list(small_scaffolds) == fin['small_scaffolds']
    for k in ('mean_coverage', 'std_dev_coverage', 'edgesupport', 'expected_links_over_mean_plus_stddev',
              'scaffold_indexer', 'tot_assembly_length', 'current_N50', 'current_L50'):
        assert getattr(param, k) == fin['param'][k], (name, k)
    # the observation lists (cut out of the device column on first access) keep BAM order, in both graphs
    for graph, snap in ((G, doc['after_loop']['G']), (G_prime, doc['after_loop']['G_prime'])):
        want_obs = {frozenset((tuple(e['u']), tuple(e['v']))): e['observations'] for e in snap}
        for u, v in graph.edges():
            d = graph[u][v]
            if d['nr_links'] is not None:
                assert 'observations' not in dict.keys(d)           # not cut yet ...
                assert d['observations'] == want_obs[frozenset((u, v))]
                assert dict.__contains__(d, 'observations') and len(d['observations']) == d['nr_links']


def test_link_data_writers_see_and_replace_the_lazy_list():
    """Writers on an edge whose 'observations' were not cut yet: an assignment wins over the column (it is not overwritten by
    a later read), pop / del / setdefault / update see the list, and a LinkData built without a column is a plain dict."""
    col = numpy.arange(100, dtype=numpy.int32)

    def fresh():
        d = CreateGraph.LinkData(nr_links=3, obs=33, obs_sq=365)
        d._col, d._lo, d._hi = col, 10, 13
        return d
    d = fresh()
    d['observations'] = [7]
    assert list(d.keys()).count('observations') == 1 and d['observations'] == [7] and dict(d)['observations'] == [7]
    assert fresh().pop('observations') == [10, 11, 12]
    d = fresh()
    del d['observations']
    assert 'observations' not in d and len(d) == 3
    assert fresh().setdefault('observations', None) == [10, 11, 12]
    d = fresh()
    d.update(gap=4)
    assert d['observations'] == [10, 11, 12] and d['gap'] == 4
    d = fresh()
    d.update(observations=[1])
    assert d['observations'] == [1]
    d = fresh()
    d.clear()
    assert len(d) == 0 and 'observations' not in d
    plain = CreateGraph.LinkData(nr_links=None)              # (built anywhere: no column, no AttributeError)
    assert 'observations' not in plain and plain.get('observations') is None and len(plain) == 1
    with pytest.raises(KeyError):
        plain['observations']


def test_link_data_is_a_complete_dict_to_whatever_looks_at_it_whole():
    col = numpy.arange(100, dtype=numpy.int32)

    def fresh():
        d = CreateGraph.LinkData(nr_links=3, obs=33, obs_sq=365)
        d._col, d._lo, d._hi = col, 10, 13
        return d
    want = dict(nr_links=3, obs=33, obs_sq=365, observations=[10, 11, 12])
    assert fresh()['observations'] == [10, 11, 12]
    assert fresh() == want and want == fresh() and not (fresh() != want)
    assert dict(fresh()) == want and fresh().copy() == want and dict(**fresh()) == want
    assert sorted(fresh()) == sorted(want) and len(fresh()) == 4
    assert 'observations' in fresh() and fresh().get('observations') == [10, 11, 12]
    assert dict(fresh().items()) == want and sorted(fresh().keys()) == sorted(want)
    plain = {}
    plain.update(fresh())                                    # what networkx's add_edges_from does with edge data
    assert plain == want
    d = fresh()
    d['observations'].append(7)                              # the cut list is the edge's own, mutable list
    assert d['observations'] == [10, 11, 12, 7]
    with pytest.raises(KeyError):
        fresh()['gap']
    d = fresh()
    d['observations'] = [1]                                  # an explicit assignment wins over the column
    assert d['observations'] == [1] and len(d) == 4


def _scaffold_summary(Scaffolds, small_scaffolds):
    out = []
    for group in (Scaffolds, small_scaffolds):
        out.append(sorted((s.s_length, tuple((c.name, c.direction, c.position) for c in s.contigs))
                          for s in group.values()))
    return out


@pytest.mark.skipif(not loader.available(), reason='reference checkout not present (build container only)')
@pytest.mark.parametrize('name', ['fr_infer', 'fr_nodup', 'fr_given', 'fr_edgecases', 'fr_noextend', 'rf_contam'])
def test_unchanged_reference_makescaffolds_accepts_the_graph(fake_gpu, name):
    mods = loader.load()
    from tests.refharness import driver
    MS = importlib.import_module('BESST.MakeScaffolds')
    importlib.import_module('BESST.lp_solve').Inf = numpy.inf     # numpy >= 2 dropped the alias the reference imports
    doc, batch = GU.load(name)
    common = dict(path_threshold=100000, score_cutoff=1.5, max_extensions=None, NO_ILP=False, FASTER_ILP=False,
                  dfs_traversal=True, multiprocess=False, development=False, plots=False, hapl_ratio=1.3,
                  hapl_threshold=3, bamfile='synthetic.bam')
    # all-reference pipeline
    rp = driver.make_param(mods, **doc['overrides'])
    driver.run_get_metrics(mods, batch, rp)
    _, _, (rG, rGp, rC, rS, rsc, rss) = driver.run_pe(mods, batch, rp, doc['fasta_names'])
    rp.information_file = io.StringIO()
    try:
        MS.Algorithm(rG, rGp, rC, rsc, rS, rss, rp.information_file, rp)
    except Exception as exc:      # the reference's own downstream code is not fully networkx-3 / Python-3 clean
        pytest.skip('reference MakeScaffolds itself fails on this scenario here: %r' % (exc,))
    want = _scaffold_summary(rS, rss)
    # drop-in graph construction, then the SAME unchanged reference stage
    param, G, G_prime, Contigs, Scaffolds, small_contigs, small_scaffolds = run_dropin(doc, batch, **common)
    param.information_file = io.StringIO()
    MS.Algorithm(G, G_prime, Contigs, small_contigs, Scaffolds, small_scaffolds, param.information_file, param)
    assert _scaffold_summary(Scaffolds, small_scaffolds) == want
    assert sum(len(s.contigs) > 1 for s in Scaffolds.values()) > 5


def _chain_state(G, G_prime, Contigs, small_contigs, Scaffolds, small_scaffolds, param):
    return dict(contigs=sorted((c.name, c.scaffold, c.position, c.direction) for c in Contigs.values()),
                small=sorted((c.name, c.scaffold, c.position, c.direction) for c in small_contigs.values()),
                scaffolds=[(k, s.s_length, [c.name for c in s.contigs]) for k, s in Scaffolds.items()],
                small_scaffolds=list(small_scaffolds), indexer=param.scaffold_indexer,
                gaps=sorted(param.gap_estimations), G_nodes=G.nodes(),
                prime_nodes=sorted(G_prime.nodes()),
                prime_links=sorted((min(u, v), max(u, v), G_prime[u][v]['nr_links'], G_prime[u][v].get('obs'))
                                   for u, v in G_prime.edges()))


@pytest.mark.skipif(not loader.available(), reason='reference checkout not present (build container only)')
@pytest.mark.parametrize('name', ['fr_infer', 'fr_given', 'rf_contam', 'fr_dense_e2', 'rf_second_lib', 'fr_edgecases', 'fr_nodup'])
def test_new_contigs_scaffolds_with_the_reference_path_search_as_hook(fake_gpu, monkeypatch, name):
    """param.extend_paths (BESST's default): the reference runs PROWithinScaf per component inside NewContigsScaffolds
    (MakeScaffolds.py:283-285).  The drop-in's NewContigsScaffolds takes that function as `within_scaffold`; with the
    reference's own function handed in, every object, both graphs and the indexer end up as the reference leaves them."""
    import copy
    from besst_amd import MakeScaffolds as OURS
    mods = loader.load()
    MS = importlib.import_module('BESST.MakeScaffolds')
    importlib.import_module('BESST.lp_solve').Inf = numpy.inf
    monkeypatch.setattr(OURS, 'chain_arrays', fake_device.fake_chain_arrays)
    doc, batch = GU.load(name)
    common = dict(path_threshold=100000, score_cutoff=1.5, max_extensions=None, NO_ILP=False, FASTER_ILP=False,
                  dfs_traversal=True, multiprocess=False, development=False, plots=False, hapl_ratio=1.3,
                  hapl_threshold=3, bamfile='synthetic.bam', path_gaps_estimated=0, gap_estimations=[])
    results, moved = [], []
    for ours in (False, True):
        param, G, G_prime, Contigs, Scaffolds, small_contigs, small_scaffolds = run_dropin(doc, batch, **common)
        param.gap_estimations = []
        info = io.StringIO()
        table = MS.GC.PreCalcMLvaluesOfdLongContigs(param.mean_ins_size, param.std_dev_ins_size, param.read_len)
        already_visited = set(G) if param.extend_paths else set()
        G = MS.RemoveIsolatedContigs(G, info)
        MS.RemoveAmbiguousRegionsUsingScore(G, G_prime, info, param, 'G')
        G = MS.RemoveIsolatedContigs(G, info)
        G, Contigs, Scaffolds = MS.RemoveLoops(G, G_prime, Scaffolds, Contigs, info, param)
        n_small = len(small_scaffolds)
        args = (G, G_prime, Contigs, small_contigs, Scaffolds, small_scaffolds, info, table, param, already_visited)
        try:
            if ours:
                OURS.NewContigsScaffolds(*args, within_scaffold=MS.PROWithinScaf)
            else:
                MS.NewContigsScaffolds(*args)
        except Exception as exc:                              # the reference's path search is not Python-3 clean everywhere
            if not ours:
                pytest.skip('reference NewContigsScaffolds itself fails on this scenario here: %r' % (exc,))
            raise
        moved.append(n_small - len(small_scaffolds))
        results.append(_chain_state(G, G_prime, Contigs, small_contigs, Scaffolds, small_scaffolds, param))
    assert moved[0] == moved[1]
    for k in results[0]:
        assert results[0][k] == results[1][k], k
    print('small scaffolds placed inside new scaffolds:', moved[0])


def test_extend_within_components_contract():
    """The order the drop-in gives the path-search hook (besst_amd.MakeScaffolds._extend_within_components): every component
    is searched and its path ends relabelled in G_prime BEFORE any walk, so a later search sees the (name, 'L'/'R') nodes
    of earlier components in G_prime while param.scaffold_indexer still holds its old value - the contract the docstring
    states (the hook must not read either); names count on from the indexer in component order."""
    from besst_amd import MakeScaffolds as OURS
    from besst_amd import nxcompat

    class P(object):
        scaffold_indexer = 40
    G, Gp = nxcompat.Graph(), nxcompat.Graph()
    for base in (1, 2, 3, 4):                                # components {1, 2} and {3, 4}: two scaffolds joined R - L
        for g in (G, Gp):
            g.add_edge((base, 'L'), (base, 'R'), nr_links=None)
    for a, b in ((1, 2), (3, 4)):
        for g in (G, Gp):
            g.add_edge((a, 'R'), (b, 'L'), nr_links=7, obs=70, obs_sq=800, observations=[10] * 7)
    Gp.add_edge((2, 'R'), (3, 'L'), nr_links=5, obs=50, obs_sq=600, observations=[10] * 5)   # a G_prime-only link between them
    seen = []

    def hook(G_, Gp_, Contigs, small_contigs, Scaffolds, small_scaffolds, param, component, table, visited):
        seen.append((sorted(n[0] for n in component if n[1] == 'L'), param.scaffold_indexer,
                     sorted(n for n in Gp_.nodes() if n[0] > 40)))
    OURS._extend_within_components(G, Gp, {}, {}, {}, {}, P(), None, set(), hook)
    assert [s[0] for s in seen] == [[1, 2], [3, 4]]
    assert [s[1] for s in seen] == [40, 40]                   # the indexer advances with the walks, not here
    assert seen[0][2] == [] and seen[1][2] == [(41, 'L'), (41, 'R')]   # the second search sees the first component renamed
    assert sorted(Gp.nodes()) == [(41, 'L'), (41, 'R'), (42, 'L'), (42, 'R')]
    assert Gp[(41, 'R')][(42, 'L')]['nr_links'] == 5          # the link between the two paths' ends travelled with them


def test_unsorted_stream_gets_the_reference_message_as_a_warning(fake_gpu, capsys):
    """libmetrics.py:237-241 refuses a BAM without an index (only a coordinate-sorted file has one); the drop-in checks the
    order of the resident stream instead and warns - on stderr and in `Information` - without refusing."""
    doc, batch = GU.load('fr_given')
    param = make_param(doc['overrides'])
    libmetrics.get_metrics(batch, param, param.information_file)
    session.close_session(batch)
    assert not hasattr(param, 'stream_unsorted_at') and 'Need indexed bamfiles' not in capsys.readouterr().err
    order = numpy.arange(len(batch))
    order[1000], order[5000] = order[5000], order[1000]       # two records change places
    shuffled = batch.take(order)
    assert (shuffled.tid[1000], shuffled.pos[1000]) != (batch.tid[1000], batch.pos[1000])
    param = make_param(doc['overrides'])
    libmetrics.get_metrics(shuffled, param, param.information_file)
    session.close_session(shuffled)
    assert param.stream_unsorted_at == 1001
    assert 'Need indexed bamfiles' in capsys.readouterr().err
    assert 'WARNING: the alignments are not sorted by coordinate (record 1001 ' in param.information_file.getvalue()
