"""GPU parity of the scaffold-graph linearisation (besst_linearize, besst_amd.MakeScaffolds) against the fixture
captured from the reference's MakeScaffolds functions and, at sizes the fixture cannot hold, against the sequential
CPU restatement."""
import io
import random

import numpy as np
import pytest

from tests import scaffold_util as SU

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', SU.case_names())
def test_device_matches_reference_fixture(name):
    from besst_amd import MakeScaffolds as MS
    case = SU.by_name(name)
    n_scaf, _, a, b, score = SU.to_arrays(case)
    SU.check_result(case, MS.linearize_arrays(n_scaf, a, b, score))


def _counts(text, what):
    return [int(line.split()[0]) for line in text.splitlines() if what in line]


@pytest.mark.parametrize('name', SU.case_names())
@pytest.mark.parametrize('one_call', [False, True])
def test_mirror_functions_leave_the_reference_graphs(name, one_call):
    """The drop-in functions on nx-1.x style graphs: nodes, link edges (in order) of G and G_prime after every step,
    and the counts and 'SCORES AMBVIVALENT' lines written to Information."""
    from besst_amd import MakeScaffolds as MS
    case = SU.by_name(name)
    G = SU.build_graph(case['nodes'], case['links'])
    G_prime = SU.build_graph(case['nodes'], case['prime_links'])
    param = SU.Param(case['extend_paths'])
    info = io.StringIO()
    if one_call:
        G, _, _ = MS.LinearizeGraph(G, G_prime, {}, {}, info, param)
    else:
        G = MS.RemoveIsolatedContigs(G, info)
        assert [list(n) for n in G.nodes()] == case['after_step1_nodes']
        MS.RemoveAmbiguousRegionsUsingScore(G, G_prime, info, param, 'G')
        assert SU.link_rows(G) == case['after_step2_links']
        assert SU.link_rows(G_prime) == case['after_step2_prime_links']
        G = MS.RemoveIsolatedContigs(G, info)
        assert [list(n) for n in G.nodes()] == case['after_step3_nodes']
        G, _, _ = MS.RemoveLoops(G, G_prime, {}, {}, info, param)
    assert [list(n) for n in G.nodes()] == case['after_step4_nodes']
    assert SU.link_rows(G) == case['after_step4_links']
    assert [list(n) for n in G_prime.nodes()] == case['after_step4_prime_nodes']
    assert SU.link_rows(G_prime) == case['after_step4_prime_links']
    text = info.getvalue()
    assert _counts(text, 'isolated contigs removed') == case['isolated_removed']
    assert _counts(text, 'cycles removed') == [case['cycles_removed']]
    amb = [[float(x) for x in line.split()[2:]] for line in text.splitlines() if line.startswith('SCORES AMBVIVALENT')]
    assert amb == case['ambivalent']
    assert text.count('A cycle in the scaffold graph') == 2 * case['cycles_removed']


def _random_arrays(rng, n_scaf, n_edges, tie_pool):
    a = rng.integers(0, 2 * n_scaf, n_edges).astype(np.int32)
    b = rng.integers(0, 2 * n_scaf, n_edges).astype(np.int32)
    keep = (a >> 1) != (b >> 1)
    a, b = a[keep], b[keep]
    # one edge per unordered node pair, like a Graph
    lo, hi = np.minimum(a, b).astype(np.int64), np.maximum(a, b).astype(np.int64)
    _, first = np.unique(lo * (2 * n_scaf) + hi, return_index=True)
    first.sort()
    a, b = a[first], b[first]
    score = np.where(rng.random(a.shape[0]) < 0.5, rng.choice(tie_pool, a.shape[0]),
                     np.round(rng.random(a.shape[0]) * 2.0, 2))
    return a, b, score.astype(np.float64)


@pytest.mark.parametrize('n_scaf,n_edges', [(2000, 30000), (300000, 450000), (1000000, 900000)])
def test_large_random_graphs_match_sequential_oracle(n_scaf, n_edges):
    from besst_amd import MakeScaffolds as MS
    from oracle import scaffold_oracle as SO
    rng = np.random.default_rng(n_scaf)
    a, b, score = _random_arrays(rng, n_scaf, n_edges, np.array([0.0, 0.5, 0.8, 1.0, 1.0, 1.25, 2.0]))
    got = MS.linearize_arrays(n_scaf, a, b, score)
    want = SO.linearize(n_scaf, a.tolist(), b.tolist(), score.tolist())
    assert got['alive2'].tolist() == want['alive2']
    assert got['present'].tolist() == want['present']
    assert got['isolated'] == want['isolated'] and got['cycles'] == want['cycles']
    assert got['ambivalent'] == [(float(t), float(s)) for t, s in want['ambivalent']]
    assert len(want['ambivalent']) > 100


def test_long_dependency_chain_takes_many_rounds():
    """A path of link-adjacent nodes with falling scores: every node has to wait for the one before it, so step 2
    needs as many rounds as the path is long (the loop must not stop at a batch boundary)."""
    from besst_amd import MakeScaffolds as MS
    from oracle import scaffold_oracle as SO
    n = 301
    # node 2k ('L' of scaffold k) carries two link edges: to scaffold k-1 and to scaffold k+1
    a = np.array([2 * k for k in range(n - 1)], np.int32)
    b = np.array([2 * (k + 1) for k in range(n - 1)], np.int32)
    score = np.array([1000.0 - 0.5 * k for k in range(n - 1)])
    got = MS.linearize_arrays(n, a, b, score)
    want = SO.linearize(n, a.tolist(), b.tolist(), score.tolist())
    assert got['alive2'].tolist() == want['alive2']
    assert got['ambivalent'] == [(float(t), float(s)) for t, s in want['ambivalent']]
    assert got['rounds'] >= 100


def test_rings_of_many_lengths_and_long_paths():
    from besst_amd import MakeScaffolds as MS
    from oracle import scaffold_oracle as SO
    rnd = random.Random(5)
    a, b, k = [], [], 0
    rings = 0
    for length in [2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 33, 64, 100, 1000, 4097]:
        for closed in (True, False):
            flip = [rnd.random() < 0.5 for _ in range(length)]
            for j in range(length - 1 + (1 if closed else 0)):
                u, v = k + j, k + (j + 1) % length
                a.append(2 * u + (0 if flip[u - k] else 1))
                b.append(2 * v + (1 if flip[v - k] else 0))
            rings += closed
            k += length
    order = list(range(len(a)))
    rnd.shuffle(order)
    a = np.array(a, np.int32)[order]
    b = np.array(b, np.int32)[order]
    score = np.full(len(order), 1.5)
    got = MS.linearize_arrays(k + 10, a, b, score)
    want = SO.linearize(k + 10, a.tolist(), b.tolist(), score.tolist())
    assert got['cycles'] == rings == want['cycles']
    assert got['present'].tolist() == want['present']
    assert got['isolated'] == [10, 0]
    # RemoveLoops alone on the same graph
    only4 = MS.linearize_arrays(k + 10, a, b, score, MS.STEP4)
    assert only4['cycles'] == rings and only4['removed_by'].tolist() == [4 if not p else 0 for p in want['present'][:k]] + [0] * 10


def test_empty_and_error_cases():
    from besst_amd import MakeScaffolds as MS
    from besst_amd import _lib
    e = np.zeros(0, np.int32)
    res = MS.linearize_arrays(0, e, e, np.zeros(0))
    assert res['isolated'] == [0, 0] and res['cycles'] == 0 and res['alive2'].shape == (0,)
    res = MS.linearize_arrays(3, e, e, np.zeros(0))
    assert res['isolated'] == [3, 0] and not res['present'].any()
    # RemoveLoops on a graph that did not go through step 2
    a = np.array([0, 0], np.int32)
    b = np.array([2, 4], np.int32)
    with pytest.raises(_lib.BesstDeviceError, match='several'):
        MS.linearize_arrays(3, a, b, np.array([1.0, 1.0]), MS.STEP4)
    with pytest.raises(ValueError):
        MS.linearize_arrays(1, a, b, np.array([1.0, 1.0]))
    # NaN and infinite scores: NaN is 'not 0 < score', +inf is a score like any other
    res = MS.linearize_arrays(3, a, b, np.array([float('nan'), float('inf')]))
    assert res['alive2'].tolist() == [False, True]
