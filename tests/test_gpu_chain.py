"""GPU: NewContigsScaffolds through the device (besst_chain_scaffolds + besst_amd.MakeScaffolds.NewContigsScaffolds) against
the reference's own results on the 19 fixture graphs, and against the sequential oracle on long random paths."""
import io

import numpy as np
import pytest

from besst_amd import Contig, MakeScaffolds as MS, Scaffold, mathstats_compat as GC, nxcompat
from oracle import scaffold_oracle as SO
from tests import chain_util as CU
from tests import scaffold_util as SU

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', CU.case_names())
def test_device_chain_extraction_matches_reference(name):
    case = CU.by_name(name)
    G = nxcompat.Graph()
    for n in case['nodes']:
        G.add_node(tuple(n), length=case['scaffolds'][str(n[0])]['s_length'])
    for s in case['scaffolds']:
        G.add_edge((int(s), 'L'), (int(s), 'R'), nr_links=None)
    for e in case['edges']:
        attrs = {k: e[k] for k in ('nr_links', 'obs', 'obs_sq', 'observations', 'avg_gap') if k in e}
        G.add_edge(tuple(e['u']), tuple(e['v']), **attrs)
    G_prime = SU.build_graph(case['nodes'], case['prime_links'])
    Contigs, Scaffolds = {}, {}
    for s, doc in case['scaffolds'].items():
        objs = []
        for cname, pos, direction, length in doc['contigs']:
            c = Contig.contig(cname, contig_scaffold=int(s), contig_direction=direction, contig_position=pos,
                              contig_length=length, contig_sequence='')
            Contigs[cname] = c
            objs.append(c)
        Scaffolds[int(s)] = Scaffold.scaffold(int(s), objs, doc['s_length'])
    param = CU.Param(case)
    table = GC.PreCalcMLvaluesOfdLongContigs(case['mean'], case['sd'], case['read_len'])
    info = io.StringIO()
    MS.NewContigsScaffolds(G, G_prime, Contigs, {}, Scaffolds, {}, info, table, param, set())
    exp = case['expect']
    assert {k: [c.scaffold, c.position, bool(c.direction)] for k, c in Contigs.items()} == exp['contigs']
    assert [[s.name, [c.name for c in s.contigs], s.s_length] for s in Scaffolds.values()] == exp['scaffolds']
    assert param.scaffold_indexer == exp['scaffold_indexer']
    assert param.gap_estimations == exp['gap_estimations']
    assert [list(n) for n in G.nodes()] == exp['nodes_left']
    assert [l for l in info.getvalue().splitlines() if l.startswith('Nr of new scaffolds')] == exp['info']
    if case['extend_paths']:
        assert [list(n) for n in G_prime.nodes()] == exp['prime_nodes']
        got = [[list(u), list(v), G_prime[u][v]['nr_links']] for u, v in G_prime.edges()
               if G_prime[u][v]['nr_links'] is not None]
        assert got == exp['prime_links']


@pytest.mark.parametrize('n_scaf,max_run', [(1000, 8), (200_000, 40), (1_000_000, 300_000)])
def test_device_list_ranking_vs_sequential_walk(n_scaf, max_run):
    """Paths of up to 300 000 scaffolds (19 doubling passes): positions, start ends and component order from the device
    equal the sequential walk's."""
    rng = np.random.default_rng(n_scaf)
    perm = rng.permutation(n_scaf)
    link = np.full(2 * n_scaf, -1, np.int32)
    gap = np.zeros(2 * n_scaf, np.int32)
    i = 0
    while i < n_scaf:
        run = int(rng.integers(1, max_run + 1))
        path = perm[i:i + run]
        sides = rng.integers(0, 2, len(path))
        for a in range(len(path) - 1):
            u = 2 * int(path[a]) + int(1 - sides[a])
            v = 2 * int(path[a + 1]) + int(sides[a + 1])
            link[u], link[v] = v, u
            gap[u] = gap[v] = int(rng.integers(1, 500))
        i += run
    slen = rng.integers(200, 20000, n_scaf).astype(np.int64)
    order_nodes = rng.permutation(2 * n_scaf)               # G.nodes() order
    order = np.empty(2 * n_scaf, np.int32)
    order[order_nodes] = np.arange(2 * n_scaf)
    terminal, beyond, lowest, passes = MS.chain_arrays(n_scaf, link, gap, slen, order)
    assert passes >= int(np.ceil(np.log2(max(2, min(max_run, n_scaf))))) - 1
    contigs = [[['c%d' % k, 0, True, int(slen[k])]] for k in range(n_scaf)]
    new, _, _ = SO.new_contigs_scaffolds(order_nodes.tolist(), link.tolist(), gap.tolist(), [None] * (2 * n_scaf),
                                         slen.tolist(), contigs, 0)
    k = np.arange(n_scaf)
    from_l = order[terminal[2 * k]] < order[terminal[2 * k + 1]]
    pos = np.where(from_l, beyond[2 * k], beyond[2 * k + 1])
    comp = np.minimum(np.minimum(lowest[2 * k], lowest[2 * k + 1]), np.minimum(order[2 * k], order[2 * k + 1]))
    comp_rank = np.unique(comp, return_inverse=True)[1]
    assert len(new) == comp_rank.max() + 1
    for sid, cl, _ in new:
        for cname, p, direction, _ in cl:
            j = int(cname[1:])
            assert comp_rank[j] == sid - 1 and pos[j] == p and bool(from_l[j]) == direction, (j, sid)
