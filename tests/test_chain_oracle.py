"""CPU: the sequential restatement of NewContigsScaffolds / UpdateInfo (oracle/scaffold_oracle.new_contigs_scaffolds)
against the state the reference itself left behind on the 19 fixture graphs (tests/golden/scaffold_chains.json.gz)."""
import pytest

from besst_amd import mathstats_compat as GC
from besst_amd.MakeScaffolds import _edge_gap
from oracle import scaffold_oracle as SO
from tests import chain_util as CU


def oracle_inputs(case):
    index = {}
    for s, _ in case['nodes']:
        index.setdefault(s, len(index))
    code = lambda n: 2 * index[n[0]] + (n[1] == 'R')
    n = len(index)
    link, gap, appended = [-1] * (2 * n), [0] * (2 * n), [None] * (2 * n)
    slen = [0] * n
    contigs = [None] * n
    for s, k in index.items():
        doc = case['scaffolds'][str(s)]
        slen[k] = doc['s_length']
        contigs[k] = [list(c) for c in doc['contigs']]
    param = CU.Param(case)
    table = GC.PreCalcMLvaluesOfdLongContigs(case['mean'], case['sd'], case['read_len'])
    for e in case['edges']:
        u, v = code(e['u']), code(e['v'])
        g, app = _edge_gap(e, slen[u >> 1], slen[v >> 1], table, param)
        g = 1 if g <= 1 else g
        link[u], link[v] = v, u
        gap[u] = gap[v] = int(g)
        appended[u] = appended[v] = app
    return index, [code(nd) for nd in case['nodes']], link, gap, appended, slen, contigs, param


@pytest.mark.parametrize('name', CU.case_names())
def test_chain_oracle_matches_reference(name):
    case = CU.by_name(name)
    index, order, link, gap, appended, slen, contigs, param = oracle_inputs(case)
    new, estimations, indexer = SO.new_contigs_scaffolds(order, link, gap, appended, slen, contigs, param.scaffold_indexer)
    exp = case['expect']
    assert indexer == exp['scaffold_indexer']
    assert estimations == exp['gap_estimations']
    old = set(int(s) for s in case['scaffolds'])
    want_new = [s for s in exp['scaffolds'] if s[0] not in old]
    assert [[sid, [c[0] for c in cl], length] for sid, cl, length in new] == want_new
    for sid, cl, _ in new:
        for cname, pos, direction, _ in cl:
            assert exp['contigs'][cname] == [sid, pos, direction], cname
