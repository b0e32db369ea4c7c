"""The Python oracle must reproduce every golden vector captured from the real reference."""
import pytest

from oracle import py_oracle as O
from tests import golden_util as GU


def oracle_state_from_layout(doc, batch, threshold):
    lay = doc['layout']
    st = O.State()
    by_scaf = {}
    for tid, name in enumerate(batch.references):
        by_scaf.setdefault(lay['scaf_id'][tid], []).append(tid)
    for sid, tids in by_scaf.items():
        s_len = lay['scaf_len'][tids[0]]
        big = s_len >= threshold
        (st.scaffolds if big else st.small_scaffolds)[sid] = dict(
            contigs=[batch.references[t] for t in tids], s_length=s_len)
        for t in tids:
            (st.contigs if big else st.small_contigs)[batch.references[t]] = dict(
                scaffold=sid, direction=bool(lay['direction'][t]), position=lay['position'][t],
                length=batch.lengths[t], coverage=None)
    return st


def run_oracle(doc, batch):
    p = O.LibParams(**doc['overrides'])
    rec = GU.rec_lists(batch)
    O.get_metrics(rec, batch.lengths, p)
    if doc['layout'] is not None:
        st = oracle_state_from_layout(doc, batch, doc['layout_threshold'])
        p.scaffold_indexer = doc['layout']['next_scaffold_id']
        p.tot_assembly_length = sum(batch.lengths)
    else:
        st = O.State()
    fasta = {n: l for n, l in zip(batch.references, batch.lengths) if n in set(doc['fasta_names'])}
    out = O.create_graph(rec, batch.references, batch.lengths, fasta, st, p)
    return p, st, out


def edge_rows(g, keys=('gap', 'score')):
    rows = []
    for u, v in g.edges():
        d = g[u][v]
        if d['nr_links'] is None:
            continue
        row = dict(u=list(u), v=list(v), nr_links=d['nr_links'], obs=d['obs'], obs_sq=d['obs_sq'])
        for k in keys:
            if k in d:
                row[k] = d[k]
        rows.append(row)
    return rows


@pytest.mark.parametrize('name', GU.scenario_names())
def test_oracle_matches_reference(name):
    doc, batch = GU.load(name)
    p, st, out = run_oracle(doc, batch)
    # library metrics (libmetrics.get_metrics) - bit exact
    for k, want in doc['metrics'].items():
        if k == 'empirical_distribution':
            got = None if p.empirical_distribution is None else \
                [p.empirical_distribution[i] for i in range(len(p.empirical_distribution))]
        else:
            got = getattr(p, k)
        assert got == want, (name, k, got, want)
    snap = doc['after_loop']
    loop = out['loop']
    # counters of the record loop
    c = snap['counter']
    assert (loop.count, loop.non_unique, loop.non_unique_for_scaf, loop.nr_of_duplicates, loop.too_long) == \
        (c['count'], c['non_unique'], c['non_unique_for_scaf'], c['nr_of_duplicates'], c['reads_with_too_long_insert'])
    assert loop.prev == (c['prev_obs1'], c['prev_obs2'])
    assert loop.fishy_reads == snap['fishy_reads']
    # coverage numerators
    for tid, cname in enumerate(batch.references):
        if cname in snap['aligned']:
            assert loop.aligned[tid] == snap['aligned'][cname]
    # fishy table
    want_fishy = {}
    for a, b, n in snap['fishy']:
        want_fishy[(tuple(a), tuple(b))] = n
    for r in loop.edges.values():
        if r.fishy:
            u, v = O.node_of(r.u), O.node_of(r.v)
            assert want_fishy[(u, v)] == r.fishy and want_fishy[(v, u)] == r.fishy
    assert sum(1 for r in loop.edges.values() if r.fishy) * 2 == len(want_fishy)
    # raw edge tables right after the loop: per-edge counts, sums, observation order, per-end lists
    for gname, bit in (('G', O.MASK_G), ('G_prime', O.MASK_GP)):
        want = {frozenset((tuple(e['u']), tuple(e['v']))): e for e in snap[gname]}
        got = {frozenset((O.node_of(r.u), O.node_of(r.v))): r for r in loop.edges.values() if r.n and r.mask & bit}
        assert set(want) == set(got), (name, gname)
        for key, e in want.items():
            r = got[key]
            assert (r.n, r.obs, r.obs_sq, r.observations) == (e['nr_links'], e['obs'], e['obs_sq'], e['observations'])
            if 'l_u' in e:
                if tuple(e['u']) == O.node_of(r.u):
                    assert (r.obs_u, r.obs_v) == (e['l_u'], e['l_v'])
                else:
                    assert (r.obs_u, r.obs_v) == (e['l_v'], e['l_u'])
    # final graphs: same edges in the same iteration order with identical attributes
    fin = doc['final']
    GU.assert_scored_rows(edge_rows(out['G']), fin['G'], doc, name)
    GU.assert_scored_rows(edge_rows(out['Gp']), fin['G_prime'], doc, name)
    assert [list(n) for n in out['G'].nodes()] == fin['G_nodes']
    assert [list(n) for n in out['Gp'].nodes()] == fin['G_prime_nodes']
    assert [[n, c['scaffold'], c['coverage']] for n, c in st.contigs.items()] == fin['contigs']
    assert [[n, c['scaffold'], c['coverage']] for n, c in st.small_contigs.items()] == fin['small_contigs']
    assert list(st.scaffolds) == fin['scaffolds'] and list(st.small_scaffolds) == fin['small_scaffolds']
    for k in ('mean_coverage', 'std_dev_coverage', 'edgesupport', 'expected_links_over_mean_plus_stddev',
              'scaffold_indexer', 'tot_assembly_length', 'current_N50', 'current_L50'):
        assert getattr(p, k) == fin['param'][k], (name, k)
