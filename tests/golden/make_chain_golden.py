#!/usr/bin/env python
"""Golden vectors for the chain-extraction step (last quarter): the reference's own
NewContigsScaffolds / UpdateInfo (BESST/MakeScaffolds.py:270-341, 344-482), imported from /root/reference through
tests/refharness, run on the linearised graphs of tests/golden/scaffold_steps.json.gz (their state after step 4).

Every scaffold of a case gets seeded contents - one to three contigs with positions, directions and lengths, scaffold
lengths from 300 bp to 40 kb so that all three gap rules of UpdateInfo are taken (table look-up for long pairs, ML
estimate, naive) - and every link edge seeded link statistics; a few edges carry a precomputed 'avg_gap'.  Stored: the
inputs and, after the call, every contig's (scaffold, position, direction), the new scaffolds (id, contig order, length),
param.scaffold_indexer, param.gap_estimations, G's remaining nodes and - for the extend_paths cases - G_prime's nodes
and link edges.

Two things about the harness, both stated here because they shape the fixture:
  * the graphs are besst_amd.nxcompat.Graph (networkx-1.x semantics: subgraph() is a copy that lists nodes in graph order),
    so "the first end node of a component" is a deterministic choice, not one of set iteration order;
  * with param.extend_paths the reference first runs PROWithinScaf (the path search, out of scope: SURVEY 8(f));
    for those cases it is replaced by a no-op so that the G_prime relabelling of :308-337 is still captured.
mathstats is the shim (tests/refharness/stubs): gaps from GapEstimator / the dValues table pin plumbing only.

    python tests/golden/make_chain_golden.py
"""
import gzip
import io
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from tests import scaffold_util as SU  # noqa: E402
from tests.refharness import driver, loader  # noqa: E402

OUT = os.path.join(HERE, 'scaffold_chains.json.gz')
MEAN, SD, READ_LEN = 2500.0, 250.0, 100


def seeded_inputs(case, seed):
    """Scaffold contents and link statistics for the step-4 state of a linearisation case."""
    rng = random.Random(seed)
    nodes = [tuple(n) for n in case['after_step4_nodes']]
    links = [(tuple(u), tuple(v)) for u, v, _ in case['after_step4_links']]
    scaffolds = {}
    for s, _ in nodes:
        if s in scaffolds:
            continue
        n_ctg = rng.choice((1, 1, 2, 3))
        target = rng.choice((300, 340, 900, 2000, 3400, 3600, 8000, 40000))
        contigs, cur = [], 0
        for c in range(n_ctg):
            length = max(120, target // n_ctg + rng.randint(-40, 40))
            contigs.append(['s%dc%d' % (s, c), cur, rng.random() < 0.5, length])
            cur += length + (rng.randint(1, 60) if c + 1 < n_ctg else 0)
        scaffolds[s] = dict(contigs=contigs, s_length=cur)
    edges = []
    for u, v in links:
        n = rng.choice((2, 3, 4, 5, 6, 9, 17, 40))
        obs = [rng.randint(300, 3800) for _ in range(n)]
        e = dict(u=list(u), v=list(v), nr_links=n, obs=sum(obs), obs_sq=sum(o * o for o in obs), observations=obs)
        if rng.random() < 0.08:
            e['avg_gap'] = rng.choice((-30, 0, 1, 57, 300.5))
        edges.append(e)
    return nodes, scaffolds, edges


def run_reference(ms, mods, nodes, scaffolds, edges, prime_links, extend_paths):
    from besst_amd import nxcompat
    G = nxcompat.Graph()
    for n in nodes:
        G.add_node(n, length=scaffolds[n[0]]['s_length'])
    for s in scaffolds:
        G.add_edge((s, 'L'), (s, 'R'), nr_links=None)
    for e in edges:
        attrs = {k: e[k] for k in ('nr_links', 'obs', 'obs_sq', 'observations', 'avg_gap') if k in e}
        G.add_edge(tuple(e['u']), tuple(e['v']), **attrs)
    G_prime = SU.build_graph([list(n) for n in nodes], prime_links)
    Contigs, Scaffolds = {}, {}
    for s, doc in scaffolds.items():
        objs = []
        for name, pos, direction, length in doc['contigs']:
            c = mods['Contig'].contig(name)
            c.scaffold, c.position, c.direction, c.length, c.sequence = s, pos, direction, length, ''
            Contigs[name] = c
            objs.append(c)
        Scaffolds[s] = mods['Scaffold'].scaffold(s, objs, doc['s_length'])
    param = driver.make_param(mods, extend_paths=extend_paths, plots=False, mean_ins_size=MEAN, std_dev_ins_size=SD,
                              read_len=READ_LEN, lognormal=False, scaffold_indexer=max(scaffolds) + 5 if scaffolds else 5)
    param.gap_estimations = []
    table = ms.GC.PreCalcMLvaluesOfdLongContigs(MEAN, SD, READ_LEN)
    info = io.StringIO()
    ms.NewContigsScaffolds(G, G_prime, Contigs, {}, Scaffolds, {}, info, table, param, set())
    return dict(
        contigs={name: [c.scaffold, c.position, bool(c.direction)] for name, c in Contigs.items()},
        scaffolds=[[s.name, [c.name for c in s.contigs], s.s_length] for s in Scaffolds.values()],
        scaffold_indexer=param.scaffold_indexer, gap_estimations=list(param.gap_estimations),
        nodes_left=[list(n) for n in G.nodes()],
        prime_nodes=[list(n) for n in G_prime.nodes()] if extend_paths else None,
        prime_links=[[list(u), list(v), G_prime[u][v]['nr_links']] for u, v in G_prime.edges()
                     if G_prime[u][v]['nr_links'] is not None] if extend_paths else None,
        info=[l for l in info.getvalue().splitlines() if l.startswith('Nr of new scaffolds')])


def main():
    mods = loader.load()
    import importlib
    ms = importlib.import_module('BESST.MakeScaffolds')
    ms.PROWithinScaf = lambda *a, **k: None               # see the module docstring
    cases = []
    for k, case in enumerate(SU.cases()):
        nodes, scaffolds, edges = seeded_inputs(case, 4242 + k)
        extend = bool(case['extend_paths']) and k % 2 == 0
        prime = [l for l in case['after_step4_prime_links']
                 if tuple(l[0]) in set(nodes) and tuple(l[1]) in set(nodes)] if extend else []
        out = run_reference(ms, mods, nodes, scaffolds, edges, prime, extend)
        cases.append(dict(name=case['name'], extend_paths=extend, nodes=[list(n) for n in nodes],
                          scaffolds={str(s): d for s, d in scaffolds.items()}, edges=edges, prime_links=prime,
                          mean=MEAN, sd=SD, read_len=READ_LEN, expect=out))
    with gzip.GzipFile(OUT, 'wb', mtime=0) as gz, io.TextIOWrapper(gz, encoding='ascii') as fh:
        json.dump(dict(generator='tests/golden/make_chain_golden.py', cases=cases), fh, separators=(',', ':'))
    print('wrote %s: %d cases, %.1f KB' % (OUT, len(cases), os.path.getsize(OUT) / 1024.0))
    for c in cases:
        print('  %-28s scaffolds %5d -> new %4d   gaps appended %4d  extend %s'
              % (c['name'], len(c['scaffolds']), len([s for s in c['expect']['scaffolds'] if s[0] > max(map(int, c['scaffolds'])) if c['scaffolds']]),
                 len(c['expect']['gap_estimations']), c['extend_paths']))


if __name__ == '__main__':
    main()
