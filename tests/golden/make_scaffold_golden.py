#!/usr/bin/env python
"""Capture golden vectors for the scaffold-graph linearisation steps from the reference.

Runs, in this container only, the reference's own
    RemoveIsolatedContigs            (MakeScaffolds.py:134-144)   step 1
    RemoveAmbiguousRegionsUsingScore (MakeScaffolds.py:206-241)   step 2  (+ remove_edges :156-204)
    RemoveIsolatedContigs                                         step 3
    RemoveLoops                      (MakeScaffolds.py:248-274)   step 4
imported from /root/reference through tests/refharness, over seeded scored scaffold graphs, and stores the
inputs (node order, link edges in G.edges() order with their scores, the G_prime edge list) and the outputs
after every step in tests/golden/scaffold_steps.json.gz.  Nothing of the reference is copied: the fixture is
graphs in, graphs out.

    python tests/golden/make_scaffold_golden.py
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

OUT = os.path.join(HERE, 'scaffold_steps.json.gz')
SCORE_POOL = (0.0, 0.0, 0.0, 0.4, 0.5, 0.8, 1.0, 1.0, 1.25, 1.6, 2.0)


def build_graph(nxg, nodes, links):
    return SU.build_graph(nodes, links)


def random_case(rng, n_scaf, mean_deg, kind):
    ids = rng.sample(range(1, 5 * n_scaf + 10), n_scaf)
    nodes = []
    for s in ids:
        sides = ['L', 'R'] if rng.random() < 0.8 else ['R', 'L']
        nodes += [(s, sides[0]), (s, sides[1])]
    links = {}

    def add(u, v, sc):
        if u[0] == v[0]:
            return
        key = frozenset((u, v))
        if key not in links:
            links[key] = (u, v, sc)
    flat = list(nodes)
    if kind in ('chains', 'mixed'):
        # long unambiguous chains and closed rings: what survives step 2 and what step 4 has to find
        order = list(ids)
        rng.shuffle(order)
        pos = 0
        while pos < len(order):
            run = order[pos:pos + rng.randint(1, 12)]
            pos += len(run)
            sides = [rng.choice('LR') for _ in run]
            for a in range(len(run) - 1):
                u = (run[a], 'R' if sides[a] == 'L' else 'L')
                v = (run[a + 1], sides[a + 1])
                add(u, v, rng.choice((1.0, 1.3, 1.7, 2.0)))
            if len(run) >= 2 and rng.random() < 0.3:
                u = (run[-1], 'R' if sides[-1] == 'L' else 'L')
                v = (run[0], sides[0])
                add(u, v, 1.5)
    if kind in ('random', 'mixed'):
        m = int(n_scaf * mean_deg)
        for _ in range(m):
            u, v = rng.choice(flat), rng.choice(flat)
            r = rng.random()
            sc = rng.choice(SCORE_POOL) if r < 0.6 else (round(rng.uniform(0.0, 2.0), 3) if r < 0.97 else -0.25)
            add(u, v, sc)
        hubs = rng.sample(flat, max(1, n_scaf // 40))
        for h in hubs:
            for _ in range(rng.randint(4, 9)):
                add(h, rng.choice(flat), rng.choice(SCORE_POOL))
    links = list(links.values())
    rng.shuffle(links)
    return nodes, links


link_rows = SU.link_rows


def run_reference(ms, mods, nxg, nodes, links, prime_links, extend_paths):
    G = build_graph(nxg, nodes, links)
    G_prime = build_graph(nxg, nodes, prime_links)
    param = driver.make_param(mods, extend_paths=extend_paths, plots=False)
    info = io.StringIO()
    case = dict(nodes=[list(n) for n in nodes], links=link_rows(G), prime_links=link_rows(G_prime),
                extend_paths=extend_paths)
    G = ms.RemoveIsolatedContigs(G, info)
    case['after_step1_nodes'] = [list(n) for n in G.nodes()]
    ms.RemoveAmbiguousRegionsUsingScore(G, G_prime, info, param, 'G')
    case['after_step2_links'] = link_rows(G)
    case['after_step2_prime_links'] = link_rows(G_prime)
    G = ms.RemoveIsolatedContigs(G, info)
    case['after_step3_nodes'] = [list(n) for n in G.nodes()]
    G, _, _ = ms.RemoveLoops(G, G_prime, {}, {}, info, param)
    case['after_step4_nodes'] = [list(n) for n in G.nodes()]
    case['after_step4_links'] = link_rows(G)
    case['after_step4_prime_nodes'] = [list(n) for n in G_prime.nodes()]
    case['after_step4_prime_links'] = link_rows(G_prime)
    text = info.getvalue()
    counts = [int(line.split()[0]) for line in text.splitlines() if 'isolated contigs removed' in line]
    case['isolated_removed'] = counts
    case['cycles_removed'] = [int(line.split()[0]) for line in text.splitlines() if 'cycles removed' in line][0]
    amb = [line.split()[2:] for line in text.splitlines() if line.startswith('SCORES AMBVIVALENT')]
    case['ambivalent'] = [[float(a), float(b)] for a, b in amb]
    return case


def main():
    mods = loader.load()
    import importlib
    ms = importlib.import_module('BESST.MakeScaffolds')
    from besst_amd import nxcompat
    cases = []
    spec = [(12, 1.0, 'random'), (30, 1.5, 'mixed'), (30, 0.6, 'chains'), (80, 2.0, 'random'), (80, 1.0, 'mixed'),
            (200, 1.2, 'mixed'), (200, 3.0, 'random'), (500, 1.0, 'mixed'), (500, 0.4, 'chains'), (1500, 1.5, 'mixed'),
            (1500, 2.5, 'random'), (3000, 1.2, 'mixed')]
    for k, (n_scaf, deg, kind) in enumerate(spec):
        rng = random.Random(20240929 + k)
        nodes, links = random_case(rng, n_scaf, deg, kind)
        # G_prime: most of G's link edges, some missing, some extra (edges only G_prime holds)
        prime = [l for l in links if rng.random() < 0.85]
        _, extra = random_case(rng, n_scaf, 0.3, 'random')
        have = set(frozenset((tuple(u), tuple(v))) for u, v, _ in prime)
        ids = set(s for s, _ in nodes)
        for u, v, sc in extra:
            if u[0] in ids and v[0] in ids and frozenset((u, v)) not in have:
                prime.append((u, v, sc))
        case = run_reference(ms, mods, nxcompat.Graph, nodes, links, prime, extend_paths=(k % 3 != 1))
        case['name'] = 'random_%02d_%s_%d' % (k, kind, n_scaf)
        cases.append(case)
    # hand-made corner cases: exact 0.8 ratio, equal tops, a node visited first as edge[1], negative scores
    hand = [
        ('ratio_exactly_0.8', [(1, 'L'), (1, 'R'), (2, 'L'), (2, 'R'), (3, 'L'), (3, 'R')],
         [((1, 'R'), (2, 'L'), 1.0), ((1, 'R'), (3, 'L'), 0.8)]),
        ('ratio_above_0.8', [(1, 'L'), (1, 'R'), (2, 'L'), (2, 'R'), (3, 'L'), (3, 'R')],
         [((1, 'R'), (2, 'L'), 1.0), ((1, 'R'), (3, 'L'), 0.81)]),
        ('equal_tops', [(1, 'L'), (1, 'R'), (2, 'L'), (2, 'R'), (3, 'L'), (3, 'R'), (4, 'L'), (4, 'R')],
         [((1, 'R'), (2, 'L'), 1.0), ((1, 'R'), (3, 'L'), 1.0), ((1, 'R'), (4, 'L'), 0.3)]),
        ('order_dependence', [(1, 'L'), (1, 'R'), (2, 'L'), (2, 'R'), (3, 'L'), (3, 'R'), (4, 'L'), (4, 'R')],
         # node (2,L) would be ambivalent (0.9 vs 0.85) unless (3,R) - visited earlier through its 2.0 edge - has
         # already dropped the 0.85 edge
         [((1, 'R'), (2, 'L'), 0.9), ((3, 'R'), (2, 'L'), 0.85), ((3, 'R'), (4, 'L'), 2.0)]),
        ('negative_and_zero', [(1, 'L'), (1, 'R'), (2, 'L'), (2, 'R'), (3, 'L'), (3, 'R')],
         [((1, 'R'), (2, 'L'), -0.5), ((1, 'R'), (3, 'L'), 0.0), ((2, 'R'), (3, 'R'), 0.7)]),
        ('ring_of_three', [(1, 'L'), (1, 'R'), (2, 'L'), (2, 'R'), (3, 'L'), (3, 'R'), (4, 'L'), (4, 'R')],
         [((1, 'R'), (2, 'L'), 1.0), ((2, 'R'), (3, 'L'), 1.1), ((3, 'R'), (1, 'L'), 1.2)]),
        ('no_links', [(5, 'L'), (5, 'R'), (9, 'R'), (9, 'L')], []),
    ]
    for name, nodes, links in hand:
        case = run_reference(ms, mods, nxcompat.Graph, nodes, links, list(links), extend_paths=True)
        case['name'] = name
        cases.append(case)
    with gzip.GzipFile(OUT, 'wb', mtime=0) as gz, io.TextIOWrapper(gz, encoding='ascii') as fh:
        json.dump(dict(generator='tests/golden/make_scaffold_golden.py', cases=cases), fh, separators=(',', ':'))
    print('wrote %s: %d cases, %d link edges in total, %.1f KB'
          % (OUT, len(cases), sum(len(c['links']) for c in cases), os.path.getsize(OUT) / 1024.0))
    for c in cases:
        print('  %-28s links %5d -> %5d -> %5d   isolated %s cycles %d ambivalent %d'
              % (c['name'], len(c['links']), len(c['after_step2_links']), len(c['after_step4_links']),
                 c['isolated_removed'], c['cycles_removed'], len(c['ambivalent'])))


if __name__ == '__main__':
    main()
