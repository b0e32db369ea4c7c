#!/usr/bin/env python
"""Capture golden vectors for ScorePaths from the reference.

Runs, in this container only, the reference's ScorePaths (ExtendLargeScaffolds.py:29-130) - imported from
/root/reference through tests/refharness - on seeded link graphs, over paths found by the reference's own
depth-first path search (ExtendLargeScaffolds.py:526) plus random alternating walks, with and without
``param.contamination_ratio`` and ``param.no_score``.  The fixture tests/golden/scorepaths.json.gz holds graphs and
paths in, ``all_paths`` out; nothing of the reference is copied.

    python tests/golden/make_scorepaths_golden.py
"""
import gzip
import importlib
import io
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from tests.refharness import driver, loader  # noqa: E402

OUT = os.path.join(HERE, 'scorepaths.json.gz')


def build_graph(nodes, links):
    from besst_amd import nxcompat
    G = nxcompat.Graph()
    for n in nodes:
        G.add_node(tuple(n), length=1000)
    for s in dict.fromkeys(s for s, _ in nodes):
        G.add_edge((s, 'L'), (s, 'R'), nr_links=None)
    for u, v, w in links:
        G.add_edge(tuple(u), tuple(v), nr_links=w, obs=100 * w, obs_sq=10000 * w, observations=[100] * w)
    return G


def random_graph(rng, n_scaf, mean_deg):
    ids = rng.sample(range(1, 4 * n_scaf + 10), n_scaf)
    nodes = []
    for s in ids:
        nodes += [(s, 'L'), (s, 'R')]
    links = {}
    order = list(ids)
    rng.shuffle(order)
    # a backbone of true adjacencies with many links, plus weaker spurious links
    for a, b in zip(order, order[1:]):
        if rng.random() < 0.85:
            links[frozenset(((a, 'R'), (b, 'L')))] = ((a, 'R'), (b, 'L'), rng.randint(5, 60))
    for a, c in zip(order, order[2:]):
        if rng.random() < 0.4:
            links.setdefault(frozenset(((a, 'R'), (c, 'L'))), ((a, 'R'), (c, 'L'), rng.randint(2, 25)))
    for _ in range(int(n_scaf * mean_deg)):
        u, v = rng.choice(nodes), rng.choice(nodes)
        if u[0] != v[0]:
            links.setdefault(frozenset((u, v)), (u, v, rng.randint(1, 12)))
    out = list(links.values())
    rng.shuffle(out)
    return nodes, out, order


def random_walk(rng, G, start, max_len):
    """Alternating walk: link edge, cross the scaffold, link edge, ... without revisiting a scaffold."""
    path, seen = [start], {start[0]}
    node = start
    while len(path) < max_len:
        nbrs = [n for n in G.neighbors(node) if n[0] != node[0] and n[0] not in seen]
        if not nbrs:
            break
        nxt = rng.choice(nbrs)
        path.append(nxt)
        seen.add(nxt[0])
        if rng.random() < 0.25 or len(path) >= max_len:
            break
        node = (nxt[0], 'R' if nxt[1] == 'L' else 'L')
        path.append(node)
    return path


def main():
    mods = loader.load()
    els = importlib.import_module('BESST.ExtendLargeScaffolds')
    cases = []
    spec = [(15, 0.5), (40, 1.0), (120, 0.8), (300, 1.5), (600, 1.0)]
    for k, (n_scaf, deg) in enumerate(spec):
        rng = random.Random(4242 + k)
        nodes, links, order = random_graph(rng, n_scaf, deg)
        G = build_graph(nodes, links)
        flat = list(G.nodes())
        paths = []
        # paths of the reference's own search from a few start nodes
        param = driver.make_param(mods, path_threshold=2000, hit_path_threshold=False)
        ends = set(rng.sample(flat, max(2, len(flat) // 3)))
        for start in rng.sample(flat, min(12, len(flat))):
            end = set(ends)
            end.discard(start)
            found = els.find_all_paths_for_start_node_DFS_dynamic_programming_ish(G, start, end, set(), 0, 2 ** 32, param)
            paths.extend(found[:40])
        n_dfs = len(paths)
        for _ in range(60 + n_scaf // 2):
            p = random_walk(rng, G, rng.choice(flat), rng.choice((2, 3, 4, 6, 10, 30, 90)))
            if len(p) >= 2:
                paths.append(p)
        if k == 2:
            # long paths (more than one 64-lane chunk) along the backbone: 100, 65 and 33 scaffolds
            for n_long, first in ((100, 0), (65, 10), (33, 50)):
                path = []
                for sc in order[first:first + n_long]:
                    path += [(sc, 'L'), (sc, 'R')]
                paths.append(path)
                paths.append(path[1:-1])
        # a few degenerate ones: a single node, a repeated node, a path through both ends of one scaffold
        paths.append([flat[0]])
        paths.append([flat[0], flat[2], flat[0], flat[2]])
        paths.append([flat[0], flat[1], flat[2], flat[3]])
        for contamination_ratio, no_score, cutoff in ((False, False, 1.5), (0.2, False, 1.5), (False, True, 0.0),
                                                      (0.3, True, 3.0)):
            param = driver.make_param(mods, contamination_ratio=contamination_ratio, no_score=no_score,
                                      score_cutoff=cutoff)
            all_paths = []
            els.ScorePaths(G, [list(p) for p in paths], all_paths, param)
            index = {}
            for i, p in enumerate(paths):
                index.setdefault(tuple(p), []).append(i)
            got = []
            taken = {}
            for score, bad, path, length in all_paths:
                key = tuple(path)
                j = taken.get(key, 0)
                taken[key] = j + 1
                got.append([score, bad, index[key][j], length])
            cases.append(dict(name='graph%d_c%s_ns%d' % (k, contamination_ratio, no_score),
                              nodes=[list(n) for n in nodes], links=[[list(u), list(v), w] for u, v, w in links],
                              paths=[[list(n) for n in p] for p in paths], n_dfs_paths=n_dfs,
                              contamination_ratio=contamination_ratio, no_score=no_score, score_cutoff=cutoff,
                              all_paths=got))
    with gzip.GzipFile(OUT, 'wb', mtime=0) as gz, io.TextIOWrapper(gz, encoding='ascii') as fh:
        json.dump(dict(generator='tests/golden/make_scorepaths_golden.py', cases=cases), fh, separators=(',', ':'))
    print('wrote %s (%.1f KB)' % (OUT, os.path.getsize(OUT) / 1024.0))
    for c in cases:
        print('  %-24s paths %4d (search %3d)  kept %4d  longest %d' % (
            c['name'], len(c['paths']), c['n_dfs_paths'], len(c['all_paths']), max(len(p) for p in c['paths'])))


if __name__ == '__main__':
    main()
