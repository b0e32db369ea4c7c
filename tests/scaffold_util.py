"""Shared by the CPU and GPU tests of the scaffold-graph linearisation steps: load the fixture captured from the
reference (tests/golden/scaffold_steps.json.gz) and turn a case into the array interface of oracle/scaffold_oracle.py."""
import gzip
import json
import os

import numpy as np

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'scaffold_steps.json.gz')
_cases = None


def cases():
    global _cases
    if _cases is None:
        with gzip.open(FIXTURE, 'rt') as fh:
            _cases = json.load(fh)['cases']
    return _cases


def case_names():
    return [c['name'] for c in cases()]


def by_name(name):
    return next(c for c in cases() if c['name'] == name)


def to_arrays(case):
    """(n_scaf, index {scaffold id: k}, a, b, score): node (s, side) -> 2 * index[s] + (side == 'R')."""
    index = {}
    for s, _ in case['nodes']:
        index.setdefault(s, len(index))

    def node(n):
        return 2 * index[n[0]] + (n[1] == 'R')
    a = np.array([node(u) for u, _, _ in case['links']], np.int32)
    b = np.array([node(v) for _, v, _ in case['links']], np.int32)
    score = np.array([sc for _, _, sc in case['links']], np.float64)
    return len(index), index, a, b, score


def check_result(case, res):
    """res: dict(alive2, present, isolated, cycles, ambivalent) as oracle/scaffold_oracle.linearize returns."""
    n_scaf, index, a, b, score = to_arrays(case)
    alive = [bool(x) for x in res['alive2']]
    present = [bool(x) for x in res['present']]
    assert [l for l, k in zip(case['links'], alive) if k] == case['after_step2_links']
    assert [n for n in case['nodes'] if present[index[n[0]]]] == case['after_step4_nodes']
    keep = [l for l, k in zip(case['links'], alive) if k and present[index[l[0][0]]] and present[index[l[1][0]]]]
    assert keep == case['after_step4_links']
    assert [int(x) for x in res['isolated']] == case['isolated_removed']
    assert int(res['cycles']) == case['cycles_removed']
    assert [[float(t), float(s)] for t, s in res['ambivalent']] == case['ambivalent']


def build_graph(nodes, links):
    """The facade graph of a case: nodes [(scaf, side)] in insertion order, the intra-scaffold edges, then the link
    edges [(u, v, score)] in listing order (which reproduces the listing as G.edges() order)."""
    from besst_amd import nxcompat
    G = nxcompat.Graph()
    for n in nodes:
        G.add_node(tuple(n), length=1000)
    seen = set()
    for s, _ in nodes:
        if s not in seen:
            seen.add(s)
            G.add_edge((s, 'L'), (s, 'R'), nr_links=None)
    for u, v, sc in links:
        G.add_edge(tuple(u), tuple(v), nr_links=7, obs=700, obs_sq=70000, observations=[100] * 7, gap=0, score=sc)
    return G


def link_rows(G):
    out = []
    for u, v in G.edges():
        d = G[u][v]
        if d['nr_links'] is None:
            continue
        out.append([list(u), list(v), d.get('score')])
    return out


class Param(object):
    def __init__(self, extend_paths):
        self.extend_paths = extend_paths
        self.plots = False
