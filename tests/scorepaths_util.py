"""Shared by the CPU and GPU tests of ScorePaths: the fixture captured from the reference and its array form."""
import gzip
import json
import os

import numpy as np

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'scorepaths.json.gz')
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


def build_graph(case):
    from besst_amd import nxcompat
    G = nxcompat.Graph()
    for n in case['nodes']:
        G.add_node(tuple(n), length=1000)
    for s in dict.fromkeys(s for s, _ in case['nodes']):
        G.add_edge((s, 'L'), (s, 'R'), nr_links=None)
    for u, v, w in case['links']:
        G.add_edge(tuple(u), tuple(v), nr_links=w, obs=100 * w, obs_sq=10000 * w, observations=[100] * w)
    return G


def to_arrays(case):
    """(n_scaffolds, row_ptr, col, weight, path_ptr, path_nodes) with node (s, side) -> 2 * index[s] + (side == 'R')."""
    index = {}
    for s, _ in case['nodes']:
        index.setdefault(s, len(index))

    def node(n):
        return 2 * index[n[0]] + (n[1] == 'R')
    n_nodes = 2 * len(index)
    adj = [[] for _ in range(n_nodes)]
    for u, v, w in case['links']:
        adj[node(u)].append((node(v), w))
        adj[node(v)].append((node(u), w))
    row_ptr = np.zeros(n_nodes + 1, np.int64)
    col, weight = [], []
    for x in range(n_nodes):
        for y, w in adj[x]:
            col.append(y)
            weight.append(w)
        row_ptr[x + 1] = len(col)
    path_ptr = np.zeros(len(case['paths']) + 1, np.int64)
    path_nodes = []
    for i, p in enumerate(case['paths']):
        path_nodes.extend(node(n) for n in p)
        path_ptr[i + 1] = len(path_nodes)
    return (len(index), row_ptr, np.array(col, np.int32), np.array(weight, np.int32), path_ptr,
            np.array(path_nodes, np.int32))


class Param(object):
    def __init__(self, case):
        self.contamination_ratio = case['contamination_ratio']
        self.no_score = case['no_score']
        self.score_cutoff = case['score_cutoff']
