"""ScorePaths of the reference's path extension on the MI355X (BESST/ExtendLargeScaffolds.py:29-130).

    ScorePaths(G, paths, all_paths, param)      same signature and effect as the reference: appends
                                                [score, bad_link_weight, path, len(path)] for every path that passes
                                                the cut-off (:118-128), in the order of ``paths``
    PathScorer(G)                               the link graph uploaded once as CSR, for many batches of paths

The connectivity weights (integer sums of nr_links, calculate_connectivity / calculate_connectivity_contamination,
:33-105) are computed by ``besst_score_paths`` (besst_amd/csrc/scorepaths.hip), one wavefront per path; the score
- a float division of the two sums - is formed here exactly as the reference forms it.  The path search itself
(BFS/DFS with a heap, :139-660) is sequential host work and stays with BESST.  No CPU path: without the library or
a GPU the calls raise ``besst_amd._lib.BesstDeviceError``.
"""
import numpy as np

from . import _lib


class PathScorer(object):
    """CSR of the link edges of ``G`` (nodes (scaffold, 'L'|'R'), link edges = nr_links is not None)."""

    def __init__(self, G, device=0):
        self.device = device
        self.index = {}
        for s, _ in G.nodes():
            self.index.setdefault(s, len(self.index))
        n_nodes = 2 * len(self.index)
        src, dst, w = [], [], []
        for u, v, d in G.edges(data=True):
            if d['nr_links'] is None:
                continue
            a, b = self.node(u), self.node(v)
            src += [a, b]
            dst += [b, a]
            w += [d['nr_links'], d['nr_links']]
        src = np.asarray(src, np.int64)
        order = np.argsort(src, kind='stable')
        self.col = np.ascontiguousarray(np.asarray(dst, np.int32)[order])
        self.weight = np.ascontiguousarray(np.asarray(w, np.int32)[order])
        self.row_ptr = np.zeros(n_nodes + 1, np.int64)
        np.cumsum(np.bincount(src, minlength=n_nodes), out=self.row_ptr[1:])
        self.n_nodes = n_nodes

    def node(self, n):
        return 2 * self.index[n[0]] + (n[1] == 'R')

    def weights(self, paths, contamination):
        """(good, bad) int64 arrays, one entry per path; good is the raw sum (not yet halved for contamination)."""
        lib = _lib.load()
        n = len(paths)
        path_ptr = np.zeros(n + 1, np.int64)
        np.cumsum([len(p) for p in paths], out=path_ptr[1:])
        nodes = np.fromiter((self.node(x) for p in paths for x in p), np.int32, count=int(path_ptr[-1]))
        good = np.zeros(max(n, 1), np.int64)
        bad = np.zeros(max(n, 1), np.int64)
        _lib.check(lib.besst_score_paths(self.device, self.n_nodes, _lib.ptr(self.row_ptr), _lib.ptr(self.col),
                                         _lib.ptr(self.weight), n, _lib.ptr(path_ptr), _lib.ptr(nodes),
                                         1 if contamination else 0, _lib.ptr(good), _lib.ptr(bad)),
                   'besst_score_paths')
        return good[:n], bad[:n]

    def score(self, paths, all_paths, param):
        if len(paths) == 0:
            return ()
        contamination = bool(param.contamination_ratio)
        good, bad = self.weights(paths, contamination)
        for path, g, b in zip(paths, good.tolist(), bad.tolist()):
            if contamination:
                g = g / 2                                    # :94 (true division: the reference runs on Python 3)
            try:
                score = g / float(b)
            except ZeroDivisionError:
                score = g
            if param.no_score and score >= param.score_cutoff:
                all_paths.append([score, b, path, len(path)])
            elif len(path) > 2 and score >= param.score_cutoff:
                all_paths.append([score, b, path, len(path)])
        return ()


def ScorePaths(G, paths, all_paths, param, device=0):
    if len(paths) == 0:
        return ()
    return PathScorer(G, device).score(paths, all_paths, param)
