"""Oracle-backed stand-in for ``besst_amd.device.GraphContext`` (test infrastructure only).

Lets the CPU suite run the product's HOST code (besst_amd.libmetrics / besst_amd.CreateGraph: graph assembly in
first-occurrence order, coverage statistics, order-dependent filters, score assembly) without a GPU, with the
device stages answered by the oracle.  It is injected by monkeypatching ``besst_amd.session.device.GraphContext``
inside tests; nothing in besst_amd/ refers to it.
"""
import numpy as np

from besst_amd import device
from besst_amd._lib import Counters, MetricsCounts
from oracle import c_oracle as CO
from oracle import py_oracle as O


class FakeGraphContext(object):
    def __init__(self, device_index=0):
        self.batches = []
        self.n_contigs = 0

    def close(self):
        pass

    def set_contigs(self, scaf_id, scaf_len, ctg_pos, ctg_len, direction, cls):
        self.table = dict(scaf_id=np.asarray(scaf_id), scaf_len=np.asarray(scaf_len), ctg_pos=np.asarray(ctg_pos),
                          ctg_len=np.asarray(ctg_len), direction=np.asarray(direction), cls=np.asarray(cls))
        self.n_contigs = len(self.table['cls'])

    def set_library(self, read_len, ins_size_threshold, min_mapq, orientation, detect_duplicate, extend_paths, no_score):
        self.lib = dict(read_len=read_len, ins_size_threshold=ins_size_threshold, min_mapq=min_mapq,
                        orientation=orientation, detect_duplicate=detect_duplicate, extend_paths=extend_paths,
                        no_score=no_score)

    def clear_records(self):
        self.batches = []

    def push_records(self, batch):
        self.batches.append(batch)

    def _batch(self):
        assert len(self.batches) == 1
        return self.batches[0]

    def stream_order(self):
        return stream_order_of(self._batch())

    def metrics_sample(self, top_mask, orientation, min_mapq, read_len, want_isize=True):
        isize, contam, c = CO.metrics_sample(self._batch(), top_mask, orientation, min_mapq, read_len, want_isize)
        counts = MetricsCounts(int(c[0]), int(c[1]), int(c[2]), int(c[3]), len(self._batch()))
        return isize, contam, counts

    def build_graph(self, lazy_observations=False):
        ids = self.table['scaf_id'][self.table['cls'] != 0]
        nb = max(1, int((int(ids.max()) if len(ids) else 1) * 2 + 1).bit_length())
        keys, payload, aligned, c = CO.record_loop(self._batch(), self.table, self.lib, nb)
        rows = CO.edge_rows(keys, payload)
        self.rows = rows
        table = device.EdgeTable(rows['key'].astype(np.uint64), rows['mask'].astype(np.uint32),
                                 rows['n'].astype(np.uint32), rows['sum_obs'].astype(np.int64),
                                 rows['sum_obs_sq'].astype(np.int64), rows['first_idx'].astype(np.uint32),
                                 rows['offset'].astype(np.uint32), nb, rows['obs_lo'].astype(np.int32),
                                 rows['obs_hi'].astype(np.int32))
        ctr = Counters(*[int(x) for x in c[:8]], int(c[8]), int(c[9]))
        return table, aligned, ctr

    def score_edges(self, rows, swap, len1, len2, mean, sigma, read_len, lognormal=None):
        return score_rows(self.rows, rows, swap, len1, len2, mean, sigma, read_len, lognormal)

    def conditional_stddevs(self, density, steps):
        return conditional_sigmas(density, steps)


def conditional_sigmas(density, steps):
    """One sigma per step of get_conditional_stddevs (CreateGraph.py:436-469), from the oracle's flattened list."""
    emp = {int(x): float(v) for x, v in enumerate(np.asarray(density).tolist()) if v != 0.0}
    emp.setdefault(len(density) - 1, 0.0)
    steps = [int(g) for g in steps]
    flat = O.conditional_stddevs(steps, emp, len(density) - 1)
    return np.array([flat[g] for g in steps], dtype=np.float64)


def stream_order_of(batch):
    """(first record in front of its predecessor in (reference id with -1 last, position) order or None, first key, last)."""
    if len(batch) == 0:
        return None, (0, 0), (0, 0)
    key = ((batch.tid.astype(np.int64) & 0xffffffff).astype(np.uint64) << np.uint64(32)) | \
        ((batch.pos.astype(np.int64) + 1) & 0xffffffff).astype(np.uint64)
    bad = np.flatnonzero(key[1:] < key[:-1])
    ends = (int(batch.tid[0]), int(batch.pos[0])), (int(batch.tid[-1]), int(batch.pos[-1]))
    return (int(bad[0]) + 1 if bad.size else None,) + ends


def score_rows(r, rows, swap, len1, len2, mean, sigma, read_len, lognormal=None):
    """GiveScoreOnEdges' per-edge numbers (CreateGraph.py:498-614) from the oracle, for rows of an edge-row dict
    (c_oracle.edge_rows layout).  lognormal = (ln_mu, ln_sigma, x_max, max_gap): the log-normal branch (:522-531)."""
    m = len(rows)
    gap = np.zeros(m)
    sd0 = np.zeros(m)
    ks = np.zeros(m, np.int32)
    flags = np.zeros(m, np.uint8)
    for j in range(m):
        i = int(rows[j])
        n = int(r['n'][i])
        lo, hi = int(r['offset'][i]), int(r['offset'][i]) + n
        a = r['obs_hi'][lo:hi] if swap[j] else r['obs_lo'][lo:hi]
        b = r['obs_lo'][lo:hi] if swap[j] else r['obs_hi'][lo:hi]
        obs = int(r['sum_obs'][i])
        mean_ = obs / float(n)
        l1, l2 = int(len1[j]), int(len2[j])
        long_enough = 2 * sigma < l1 and 2 * sigma < l2
        if long_enough and lognormal is not None:
            samples = [int(x) + int(y) for x, y in zip(a, b)]
            g = min(O.lognormal_gap_estimator(lognormal[0], lognormal[1], read_len, samples, l1, l2), lognormal[3])
        else:
            g = O.gap_estimator(mean, sigma, read_len, mean_, l1, l2) if long_enough else (n * mean - obs) / float(n)
        gap[j] = g
        flags[j] = (1 if long_enough else 0) | (2 if (-g > l1 or -g > l2) else 0)
        sd0[j] = O.tr_sk_std_dev(mean, sigma, read_len, l1, l2, g) if long_enough and lognormal is None else 2.0 ** 32
        s1 = sorted(int(x) for x in a)
        m1 = sum(s1) / float(n)
        mx = int(max(b))
        s2 = sorted(mx - int(x) for x in b)
        m2 = sum(s2) / float(n)
        ks[j] = O.ks_h([x - m1 for x in s1], [x - m2 for x in s2])
    return gap, sd0, ks, flags


class OracleRankEngine(object):
    """Stand-in for besst_amd.sharded.HipRankEngine: one rank's kernel stages answered by the oracle on CPU tensors, so
    that the sharded drop-in (besst_amd.sharded: leader / followers, the collective build and score stages) runs under gloo
    without a GPU.  Injected by `sharded.RankEngine = OracleRankEngine` inside the test processes."""

    def __init__(self, part, n_contigs):
        self.part, self.n_contigs = part, n_contigs

    @classmethod
    def from_batch(cls, part, n_contigs):
        return cls(part, n_contigs)

    def stream_order(self):
        return stream_order_of(self.part)

    def metrics_backend(self, top_mask):
        from tests import dist_util as DU
        return DU.OracleMetricsBackend(self.part, top_mask)

    def probe_tuples(self, table, lib, node_bits):
        from tests import dist_util as DU
        self.table, self.lib, self.node_bits = table, lib, node_bits
        res = O.LoopResult(len(table['cls']))
        res.tuples = []
        O.record_loop(DU.rec_lists(self.part), DU.table_lists(table), DU.oracle_params(lib), res=res)
        return len(res.tuples)

    def make_job(self, rank, world, group, pair_capacity, tuple_capacity):
        import torch
        from besst_amd import distributed
        from tests import dist_util as DU
        backend = DU.OracleBackend(self.part, self.table, self.lib, self.node_bits, rank, world, pair_capacity)
        return distributed.ShardedGraphBuild(torch.device('cpu'), None, rank, world, backend=backend, group=group)

    def local_table(self, job):
        rows = job.backend.rows
        keys = sorted(rows)
        n = np.array([rows[k]['n'] for k in keys], dtype=np.uint32)
        off = (np.cumsum(n.astype(np.int64)) - n).astype(np.uint32)
        lo = np.array([x for k in keys for x in (rows[k]['lo'] if not k & 1 else [0] * rows[k]['n'])], dtype=np.int32)
        hi = np.array([x for k in keys for x in (rows[k]['hi'] if not k & 1 else [0] * rows[k]['n'])], dtype=np.int32)
        self.rows = dict(n=n, offset=off, obs_lo=lo, obs_hi=hi,
                         sum_obs=np.array([rows[k]['s'] if not k & 1 else 0 for k in keys], dtype=np.int64))
        return device.EdgeTable(np.array(keys, dtype=np.uint64), np.array([rows[k]['mask'] for k in keys], dtype=np.uint32), n,
                                self.rows['sum_obs'],
                                np.array([rows[k]['s2'] if not k & 1 else 0 for k in keys], dtype=np.int64),
                                np.array([rows[k]['first'] for k in keys], dtype=np.uint32), off, self.node_bits, lo, hi)

    def score(self, job, rows, swap, len1, len2, mean, sigma, read_len, lognormal=None):
        return score_rows(self.rows, rows, swap, len1, len2, mean, sigma, read_len, lognormal)

    def conditional_stddevs(self, density, steps):
        return conditional_sigmas(density, steps)

    def close(self):
        pass


def fake_chain_arrays(n_scaffolds, link, gap, scaffold_length, node_order, device=0):
    """Sequential stand-in for besst_amd.MakeScaffolds.chain_arrays (besst_chain_scaffolds, csrc/chain.hip): per scaffold
    end the terminal of the path on that side, what lies beyond it and the smallest node order there - by walking."""
    n = int(n_scaffolds)
    terminal = np.zeros(2 * n, np.int32)
    beyond = np.zeros(2 * n, np.int64)
    lowest = np.full(2 * n, 0x7fffffff, np.int32)
    for h in range(2 * n):
        cur, dist, low, steps = h, 0, 0x7fffffff, 0
        while 0 <= link[cur] < 2 * n and steps <= 2 * n:
            v = int(link[cur])
            far = v ^ 1
            dist += int(gap[cur]) + int(scaffold_length[v >> 1])
            low = min(low, int(node_order[v]), int(node_order[far]))
            cur = far
            steps += 1
        terminal[h], beyond[h], lowest[h] = cur, dist, low
    return terminal, beyond, lowest, 0
