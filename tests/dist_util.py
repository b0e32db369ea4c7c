"""Oracle-backed stand-in for the per-rank kernel stages (test infrastructure only).

``OracleBackend`` implements the interface ``besst_amd.distributed.ShardedGraphBuild`` drives, with the pure
Python oracle instead of HIP kernels and CPU tensors instead of HBM, so the multi-rank orchestration (tail
gather, carry resolution, region packing, all-to-all, unpack order, reductions) can be run under gloo with
world_size 2 on a machine without GPUs.  The region byte layout is the one csrc/sortreduce.hip writes.
"""
import numpy as np
import torch

from besst_amd import _lib
from oracle import py_oracle as O


def split_batch(batch, world):
    n = len(batch)
    cuts = [n * r // world for r in range(world + 1)]
    return [batch.slice(cuts[r], cuts[r + 1]) for r in range(world)]


def table_lists(table):
    return dict(cls=table['cls'].tolist(), scaf=table['scaf_id'].tolist(), slen=table['scaf_len'].tolist(),
                cpos=table['ctg_pos'].tolist(), clen=table['ctg_len'].tolist(),
                cdir=[bool(x) for x in table['direction'].tolist()])


def rec_lists(batch):
    return {k: getattr(batch, k).tolist() for k in ('tid', 'mtid', 'pos', 'mpos', 'flag', 'mapq', 'qlen')}


def oracle_params(lib):
    return O.LibParams(read_len=lib['read_len'], ins_size_threshold=lib['ins_size_threshold'], min_mapq=lib['min_mapq'],
                       orientation=lib['orientation'], detect_duplicate=lib['detect_duplicate'],
                       extend_paths=lib['extend_paths'], no_score=lib['no_score'])


class OracleBackend(object):
    def __init__(self, batch_slice, table, lib, node_bits, rank, world, pair_capacity):
        self.lib = _lib.load()
        self.rec = rec_lists(batch_slice)
        self.tab = table_lists(table)
        self.p = oracle_params(lib)
        self.nb = node_bits
        self.rank, self.world = rank, world
        self.pair_cap = pair_capacity
        self.region = 64 + pair_capacity * 20
        self.n_contigs = len(self.tab['cls'])
        self.aligned = torch.zeros(self.n_contigs, dtype=torch.int64)
        self.counter_words = torch.zeros(8, dtype=torch.int64)
        self.rows = None
        self.overflow = False

    def reset(self):
        self.aligned.zero_()
        self.counter_words.zero_()

    def classify_scan(self):
        # a dry pass only to learn the slice's tail; prev_obs does not influence which records reach CreateEdge
        res = O.record_loop(self.rec, self.tab, self.p)
        self._tail = torch.tensor([1 if res.n_reach else 0, res.prev[0] if res.n_reach else 0,
                                   res.prev[1] if res.n_reach else 0, 0], dtype=torch.int32)

    def classify_tail(self):
        return self._tail

    def classify_emit(self, tails):
        t = tails.numpy().reshape(self.world, 4)
        carry = (-1, -1)
        for j in range(self.rank):
            if t[j, 0]:
                carry = (int(t[j, 1]), int(t[j, 2]))
        res = O.LoopResult(self.n_contigs)
        res.prev = carry
        res.tuples = []
        O.record_loop(self.rec, self.tab, self.p, res=res)
        self.res = res
        self.aligned += torch.tensor(res.aligned, dtype=torch.int64)
        self.counter_words += torch.tensor([res.count, res.non_unique, res.non_unique_for_scaf, res.nr_of_duplicates,
                                            res.too_long, res.fishy_reads, len(res.tuples), res.n_reach],
                                           dtype=torch.int64)

    def _key(self, u, v, fishy):
        return (((u << self.nb) | v) << 1) | fishy

    def partition(self):
        buf = np.zeros(self.world * self.region, dtype=np.uint8)
        per = [[] for _ in range(self.world)]
        for i, (u, v, fishy, ou, ov, mask) in enumerate(self.res.tuples):
            owner = self.lib.besst_owner_of_scaffold(u >> 1, self.world)
            per[owner].append((self._key(u, v, fishy), ou | ((ov | (mask << 30)) << 32), i))
        for d in range(self.world):
            base = d * self.region
            cnt = min(len(per[d]), self.pair_cap)
            buf[base:base + 16].view(np.uint32)[:] = [cnt, len(per[d]), len(self.res.tuples), 0]
            k = buf[base + 64:base + 64 + self.pair_cap * 8].view(np.uint64)
            pl = buf[base + 64 + self.pair_cap * 8:base + 64 + self.pair_cap * 16].view(np.uint64)
            ix = buf[base + 64 + self.pair_cap * 16:base + 64 + self.pair_cap * 20].view(np.uint32)
            for j in range(cnt):
                k[j], pl[j], ix[j] = per[d][j]
        return torch.from_numpy(buf)

    def unpack(self, recv):
        buf = recv.numpy()
        self.recv = []
        gbase = 0
        for s in range(self.world):
            base = s * self.region
            cnt, want, n_out, _ = buf[base:base + 16].view(np.uint32).tolist()
            self.overflow |= want > cnt
            k = buf[base + 64:base + 64 + self.pair_cap * 8].view(np.uint64)
            pl = buf[base + 64 + self.pair_cap * 8:base + 64 + self.pair_cap * 16].view(np.uint64)
            ix = buf[base + 64 + self.pair_cap * 16:base + 64 + self.pair_cap * 20].view(np.uint32)
            for j in range(cnt):
                self.recv.append((int(k[j]), int(pl[j]), gbase + int(ix[j])))
            gbase += n_out

    def reduce(self):
        rows = {}
        for key, pl, g in self.recv:         # arrival order = global BAM order
            lo, hi = pl & 0xffffffff, pl >> 32
            r = rows.setdefault(key, dict(n=0, s=0, s2=0, first=g, mask=hi >> 30, lo=[], hi=[]))
            o_lo, o_hi = lo, hi & 0x3fffffff
            r['n'] += 1
            r['s'] += o_lo + o_hi
            r['s2'] += (o_lo + o_hi) ** 2
            r['lo'].append(o_lo)
            r['hi'].append(o_hi)
        self.rows = rows

    def pack_for_allreduce(self):
        self._sum_buf = torch.cat([self.aligned, self.counter_words])
        return self._sum_buf

    def unpack_after_allreduce(self):
        self.aligned.copy_(self._sum_buf[:self.n_contigs])
        self.counter_words.copy_(self._sum_buf[self.n_contigs:])

    def overflowed(self):
        return self.overflow

    def sizes(self):
        return len(self.recv), len(self.rows)

    def local_table(self):
        return self.rows


def rows_from_table(table):
    """{key: dict(n, s, s2, first, mask, lo, hi)} from a device EdgeTable (fishy rows keep empty lists)."""
    out = {}
    for i in range(len(table)):
        lo, hi = int(table.offset[i]), int(table.offset[i]) + int(table.n[i])
        out[int(table.key[i])] = dict(n=int(table.n[i]), s=int(table.sum_obs[i]), s2=int(table.sum_obs_sq[i]),
                                      first=int(table.first_idx[i]), mask=int(table.mask[i]),
                                      lo=table.obs_lo[lo:hi].tolist(), hi=table.obs_hi[lo:hi].tolist())
    return out


def expected_rows(batch, table, lib, node_bits):
    """Single-process oracle result in the same row format (first = global emit index)."""
    res = O.LoopResult(len(table['cls']))
    res.tuples = []
    O.record_loop(rec_lists(batch), table_lists(table), oracle_params(lib), res=res)
    rows = {}
    for g, (u, v, fishy, ou, ov, mask) in enumerate(res.tuples):
        key = (((u << node_bits) | v) << 1) | fishy
        r = rows.setdefault(key, dict(n=0, s=0, s2=0, first=g, mask=mask, lo=[], hi=[]))
        r['n'] += 1
        r['s'] += ou + ov
        r['s2'] += (ou + ov) ** 2
        r['lo'].append(ou)
        r['hi'].append(ov)
    return rows, res


class OracleMetricsBackend(object):
    """Stand-in for pipeline.DeviceMetricsSampler: the predicates of libmetrics' scans restated in numpy
    (bam_parser.py:22-29, libmetrics.py:63-84,293-303), with running counts so that a slice can continue the scan
    of the slices before it."""
    CAP = 1000000

    def __init__(self, batch_slice, top_mask):
        b = batch_slice
        self.b = b
        tid = b.tid.astype(np.int64)
        ok = (tid >= 0) & (tid < len(top_mask))
        self.top = np.zeros(len(b), bool)
        self.top[ok] = np.asarray(top_mask, bool)[tid[ok]]
        self.at = np.abs(b.tlen.astype(np.int64))

    def _flags(self, orientation, min_mapq, read_len):
        b = self.b
        f = b.flag.astype(np.int64)
        rev, mrev = (f & 0x10) != 0, (f & 0x20) != 0
        tl = b.tlen.astype(np.int64)
        base = ((f & 0x80) != 0) & (b.tid == b.mtid) & ((f & 0x8) == 0) & (b.mapq.astype(np.int64) > min_mapq) & \
               ((f & 0x100) == 0)
        innie = base & ((rev & ~mrev & (tl < 0)) | (~rev & mrev & (tl > 0)))
        outie = base & ((rev & ~mrev & (tl > 0)) | (~rev & mrev & (tl < 0)))
        rf = orientation == 'rf'
        a = self.top & (outie if rf else innie)
        c = self.top & ((f & 0x4) == 0)
        if rf:
            d = self.top & innie & (read_len < self.at.astype(np.float64))
        else:
            d = self.top & outie & (read_len < self.at.astype(np.float64) + 2.0 * read_len)
        return a, self.top, c, d

    def count(self, orientation, min_mapq, read_len):
        a, b, c, d = self._flags(orientation, min_mapq, read_len)
        return torch.tensor([int(a.sum()), int(b.sum()), int(d.sum())], dtype=torch.int64)

    def emit(self, before, orientation, min_mapq, read_len, want_isize=True):
        a, b, c, d = self._flags(orientation, min_mapq, read_len)
        pa0, pb0, pd0 = [int(x) for x in before.tolist()]
        samples = np.zeros(2 * self.CAP, np.int32)
        state = np.zeros(8, np.int64)
        if want_isize:
            pos = pa0 + np.cumsum(a) - 1
            sel = a & (pos < self.CAP)
            samples[pos[sel]] = self.at[sel]
        inside = b & (pb0 + np.cumsum(b) - 1 < self.CAP)
        posd = pd0 + np.cumsum(d) - 1
        seld = d & inside
        samples[self.CAP + posd[seld]] = self.at[seld]
        state[0], state[1], state[2] = pa0 + int(a.sum()), pb0 + int(b.sum()), pd0 + int(d.sum())
        state[3], state[4], state[5] = int((c & inside).sum()), int(seld.sum()), len(self.b)
        return torch.from_numpy(samples), torch.from_numpy(state)
