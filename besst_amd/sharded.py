"""The drop-in's two entry points over a process group: ``libmetrics.get_metrics`` and ``CreateGraph.PE`` on a record
stream that is sharded across the GPUs of one node (the reference's seam runBESST:162-182,
BESST/libmetrics.py:226, BESST/CreateGraph.py:45).

One process per GPU (torchrun); every rank makes the SAME calls the single-GPU drop-in makes:

    records = bamio.open_bam(path)                      # rank r ingests slice r of the file on its GPU
    libmetrics.get_metrics(records, param, Information) # ShardedMetricsSample + the unchanged host finishing: every rank
                                                        # ends with the identical `param`
    G, G_prime = CreateGraph.PE(...)                    # rank 0 LEADS: it runs the host side of PE (objects, filters in
                                                        # first-occurrence order, graph assembly) and returns (G, G_prime);
                                                        # the other ranks FOLLOW: they take part in the collective stages
                                                        # rank 0 asks for and return empty graphs

Stages of PE that are collective (rank 0 broadcasts a command, every rank executes it):
    build   contig table + library constants from rank 0 -> every rank runs distributed.ShardedGraphBuild.step on its
            slice (stream-order slices -> owner partition -> ONE all-to-all -> per-owner sort/reduce), then the owners'
            edge rows are gathered to rank 0 (keys are disjoint; first_idx = global emit index restores the reference's
            first-occurrence order, CreateGraph.py:842-849)
    score   rank 0 has filtered the edges (the dense-region rule is order dependent, CreateGraph.py:355-404, so it runs
            on the gathered table); the rows that are left are scored by their OWNERS, where the observations lie
            (GiveScoreOnEdges, :498-614), and the results gathered
    done    the fields PE leaves in `param` travel to the followers; abort: rank 0 left PE with an exception / sys.exit

The kernel stages of a rank sit behind ``RankEngine`` (HipRankEngine: the HIP library on the rank's GPU).  The CPU suite
injects an oracle-backed engine (tests/fake_device.py) to run exactly this orchestration under gloo with world_size 2.
"""
import sys

import numpy as np

from . import _lib
from ._lib import Counters
from .device import EdgeTable

HEAD_RECORDS = 1000                 # the read-length step looks at the first 1000 records (libmetrics.py:246-273)
# what CreateGraph.PE leaves in `param`; sent to the followers with 'done'
PARAM_FIELDS = ('mean_coverage', 'std_dev_coverage', 'edgesupport', 'expected_links_over_mean_plus_stddev',
                'scaffold_indexer', 'tot_assembly_length', 'current_N50', 'current_L50')


def _dist():
    import torch.distributed as dist
    return dist


def active_group():
    """(rank, world) when the sharded form applies, None otherwise (without importing torch when nobody has).  Sharding is
    OPT-IN: a process that uses torch.distributed for something of its own and calls get_metrics / PE on one rank must not
    be pulled into collectives.  It applies when torch.distributed is initialised with more than one rank AND the caller
    has said so - ``sharded.enable(group=None)`` (what besst_amd.cli does under torchrun), or BESST_SHARDED=1 in the
    environment; BESST_SHARDED=0 switches it off whatever the code says, BESST_SHARDED=force applies it with ONE rank too
    (the whole orchestration over RCCL on a single GPU: what a one-GPU box can check of the device-side transport).  A
    bamio.ShardedBam handed to get_metrics / PE is sharded by construction."""
    import os
    mode = os.environ.get('BESST_SHARDED')
    if mode == '0' or 'torch.distributed' not in sys.modules:
        return None
    if mode is None and not _ENABLED:
        return None
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()):
        return None
    world = dist.get_world_size(PROCESS_GROUP)
    if world < 2 and mode != 'force':
        return None
    return dist.get_rank(PROCESS_GROUP), world


# the process group the sharded sessions use (None: the default group); set by enable() before the first get_metrics call
PROCESS_GROUP = None
_ENABLED = False


def enable(group=None):
    """Every rank of `group` (default: the whole job) will make the drop-in's calls together from here on: open_bam ingests
    a slice per rank, get_metrics and PE are collective (see the module docstring)."""
    global PROCESS_GROUP, _ENABLED
    PROCESS_GROUP, _ENABLED = group, True


def disable():
    global PROCESS_GROUP, _ENABLED
    PROCESS_GROUP, _ENABLED = None, False


class RemoteAbort(_lib.BesstDeviceError):
    """Rank 0 left CreateGraph.PE with an exception; the followers raise this."""


RankFailure = _lib.RankFailure      # a collective stage failed on some rank; raised on every rank


def _src(group):
    dist = _dist()
    return dist.get_global_rank(group, 0) if group is not None else 0


def _agree(group, world, error):
    """Every rank reports None or an error text; all raise together when anybody failed (nobody is left in a collective)."""
    if world == 1:
        if error is not None:
            raise RankFailure(error)
        return
    out = [None] * world
    _dist().all_gather_object(out, error, group=group)
    bad = [(r, e) for r, e in enumerate(out) if e is not None]
    if bad:
        raise RankFailure('; '.join('rank %d: %s' % b for b in bad))


def node_bits_of(table):
    """Width of a node code (scaffold id * 2 + side) over the scaffolds present in the table."""
    present = np.asarray(table['cls']) != 0
    ids = np.asarray(table['scaf_id'])[present]
    top = int(ids.max()) if ids.size else 1
    return max(1, int(top * 2 + 1).bit_length())


def union_tables(tables):
    """The owners' edge tables as one (keys are disjoint: every key has one owner).  -> (EdgeTable, owner of every row,
    the row's index in its owner's table).  Observation offsets are shifted to the concatenated columns."""
    node_bits = tables[0].node_bits
    base, offs = 0, []
    for t in tables:
        offs.append(t.offset.astype(np.int64) + base)
        base += int(t.obs_lo.shape[0])
    if base >= 1 << 32:
        raise _lib.BesstDeviceError('gathered edge table: %d observations do not fit the 32-bit offsets' % base)

    def cat(name, dtype):
        return np.concatenate([np.asarray(getattr(t, name), dtype=dtype) for t in tables])
    table = EdgeTable(cat('key', np.uint64), cat('mask', np.uint32), cat('n', np.uint32), cat('sum_obs', np.int64),
                      cat('sum_obs_sq', np.int64), cat('first_idx', np.uint32),
                      np.concatenate(offs).astype(np.uint32), node_bits, cat('obs_lo', np.int32), cat('obs_hi', np.int32))
    owner = np.concatenate([np.full(len(t), r, dtype=np.int32) for r, t in enumerate(tables)])
    local = np.concatenate([np.arange(len(t), dtype=np.int64) for t in tables])
    return table, owner, local


class HipRankEngine(object):
    """One rank's kernel stages on its GPU: pipeline.DeviceMetricsSampler for the library scans,
    distributed.ShardedGraphBuild / HipBackend for the graph build, DeviceGraphBuilder.score_edges for its rows."""

    def __init__(self, device, rec, n_contigs, keep=None):
        self.device, self.rec, self.n_contigs = device, rec, int(n_contigs)
        self.keep = keep                                     # whatever owns the record memory (a ResidentBam slice)
        self._sampler = None

    @classmethod
    def from_batch(cls, part, n_contigs):
        """Over a slice of host records: uploaded to the process's current device."""
        from . import pipeline
        dev = _cuda_device()
        return cls(dev, pipeline.DeviceRecords(part, dev), n_contigs)

    def metrics_backend(self, top_mask):
        from . import pipeline
        if self._sampler is None:
            self._sampler = pipeline.DeviceMetricsSampler(self.device, self.rec, self.n_contigs)
        self._sampler.set_top(top_mask)
        return self._sampler

    def stream_order(self):
        """(first record of the slice that lies in front of its predecessor or None, the slice's first (tid, pos), its last)."""
        import ctypes as C
        import torch
        rec, lib = self.rec, _lib.load()
        if rec.n == 0:
            return None, (0, 0), (0, 0)
        word = torch.zeros(1, dtype=torch.int64, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(lib.besst_dev_stream_order(C.c_void_p(stream), rec.n, C.c_void_p(rec.tid.data_ptr()),
                                              C.c_void_p(rec.pos.data_ptr()), C.c_void_p(word.data_ptr())), 'dev_stream_order')
        ends = torch.stack([rec.tid[0], rec.pos[0], rec.tid[-1], rec.pos[-1]]).cpu().tolist()
        first = int(word.item())
        return (None if first < 0 else first), (ends[0], ends[1]), (ends[2], ends[3])

    def probe_tuples(self, table, lib, node_bits):
        """Tuples this slice emits (one untimed local pass, no collective): sizes the exchange regions."""
        from . import distributed
        self._sampler = None                                 # (its 8 MB of samples and workspace are not needed any more)
        self.inputs = distributed.BuildInputs(self.rec, self.n_contigs, node_bits, lib, table)
        return distributed.ShardedGraphBuild.probe_tuples(self.device, self.inputs)

    def make_job(self, rank, world, group, pair_capacity, tuple_capacity):
        from . import distributed
        backend = distributed.HipBackend(self.device, self.inputs, rank, world, pair_capacity, tuple_capacity)
        return distributed.ShardedGraphBuild(self.device, self.inputs, rank, world, backend=backend, group=group)

    def synchronize(self):
        import torch
        torch.cuda.synchronize(self.device)

    def local_table(self, job):
        return job.backend.local_table()

    def score(self, job, rows, swap, len1, len2, mean, sigma, read_len, lognormal=None):
        return job.backend.gb.score_edges(rows, swap, len1, len2, mean, sigma, read_len, lognormal=lognormal)

    def conditional_stddevs(self, density, steps):
        """get_conditional_stddevs' sigmas on this rank's GPU (besst_dev_conditional_stddevs)."""
        import ctypes as C
        import torch
        lib = _lib.load()
        f = torch.from_numpy(np.ascontiguousarray(density, dtype=np.float64)).to(self.device)
        st = torch.from_numpy(np.ascontiguousarray(steps, dtype=np.int32)).to(self.device)
        out = torch.empty(int(st.shape[0]), dtype=torch.float64, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(lib.besst_dev_conditional_stddevs(C.c_void_p(stream), C.c_void_p(f.data_ptr()), int(f.shape[0]) - 1,
                                                     C.c_void_p(st.data_ptr()), int(st.shape[0]),
                                                     C.c_void_p(out.data_ptr())), 'dev_conditional_stddevs')
        return out.cpu().numpy()

    def close(self):
        self._sampler = None
        self.inputs = None
        if self.keep is not None and hasattr(self.keep, 'close'):
            self.keep.close()
        self.keep = self.rec = None


# what builds a rank's engine from its slice of the stream; tests put an oracle-backed class here
RankEngine = HipRankEngine


class ShardedHead(object):
    """What the host side of get_metrics / PE reads of the `bam_file`: the header, the GLOBAL record count and
    rlen / alen / qlen of the first 1000 records of the whole stream (gathered from the slices in rank order)."""

    def __init__(self, references, lengths, n_local, head, group, world):
        self.references, self.lengths = tuple(references), tuple(int(x) for x in lengths)
        counts = [None] * world
        _dist().all_gather_object(counts, (int(n_local), head), group=group)
        self.slice_records = [c[0] for c in counts]
        self._n = int(sum(self.slice_records))
        cols = []
        for k in range(3):
            parts = [np.asarray(c[1][k]) for c in counts if c[1][k] is not None]
            cols.append(np.concatenate(parts)[:HEAD_RECORDS] if parts else None)
        self.rlen, self.alen, self.qlen = cols

    def __len__(self):
        return self._n


class _Counts(object):
    def __init__(self, d):
        self.__dict__.update(d)


class ShardedContext(object):
    """device.GraphContext's interface for CreateGraph.PE, executed by all ranks: rank 0 calls set_contigs /
    set_library / build_graph / score_edges as on one GPU, the other ranks sit in follow()."""

    def __init__(self, engine, rank, world, group):
        self.engine, self.rank, self.world, self.group = engine, rank, world, group
        self.job = None
        self._table_cols = self._lib = None
        self._owner = self._local = None
        self.stats = {}

    # ---- leader side (rank 0): GraphContext's methods --------------------------------------------------------
    def _send(self, cmd):
        _dist().broadcast_object_list([cmd], src=_src(self.group), group=self.group)

    def set_contigs(self, **cols):
        self._table_cols = {k: np.ascontiguousarray(v) for k, v in cols.items()}
        self.n_contigs = int(self._table_cols['cls'].shape[0])

    def set_library(self, read_len, ins_size_threshold, min_mapq, orientation, detect_duplicate, extend_paths, no_score):
        self._lib = dict(read_len=float(read_len), ins_size_threshold=float(ins_size_threshold), min_mapq=int(min_mapq),
                         orientation=orientation, detect_duplicate=bool(detect_duplicate), extend_paths=bool(extend_paths),
                         no_score=bool(no_score))

    def build_graph(self, lazy_observations=False):
        self._send(('build', self._table_cols, self._lib))
        return self._build(self._table_cols, self._lib)

    def conditional_stddevs(self, density, steps):
        return self.engine.conditional_stddevs(density, steps)          # rank 0 alone: a table of ~40 numbers

    def score_edges(self, rows, swap, len1, len2, mean, sigma, read_len, lognormal=None):
        rows = np.asarray(rows, dtype=np.int64)
        swap, len1, len2 = np.asarray(swap, np.uint8), np.asarray(len1, np.int64), np.asarray(len2, np.int64)
        owner = self._owner[rows]
        requests, where = [], []
        for r in range(self.world):
            at = np.flatnonzero(owner == r)
            where.append(at)
            requests.append((self._local[rows[at]].astype(np.uint32), swap[at], len1[at], len2[at],
                             float(mean), float(sigma), float(read_len), lognormal))
        self._send(('score',))
        results = self._score(requests)
        m = int(rows.shape[0])
        gap, sd0 = np.zeros(m, np.float64), np.zeros(m, np.float64)
        ks_h, flags = np.zeros(m, np.int32), np.zeros(m, np.uint8)
        for at, res in zip(where, results):
            gap[at], sd0[at], ks_h[at], flags[at] = res
        return gap, sd0, ks_h, flags

    def done(self, param):
        self._send(('done', {k: getattr(param, k) for k in PARAM_FIELDS if hasattr(param, k)}))

    def abort(self, exc):
        kind = 'exit' if isinstance(exc, SystemExit) else 'error'
        code = getattr(exc, 'code', None)
        self._send(('abort', kind, code if kind == 'exit' and isinstance(code, (int, str, type(None))) else
                    '%s: %s' % (type(exc).__name__, exc)))

    # ---- follower side ------------------------------------------------------------------------------------------
    def follow(self, param):
        """Execute rank 0's commands until it is done with PE."""
        while True:
            box = [None]
            _dist().broadcast_object_list(box, src=_src(self.group), group=self.group)
            cmd = box[0]
            if cmd[0] == 'build':
                self._build(cmd[1], cmd[2])
            elif cmd[0] == 'score':
                self._score(None)
            elif cmd[0] == 'done':
                for k, v in cmd[1].items():
                    setattr(param, k, v)
                return
            elif cmd[0] == 'abort':
                if cmd[1] == 'exit':
                    # rank 0 has written the reference's message (sys.exit(str) prints it); the followers leave quietly
                    raise SystemExit(0 if cmd[2] is None or isinstance(cmd[2], str) else cmd[2])
                raise RemoteAbort('rank 0 left CreateGraph.PE: %s' % (cmd[2],))
            else:
                raise _lib.BesstDeviceError('sharded PE: unknown command %r' % (cmd[0],))

    # ---- the collective stages (every rank) ---------------------------------------------------------------------
    def _build(self, table, lib):
        from time import perf_counter
        t0 = perf_counter()
        eng, world, group = self.engine, self.world, self.group
        self.job = None
        node_bits = node_bits_of(table)
        # 1. local probe pass (no collective): how many tuples this slice emits -> the regions' capacity, on every rank
        n_out, err = 0, None
        try:
            n_out = int(eng.probe_tuples(table, lib, node_bits))
        except Exception as e:
            err = '%s: %s' % (type(e).__name__, e)
        outs = [None] * world
        _dist().all_gather_object(outs, (n_out, err), group=group)
        bad = [(r, o[1]) for r, o in enumerate(outs) if o[1] is not None]
        if bad:
            raise RankFailure('; '.join('rank %d: %s' % b for b in bad))
        pair_cap = max(int(o[0] * 1.5 / world) + 4096 for o in outs)
        # 2. buffers (local), agreed before the first data-path collective
        try:
            self.job = eng.make_job(self.rank, world, group, pair_cap, int(n_out * 1.25) + 4096)
        except Exception as e:
            err = '%s: %s' % (type(e).__name__, e)
        _agree(group, world, err)
        # 3. the step: slices -> owners -> rows; a region that overflowed grows and the step repeats (all ranks together).
        # What fails on one rank behind the step's collectives - the capacity check's verdict, the download of the rows -
        # is agreed on before the gather: nobody is left waiting for a rank that has gone.
        # (a rank that fails locally inside step() - after its collectives: the per-owner sort and reduction - must not
        # go on to _agree's all-gather while the others sit in check_capacity's all-reduce: the local verdict is agreed on
        # between the two, and check_capacity agrees on its own allocation before it repeats the step)
        mine = None
        try:
            self.job.step()
        except RankFailure:
            raise
        except Exception as e:
            err = '%s: %s' % (type(e).__name__, e)
        _agree(group, world, err)
        self.job.check_capacity()                            # (raises RankFailure on every rank, or on none)
        t1 = perf_counter()
        if err is None:
            try:
                mine = eng.local_table(self.job)
            except Exception as e:
                err = '%s: %s' % (type(e).__name__, e)
        _agree(group, world, err)
        tables = [None] * world if self.rank == 0 else None
        _dist().gather_object(mine, tables, dst=_src(group), group=group)
        self.stats = dict(build_s=t1 - t0, gather_s=perf_counter() - t1, pair_capacity=pair_cap)
        if self.rank != 0:
            return None
        table_u, self._owner, self._local = union_tables(tables)
        b = self.job.backend
        aligned = np.asarray(b.aligned.cpu().numpy(), dtype=np.int64)
        words = [int(x) for x in b.counter_words.cpu().tolist()[:8]]
        prev = self.job.final_prev_obs()
        ctr = Counters(*words, int(prev[0]), int(prev[1]))
        return table_u, aligned, ctr

    def _score(self, requests):
        dist = _dist()
        box = [None]
        dist.scatter_object_list(box, requests if self.rank == 0 else None, src=_src(self.group), group=self.group)
        rows, swap, len1, len2, mean, sigma, read_len, lognormal = box[0]
        res, err = None, None
        try:
            if rows.shape[0]:
                res = tuple(np.asarray(a) for a in self.engine.score(self.job, rows, swap, len1, len2, mean, sigma, read_len,
                                                                     lognormal=lognormal))
            else:
                res = (np.zeros(0), np.zeros(0), np.zeros(0, np.int32), np.zeros(0, np.uint8))
        except Exception as e:
            err = '%s: %s' % (type(e).__name__, e)
        # (EVERY rank learns of a failure: a follower whose own rows were fine would otherwise go back to waiting for the
        # next command of a rank 0 that has left)
        _agree(self.group, self.world, err)
        out = [None] * self.world if self.rank == 0 else None
        dist.gather_object(res, out, dst=_src(self.group), group=self.group)
        return out if self.rank == 0 else None

    def close(self):
        self.job = None
        if self.engine is not None:
            self.engine.close()
            self.engine = None


class ShardedSession(object):
    """session.Session's interface over the ranks of a process group (see the module docstring)."""

    def __init__(self, head, engine, rank, world, group):
        self.batch = head
        self.rank, self.world, self.group = rank, world, group
        self.ctx = ShardedContext(engine, rank, world, group)
        self.is_follower = rank != 0

    def metrics_sample(self, top_mask, orientation, min_mapq, read_len, want_isize):
        from . import distributed
        backend = self.ctx.engine.metrics_backend(top_mask)
        job = distributed.ShardedMetricsSample(backend, self.rank, self.world, self.group)
        isize, contam, counts = job.sample(orientation, min_mapq, read_len, want_isize)
        return isize, contam, _Counts(counts)

    def stream_order(self):
        """Index (in the whole stream) of the first record that breaks the coordinate order, or None: every slice checks
        itself, and its first record against the last record of the slice before it."""
        local, err = None, None
        try:
            local = self.ctx.engine.stream_order()
        except Exception as e:
            err = '%s: %s' % (type(e).__name__, e)
        n_local = self.batch.slice_records[self.rank]
        out = [None] * self.world
        _dist().all_gather_object(out, (local, n_local, err), group=self.group)
        bad = [(r, o[2]) for r, o in enumerate(out) if o[2] is not None]
        if bad:
            raise RankFailure('; '.join('rank %d: %s' % b for b in bad))
        out = [o[:2] for o in out]
        key = lambda k: ((k[0] & 0xffffffff) << 32) | ((k[1] + 1) & 0xffffffff)
        base, prev_last = 0, None
        for (first, first_key, last_key), n in out:
            if n:
                if prev_last is not None and key(first_key) < key(prev_last):
                    return base
                if first is not None:
                    return base + first
                prev_last = last_key
            base += n
        return None

    # PE's epilogue on rank 0 / the whole of PE on the others
    def follow(self, param):
        self.ctx.follow(param)

    def done(self, param):
        self.ctx.done(param)

    def abort(self, exc):
        self.ctx.abort(exc)

    def close(self):
        self.ctx.close()


def _cuda_device():
    import torch
    return torch.device('cuda', torch.cuda.current_device())


def session_for_batch(batch, rank, world, group=None):
    """A host RecordBatch that every rank holds (tests, pysam-like inputs materialised on every rank): rank r takes the
    r-th contiguous slice of the stream."""
    n = len(batch)
    lo, hi = n * rank // world, n * (rank + 1) // world
    part = batch.slice(lo, hi)
    k = min(HEAD_RECORDS, hi - lo)
    head = tuple(None if col is None else np.asarray(col[:k]) for col in
                 ((part.rlen if part.rlen is not None else part.qlen), (part.alen if part.alen is not None else part.qlen),
                  part.qlen))
    engine = RankEngine.from_batch(part, len(batch.references))
    return ShardedSession(ShardedHead(batch.references, batch.lengths, hi - lo, head, group, world), engine, rank, world, group)


def session_for_bam(sbam):
    """Over a bamio.ShardedBam: the slice is in HBM already."""
    return ShardedSession(sbam.head, sbam.engine, sbam.rank, sbam.world, sbam.group)
