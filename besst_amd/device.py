"""Thin Python handle on a ``besst_ctx`` (host-buffer layer of the C ABI).

Everything here is marshalling: numpy columns in, numpy result tables out.  The arithmetic
lives in the HIP kernels behind libbesst_amd.so; nothing in this module computes on the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import BesstDeviceError, Counters, LibParams, MetricsCounts

CLS_ABSENT, CLS_LARGE, CLS_SMALL = 0, 1, 2
MASK_G, MASK_GPRIME = 1, 2
SAMPLE_CAP = 1000000


class EdgeTable(object):
    """Edge rows as returned by the device, sorted by key.

    ``u``/``v`` are node codes (``scaffold_id * 2 + (side == 'R')``) with ``u < v``;
    ``is_fishy`` rows carry the BWA-quirk counts of CreateGraph.py:141-163, the other rows the
    link statistics of CreateEdge (:842-862).  ``obs_lo``/``obs_hi`` hold per-link observations
    grouped by row (slice ``offset[i] : offset[i] + n[i]``), in BAM order; ``observation_sums()`` is
    their sum, one value per link - an edge's `observations`.  A table that comes from a GraphContext
    holds the three columns lazily (``source``: the context's ObservationSource): they cross PCIe when
    something reads them.
    """

    def __init__(self, key, mask, n, sum_obs, sum_obs_sq, first_idx, offset, node_bits, obs_lo, obs_hi, source=None):
        self.key, self.mask, self.n = key, mask, n
        self.sum_obs, self.sum_obs_sq = sum_obs, sum_obs_sq
        self.first_idx, self.offset = first_idx, offset
        self.node_bits = node_bits
        self._obs_lo, self._obs_hi, self._sums, self._source = obs_lo, obs_hi, None, source
        pair = key >> np.uint64(1)
        self.is_fishy = (key & np.uint64(1)).astype(bool)
        self.u = (pair >> np.uint64(node_bits)).astype(np.int64)
        self.v = (pair & np.uint64((1 << node_bits) - 1)).astype(np.int64)

    def _ends(self):
        if self._obs_lo is None:
            self._obs_lo, self._obs_hi = self._source.ends()
        return self._obs_lo, self._obs_hi

    @property
    def obs_lo(self):
        return self._ends()[0]

    @property
    def obs_hi(self):
        return self._ends()[1]

    def observation_sums(self):
        if self._sums is None:
            self._sums = self._source.sums() if self._obs_lo is None else self._obs_lo + self._obs_hi
        return self._sums

    def __len__(self):
        return int(self.key.shape[0])

    def __getstate__(self):                                  # (pickled by the sharded build's gather: plain columns)
        d = dict(self.__dict__)
        d['_obs_lo'], d['_obs_hi'] = self._ends()
        d['_source'] = None
        return d


class ObservationSource(object):
    """The observation columns of the table a GraphContext has built, while they are still on the device.  The one
    column the graphs need - obs_lo + obs_hi per link - is summed on the device and fetched by a background thread
    (besst_ctx_fetch_observation_sums: its own stream; ctypes releases the interpreter lock for the call) from the moment
    the table exists, under CreateGraph.PE's host work; the two columns apart are fetched when somebody asks.  The context
    finishes a pending fetch before its records or the context itself go."""

    def __init__(self, ctx, n_tuples):
        import threading
        self._ctx, self._n = ctx, int(n_tuples)
        self._sums, self._error, self._ends_cols = None, None, None
        self._thread = threading.Thread(target=self._fetch, name='besst-observations', daemon=True)
        self._thread.start()

    def _fetch(self):
        try:
            out = np.empty(self._n, dtype=np.int32)
            _lib.check(self._ctx._lib.besst_ctx_fetch_observation_sums(self._ctx._ctx, _lib.ptr(out)), 'fetch_observation_sums')
            self._sums = out
        except BaseException as e:                           # handed to whoever joins
            self._error = e

    def finish(self):
        t = self._thread
        if t is not None:
            t.join()
            self._thread = None
        if self._error is not None:
            raise self._error

    def sums(self):
        self.finish()
        return self._sums

    def ends(self):
        if self._ends_cols is None:
            self.finish()
            ctx = self._ctx
            if ctx is None or not ctx._ctx:
                raise BesstDeviceError('the observation columns were not fetched before the context was closed')
            lo = np.empty(self._n, dtype=np.int32)
            hi = np.empty(self._n, dtype=np.int32)
            _lib.check(ctx._lib.besst_ctx_fetch_observations(ctx._ctx, _lib.ptr(lo), _lib.ptr(hi)), 'fetch_observations')
            self._ends_cols = (lo, hi)
        return self._ends_cols

    def detach(self):
        """The context is about to go (or to build another table): what is pending is finished, nothing refers to it after.
        A fetch that failed is the business of whoever READS the observations (sums / ends raise it): the unrelated call
        that happens to detach the source - push_records, build_graph, close - must not fail with the previous table's
        error, so it is only noted on stderr here and stays with the source."""
        t = self._thread
        if t is not None:
            t.join()
            self._thread = None
        if self._error is not None and not getattr(self, '_reported', False):
            self._reported = True
            import sys
            sys.stderr.write('besst_amd: the background fetch of a table\'s observations failed (%s); its readers will see '
                             'the error\n' % (self._error,))
        self._ctx = None


# Wall time spent inside the C-ABI calls of GraphContext (seconds per method), filled only while `CALL_SECONDS` is a dict:
# bench.py uses it to split the drop-in's wall time into library (device + transfers) and Python host time.
CALL_SECONDS = None


def _timed(fn):
    import functools
    import time

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        acc = CALL_SECONDS
        if acc is None:
            return fn(self, *args, **kwargs)
        t0 = time.perf_counter()
        try:
            return fn(self, *args, **kwargs)
        finally:
            acc[fn.__name__] = acc.get(fn.__name__, 0.0) + time.perf_counter() - t0
    return wrapper


class _DeviceView(object):
    """__cuda_array_interface__ over a span of a context's HBM (keeps the context alive)."""

    def __init__(self, owner, ptr, n, typestr):
        self.owner = owner
        self.__cuda_array_interface__ = {'shape': (n,), 'typestr': typestr, 'data': (ptr, False), 'version': 2, 'strides': None}


class GraphContext(object):
    def __init__(self, device=0):
        self._lib = _lib.load()
        self._ctx = self._lib.besst_ctx_create(int(device))
        if not self._ctx:
            raise BesstDeviceError('besst_ctx_create(%d) failed: %s' % (device, _lib.last_error()))
        self.device = int(device)
        self.n_contigs = 0

    def _detach_observations(self):
        src, self._observations = getattr(self, '_observations', None), None
        if src is not None:
            src.detach()

    def close(self):
        if getattr(self, '_ctx', None):
            try:
                self._detach_observations()
            finally:
                self._lib.besst_ctx_destroy(self._ctx)
                self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # ---- inputs ----------------------------------------------------------------------------------
    @_timed
    def set_contigs(self, scaf_id, scaf_len, ctg_pos, ctg_len, direction, cls):
        cols = [_lib.as_col(scaf_id, np.int32), _lib.as_col(scaf_len, np.int32), _lib.as_col(ctg_pos, np.int32),
                _lib.as_col(ctg_len, np.int32), _lib.as_col(direction, np.uint8), _lib.as_col(cls, np.uint8)]
        n = cols[0].shape[0]
        if any(c.shape != (n,) for c in cols):
            raise ValueError('contig table columns differ in length')
        _lib.check(self._lib.besst_ctx_set_contigs(self._ctx, n, *[_lib.ptr(c) for c in cols]), 'set_contigs')
        self.n_contigs = n

    def set_library(self, read_len, ins_size_threshold, min_mapq, orientation, detect_duplicate, extend_paths,
                    no_score):
        p = LibParams(float(read_len), float(ins_size_threshold), int(min_mapq),
                      {'fr': 0, 'rf': 1}[orientation], int(bool(detect_duplicate)), int(bool(extend_paths)),
                      int(bool(no_score)), 0)
        _lib.check(self._lib.besst_ctx_set_library(self._ctx, C.byref(p)), 'set_library')

    def clear_records(self):
        self._detach_observations()
        _lib.check(self._lib.besst_ctx_clear_records(self._ctx), 'clear_records')

    @_timed
    def push_records(self, batch):
        """``batch``: a RecordBatch (or anything with the eight SoA columns)."""
        self._detach_observations()
        cols = [_lib.as_col(batch.tid, np.int32), _lib.as_col(batch.mtid, np.int32),
                _lib.as_col(batch.pos, np.int32), _lib.as_col(batch.mpos, np.int32),
                _lib.as_col(batch.tlen, np.int32), _lib.as_col(batch.flag, np.uint16),
                _lib.as_col(batch.mapq, np.uint8), _lib.as_col(batch.qlen, np.uint16)]
        n = cols[0].shape[0]
        _lib.check(self._lib.besst_ctx_push_records(self._ctx, n, *[_lib.ptr(c) for c in cols]), 'push_records')

    @_timed
    def push_bam(self, handle, chunk_records=0, head_records=1000, mode=None, chunk_blocks=0, part=None, first_skip=None):
        """Stream an open besst_bam (bamio) into the context.  mode 'device': BGZF inflate + record decode on the GPU, the
        compressed file crosses PCIe (besst_ctx_push_bam_device; any BGZF block layout); 'host': inflate + decode
        on the reader's host threads into pinned staging, copies under the next chunk's decode (besst_ctx_push_bam); 'auto' (default; BESST_INGEST overrides): the device form, and the host form when the library answers
        BESST_ERR_UNSUPPORTED (``stats.on_device`` tells which one ran).  part = (r, W): only the r-th of W parts of the
        file's records (cut at BGZF block boundaries; multi-GPU ingest: rank r's slice of the stream) - device form only, and
        only for files in htslib's layout, where a block boundary is a record boundary (else BesstDeviceError, status 5) -
        unless first_skip is given: the SLICE form for any block layout (besst_ctx_push_bam_device_slice: -1 = guess where the
        slice's first record begins, >= 0 = it begins that many inflated bytes in); ``self.slice_boundary`` then holds
        (offset used, bytes of the slice's last record that lie in the next slice) for distributed.ingest_slice's check.
        -> (IngestStats, head rlen, head alen, head qlen)."""
        import os
        from ._lib import IngestStats
        self._detach_observations()
        mode = mode or os.environ.get('BESST_INGEST', 'auto')
        if mode not in ('auto', 'device', 'host'):
            raise ValueError("push_bam: mode must be 'auto', 'device' or 'host'")
        stats = IngestStats()
        rlen = np.zeros(head_records, dtype=np.int32)
        alen = np.zeros(head_records, dtype=np.int32)
        qlen = np.zeros(head_records, dtype=np.uint16)
        done = False
        if part is not None:
            r, w = int(part[0]), int(part[1])
            if mode == 'host':
                raise ValueError('push_bam: a part of a file is read by the device form only')
            mode = 'device'
        else:
            r, w = 0, 1
        if mode in ('auto', 'device'):
            if first_skip is not None:
                if part is None:
                    raise ValueError('push_bam: first_skip goes with part=(r, W)')
                bound = np.zeros(2, dtype=np.int64)
                rc = self._lib.besst_ctx_push_bam_device_slice(self._ctx, handle, r, w, int(chunk_blocks), int(first_skip),
                                                               _lib.ptr(bound), int(head_records), _lib.ptr(rlen), _lib.ptr(alen),
                                                               _lib.ptr(qlen), C.byref(stats))
                self.slice_boundary = (int(bound[0]), int(bound[1]))
            else:
                rc = self._lib.besst_ctx_push_bam_device_part(self._ctx, handle, r, w, int(chunk_blocks), int(head_records),
                                                              _lib.ptr(rlen), _lib.ptr(alen), _lib.ptr(qlen), C.byref(stats))
            if rc == 0:
                done = True
            elif rc not in (_lib.ERR_UNSUPPORTED, _lib.ERR_NOMEM) or mode == 'device':
                _lib.check(rc, 'push_bam_device')
            else:
                # (out of memory: the device form wants up to three slots of scratch - ~1.8 GB of HBM (1.35 of it the inflate kernel's symbol buffer) and 170 MB pinned each -, the host form 2 x 25 B x chunk_records of pinned staging; context and reader are untouched)
                self.ingest_fallback = _lib.last_error()
        if not done:
            _lib.check(self._lib.besst_ctx_push_bam(self._ctx, handle, int(chunk_records), int(head_records), _lib.ptr(rlen),
                                                    _lib.ptr(alen), _lib.ptr(qlen), C.byref(stats)), 'push_bam')
        k = min(head_records, stats.records)
        return stats, rlen[:k], alen[:k], qlen[:k]

    def record_tensors(self):
        """The resident columns as torch tensors that VIEW the context's memory (no copy; valid while the context lives and
        its records do not change) - what pipeline.DeviceRecords.from_columns and the sharded build take."""
        import torch
        n = C.c_int64(0)
        ptrs = np.zeros(8, dtype=np.uint64)
        _lib.check(self._lib.besst_ctx_record_pointers(self._ctx, C.byref(n), _lib.ptr(ptrs)), 'record_pointers')
        spec = (('tid', '<i4'), ('mtid', '<i4'), ('pos', '<i4'), ('mpos', '<i4'), ('tlen', '<i4'), ('flag', '<u2'), ('mapq', '|u1'),
                ('qlen', '<u2'))
        out = {}
        for (name, typestr), ptr in zip(spec, ptrs.tolist()):
            view = _DeviceView(self, int(ptr), int(n.value), typestr)
            t = torch.as_tensor(view, device=torch.device('cuda', self.device))
            if name in ('flag', 'qlen') and t.dtype != torch.uint16:
                t = t.view(torch.uint16)
            out[name] = t
        return out

    def fetch_records(self, first=0, n=None):
        """The resident records as host columns (dict of numpy arrays)."""
        total = C.c_int64(0)
        _lib.check(self._lib.besst_ctx_record_count(self._ctx, C.byref(total)), 'record_count')
        n = total.value - first if n is None else n
        spec = (('tid', np.int32), ('mtid', np.int32), ('pos', np.int32), ('mpos', np.int32), ('tlen', np.int32),
                ('flag', np.uint16), ('mapq', np.uint8), ('qlen', np.uint16))
        cols = {k: np.empty(n, dtype=dt) for k, dt in spec}
        _lib.check(self._lib.besst_ctx_fetch_records(self._ctx, int(first), int(n), *[_lib.ptr(cols[k]) for k, _ in spec]),
                   'fetch_records')
        return cols

    # ---- library statistics ----------------------------------------------------------------------
    @_timed
    def stream_order(self):
        """-> (index of the first record that lies in front of its predecessor in (reference, position) order or None,
        (tid, pos) of the first resident record, of the last) - besst_ctx_stream_order."""
        first = C.c_int64(-1)
        keys = np.zeros(4, dtype=np.int32)
        _lib.check(self._lib.besst_ctx_stream_order(self._ctx, C.byref(first), _lib.ptr(keys[:2]), _lib.ptr(keys[2:])),
                   'stream_order')
        return (None if first.value < 0 else int(first.value)), (int(keys[0]), int(keys[1])), (int(keys[2]), int(keys[3]))

    @_timed
    def metrics_sample(self, top_mask, orientation, min_mapq, read_len, want_isize=True):
        top = _lib.as_col(top_mask, np.uint8)
        isize = np.empty(SAMPLE_CAP, dtype=np.int32)
        contam = np.empty(SAMPLE_CAP, dtype=np.int32)
        counts = MetricsCounts()
        _lib.check(self._lib.besst_ctx_metrics_sample(
            self._ctx, _lib.ptr(top), {'fr': 0, 'rf': 1}[orientation], int(min_mapq), float(read_len),
            int(bool(want_isize)), _lib.ptr(isize), _lib.ptr(contam), C.byref(counts)), 'metrics_sample')
        return isize[:counts.n_isize], contam[:counts.n_contam], counts

    @_timed
    def gap_condition_table(self, mean, sigma, read_len, contig_len, d_lower, n):
        out = np.zeros(int(n), dtype=np.float64)
        _lib.check(self._lib.besst_ctx_gap_condition_table(self._ctx, float(mean), float(sigma), float(read_len),
                                                           float(contig_len), int(d_lower), int(n), _lib.ptr(out)),
                   'gap_condition_table')
        return out

    @_timed
    def value_histogram(self, values, n_bins):
        vals = _lib.as_col(values, np.int32)
        hist = np.zeros(int(n_bins), dtype=np.int64)
        overflow = np.zeros(1, dtype=np.int64)
        _lib.check(self._lib.besst_ctx_value_histogram(self._ctx, _lib.ptr(vals), vals.shape[0], int(n_bins),
                                                       _lib.ptr(hist), _lib.ptr(overflow)), 'value_histogram')
        return hist, int(overflow[0])

    # ---- graph build -----------------------------------------------------------------------------
    @_timed
    def build_graph(self, lazy_observations=False):
        """-> (EdgeTable, coverage numerators, Counters).  lazy_observations (what CreateGraph.PE asks for): the observation
        columns stay on the device - their per-link sum crosses PCIe in the background (ObservationSource) and obs_lo /
        obs_hi only if somebody reads them while the context lives; default: the two columns are fetched with the table."""
        self._detach_observations()                          # (an earlier table's columns, before they are overwritten)
        _lib.check(self._lib.besst_ctx_build_graph(self._ctx), 'build_graph')
        rows, tuples = C.c_int64(), C.c_int64()
        _lib.check(self._lib.besst_ctx_edge_count(self._ctx, C.byref(rows), C.byref(tuples)), 'edge_count')
        r, L = rows.value, tuples.value
        key = np.empty(r, dtype=np.uint64)
        mask = np.empty(r, dtype=np.uint32)
        n = np.empty(r, dtype=np.uint32)
        s1 = np.empty(r, dtype=np.int64)
        s2 = np.empty(r, dtype=np.int64)
        first = np.empty(r, dtype=np.uint32)
        off = np.empty(r, dtype=np.uint32)
        nb = C.c_int32()
        _lib.check(self._lib.besst_ctx_fetch_edges(self._ctx, _lib.ptr(key), _lib.ptr(mask), _lib.ptr(n),
                                                   _lib.ptr(s1), _lib.ptr(s2), _lib.ptr(first), _lib.ptr(off),
                                                   C.byref(nb)), 'fetch_edges')
        lo = hi = None
        if lazy_observations:
            # the observation columns stay on the device: their sum starts crossing in the background now
            self._observations = ObservationSource(self, L)
        else:
            lo = np.empty(L, dtype=np.int32)
            hi = np.empty(L, dtype=np.int32)
            _lib.check(self._lib.besst_ctx_fetch_observations(self._ctx, _lib.ptr(lo), _lib.ptr(hi)), 'fetch_observations')
        aligned = np.empty(self.n_contigs, dtype=np.int64)
        _lib.check(self._lib.besst_ctx_fetch_coverage(self._ctx, _lib.ptr(aligned)), 'fetch_coverage')
        ctr = Counters()
        _lib.check(self._lib.besst_ctx_fetch_counters(self._ctx, C.byref(ctr)), 'fetch_counters')
        return EdgeTable(key, mask, n, s1, s2, first, off, nb.value, lo, hi, source=self._observations), aligned, ctr

    @_timed
    def score_edges(self, rows, swap, len1, len2, mean, sigma, read_len, lognormal=None):
        """Per-edge numbers of GiveScoreOnEdges -> (gap, sd0, ks_h, flags).  lognormal = (ln_mu, ln_sigma, x_max, max_gap)
        selects the log-normal gap estimator (param.lognormal, CreateGraph.py:522-531); sd0 is then 2**32 throughout -
        the caller looks the conditional sigma up with the gap (conditional_stddevs)."""
        rows = _lib.as_col(rows, np.uint32)
        swap = _lib.as_col(swap, np.uint8)
        len1 = _lib.as_col(len1, np.int32)
        len2 = _lib.as_col(len2, np.int32)
        m = rows.shape[0]
        gap = np.empty(m, dtype=np.float64)
        ks_h = np.empty(m, dtype=np.int32)
        flags = np.empty(m, dtype=np.uint8)
        if lognormal is not None:
            ln_mu, ln_sigma, x_max, max_gap = lognormal
            _lib.check(self._lib.besst_ctx_score_edges_lognormal(
                self._ctx, m, _lib.ptr(rows), _lib.ptr(swap), _lib.ptr(len1), _lib.ptr(len2), float(mean), float(sigma),
                float(read_len), float(ln_mu), float(ln_sigma), int(x_max), int(max_gap), _lib.ptr(gap), _lib.ptr(ks_h),
                _lib.ptr(flags)), 'score_edges_lognormal')
            return gap, np.full(m, 2.0 ** 32), ks_h, flags
        sd0 = np.empty(m, dtype=np.float64)
        _lib.check(self._lib.besst_ctx_score_edges(self._ctx, m, _lib.ptr(rows), _lib.ptr(swap), _lib.ptr(len1),
                                                   _lib.ptr(len2), float(mean), float(sigma), float(read_len),
                                                   _lib.ptr(gap), _lib.ptr(sd0), _lib.ptr(ks_h), _lib.ptr(flags)),
                   'score_edges')
        return gap, sd0, ks_h, flags

    @_timed
    def conditional_stddevs(self, density, steps):
        """sigma of density[x] * max(0, x - gap + 1) over x for every gap of `steps` (get_conditional_stddevs,
        CreateGraph.py:436-469); density: float64, index = insert size."""
        density = _lib.as_col(density, np.float64)
        steps = _lib.as_col(steps, np.int32)
        out = np.empty(steps.shape[0], dtype=np.float64)
        _lib.check(self._lib.besst_ctx_conditional_stddevs(self._ctx, _lib.ptr(density), density.shape[0] - 1,
                                                           _lib.ptr(steps), steps.shape[0], _lib.ptr(out)),
                   'conditional_stddevs')
        return out
