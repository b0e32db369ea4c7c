"""Device-resident graph build on caller-owned HBM (torch tensors) through the besst_dev_* layer.

torch is used for what it is good at here - device memory, streams and (in distributed.py)
the RCCL process group.  Every kernel is a hand-written HIP kernel behind the C ABI; tensors
only lend their ``data_ptr()``.  Used by bench.py, the multi-GPU path and the GPU tests; the
ctypes drop-in (CreateGraph.PE) uses the host-buffer layer in device.py instead.
"""
import ctypes as C
import os
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import Counters, LibParams

COUNTER_BYTES = C.sizeof(Counters)
# include/besst_amd.h: what *n_rows carries when stage 2 could not build the table, and the flag that answers the second
ROWS_SORT_FAILED = 0xFFFFFFFF
ROWS_RUN_OVERFLOW = 0xFFFFFFFE
REDUCE_NO_RUNS = 1


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _from_np(arr, device):
    """Upload a numpy column; 16-bit unsigned columns travel as int16 bit patterns."""
    if arr.dtype == np.uint16:
        arr = arr.view(np.int16)
    return torch.from_numpy(np.ascontiguousarray(arr)).to(device, non_blocking=False)


class DeviceRecords(object):
    """Record columns resident in HBM."""

    def __init__(self, batch, device):
        self.n = len(batch)
        self.tid = _from_np(batch.tid, device)
        self.mtid = _from_np(batch.mtid, device)
        self.pos = _from_np(batch.pos, device)
        self.mpos = _from_np(batch.mpos, device)
        self.tlen = _from_np(batch.tlen, device)
        self.flag = _from_np(batch.flag, device)
        self.mapq = _from_np(batch.mapq, device)
        self.qlen = _from_np(batch.qlen, device)

        self._make_bits()

    @classmethod
    def from_columns(cls, cols, copy=False):
        """Columns that already live on the device (synth.simulate_library_device)."""
        self = cls.__new__(cls)
        self.n = int(cols['tid'].shape[0])
        for k in ('tid', 'mtid', 'pos', 'mpos', 'tlen', 'flag', 'mapq', 'qlen'):
            setattr(self, k, cols[k].clone() if copy else cols[k])
        self._make_bits()
        return self

    def _make_bits(self):
        """The ninth column of the resident layout: one bit per record, set where the mate lies on another reference
        (tid != mtid) - made once, here, when the records become resident (besst_dev_mate_bits; 1/8 byte per record).  The
        record loop then reads `mtid` only where a lane holds such a record (besst_lib_params.mate_bits).
        BESST_MATE_BITS=0: no bit column, the loop compares tid and mtid itself (tests compare the two forms)."""
        self.mate_bits = None
        if self.n == 0 or not self.tid.is_cuda or os.environ.get('BESST_MATE_BITS') == '0':
            return
        lib = _lib.load()
        dev = self.tid.device
        bits = torch.empty(int(lib.besst_dev_mate_bits_bytes(self.n)), dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.besst_dev_mate_bits(C.c_void_p(stream), self.n, _p(self.tid), _p(self.mtid), _p(bits)), 'dev_mate_bits')
        self.mate_bits = bits

    @property
    def mate_bits_ptr(self):
        return None if self.mate_bits is None else self.mate_bits.data_ptr()

    @property
    def graph_bytes(self):
        return self.n * 19


class DeviceGraphBuilder(object):
    """classify -> (optional exchange) -> sort/reduce on one GPU, no host round trip inside."""

    def __init__(self, device, n_contigs, node_bits, lib, record_capacity, tuple_capacity):
        self.lib = _lib.load()
        self.device = device
        self.n_contigs = int(n_contigs)
        self.node_bits = int(node_bits)
        self.params = LibParams(float(lib['read_len']), float(lib['ins_size_threshold']), int(lib['min_mapq']),
                                {'fr': 0, 'rf': 1}[lib['orientation']], int(bool(lib['detect_duplicate'])),
                                int(bool(lib['extend_paths'])), int(bool(lib['no_score'])), 0)
        self.rec_cap = int(max(1, record_capacity))
        self.tup_cap = int(max(1, tuple_capacity))
        u8 = dict(dtype=torch.uint8, device=device)
        i64 = dict(dtype=torch.int64, device=device)
        i32 = dict(dtype=torch.int32, device=device)
        self.table = torch.zeros(self.lib.besst_dev_contig_table_bytes(self.n_contigs), **u8)
        # ONE int64 block holds everything a pass starts from zero: coverage numerators | 8 summable counter words |
        # prev_obs words, carry[2], n_out, n_rows | 2 spare words.  reset() is a single device copy from a template,
        # and the multi-GPU path all-reduces state[:n_contigs + 8] in place (coverage and counters in one call).
        self.small_words = (COUNTER_BYTES + 32) // 8
        self.state = torch.zeros(self.n_contigs + self.small_words + 2, **i64)
        self.aligned = self.state[:self.n_contigs]
        self.small = self.state[self.n_contigs:self.n_contigs + self.small_words].view(torch.uint8)
        self.spare = self.state[self.n_contigs + self.small_words:].view(torch.int32)     # 4 x int32 for callers
        self.keys = torch.empty(self.rec_cap, **i64)
        self.payload = torch.empty(self.rec_cap, **i64)
        self.ws1 = torch.empty(self.lib.besst_dev_classify_workspace_bytes(self.rec_cap), **u8)
        self.ws2 = torch.empty(self.lib.besst_dev_reduce_workspace_bytes(self.tup_cap), **u8)
        c = self.tup_cap
        self.row_key = torch.empty(c, **i64)
        self.row_mask = torch.empty(c, **i32)
        self.row_n = torch.empty(c, **i32)
        self.row_sum = torch.empty(c, **i64)
        self.row_sum_sq = torch.empty(c, **i64)
        self.row_first = torch.empty(c, **i32)
        self.row_offset = torch.empty(c, **i32)
        self.obs_lo = torch.empty(c, **i32)
        self.obs_hi = torch.empty(c, **i32)
        init = torch.zeros(self.state.numel(), dtype=torch.int64)
        init_small = init[self.n_contigs:self.n_contigs + self.small_words].view(torch.uint8)
        init_small[COUNTER_BYTES:COUNTER_BYTES + 8] = torch.from_numpy(np.array([-1, -1], dtype=np.int32).view(np.uint8))
        self._init = init.to(device)
        self._args = {}
        # per record set: the marshalled argument list and the form of the record loop, dropped with the record set
        # (keyed by the object, weakly: an id() could be inherited by a later record set at the same address)
        self._rec_args = weakref.WeakKeyDictionary()
        self._paths = weakref.WeakKeyDictionary()
        self.key_base, self.key_bits = 0, 2 * self.node_bits + 1
        self._density = torch.zeros(2, dtype=torch.int64, device=device)
        self._presorted = False
        self._last_rec = None
        self.keys_valid = False                              # self.keys holds the last pass's dense tuple stream
        self.payload_valid = False                           # ... and self.payload (include/besst_amd.h, besst_presort: not after a pass whose record loop kept its tuples in the block segments)
        self.candidate_share = None
        # BESST_REDUCE_* flags of this builder's stage-2 calls.  A large stream is first reduced in the run-grouped form;
        # if its keys do not cluster (read_sizes() sees BESST_ROWS_RUN_OVERFLOW) the builder repeats the call with
        # BESST_REDUCE_NO_RUNS and keeps that flag for the passes that follow.
        self.sort_flags = 0
        self._redo = None

    # device addresses inside the small block
    def _small(self, off):
        return C.c_void_p(self.small.data_ptr() + off)

    @property
    def _carry(self):
        return self._small(COUNTER_BYTES)

    @property
    def _n_out(self):
        return self._small(COUNTER_BYTES + 8)

    @property
    def _n_rows(self):
        """Device address of the row count.  Read it through read_sizes(): the word also carries ROWS_RUN_OVERFLOW /
        ROWS_SORT_FAILED, which only read_sizes() acts on."""
        return self._small(COUNTER_BYTES + 12)

    def set_contigs(self, scaf_id, scaf_len, ctg_pos, ctg_len, direction, cls):
        cols = [_lib.as_col(scaf_id, np.int32), _lib.as_col(scaf_len, np.int32), _lib.as_col(ctg_pos, np.int32),
                _lib.as_col(ctg_len, np.int32), _lib.as_col(direction, np.uint8), _lib.as_col(cls, np.uint8)]
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self.lib.besst_dev_pack_contigs(C.c_void_p(stream), self.n_contigs, *[_lib.ptr(c) for c in cols],
                                                   _p(self.table)), 'pack_contigs')
        # key range of this table (include/besst_amd.h, besst_dev_reduce): scaffold ids of a later library start far
        # above 1, so the sort works on key - key_base
        present = cols[5] != 0
        if present.any():
            lo, hi = int(cols[0][present].min()), int(cols[0][present].max())
            top = hi * 2 + 1
            self.key_base = ((lo * 2) << self.node_bits) << 1
            self.key_bits = max(1, int(((((top << self.node_bits) | top) << 1) | 1) - self.key_base).bit_length())
            self._args.pop('reduce', None)
            self._args.pop('presort', None)

    def reset(self):
        """Zero coverage / counters, prev_obs = (-1, -1) (CreateGraph.py:89-99): the state block restored from its template."""
        args = self._args.get('reset')
        if args is None:
            args = self._args['reset'] = (_p(self.state), _p(self._init), self.state.numel() * 8)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self.lib.besst_dev_restore_state(C.c_void_p(stream), *args), 'restore_state')

    def record_path(self, rec):
        """Which form of the record loop serves this record set (include/besst_amd.h, besst_lib_params.record_path):
        sampled once per record set with besst_dev_candidate_density (synchronises), BESST_RECORD_PATH overrides."""
        path = self._paths.get(rec)
        if path is None:
            forced = os.environ.get('BESST_RECORD_PATH')
            if forced in ('0', '1'):
                path = int(forced)
            else:
                share, out = C.c_double(0.0), C.c_int32(0)
                stream = torch.cuda.current_stream(self.device).cuda_stream
                _lib.check(self.lib.besst_dev_candidate_density(C.c_void_p(stream), rec.n, _p(rec.tid), _p(rec.mtid),
                                                                4 << 20, C.c_void_p(self._density.data_ptr()),
                                                                C.byref(share), C.byref(out)), 'candidate_density')
                path = int(out.value)
                self.candidate_share = float(share.value)
            self._paths[rec] = path
        return path

    def classify(self, rec, presort=False):
        import weakref
        self._last_rec = weakref.ref(rec)
        self.params.record_path = self.record_path(rec)
        self.params.mate_bits = rec.mate_bits_ptr            # (the struct the marshalled argument lists point to)
        # argument lists are marshalled once per record set (every buffer is allocated once)
        args = self._rec_args.get(rec)
        if args is None:
            args = self._rec_args[rec] = (
                rec.n, _p(rec.tid), _p(rec.mtid), _p(rec.pos), _p(rec.mpos), _p(rec.flag), _p(rec.mapq), _p(rec.qlen),
                self.n_contigs, _p(self.table), C.byref(self.params), self.node_bits, self._carry, _p(self.aligned),
                _p(self.keys), _p(self.payload), self._n_out, self._small(0), _p(self.ws1), self.ws1.numel())
        stream = torch.cuda.current_stream(self.device).cuda_stream
        ref = self._presort_ref(presort)
        _lib.check(self.lib.besst_dev_classify_presort(C.c_void_p(stream), *args, ref), 'dev_classify')
        # the next reduce() of the builder's own tuples may trust the table (and, spec.segmented, must read the tuples
        # from the block segments: self.keys was not written then)
        self._presorted = ref is not None
        self.keys_valid = not (ref is not None and self._args['presort'][0].segmented)
        # (segmented with in_record_loop 1: stage 2 writes the dense payload; 2 or 3: nobody does)
        self.payload_valid = self.keys_valid or self._args['presort'][0].in_record_loop == 1

    def _presort_ref(self, on):
        """The hand-over of the sort's digit histograms from stage 1 to stage 2 (include/besst_amd.h, besst_presort)
        for this builder's own tuple buffers, or NULL when stage 2 would not use it."""
        if not on:
            return None
        spec = self._args.get('presort')
        if spec is None:
            spec = _lib.Presort()
            used = self.lib.besst_dev_reduce_presort(self.tup_cap, self.key_bits, self.key_base, _p(self.ws2),
                                                     self.ws2.numel(), C.byref(spec))
            spec = self._args['presort'] = (spec, bool(used == 1 and spec.table))
        if not spec[1]:
            return None
        spec[0].segmented = 1                               # asked for anew on every pass (classify answers in place)
        # the record loop has to know which form of stage 2 follows (with BESST_REDUCE_NO_RUNS it counts the sort's
        # digits and writes keys; without, it groups the runs itself)
        spec[0].flags = self.sort_flags
        return C.byref(spec[0])

    def reduce(self, keys=None, payload=None, n_tuples_ptr=None, capacity=None, first_map=None):
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if keys is None and payload is None and n_tuples_ptr is None and capacity is None and first_map is None:
            args = self._args.get('reduce')
            if args is None:
                args = self._args['reduce'] = (
                    self.tup_cap, self._n_out, self.key_bits, _p(self.keys), _p(self.payload),
                    _p(self.row_key), _p(self.row_mask), _p(self.row_n), _p(self.row_sum), _p(self.row_sum_sq),
                    _p(self.row_first), _p(self.row_offset), _p(self.obs_lo), _p(self.obs_hi), self._n_rows,
                    _p(self.ws2), self.ws2.numel(), None, self.key_base)
            if self._presorted:
                self._presorted = False
                spec = self._args['presort'][0]

                def again():
                    # (valid for reset -> classify -> reduce passes, which is what step() runs: the repeat below resets the
                    # state and classifies the LAST record set only - a caller that classified several record sets into one
                    # state without reset() must repeat them itself, with sort_flags |= REDUCE_NO_RUNS set beforehand)
                    spec.flags = self.sort_flags
                    if spec.in_record_loop >= 2 and (spec.flags & REDUCE_NO_RUNS):
                        # the record loop handed its segments over without the sort's digit counts (stage 2 was going to
                        # group runs): the pass is repeated from its start, counting this time
                        rec = self._last_rec() if self._last_rec is not None else None
                        if rec is None:
                            raise _lib.BesstDeviceError('reduce: the record set of the pass to repeat is gone')
                        self.reset()
                        self.classify(rec, presort=True)
                        self._presorted = False
                    st = torch.cuda.current_stream(self.device).cuda_stream
                    _lib.check(self.lib.besst_dev_reduce_presorted(C.c_void_p(st), *args, C.byref(spec)), 'dev_reduce')
            else:
                def again():
                    st = torch.cuda.current_stream(self.device).cuda_stream
                    _lib.check(self.lib.besst_dev_reduce_flags(C.c_void_p(st), *args, self.sort_flags), 'dev_reduce')
            self._redo = again
            again()
            return
        self._presorted = False
        if (keys is None and not self.keys_valid) or (payload is None and keys is not None and not self.payload_valid):
            raise _lib.BesstDeviceError('reduce: the last classify(presort=True) left its tuples in the block segments; '
                                        'self.keys holds an earlier pass (classify without presort to get the dense stream)')
        keys = self.keys if keys is None else keys
        payload = self.payload if payload is None else payload
        cap = self.tup_cap if capacity is None else int(capacity)
        args = (cap, n_tuples_ptr or self._n_out, self.key_bits, _p(keys), _p(payload),
                _p(self.row_key), _p(self.row_mask), _p(self.row_n), _p(self.row_sum), _p(self.row_sum_sq),
                _p(self.row_first), _p(self.row_offset), _p(self.obs_lo), _p(self.obs_hi), self._n_rows,
                _p(self.ws2), self.ws2.numel(), _p(first_map), self.key_base)

        def again():
            st = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self.lib.besst_dev_reduce_flags(C.c_void_p(st), *args, self.sort_flags), 'dev_reduce')
        self._redo = again
        again()

    def lognormal_tables(self, ln_mu, ln_sigma, x_max):
        """Prefix tables of the log-normal pmf on this builder's device (besst_dev_lognormal_tables), kept while the
        library's parameters stay the same -> (F0, F1) float64 tensors of x_max + 1 entries."""
        key = (float(ln_mu), float(ln_sigma), int(x_max))
        if getattr(self, '_ln_key', None) != key:
            dev = self.device
            F = torch.empty(2 * (key[2] + 1), dtype=torch.float64, device=dev)
            ws = torch.empty(max(256, int(self.lib.besst_dev_lognormal_tables_workspace_bytes(key[2]))), dtype=torch.uint8,
                             device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(self.lib.besst_dev_lognormal_tables(C.c_void_p(stream), key[0], key[1], key[2], _p(F),
                                                           C.c_void_p(F.data_ptr() + 8 * (key[2] + 1)), _p(ws), ws.numel()),
                       'dev_lognormal_tables')
            torch.cuda.current_stream(dev).synchronize()         # (ws goes out of scope)
            self._ln_key, self._ln_F = key, F
        return self._ln_F[:key[2] + 1], self._ln_F[key[2] + 1:]

    def score_edges(self, rows, swap, len1, len2, mean, sigma, read_len, lognormal=None):
        """Score rows of the table this builder holds (besst_dev_score_edges / _lognormal) -> (gap, sd0, ks_h, flags)
        numpy arrays.  lognormal = (ln_mu, ln_sigma, x_max, max_gap): the log-normal gap estimator (sd0 is then 2**32).
        Not part of the timed graph build; synchronises (the scratch layout needs the link counts on the host)."""
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        m = int(rows.shape[0])
        if m == 0:
            return (np.zeros(0), np.zeros(0), np.zeros(0, np.int32), np.zeros(0, np.uint8))
        dev = self.device
        n_links = self.row_n.cpu().numpy().view(np.uint32)[rows].astype(np.int64)
        if (n_links < 1).any():
            raise _lib.BesstDeviceError('score_edges: empty row')
        npow = np.where(n_links > 1, 1 << np.ceil(np.log2(np.maximum(n_links, 2))).astype(np.int64), 1)
        big = np.where(npow > 8192, 2 * npow, 0)
        big_off = (np.cumsum(big) - big).astype(np.uint64)
        off_bytes = (m * 8 + 255) // 256 * 256
        ws = torch.zeros(off_bytes + (int(big.sum()) * 4 + 256 + 255) // 256 * 256 + (off_bytes if lognormal is not None else 0),
                         dtype=torch.uint8, device=dev)
        ws[:m * 8] = torch.from_numpy(big_off.view(np.uint8)).to(dev)
        d = [torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
             for a, dt in ((rows.view(np.int32), np.int32), (swap, np.uint8), (len1, np.int32), (len2, np.int32))]
        gap = torch.zeros(m, dtype=torch.float64, device=dev)
        sd0 = torch.zeros(m, dtype=torch.float64, device=dev)
        ks = torch.zeros(m, dtype=torch.int32, device=dev)
        flags = torch.zeros(m, dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        if lognormal is not None:
            ln_mu, ln_sigma, x_max, max_gap = lognormal
            F0, F1 = self.lognormal_tables(ln_mu, ln_sigma, x_max)
            _lib.check(self.lib.besst_dev_score_edges_lognormal(
                C.c_void_p(stream), m, _p(d[0]), _p(d[1]), _p(d[2]), _p(d[3]), _p(self.row_n), _p(self.row_sum),
                _p(self.row_offset), _p(self.obs_lo), _p(self.obs_hi), float(mean), float(sigma), float(read_len),
                float(ln_mu), float(ln_sigma), int(x_max), _p(F0), _p(F1), int(max_gap), _p(gap), _p(ks), _p(flags), _p(ws),
                ws.numel()), 'dev_score_edges_lognormal')
            return gap.cpu().numpy(), np.full(m, 2.0 ** 32), ks.cpu().numpy(), flags.cpu().numpy()
        _lib.check(self.lib.besst_dev_score_edges(
            C.c_void_p(stream), m, _p(d[0]), _p(d[1]), _p(d[2]), _p(d[3]), _p(self.row_n), _p(self.row_sum),
            _p(self.row_offset), _p(self.obs_lo), _p(self.obs_hi), float(mean), float(sigma), float(read_len),
            _p(gap), _p(sd0), _p(ks), _p(flags), _p(ws), ws.numel()), 'dev_score_edges')
        return gap.cpu().numpy(), sd0.cpu().numpy(), ks.cpu().numpy(), flags.cpu().numpy()

    def step(self, rec):
        self.reset()
        self.classify(rec, presort=True)
        self.reduce()

    def read_sizes(self):
        """(n_tuples, n_rows) - synchronises.  Where stage 2 reported a stream that does not fit its run-grouped form,
        the call is repeated tuple by tuple (once; ``sort_flags`` remembers); a sort that gave up raises."""
        for attempt in range(2):
            raw = self.small.cpu().numpy()
            n_out, n_rows = np.frombuffer(raw[COUNTER_BYTES + 8:COUNTER_BYTES + 16].tobytes(), dtype=np.uint32)
            if int(n_rows) == ROWS_RUN_OVERFLOW and attempt == 0 and self._redo is not None \
                    and not (self.sort_flags & REDUCE_NO_RUNS):
                self.sort_flags |= REDUCE_NO_RUNS
                self._redo()
                continue
            break
        if int(n_rows) in (ROWS_SORT_FAILED, ROWS_RUN_OVERFLOW):
            raise _lib.BesstDeviceError('stage 2 could not build the edge table (n_rows word 0x%08x): %s' % (
                int(n_rows), 'a chained-scan look-back gave up' if int(n_rows) == ROWS_SORT_FAILED else 'run overflow'))
        return int(n_out), int(n_rows)

    def read_counters(self):
        raw = self.small.cpu().numpy().tobytes()
        ctr = Counters.from_buffer_copy(raw[:COUNTER_BYTES])
        carry = np.frombuffer(raw[COUNTER_BYTES:COUNTER_BYTES + 8], dtype=np.int32)
        ctr.prev_obs1, ctr.prev_obs2 = int(carry[0]), int(carry[1])
        return ctr

    def fetch_table(self):
        """EdgeTable on the host (synchronises)."""
        from .device import EdgeTable
        L, r = self.read_sizes()
        if L > self.tup_cap:
            raise _lib.BesstDeviceError('tuple capacity %d exceeded (%d tuples)' % (self.tup_cap, L))

        def h(t, n, dt):
            return t[:n].cpu().numpy().view(dt)
        return EdgeTable(h(self.row_key, r, np.uint64), h(self.row_mask, r, np.uint32), h(self.row_n, r, np.uint32),
                         h(self.row_sum, r, np.int64), h(self.row_sum_sq, r, np.int64),
                         h(self.row_first, r, np.uint32), h(self.row_offset, r, np.uint32), self.node_bits,
                         h(self.obs_lo, L, np.int32), h(self.obs_hi, L, np.int32))


SAMPLE_CAP = 1000000     # libmetrics' "first 1,000,000" cut-offs (libmetrics.py:83,302)


class DeviceMetricsSampler(object):
    """libmetrics scans of one resident slice through besst_dev_metrics_sample (see include/besst_amd.h).

    ``samples`` holds the insert-size sample in [0, 1e6) and the contamination sample in [1e6, 2e6); ``state`` is
    the 6-word running state of the scan (counts of earlier slices in, counts including this slice out)."""

    def __init__(self, device, rec, n_contigs):
        self.lib = _lib.load()
        self.device = device
        self.rec = rec
        self.n_contigs = int(n_contigs)
        self.samples = torch.zeros(2 * SAMPLE_CAP, dtype=torch.int32, device=device)
        self.state = torch.zeros(8, dtype=torch.int64, device=device)
        self.top = torch.zeros(self.n_contigs, dtype=torch.uint8, device=device)
        self.ws = torch.empty(self.lib.besst_dev_metrics_workspace_bytes(max(1, rec.n)), dtype=torch.uint8, device=device)

    def set_top(self, top_mask):
        self.top.copy_(torch.from_numpy(np.ascontiguousarray(top_mask, np.uint8)))

    def _launch(self, orientation, min_mapq, read_len, count_only, want_isize):
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rec = self.rec
        isize = C.c_void_p(self.samples.data_ptr()) if want_isize else None
        contam = C.c_void_p(self.samples.data_ptr() + 4 * SAMPLE_CAP)
        _lib.check(self.lib.besst_dev_metrics_sample(
            C.c_void_p(stream), rec.n, _p(rec.tid), _p(rec.mtid), _p(rec.tlen), _p(rec.flag), _p(rec.mapq),
            self.n_contigs, _p(self.top), {'fr': 0, 'rf': 1}[orientation], int(min_mapq), float(read_len),
            1 if count_only else 0, isize, contam, _p(self.state), _p(self.ws), self.ws.numel()), 'dev_metrics_sample')

    def count(self, orientation, min_mapq, read_len):
        """state[0:3] <- this slice's (isize-qualifying, top-contig, contamination) record counts."""
        self.state.zero_()
        self._launch(orientation, min_mapq, read_len, True, True)
        return self.state[:3]

    def emit(self, before, orientation, min_mapq, read_len, want_isize=True):
        """Write this slice's share of the two samples, given the counts of the slices before it."""
        self.samples.zero_()
        self.state.zero_()
        self.state[:3] = before
        self._launch(orientation, min_mapq, read_len, False, want_isize)
        return self.samples, self.state


_POOL_STREAMS = {}


class PassPool(object):
    """Independent graph-build passes in flight on separate HIP streams (one builder and workspace per slot).

    A pass over a sparse (paired-end) library is one bandwidth-bound kernel followed by a chain of short,
    latency-bound ones; with several libraries to process (BASELINE configs 4 and 5 have two and three), the short
    kernels of one pass run under the streaming kernel of another.  ``submit(rec)`` enqueues a pass on the next slot
    and returns that slot's builder, whose outputs are valid once its stream - or the device - has been
    synchronised and until the slot is reused ``in_flight`` submissions later."""

    def __init__(self, device, n_contigs, node_bits, lib, record_capacity, tuple_capacity, in_flight=3):
        self.device = device
        # One PRIORITY LEVEL per stream where there are enough levels: the runtime spreads a process's streams over a few
        # hardware queues per level, in creation order, and two passes that land on one queue run one after the other (the
        # "three in flight" figure of one bench run read 3.39 ms, of the next 1.58).  Levels are used for placement only - the
        # passes are peers.
        try:
            least, greatest = torch.cuda.Stream.priority_range()
            levels = list(range(greatest, least + 1))
        except Exception:                                    # (older torch: no priority_range)
            levels = [0]
        n = max(1, int(in_flight))
        # (and the streams are the PROCESS's, made once per device and position: a second pool - another library, another
        # config - that made three more would shift everybody's queues again)
        self.streams = []
        for k in range(n):
            key = (str(device), k)
            if key not in _POOL_STREAMS:
                _POOL_STREAMS[key] = torch.cuda.Stream(device, priority=levels[k % len(levels)])
            self.streams.append(_POOL_STREAMS[key])
        self.builders = []
        for st in self.streams:
            with torch.cuda.stream(st):
                self.builders.append(DeviceGraphBuilder(device, n_contigs, node_bits, lib, record_capacity,
                                                        tuple_capacity))
        self._next = 0

    def set_contigs(self, **table):
        for st, gb in zip(self.streams, self.builders):
            with torch.cuda.stream(st):
                gb.set_contigs(**table)

    def submit(self, rec):
        slot = self._next % len(self.builders)
        self._next += 1
        with torch.cuda.stream(self.streams[slot]):
            self.builders[slot].step(rec)
        return self.builders[slot]

    def synchronize(self):
        for st in self.streams:
            st.synchronize()


def prof_collect():
    lib = _lib.load()
    n = lib.besst_prof_slots()
    ms = np.zeros(n, dtype=np.float64)
    cnt = np.zeros(n, dtype=np.int64)
    _lib.check(lib.besst_prof_collect(n, _lib.ptr(ms), _lib.ptr(cnt)), 'prof_collect')
    return {lib.besst_prof_slot_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(n) if cnt[i]}
