"""TEST INFRASTRUCTURE - ctypes wrapper of oracle/libbesst_oracle.so (see besst_oracle.c)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libbesst_oracle.so')
_lib = None


class Params(C.Structure):
    _fields_ = [('read_len', C.c_double), ('ins_size_threshold', C.c_double), ('min_mapq', C.c_int32),
                ('rf', C.c_int32), ('detect_duplicate', C.c_int32), ('extend_paths', C.c_int32),
                ('no_score', C.c_int32), ('node_bits', C.c_int32)]


def load():
    global _lib
    if _lib is None:
        if not os.path.isfile(_SO):
            subprocess.check_call(['make', '-s', '-C', _HERE])
        _lib = C.CDLL(_SO)
        _lib.oracle_record_loop.restype = C.c_int64
        _lib.oracle_record_loop_mt.restype = C.c_int64
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def record_loop(batch, table, lib, node_bits, prev=(-1, -1), threads=1):
    """Ordered tuple stream + coverage + counters for a RecordBatch.  table: dict of numpy columns.
    threads > 1 runs the slice-parallel variant (same results; bench.py's "all host cores" baseline)."""
    L = load()
    n = len(batch)
    cols = [np.ascontiguousarray(batch.tid, np.int32), np.ascontiguousarray(batch.mtid, np.int32),
            np.ascontiguousarray(batch.pos, np.int32), np.ascontiguousarray(batch.mpos, np.int32),
            np.ascontiguousarray(batch.flag, np.uint16), np.ascontiguousarray(batch.mapq, np.uint8),
            np.ascontiguousarray(batch.qlen, np.uint16)]
    tab = [np.ascontiguousarray(table['cls'], np.uint8), np.ascontiguousarray(table['scaf_id'], np.int32),
           np.ascontiguousarray(table['scaf_len'], np.int32), np.ascontiguousarray(table['ctg_pos'], np.int32),
           np.ascontiguousarray(table['ctg_len'], np.int32), np.ascontiguousarray(table['direction'], np.uint8)]
    nc = tab[0].shape[0]
    p = Params(float(lib['read_len']), float(lib['ins_size_threshold']), int(lib['min_mapq']),
               1 if lib['orientation'] == 'rf' else 0, int(bool(lib['detect_duplicate'])),
               int(bool(lib['extend_paths'])), int(bool(lib['no_score'])), int(node_bits))
    aligned = np.zeros(nc, np.int64)
    counters = np.zeros(10, np.int64)
    counters[8], counters[9] = prev
    keys = np.empty(max(n, 1), np.uint64)
    payload = np.empty(max(n, 1), np.uint64)
    if threads > 1:
        nt = L.oracle_record_loop_mt(C.c_int(int(threads)), C.c_int64(n), *[_p(c) for c in cols], C.c_int64(nc),
                                     *[_p(t) for t in tab], C.byref(p), _p(aligned), _p(counters), _p(keys), _p(payload))
    else:
        nt = L.oracle_record_loop(C.c_int64(n), *[_p(c) for c in cols], C.c_int64(nc), *[_p(t) for t in tab],
                                  C.byref(p), _p(aligned), _p(counters), _p(keys), _p(payload))
    return keys[:nt].copy(), payload[:nt].copy(), aligned, counters


def edge_rows(keys, payload):
    """Aggregate the ordered tuple stream into rows sorted by key (numpy; exact integer sums)."""
    order = np.argsort(keys, kind='stable')
    k = keys[order]
    lo = (payload[order] & np.uint64(0xffffffff)).astype(np.int64)
    hi32 = (payload[order] >> np.uint64(32)).astype(np.int64)
    hi = hi32 & 0x3fffffff
    mask = hi32 >> 30
    heads = np.ones(k.shape[0], bool)
    heads[1:] = k[1:] != k[:-1]
    starts = np.nonzero(heads)[0]
    o = lo + hi
    return dict(key=k[starts], n=np.diff(np.append(starts, k.shape[0])), sum_obs=np.add.reduceat(o, starts) if len(starts) else o[:0],
                sum_obs_sq=np.add.reduceat(o * o, starts) if len(starts) else o[:0], first_idx=order[starts],
                offset=starts, mask=mask[starts], obs_lo=lo, obs_hi=hi)


def metrics_sample(batch, top_mask, orientation, min_mapq, read_len, want_isize=True):
    L = load()
    n = len(batch)
    cols = [np.ascontiguousarray(batch.tid, np.int32), np.ascontiguousarray(batch.mtid, np.int32),
            np.ascontiguousarray(batch.tlen, np.int32), np.ascontiguousarray(batch.flag, np.uint16),
            np.ascontiguousarray(batch.mapq, np.uint8)]
    top = np.ascontiguousarray(top_mask, np.uint8)
    isize = np.empty(1000000, np.int32)
    contam = np.empty(1000000, np.int32)
    counts = np.zeros(4, np.int64)
    L.oracle_metrics_sample(C.c_int64(n), *[_p(c) for c in cols], C.c_int64(top.shape[0]), _p(top),
                            C.c_int(1 if orientation == 'rf' else 0), C.c_int32(int(min_mapq)), C.c_double(float(read_len)),
                            C.c_int(int(bool(want_isize))), _p(isize), _p(contam), _p(counts))
    return isize[:counts[0]].copy(), contam[:counts[1]].copy(), counts
