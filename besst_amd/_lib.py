"""ctypes binding of libbesst_amd.so (the C ABI declared in include/besst_amd.h).

The reference reaches its one native helper the same way - ``ctypes.CDLL`` plus a
caller-allocated result struct (BESST/diploid/wrapper_sw.py:12-24).  There is no CPU
fallback: if the shared library is missing or no MI355X is visible, loading or context
creation raises :class:`BesstDeviceError`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BESST_AMD_LIB: another build of the same library (A/B runs of kernel variants, tools/variant.sh)
LIB_PATH = os.environ.get('BESST_AMD_LIB') or os.path.join(_HERE, 'libbesst_amd.so')


class BesstDeviceError(RuntimeError):
    """``status``: the library's return code (include/besst_amd.h, BESST_ERR_*) when the error comes from a C-ABI call, else
    None."""

    def __init__(self, message, status=None):
        RuntimeError.__init__(self, message)
        self.status = status


class RankFailure(BesstDeviceError):
    """A collective stage of the sharded path failed on some rank; raised on EVERY rank (besst_amd.sharded / distributed)."""
    on_every_rank = True


ERR_UNSUPPORTED = 5     # include/besst_amd.h: BESST_ERR_UNSUPPORTED
ERR_NOMEM = 4           # include/besst_amd.h: BESST_ERR_NOMEM


class LibParams(C.Structure):
    _fields_ = [('read_len', C.c_double), ('ins_size_threshold', C.c_double), ('min_mapq', C.c_int32),
                ('orientation', C.c_int32), ('detect_duplicate', C.c_int32), ('extend_paths', C.c_int32),
                ('no_score', C.c_int32), ('record_path', C.c_int32), ('mate_bits', C.c_void_p)]


class Presort(C.Structure):          # include/besst_amd.h: besst_presort
    _fields_ = [('table', C.c_void_p), ('rows', C.c_int32), ('shift', C.c_int32), ('key_base', C.c_uint64),
                ('capacity', C.c_uint32), ('flags', C.c_uint32), ('segmented', C.c_int32), ('in_record_loop', C.c_int32),
                ('seg_keys', C.c_void_p), ('seg_payload', C.c_void_p), ('seg_offsets', C.c_void_p), ('seg_skip', C.c_void_p),
                ('seg_blocks', C.c_uint32), ('seg_tile', C.c_uint32), ('payload_out', C.c_void_p),
                ('seg_chunk_first', C.c_void_p), ('seg_run_offsets', C.c_void_p), ('seg_summ', C.c_void_p),
                ('seg_summ_stride', C.c_uint32), ('seg_run_status', C.c_void_p)]


class Counters(C.Structure):
    _fields_ = [('count', C.c_int64), ('non_unique', C.c_int64), ('non_unique_for_scaf', C.c_int64),
                ('nr_of_duplicates', C.c_int64), ('reads_with_too_long_insert', C.c_int64),
                ('fishy_reads', C.c_int64), ('n_tuples', C.c_int64), ('n_reach', C.c_int64),
                ('prev_obs1', C.c_int32), ('prev_obs2', C.c_int32)]


class IngestStats(C.Structure):      # include/besst_amd.h: besst_ingest_stats
    _fields_ = [('records', C.c_int64), ('chunks', C.c_int64), ('bytes_h2d', C.c_int64), ('seconds', C.c_double),
                ('decode_seconds', C.c_double), ('copy_wait_seconds', C.c_double), ('inflated_bytes', C.c_int64),
                ('blocks', C.c_int64), ('on_device', C.c_int32), ('starts_repaired', C.c_int32)]


class MetricsCounts(C.Structure):
    _fields_ = [('n_isize', C.c_int64), ('n_contam', C.c_int64), ('counter_total', C.c_int64),
                ('sample_counter', C.c_int64), ('records_scanned', C.c_int64)]


def effective_cpus():
    """CPUs this process may actually use: the affinity mask, cut down by the cgroup's CFS quota when there is one
    (cpu.max / cpu.cfs_quota_us).  os.cpu_count() reports the machine - on a container limited to 16 CPUs of a 256-CPU
    host, 128 busy threads get the whole group throttled for most of every scheduling period (the BAM reader ran at a
    third of its 32-thread rate there)."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as fh:                       # cgroup v2: "<quota|max> <period>"
            q, period = fh.read().split()[:2]
            if q != 'max':
                quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as fq, open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as fp:
                q, period = float(fq.read()), float(fp.read())
                if q > 0 and period > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n)


_P = C.c_void_p
_SIGNATURES = {
    'besst_abi_version': (C.c_int, []),
    'besst_last_error': (C.c_char_p, []),
    'besst_release_cached_memory': (None, []),
    'besst_device_count': (C.c_int, []),
    'besst_prof_enable': (None, [C.c_uint32]),
    'besst_prof_sample_every': (None, [C.c_uint32]),
    'besst_prof_slots': (C.c_int, []),
    'besst_prof_slot_name': (C.c_char_p, [C.c_int]),
    'besst_prof_collect': (C.c_int, [C.c_int, _P, _P]),
    'besst_ctx_create': (_P, [C.c_int]),
    'besst_ctx_destroy': (None, [_P]),
    'besst_ctx_set_contigs': (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P]),
    'besst_ctx_set_library': (C.c_int, [_P, C.POINTER(LibParams)]),
    'besst_ctx_clear_records': (C.c_int, [_P]),
    'besst_ctx_push_records': (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P]),
    'besst_ctx_record_count': (C.c_int, [_P, C.POINTER(C.c_int64)]),
    'besst_ctx_record_pointers': (C.c_int, [_P, C.POINTER(C.c_int64), _P]),
    'besst_ctx_fetch_records': (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P]),
    'besst_ctx_push_bam': (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, _P, _P, C.POINTER(IngestStats)]),
    'besst_ctx_push_bam_device': (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, _P, _P, C.POINTER(IngestStats)]),
    'besst_ctx_push_bam_device_part': (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int64, C.c_int64, _P, _P, _P,
                                                 C.POINTER(IngestStats)]),
    'besst_ctx_push_bam_device_slice': (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int64, C.c_int64, _P, C.c_int64, _P, _P, _P,
                                                  C.POINTER(IngestStats)]),
    'besst_bgzf_inflate_device': (C.c_int, [C.c_int, _P, C.c_size_t, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    'besst_ctx_metrics_sample': (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_double, C.c_int32, _P, _P,
                                           C.POINTER(MetricsCounts)]),
    'besst_ctx_stream_order': (C.c_int, [_P, C.POINTER(C.c_int64), _P, _P]),
    'besst_dev_stream_order': (C.c_int, [_P, C.c_int64, _P, _P, _P]),
    'besst_ctx_value_histogram': (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, _P]),
    'besst_ctx_build_graph': (C.c_int, [_P]),
    'besst_ctx_edge_count': (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'besst_ctx_fetch_edges': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(C.c_int32)]),
    'besst_ctx_fetch_observations': (C.c_int, [_P, _P, _P]),
    'besst_ctx_fetch_observation_sums': (C.c_int, [_P, _P]),
    'besst_ctx_fetch_coverage': (C.c_int, [_P, _P]),
    'besst_ctx_fetch_counters': (C.c_int, [_P, C.POINTER(Counters)]),
    'besst_ctx_score_edges': (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, C.c_double, C.c_double, C.c_double,
                                        _P, _P, _P, _P]),
    'besst_ctx_score_edges_lognormal': (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, C.c_double, C.c_double, C.c_double,
                                                  C.c_double, C.c_double, C.c_int64, C.c_int32, _P, _P, _P]),
    'besst_ctx_conditional_stddevs': (C.c_int, [_P, _P, C.c_int64, _P, C.c_int32, _P]),
    'besst_host_isize_stats': (C.c_int, [_P, C.c_int64, C.c_int32, C.c_double, _P, _P, _P]),
    'besst_host_contam_stats': (C.c_int, [_P, C.c_int64, C.c_int32, C.c_double, _P, _P]),
    'besst_host_getdistr': (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_double, _P, C.c_int64, _P, C.c_int64, _P, _P]),
    'besst_bam_open': (_P, [C.c_char_p, C.c_int]),
    'besst_bam_close': (None, [_P]),
    'besst_bam_n_references': (C.c_int64, [_P]),
    'besst_bam_clamped_records': (C.c_int64, [_P]),
    'besst_bam_reference_name': (C.c_char_p, [_P, C.c_int64]),
    'besst_bam_reference_names': (C.c_int64, [_P, _P, C.c_int64]),
    'besst_bam_reference_lengths': (C.c_int, [_P, _P]),
    'besst_bam_read_records': (C.c_int64, [_P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'besst_bam_write_records': (C.c_int, [C.c_char_p, C.c_int64, _P, _P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                          C.c_int, C.c_int]),
    'besst_dev_classify_workspace_bytes': (C.c_size_t, [C.c_int64]),
    'besst_dev_reduce_workspace_bytes': (C.c_size_t, [C.c_int64]),
    'besst_dev_contig_table_bytes': (C.c_size_t, [C.c_int64]),
    'besst_dev_restore_state': (C.c_int, [_P, _P, _P, C.c_int64]),
    'besst_dev_pack_contigs': (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P, _P]),
    'besst_dev_classify': (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, C.c_int64, _P,
                                     C.POINTER(LibParams), C.c_int32, _P, _P, _P, _P, _P, _P, _P, C.c_size_t]),
    'besst_dev_candidate_density': (C.c_int, [_P, C.c_int64, _P, _P, C.c_int64, _P, C.POINTER(C.c_double),
                                              C.POINTER(C.c_int32)]),
    'besst_dev_reduce': (C.c_int, [_P, C.c_int64, _P, C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                   _P, _P, C.c_size_t, _P, C.c_uint64]),
    'besst_dev_reduce_flags': (C.c_int, [_P, C.c_int64, _P, C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                         _P, _P, C.c_size_t, _P, C.c_uint64, C.c_uint32]),
    'besst_dev_reduce_presort': (C.c_int, [C.c_int64, C.c_int32, C.c_uint64, _P, C.c_size_t, C.POINTER(Presort)]),
    'besst_dev_classify_presort': (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, C.c_int64, _P,
                                             C.POINTER(LibParams), C.c_int32, _P, _P, _P, _P, _P, _P, _P, C.c_size_t,
                                             C.POINTER(Presort)]),
    'besst_dev_reduce_presorted': (C.c_int, [_P, C.c_int64, _P, C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                             _P, _P, C.c_size_t, _P, C.c_uint64, C.POINTER(Presort)]),
    'besst_dev_classify_scan': (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, C.c_int64, _P,
                                          C.POINTER(LibParams), C.c_int32, _P, _P, _P, C.c_size_t]),
    'besst_dev_classify_tail': (C.c_int, [_P, C.c_int64, _P, _P, C.c_size_t]),
    'besst_dev_resolve_carry': (C.c_int, [_P, _P, C.c_int32, _P]),
    'besst_dev_classify_emit': (C.c_int, [_P, C.c_int64, C.c_int32, _P, _P, _P, _P, _P, _P, C.c_size_t, C.c_int64,
                                          _P, _P, _P, C.c_int32, _P]),
    'besst_dev_gap_condition_table': (C.c_int, [_P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32, _P]),
    'besst_ctx_gap_condition_table': (C.c_int, [_P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32, _P]),
    'besst_dev_chain_workspace_bytes': (C.c_size_t, [C.c_int64]),
    'besst_dev_chain_scaffolds': (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, C.c_size_t, _P, _P, _P, _P]),
    'besst_chain_scaffolds': (C.c_int, [C.c_int, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P]),
    'besst_dev_score_edges': (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_double, C.c_double,
                                        C.c_double, _P, _P, _P, _P, _P, C.c_size_t]),
    'besst_dev_mate_bits_bytes': (C.c_size_t, [C.c_int64]),
    'besst_dev_mate_bits': (C.c_int, [_P, C.c_int64, _P, _P, _P]),
    'besst_dev_lognormal_tables_workspace_bytes': (C.c_size_t, [C.c_int64]),
    'besst_dev_lognormal_tables': (C.c_int, [_P, C.c_double, C.c_double, C.c_int64, _P, _P, _P, C.c_size_t]),
    'besst_dev_score_edges_lognormal': (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_double,
                                                  C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64, _P, _P,
                                                  C.c_int32, _P, _P, _P, _P, C.c_size_t]),
    'besst_dev_conditional_stddevs': (C.c_int, [_P, _P, C.c_int64, _P, C.c_int32, _P]),
    'besst_dev_metrics_workspace_bytes': (C.c_size_t, [C.c_int64]),
    'besst_dev_metrics_sample': (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, C.c_int64, _P, C.c_int32, C.c_int32,
                                           C.c_double, C.c_int32, _P, _P, _P, _P, C.c_size_t]),
    'besst_dev_exchange_region_bytes': (C.c_size_t, [C.c_int64]),
    'besst_owner_of_scaffold': (C.c_uint32, [C.c_uint32, C.c_uint32]),
    'besst_dev_exchange_stride_bytes': (C.c_size_t, [C.c_int64, C.c_int64]),
    'besst_dev_partition': (C.c_int, [_P, C.c_int64, _P, C.c_int32, C.c_int32, _P, _P, C.c_int64, _P, _P,
                                      C.c_size_t, _P, C.c_int64, _P]),
    'besst_dev_unpack': (C.c_int, [_P, C.c_int32, C.c_int64, _P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int32,
                                   C.c_int32, C.c_int32, _P, _P]),
    'besst_linearize': (C.c_int, [C.c_int, C.c_int32, C.c_int64, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'besst_score_paths': (C.c_int, [C.c_int, C.c_int64, _P, _P, _P, C.c_int64, _P, _P, C.c_int32, _P, _P]),
    'besst_dev_score_paths': (C.c_int, [_P, C.c_int64, _P, _P, _P, C.c_int64, _P, _P, C.c_int32, _P, _P]),
    'besst_dev_linearize_workspace_bytes': (C.c_size_t, [C.c_int64, C.c_int64]),
    'besst_dev_linearize': (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int64, _P, _P, _P, _P, C.c_size_t, _P, _P, _P, _P, _P, _P,
                                      _P]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGNATURES)


def load():
    """Load libbesst_amd.so (no GPU needed for loading; needed for any compute call)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise BesstDeviceError(
            '%s is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(besst_amd has no CPU fallback)' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.besst_abi_version() != 3:
        raise BesstDeviceError('libbesst_amd.so ABI version mismatch')
    _lib = lib
    return lib


def last_error():
    return load().besst_last_error().decode('utf-8', 'replace')


def check(status, what):
    if status != 0:
        raise BesstDeviceError('%s failed (status %d): %s' % (what, status, last_error()), status=int(status))


def ptr(arr):
    """Raw pointer of a C-contiguous numpy array (or None)."""
    if arr is None:
        return None
    if not arr.flags['C_CONTIGUOUS']:
        raise ValueError('array must be C-contiguous')
    return arr.ctypes.data_as(C.c_void_p)


def as_col(arr, dtype):
    return np.ascontiguousarray(arr, dtype=dtype)
