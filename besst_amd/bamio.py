"""BAM file -> RecordBatch through the library's own BGZF/BAM reader (no pysam, no htslib).

This is the front-end the reference gets from ``pysam.Samfile(param.bamfile, 'rb')`` (runBESST:162): header
references/lengths plus, per record, exactly the attributes the hot path reads.  Inflate runs on host threads;
the decoded columns are the same SoA layout the device kernels consume.
"""
import os

import numpy as np

from . import _lib
from .records import RecordBatch


def read_bam(path, threads=None, chunk_records=8_000_000):
    lib = _lib.load()
    # inflate, the record walk and the column fill are block/record parallel; beyond ~32-64 threads waking the pool
    # costs more than it buys (76 M records/s at 64 threads, 18 M at 256 on the 256-core bench host)
    threads = threads or min(32, os.cpu_count() or 1)
    handle = lib.besst_bam_open(os.fsencode(path), int(threads))
    if not handle:
        raise IOError('cannot read BAM %s: %s' % (path, _lib.last_error()))
    try:
        n_ref = lib.besst_bam_n_references(handle)
        names = [lib.besst_bam_reference_name(handle, i).decode() for i in range(n_ref)]
        lengths = np.zeros(max(n_ref, 1), dtype=np.int32)
        _lib.check(lib.besst_bam_reference_lengths(handle, _lib.ptr(lengths)), 'bam_reference_lengths')
        spec = (('tid', np.int32), ('mtid', np.int32), ('pos', np.int32), ('mpos', np.int32), ('tlen', np.int32),
                ('flag', np.uint16), ('mapq', np.uint8), ('qlen', np.uint16), ('rlen', np.int32), ('alen', np.int32))
        parts = {k: [] for k, _ in spec}
        while True:
            bufs = [np.empty(chunk_records, dtype=dt) for _, dt in spec]
            got = lib.besst_bam_read_records(handle, chunk_records, *[_lib.ptr(b) for b in bufs])
            if got < 0:
                raise IOError('error while reading %s: %s' % (path, _lib.last_error()))
            if got == 0:
                break
            for (k, _), b in zip(spec, bufs):
                parts[k].append(b[:got])
        cols = {k: (np.concatenate(v) if v else np.empty(0, dtype=dt)) for (k, dt), v in zip(spec, parts.values())}
        clamped = lib.besst_bam_clamped_records(handle)
        if clamped > 0:
            import warnings
            warnings.warn('%s: %d record(s) align more than 65535 query bases; their qlen (the coverage numerator of '
                          'CreateGraph.py:138-139) is stored as 65535' % (path, clamped))
    finally:
        lib.besst_bam_close(handle)
    return RecordBatch(names, lengths[:n_ref].tolist(), **cols)
