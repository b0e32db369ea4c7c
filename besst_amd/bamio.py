"""BAM file -> RecordBatch through the library's own BGZF/BAM reader (no pysam, no htslib).

This is the front-end the reference gets from ``pysam.Samfile(param.bamfile, 'rb')`` (runBESST:162): header
references/lengths plus, per record, exactly the attributes the hot path reads.  Inflate runs on host threads;
the decoded columns are the same SoA layout the device kernels consume.
"""
import os

import numpy as np

from . import _lib
from .records import RecordBatch


def reader_threads():
    """Workers of the reader's pool: twice the CPUs the process may use (_lib.effective_cpus: affinity and cgroup quota -
    a worker blocked in a page fault leaves its share unused), at most 128."""
    return min(128, 2 * _lib.effective_cpus())


def _open(lib, path, threads):
    handle = lib.besst_bam_open(os.fsencode(path), int(threads))
    if not handle:
        raise IOError('cannot read BAM %s: %s' % (path, _lib.last_error()))
    names, lengths = _header(lib, handle)
    return handle, names, lengths


def _header(lib, handle):
    """(reference names, lengths) of an open reader; the names in one call (a C5 header has 2 M of them)."""
    n_ref = lib.besst_bam_n_references(handle)
    need = lib.besst_bam_reference_names(handle, None, 0)
    buf = np.zeros(max(int(need), 1), dtype=np.uint8)
    lib.besst_bam_reference_names(handle, _lib.ptr(buf), int(need))
    names = buf[:need].tobytes().decode().split('\0')[:n_ref] if n_ref else []
    lengths = np.zeros(max(n_ref, 1), dtype=np.int32)
    _lib.check(lib.besst_bam_reference_lengths(handle, _lib.ptr(lengths)), 'bam_reference_lengths')
    return names, lengths[:n_ref].tolist()


class ResidentBam(object):
    """A BAM file whose records went straight to HBM - besst_ctx_push_bam_device: the compressed file is uploaded and
    inflated + decoded on the GPU (any BGZF block layout); else besst_ctx_push_bam: decode on host threads, pinned
    staging, copies under the next chunk's decode; ``mode`` as in GraphContext.push_bam - the `bam_file` argument for libmetrics.get_metrics and CreateGraph.PE when nothing
    on the host needs the record columns; ``part=(r, W)``: rank r's slice of the stream (htslib's layout; with ``first_skip`` - -1: guess, >= 0: inflated bytes in front of the slice's first record - the SLICE form for any block layout, ``boundary`` = (offset used, bytes of the last record in the next slice): distributed.ingest_slice runs the check between ranks; multi-GPU ingest: ``len()`` and the
    head arrays then describe that slice only - such an object is for distributed.ingest_slice, not for
    libmetrics.get_metrics, whose < 1000-record check and read-length step are about the whole file).  It carries what the host side of those two does read: the header
    (``references``, ``lengths``), the record count (``len()``) and ``rlen`` / ``alen`` / ``qlen`` of the first 1000
    records (the read-length step, libmetrics.py:246-273); ``ctx`` is the GraphContext that holds the records and
    ``ingest`` the timings of the upload."""

    def __init__(self, path, device_index=0, threads=None, chunk_records=4 << 20, mode=None, chunk_blocks=0, part=None,
                 first_skip=None):
        from . import device
        lib = _lib.load()
        threads = threads or reader_threads()
        handle, self.references, self.lengths = _open(lib, path, threads)
        self.path = path
        self.ctx = None
        try:
            self.ctx = device.GraphContext(device_index)     # (inside the try: a context that cannot be made must not leak the reader)
            zeros = np.zeros(len(self.references), dtype=np.int32)
            self.ctx.set_contigs(scaf_id=zeros, scaf_len=zeros, ctg_pos=zeros, ctg_len=zeros, direction=zeros, cls=zeros)
            self.ingest, self.rlen, self.alen, self.qlen = self.ctx.push_bam(handle, chunk_records, mode=mode, chunk_blocks=chunk_blocks, part=part,
                                                                              first_skip=first_skip)
            self.boundary = getattr(self.ctx, 'slice_boundary', None)
            clamped = lib.besst_bam_clamped_records(handle)
        except Exception:
            if self.ctx is not None:
                self.ctx.close()
            raise
        finally:
            lib.besst_bam_close(handle)
        self._n = int(self.ingest.records)
        if clamped > 0:
            import warnings
            warnings.warn('%s: %d record(s) align more than 65535 query bases; their qlen is stored as 65535' % (path, clamped))

    def __len__(self):
        return self._n

    def close(self):
        self.ctx.close()


class ShardedBam(object):
    """The `bam_file` argument of libmetrics.get_metrics / CreateGraph.PE under a process group (one process per GPU):
    this rank's slice of the file, inflated and decoded on its GPU (distributed.ingest_slice: any BGZF block layout, the
    ranks settle where each slice's first record begins), plus what the host side reads of the whole file - header,
    global record count, rlen / alen / qlen of the stream's first 1000 records (sharded.ShardedHead).  ``ingest``: the
    slice's timings."""

    def __init__(self, path, group=None, device_index=None, threads=None, chunk_blocks=0):
        import torch
        import torch.distributed as dist
        from . import distributed, pipeline, sharded
        self.path, self.group = path, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        dev = torch.cuda.current_device() if device_index is None else int(device_index)
        self.slice, cols = distributed.ingest_slice(path, self.rank, self.world, device_index=dev, threads=threads,
                                                    chunk_blocks=chunk_blocks, group=group)
        self.ingest = self.slice.ingest
        self.references, self.lengths = self.slice.references, self.slice.lengths
        device = torch.device('cuda', dev)
        rec = pipeline.DeviceRecords.from_columns(cols)
        self.engine = sharded.HipRankEngine(device, rec, len(self.references), keep=self.slice)
        self.head = sharded.ShardedHead(self.references, self.lengths, len(self.slice),
                                        (self.slice.rlen, self.slice.alen, self.slice.qlen), group, self.world)
        self.rlen, self.alen, self.qlen = self.head.rlen, self.head.alen, self.head.qlen

    def __len__(self):
        return len(self.head)

    def close(self):
        if self.engine is not None:
            self.engine.close()
            self.engine = None


def open_bam(path, threads=None, **kw):
    """What a caller of the two entry points opens in place of ``pysam.Samfile(path, 'rb')`` (runBESST:162): a ResidentBam,
    or - under a process group for which sharding has been switched on (sharded.enable() / BESST_SHARDED=1: torchrun, one
    process per GPU) - this rank's ShardedBam."""
    from . import sharded
    if sharded.active_group() is not None:
        return ShardedBam(path, group=sharded.PROCESS_GROUP, threads=threads,
                          **{k: v for k, v in kw.items() if k in ('device_index', 'chunk_blocks')})
    return ResidentBam(path, threads=threads, **kw)


def inflate_bgzf_device(data, device_index=0, out_cap=None):
    """Test hook: the BGZF blocks of ``data`` (bytes) inflated by the GPU kernel, concatenated."""
    import ctypes as C
    lib = _lib.load()
    src = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(out_cap if out_cap is not None else max(1, 66 * len(data)), dtype=np.uint8)
    n = C.c_size_t(0)
    _lib.check(lib.besst_bgzf_inflate_device(int(device_index), _lib.ptr(src), len(data), _lib.ptr(out), out.size, C.byref(n)),
               'bgzf_inflate_device')
    return out[:n.value].tobytes()


def write_bam(path, batch, threads=None, level=1, realistic=False):
    """Test / bench scaffolding (besst_bam_write_records): the batch as a BAM file in htslib's block layout.  realistic:
    pseudo-random bases and slowly changing qualities instead of constant bytes (the file compresses ~3 x, not ~13 x)."""
    if realistic:
        level = int(level) | 16
    import ctypes as C
    lib = _lib.load()
    names = (C.c_char_p * len(batch.references))(*[n.encode() for n in batch.references])
    lengths = np.ascontiguousarray(batch.lengths, dtype=np.int32)
    rlen = batch.rlen if batch.rlen is not None else batch.qlen
    cols = [_lib.as_col(batch.tid, np.int32), _lib.as_col(batch.mtid, np.int32), _lib.as_col(batch.pos, np.int32),
            _lib.as_col(batch.mpos, np.int32), _lib.as_col(batch.tlen, np.int32), _lib.as_col(batch.flag, np.uint16),
            _lib.as_col(batch.mapq, np.uint8), _lib.as_col(batch.qlen, np.uint16), _lib.as_col(rlen, np.int32)]
    _lib.check(lib.besst_bam_write_records(os.fsencode(path), len(batch.references), names, _lib.ptr(lengths), len(batch),
                                           *[_lib.ptr(c) for c in cols], int(threads or reader_threads()),
                                           int(level)), 'bam_write_records')


def read_bam(path, threads=None, chunk_records=8_000_000):
    lib = _lib.load()
    threads = threads or reader_threads()
    handle = lib.besst_bam_open(os.fsencode(path), int(threads))
    if not handle:
        raise IOError('cannot read BAM %s: %s' % (path, _lib.last_error()))
    try:
        names, lengths = _header(lib, handle)
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
    return RecordBatch(names, lengths, **cols)
