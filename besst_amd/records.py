"""Alignment records as a flat structure-of-arrays (the HBM input schema).

The reference walks a ``pysam.Samfile`` one ``AlignedRead`` at a time and touches
exactly nine attributes plus seven flag bits (bam_parser.py:22-36;
CreateGraph.py:119-120,138-141,169,813-827; libmetrics.py:65-79,258-262,294-300).
:class:`RecordBatch` holds those columns contiguously so they can be handed to
the device unchanged:

    tid i32, mtid i32, pos i32, mpos i32, tlen i32, flag u16, mapq u8, qlen u16

(``rlen``/``alen`` are only consulted for the first 1000 records, libmetrics.py:246-273,
and are kept as optional host-side columns.)

A batch also quacks like the ``bam_file`` argument of the reference entry points
(``references``, ``lengths``, iteration, ``reset()``, ``fetch()``), yielding
lightweight record views with the pysam 0.8 attribute names.  The reference
harness under ``tests/refharness`` feeds the very same object to the reference
code, which is how golden vectors and the device path see identical inputs.
"""
import numpy as np

FLAG_PAIRED = 0x1
FLAG_PROPER = 0x2
FLAG_UNMAPPED = 0x4
FLAG_MATE_UNMAPPED = 0x8
FLAG_REVERSE = 0x10
FLAG_MATE_REVERSE = 0x20
FLAG_READ1 = 0x40
FLAG_READ2 = 0x80
FLAG_SECONDARY = 0x100

_COLUMNS = (('tid', np.int32), ('mtid', np.int32), ('pos', np.int32), ('mpos', np.int32),
            ('tlen', np.int32), ('flag', np.uint16), ('mapq', np.uint8), ('qlen', np.uint16))


class RecordView(object):
    """One alignment record with pysam-0.8 style attribute names."""
    __slots__ = ('rname', 'mrnm', 'pos', 'mpos', 'tlen', 'flag', 'mapq', 'qlen', 'rlen', 'alen')

    def __init__(self, rname, mrnm, pos, mpos, tlen, flag, mapq, qlen, rlen, alen):
        self.rname = rname
        self.mrnm = mrnm
        self.pos = pos
        self.mpos = mpos
        self.tlen = tlen
        self.flag = flag
        self.mapq = mapq
        self.qlen = qlen
        self.rlen = rlen
        self.alen = alen

    # pysam aliases
    tid = property(lambda self: self.rname)
    rnext = property(lambda self: self.mrnm)
    is_unmapped = property(lambda self: bool(self.flag & FLAG_UNMAPPED))
    mate_is_unmapped = property(lambda self: bool(self.flag & FLAG_MATE_UNMAPPED))
    is_reverse = property(lambda self: bool(self.flag & FLAG_REVERSE))
    mate_is_reverse = property(lambda self: bool(self.flag & FLAG_MATE_REVERSE))
    is_read1 = property(lambda self: bool(self.flag & FLAG_READ1))
    is_read2 = property(lambda self: bool(self.flag & FLAG_READ2))
    is_secondary = property(lambda self: bool(self.flag & FLAG_SECONDARY))


class RecordBatch(object):
    """SoA batch of alignment records plus the BAM header's reference table."""

    def __init__(self, references, lengths, tid, mtid, pos, mpos, tlen, flag, mapq, qlen,
                 rlen=None, alen=None):
        self.references = tuple(references)
        self.lengths = tuple(int(x) for x in lengths)
        if len(self.references) != len(self.lengths):
            raise ValueError('references and lengths differ in size')
        cols = dict(tid=tid, mtid=mtid, pos=pos, mpos=mpos, tlen=tlen, flag=flag, mapq=mapq, qlen=qlen)
        n = len(cols['tid'])
        for name, dtype in _COLUMNS:
            raw = np.asarray(cols[name])
            if raw.shape != (n,):
                raise ValueError('column %s has shape %r, expected (%d,)' % (name, raw.shape, n))
            arr = np.ascontiguousarray(raw, dtype=dtype)
            if raw.dtype != arr.dtype and n and not np.array_equal(arr.astype(np.int64), raw.astype(np.int64)):
                raise ValueError('column %s does not fit %s' % (name, np.dtype(dtype).name))
            setattr(self, name, arr)
        self.rlen = None if rlen is None else np.ascontiguousarray(rlen, dtype=np.int32)
        self.alen = None if alen is None else np.ascontiguousarray(alen, dtype=np.int32)

    def __len__(self):
        return int(self.tid.shape[0])

    @property
    def nbytes(self):
        return sum(getattr(self, name).nbytes for name, _ in _COLUMNS)

    # ---- pysam.Samfile surface used by the reference hot path -------------------------------
    def __iter__(self):
        tid = self.tid.tolist()
        mtid = self.mtid.tolist()
        pos = self.pos.tolist()
        mpos = self.mpos.tolist()
        tlen = self.tlen.tolist()
        flag = self.flag.tolist()
        mapq = self.mapq.tolist()
        qlen = self.qlen.tolist()
        rlen = self.rlen.tolist() if self.rlen is not None else qlen
        alen = self.alen.tolist() if self.alen is not None else qlen
        for i in range(len(tid)):
            yield RecordView(tid[i], mtid[i], pos[i], mpos[i], tlen[i], flag[i], mapq[i], qlen[i],
                             rlen[i], alen[i])

    def reset(self):
        return None

    def fetch(self, reference=None):
        if reference is not None and reference not in self.references:
            raise ValueError('unknown reference %r' % (reference,))
        return iter(())

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    # ---- helpers ---------------------------------------------------------------------------
    def slice(self, start, stop):
        kw = {name: getattr(self, name)[start:stop] for name, _ in _COLUMNS}
        rlen = None if self.rlen is None else self.rlen[start:stop]
        alen = None if self.alen is None else self.alen[start:stop]
        return RecordBatch(self.references, self.lengths, rlen=rlen, alen=alen, **kw)

    def take(self, index):
        """Rows `index` (any numpy index expression) as a new batch."""
        kw = {name: np.ascontiguousarray(getattr(self, name)[index]) for name, _ in _COLUMNS}
        rlen = None if self.rlen is None else np.ascontiguousarray(self.rlen[index])
        alen = None if self.alen is None else np.ascontiguousarray(self.alen[index])
        return RecordBatch(self.references, self.lengths, rlen=rlen, alen=alen, **kw)

    @classmethod
    def concatenate(cls, batches):
        """The batches' records one after the other (same reference table)."""
        first = batches[0]
        kw = {name: np.concatenate([getattr(b, name) for b in batches]) for name, _ in _COLUMNS}
        return cls(first.references, first.lengths, **kw)

    @classmethod
    def from_pysam_like(cls, bam_file):
        """Materialise any pysam-like iterable (slow host loop; compatibility path only)."""
        if isinstance(bam_file, RecordBatch):
            return bam_file
        cols = {name: [] for name, _ in _COLUMNS}
        rlen, alen = [], []
        for read in bam_file:
            cols['tid'].append(read.rname)
            cols['mtid'].append(read.mrnm)
            cols['pos'].append(read.pos)
            cols['mpos'].append(read.mpos)
            cols['tlen'].append(read.tlen)
            flag = getattr(read, 'flag', None)
            if flag is None:
                flag = (FLAG_UNMAPPED * bool(read.is_unmapped) | FLAG_MATE_UNMAPPED * bool(read.mate_is_unmapped)
                        | FLAG_REVERSE * bool(read.is_reverse) | FLAG_MATE_REVERSE * bool(read.mate_is_reverse)
                        | FLAG_READ1 * bool(read.is_read1) | FLAG_READ2 * bool(read.is_read2)
                        | FLAG_SECONDARY * bool(read.is_secondary))
            cols['flag'].append(flag)
            cols['mapq'].append(read.mapq)
            cols['qlen'].append(read.qlen)
            rlen.append(read.rlen)
            alen.append(read.alen if read.alen is not None else 0)
        if hasattr(bam_file, 'reset'):
            bam_file.reset()
        return cls(bam_file.references, bam_file.lengths, rlen=rlen, alen=alen, **cols)
