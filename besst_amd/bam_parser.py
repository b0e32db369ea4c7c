"""Per-record predicates of the library-statistics pass.

Scalar forms mirror bam_parser.py:22-36 of the reference (same names, same
argument meaning: a record with pysam-0.8 attributes and a mapq threshold that is
compared with a STRICT ``>``).  The ``*_mask`` forms evaluate the same predicates
over flat SoA columns and are what the host code and tests use; the device
kernels in csrc/metrics.hip implement the identical bit logic.
"""
import numpy as np

from .records import (FLAG_MATE_REVERSE, FLAG_MATE_UNMAPPED, FLAG_READ2, FLAG_REVERSE,
                      FLAG_SECONDARY, FLAG_UNMAPPED)


def _oriented(read, sign):
    rev, mrev = read.is_reverse, read.mate_is_reverse
    t = read.tlen * sign
    return ((rev and not mrev and t < 0) or (not rev and mrev and t > 0)) \
        and read.is_read2 and read.rname == read.mrnm


def is_proper_aligned_unique_innie(read, mapq_threshold):
    return bool(_oriented(read, 1) and not read.mate_is_unmapped
                and read.mapq > mapq_threshold and not read.is_secondary)


def is_proper_aligned_unique_outie(read, mapq_threshold):
    return bool(_oriented(read, -1) and not read.mate_is_unmapped
                and read.mapq > mapq_threshold and not read.is_secondary)


def is_unique_read_link(read, mapq_threshold):
    return bool(not read.is_unmapped and not read.mate_is_unmapped and read.rname != read.mrnm
                and read.mapq > mapq_threshold and not read.is_secondary)


def _oriented_mask(tid, mtid, tlen, flag, sign):
    rev = (flag & FLAG_REVERSE) != 0
    mrev = (flag & FLAG_MATE_REVERSE) != 0
    t = tlen.astype(np.int64) * sign
    return ((rev & ~mrev & (t < 0)) | (~rev & mrev & (t > 0))) & ((flag & FLAG_READ2) != 0) & (tid == mtid)


def innie_mask(tid, mtid, tlen, flag, mapq, mapq_threshold):
    return _oriented_mask(tid, mtid, tlen, flag, 1) & ((flag & (FLAG_MATE_UNMAPPED | FLAG_SECONDARY)) == 0) \
        & (mapq.astype(np.int64) > mapq_threshold)


def outie_mask(tid, mtid, tlen, flag, mapq, mapq_threshold):
    return _oriented_mask(tid, mtid, tlen, flag, -1) & ((flag & (FLAG_MATE_UNMAPPED | FLAG_SECONDARY)) == 0) \
        & (mapq.astype(np.int64) > mapq_threshold)


def unique_read_link_mask(tid, mtid, flag, mapq, mapq_threshold):
    return ((flag & (FLAG_UNMAPPED | FLAG_MATE_UNMAPPED | FLAG_SECONDARY)) == 0) & (tid != mtid) \
        & (mapq.astype(np.int64) > mapq_threshold)
