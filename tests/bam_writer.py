"""Minimal BAM writer for tests: serialises a RecordBatch as BGZF-compressed BAM (SAM spec v1 section 4)."""
import struct
import zlib

import numpy as np


LIBDEFLATE_LEVEL = None              # set (0-12): the blocks are compressed by libdeflate - what htslib writes them with - instead of zlib


def _bgzf_block(data):
    if LIBDEFLATE_LEVEL is not None:
        from tests import libdeflate_util
        payload = libdeflate_util.deflate(data, LIBDEFLATE_LEVEL)
    else:
        comp = zlib.compressobj(6, zlib.DEFLATED, -15)
        payload = comp.compress(data) + comp.flush()
    bsize = len(payload) + 25
    head = struct.pack('<BBBBIBBHBBHH', 31, 139, 8, 4, 0, 0, 255, 6, ord('B'), ord('C'), 2, bsize)
    return head + payload + struct.pack('<II', zlib.crc32(data) & 0xffffffff, len(data))


def write_bam(path, batch, soft_clip=None, block_bytes=60000, align_records=False, decoys=False):
    """Write ``batch`` as BAM.  CIGAR per record: [soft clip S] qlen M, so that qlen/rlen/alen round-trip:
    rlen = qlen + clip (or 0 when batch.rlen is 0), alen = qlen.
    align_records=False cuts BGZF blocks at arbitrary bytes (records straddle blocks); True flushes the block
    before a record that would not fit, like htslib's bam_write1, so that every block starts with a record.
    decoys=True makes the qualities of every record with room for them spell the fixed fields of a record (legal
    reference ids and positions, a NUL-closed name, a length that points far beyond the file): bytes that a reader
    which looks for record starts inside a block must not take for one."""
    decoy = struct.pack('<IiiBBHHHIiii', 1 << 26, 0, 5, 3, 30, 4680, 0, 99, 0, 0, 7, 0) + b'zz\x00'
    out = bytearray()
    cuts = []                        # block boundaries when align_records
    text = b'@HD\tVN:1.0\tSO:coordinate\n'
    hdr = b'BAM\x01' + struct.pack('<I', len(text)) + text + struct.pack('<I', len(batch.references))
    for name, length in zip(batch.references, batch.lengths):
        nm = name.encode() + b'\x00'
        hdr += struct.pack('<I', len(nm)) + nm + struct.pack('<I', length)
    out += hdr
    rlen = batch.rlen if batch.rlen is not None else batch.qlen.astype(np.int32)
    for i in range(len(batch)):
        q = int(batch.qlen[i])
        seq_len = int(rlen[i])
        clip = max(0, seq_len - q) if seq_len else 0
        cigar = []
        if clip:
            cigar.append((clip << 4) | 4)
        if q:
            cigar.append((q << 4) | 0)
        name = b'r%d\x00' % i
        body = struct.pack('<iiBBHHHIiii', int(batch.tid[i]), int(batch.pos[i]), len(name), int(batch.mapq[i]), 4680,
                           len(cigar), int(batch.flag[i]), seq_len, int(batch.mtid[i]), int(batch.mpos[i]),
                           int(batch.tlen[i]))
        body += name + b''.join(struct.pack('<I', c) for c in cigar)
        qual = b'\xff' * seq_len
        if decoys and seq_len >= len(decoy):
            qual = decoy + qual[len(decoy):]
        body += b'\x11' * ((seq_len + 1) // 2) + qual
        if align_records and (not cuts or len(out) + 4 + len(body) - cuts[-1] > block_bytes):
            cuts.append(len(out))    # the header gets blocks of its own, then a new block whenever one is full
        out += struct.pack('<I', len(body)) + body
    with open(path, 'wb') as fh:
        if align_records:
            bounds = [0] + [c for c in cuts if c > 0] + [len(out)]
            for lo, hi in zip(bounds[:-1], bounds[1:]):
                for off in range(lo, hi, 65000):         # only the header can exceed one block
                    fh.write(_bgzf_block(bytes(out[off:min(off + 65000, hi)])))
        else:
            for off in range(0, len(out), block_bytes):
                fh.write(_bgzf_block(bytes(out[off:off + block_bytes])))
        fh.write(_bgzf_block(b''))      # BGZF EOF marker
