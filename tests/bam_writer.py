"""Minimal BAM writer for tests: serialises a RecordBatch as BGZF-compressed BAM (SAM spec v1 section 4)."""
import struct
import zlib

import numpy as np


def _bgzf_block(data):
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    payload = comp.compress(data) + comp.flush()
    bsize = len(payload) + 25
    head = struct.pack('<BBBBIBBHBBHH', 31, 139, 8, 4, 0, 0, 255, 6, ord('B'), ord('C'), 2, bsize)
    return head + payload + struct.pack('<II', zlib.crc32(data) & 0xffffffff, len(data))


def write_bam(path, batch, soft_clip=None, block_bytes=60000):
    """Write ``batch`` as BAM.  CIGAR per record: [soft clip S] qlen M, so that qlen/rlen/alen round-trip:
    rlen = qlen + clip (or 0 when batch.rlen is 0), alen = qlen."""
    out = bytearray()
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
        body += b'\x11' * ((seq_len + 1) // 2) + b'\xff' * seq_len
        out += struct.pack('<I', len(body)) + body
    with open(path, 'wb') as fh:
        for off in range(0, len(out), block_bytes):
            fh.write(_bgzf_block(bytes(out[off:off + block_bytes])))
        fh.write(_bgzf_block(b''))      # BGZF EOF marker
