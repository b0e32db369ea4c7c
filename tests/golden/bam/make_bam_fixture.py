#!/usr/bin/env python
"""Hand-assembled BAM fixtures for the native BAM front-end (besst_amd/csrc/bam_reader.hip).

Everything is packed field by field from the SAM/BAM specification (SAMv1 section 4: BGZF blocks = gzip members with
the 'BC' extra subfield; BAM header magic / l_text / n_ref / references; alignment records block_size, refID, pos,
l_read_name, mapq, bin, n_cigar_op, flag, l_seq, next_refID, next_pos, tlen, read_name, cigar, seq, qual, aux) with
struct.pack and zlib only.  It does NOT use tests/bam_writer.py (the writer that lives beside the reader), so a shared
misunderstanding of the format cannot hide.  The expected per-record values are written down here by hand from the
definitions of pysam 0.8.4's AlignedRead properties, which is what the reference reads (CreateGraph.py:138;
libmetrics.py:258-262):

  qlen = query_alignment_length = qend - qstart; qstart = leading soft clips (leading hard clips skipped);
         qend = l_seq (or, without a sequence, the M/I/S/=/X total of the CIGAR) minus trailing soft clips
  rlen = query_length = l_seq (0 when the sequence is '*')
  alen = reference_length = bases of reference consumed by M/D/N/=/X; None (here 0) without a CIGAR
  (pysam 0.8.4 predates the CG:B,I long-CIGAR convention: it reads the placeholder CIGAR <l_seq>S<ref_len>N as it
  stands, i.e. qlen 0 and alen = ref_len)

Two files with the same records:
  handmade_a.bam  header in its own block, a few records per block, a record cut in two by a block boundary, an empty
                  block in the middle, the standard EOF marker
  handmade_b.bam  97 payload bytes per block (the header and nearly every record straddle blocks), NO EOF marker
Run:  python tests/golden/bam/make_bam_fixture.py   (writes the .bam files and handmade_bam.json next to this script)
"""
import json
import os
import struct
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
OPS = 'MIDNSHP=X'
REFS = [('ctgA', 5000), ('ctgB', 12345), ('a_rather_long_reference_name_to_cross_a_block_boundary', 70000)]


def bgzf_block(payload):
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    data = comp.compress(payload) + comp.flush()
    bsize = 12 + 6 + len(data) + 8 - 1
    head = struct.pack('<BBBBIBBH', 31, 139, 8, 4, 0, 0, 255, 6) + struct.pack('<BBHH', ord('B'), ord('C'), 2, bsize)
    return head + data + struct.pack('<II', zlib.crc32(payload) & 0xffffffff, len(payload))


EOF_BLOCK = bgzf_block(b'')


def reg2bin(beg, end):
    end -= 1
    for shift, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return off + (beg >> shift)
    return 0


def cigar_ops(text):
    out, num = [], ''
    for ch in text:
        if ch.isdigit():
            num += ch
        else:
            out.append((OPS.index(ch), int(num)))
            num = ''
    return out


def record(name, flag, tid, pos, mapq, cigar, mtid, mpos, tlen, l_seq, aux=b''):
    ops = cigar_ops(cigar) if cigar != '*' else []
    ref_len = sum(n for op, n in ops if op in (0, 2, 3, 7, 8))
    name_b = name.encode() + b'\0'
    cig_b = b''.join(struct.pack('<I', (n << 4) | op) for op, n in ops)
    seq_b = bytes([0x12] * ((l_seq + 1) // 2))          # 'AC' pairs
    qual_b = bytes([30] * l_seq)
    body = struct.pack('<iiBBHHHIiii', tid, pos, len(name_b), mapq, reg2bin(pos, pos + max(ref_len, 1)), len(ops), flag,
                       l_seq, mtid, mpos, tlen) + name_b + cig_b + seq_b + qual_b + aux
    return struct.pack('<I', len(body)) + body


def aux_tags():
    return (b'NMC' + bytes([3]) + b'MDZ' + b'50A49\0' + b'ASi' + struct.pack('<i', 97) + b'XSs' + struct.pack('<h', -5)
            + b'ZBB' + b'S' + struct.pack('<I', 3) + struct.pack('<HHH', 1, 2, 3) + b'XFf' + struct.pack('<f', 1.5))


# name, flag, tid, pos, mapq, cigar, mtid, mpos, tlen, l_seq, aux  ->  expected qlen, rlen, alen
CASES = [
    (('r01_plain', 0x63, 0, 100, 60, '100M', 0, 400, 400, 100, b''), (100, 100, 100)),
    (('r02_softclip', 0x93, 0, 400, 60, '5S90M5S', 0, 100, -400, 100, aux_tags()), (90, 100, 90)),
    (('r03_hard_and_soft', 0x63, 0, 700, 37, '3H10S80M10S2H', 1, 50, 0, 100, b''), (80, 100, 80)),
    (('r04_insertion', 0xa3, 0, 900, 60, '50M2I48M', 1, 900, 0, 100, b''), (100, 100, 98)),
    (('r05_deletion', 0x53, 1, 10, 11, '50M3D50M', 0, 4000, 0, 100, aux_tags()), (100, 100, 103)),
    (('r06_skip', 0x63, 1, 300, 10, '30M1000N70M', 1, 2000, 1800, 100, b''), (100, 100, 1100)),
    (('r07_eq_x', 0x93, 1, 2000, 0, '50=2X48=', 1, 300, -1800, 100, b''), (100, 100, 100)),
    (('r08_padding', 0x61, 1, 2500, 60, '40M5P60M', 2, 10, 0, 100, b''), (100, 100, 100)),
    # BWA: unmapped read placed at its mate's position, CIGAR '*': query_alignment_length = l_seq, no reference length
    (('r09_unmapped_placed', 0x65, 1, 3000, 0, '*', 1, 3000, 0, 100, b''), (100, 100, 0)),
    (('r10_no_sequence', 0xa1, 1, 3500, 60, '100M', 2, 500, 0, 0, b''), (100, 0, 100)),
    (('r11_no_sequence_clipped', 0x91, 1, 3600, 60, '10S80M10S', 2, 600, 0, 0, b''), (80, 0, 80)),
    # long-CIGAR convention: placeholder CIGAR + the real operations in CG:B,I (read as it stands by pysam 0.8.4)
    (('r12_cg_tag', 0x61, 2, 1000, 60, '100S150N', 2, 5000, 0, 100,
      b'CGBI' + struct.pack('<I', 3) + struct.pack('<III', (60 << 4) | 0, (50 << 4) | 2, (40 << 4) | 0)), (0, 100, 150)),
    (('n' * 254, 0xa3, 2, 5000, 60, '100M', 2, 1000, 0, 100, b''), (100, 100, 100)),
    (('x', 0x63, 2, 60000, 60, '20S30M50S', 2, 60200, 300, 100, aux_tags()), (30, 100, 30)),
    (('r15_l_seq_0_no_cigar', 0x4d, -1, -1, 0, '*', -1, -1, 0, 0, b''), (0, 0, 0)),
]


def bam_bytes():
    text = '@HD\tVN:1.4\tSO:coordinate\n' + ''.join('@SQ\tSN:%s\tLN:%d\n' % r for r in REFS)
    head = b'BAM\1' + struct.pack('<i', len(text)) + text.encode() + struct.pack('<i', len(REFS))
    for name, length in REFS:
        head += struct.pack('<i', len(name) + 1) + name.encode() + b'\0' + struct.pack('<i', length)
    return head, [record(*args) for args, _ in CASES]


def main():
    head, recs = bam_bytes()
    # ---- a: header alone, three records per block, record 7 cut in two, an empty block in the middle, EOF marker
    blocks = [bgzf_block(head), bgzf_block(b''.join(recs[:3])), bgzf_block(b''.join(recs[3:6]) + recs[6][:41]), EOF_BLOCK,
              bgzf_block(recs[6][41:] + b''.join(recs[7:12])), bgzf_block(b''.join(recs[12:])), EOF_BLOCK]
    with open(os.path.join(HERE, 'handmade_a.bam'), 'wb') as fh:
        fh.write(b''.join(blocks))
    # ---- b: 97-byte payloads, no EOF marker
    stream = head + b''.join(recs)
    with open(os.path.join(HERE, 'handmade_b.bam'), 'wb') as fh:
        for off in range(0, len(stream), 97):
            fh.write(bgzf_block(stream[off:off + 97]))
    doc = dict(references=[r[0] for r in REFS], lengths=[r[1] for r in REFS], records=[])
    for (name, flag, tid, pos, mapq, cigar, mtid, mpos, tlen, l_seq, _), (qlen, rlen, alen) in CASES:
        doc['records'].append(dict(name=name if len(name) < 40 else name[:8] + '..%d' % len(name), cigar=cigar, tid=tid,
                                   pos=pos, mapq=mapq, flag=flag, mtid=mtid, mpos=mpos, tlen=tlen, qlen=qlen, rlen=rlen,
                                   alen=alen))
    with open(os.path.join(HERE, 'handmade_bam.json'), 'w') as fh:
        json.dump(doc, fh, indent=1)
    print('wrote %d records, %d + %d bytes' % (len(recs), os.path.getsize(os.path.join(HERE, 'handmade_a.bam')),
                                                os.path.getsize(os.path.join(HERE, 'handmade_b.bam'))))


if __name__ == '__main__':
    main()
