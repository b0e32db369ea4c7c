"""Host BAM front-end: a BAM written record by record must decode to the same SoA columns."""
import os
import tempfile

import numpy as np
import pytest

from besst_amd import bamio, synth
from tests import bam_writer


@pytest.mark.parametrize('threads,align,block_bytes,chunk', [(1, False, 60000, 5000), (4, False, 60000, 5000),
                                                             (4, True, 60000, 5000), (3, True, 3000, 777),
                                                             (2, False, 1000, 100000), (1, True, 60000, 1)])
def test_round_trip(threads, align, block_bytes, chunk):
    """Records straddling BGZF blocks (sequential walk) and htslib-style block-aligned records (block-parallel
    speculative walk), small blocks, and read chunks that end in the middle of a block."""
    asm = synth.make_assembly(120, 1500, 31)
    batch = synth.simulate_library(asm, synth.LibrarySpec('rf', 1500.0, 150.0, contam_frac=0.2), 9000, 32)
    if chunk == 1:
        batch = batch.slice(0, 600)
    batch.rlen[::9] = 0              # sequence absent -> rlen 0 (libmetrics.py:258-263 falls back to alen)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'x.bam')
        bam_writer.write_bam(path, batch, block_bytes=block_bytes, align_records=align)
        got = bamio.read_bam(path, threads=threads, chunk_records=chunk)
    assert got.references == batch.references and got.lengths == batch.lengths
    for col in ('tid', 'mtid', 'pos', 'mpos', 'tlen', 'flag', 'mapq', 'qlen'):
        assert np.array_equal(getattr(got, col), getattr(batch, col)), col
    assert np.array_equal(got.rlen, batch.rlen)
    assert np.array_equal(got.alen, batch.qlen.astype(np.int32))


def test_rejects_non_bam():
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'junk.bam')
        with open(path, 'wb') as fh:
            fh.write(b'this is not a bam file at all, not even gzip')
        with pytest.raises(IOError):
            bamio.read_bam(path)
    with pytest.raises(IOError):
        bamio.read_bam('/nonexistent/file.bam')


@pytest.mark.parametrize('name', ['handmade_a.bam', 'handmade_b.bam'])
@pytest.mark.parametrize('threads,chunk', [(1, 8_000_000), (4, 3), (2, 1)])
def test_hand_assembled_bam(name, threads, chunk):
    """BAM bytes the reader's author did not write with the reader's own writer: tests/golden/bam/make_bam_fixture.py packs
    header, records and BGZF blocks field by field from the SAM/BAM specification.  Covered: CIGAR I/D/N/S/H/P/=/X, a
    record without CIGAR (qlen = l_seq, as pysam 0.8.4 reports it), records without sequence, the CG:B,I placeholder,
    254-character and 1-character read names, auxiliary fields of several types, a record (and in _b nearly every record
    and the header) straddling BGZF blocks, an empty block in the middle of the file, a missing EOF marker."""
    import json
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'bam')
    with open(os.path.join(here, 'handmade_bam.json')) as fh:
        want = json.load(fh)
    got = bamio.read_bam(os.path.join(here, name), threads=threads, chunk_records=chunk)
    assert list(got.references) == want['references'] and list(got.lengths) == want['lengths']
    assert len(got) == len(want['records'])
    for i, r in enumerate(want['records']):
        for col in ('tid', 'mtid', 'pos', 'mpos', 'tlen', 'flag', 'mapq', 'qlen', 'rlen', 'alen'):
            assert int(getattr(got, col)[i]) == r[col], (r['name'], r['cigar'], col, int(getattr(got, col)[i]), r[col])


@pytest.mark.parametrize('threads', [1, 5])
def test_native_writer_agrees_with_the_python_writer(threads):
    """besst_bam_write_records (bench / test scaffolding, parallel deflate) lays the records out as tests/bam_writer.py
    does with align_records=True: both files inflate to the same bytes, and both read back to the batch."""
    import gzip
    asm = synth.make_assembly(80, 1500, 41)
    batch = synth.simulate_library(asm, synth.LibrarySpec('fr', 500.0, 50.0), 7000, 42)
    batch.rlen[::7] = 0
    with tempfile.TemporaryDirectory() as tmp:
        a, b = os.path.join(tmp, 'native.bam'), os.path.join(tmp, 'python.bam')
        bamio.write_bam(a, batch, threads=threads, level=6)
        bam_writer.write_bam(b, batch, align_records=True)
        with gzip.open(a, 'rb') as fa, gzip.open(b, 'rb') as fb:     # BGZF is a series of gzip members
            assert fa.read() == fb.read()
        got = bamio.read_bam(a, threads=3, chunk_records=1000)
    for col in ('tid', 'mtid', 'pos', 'mpos', 'tlen', 'flag', 'mapq', 'qlen', 'rlen'):
        assert np.array_equal(getattr(got, col), getattr(batch, col)), col


def test_a_wrong_block_crc_is_an_error():
    """BGZF keeps the CRC-32 of every block's inflated bytes; htslib (what the reference reads through) checks it, and so
    does the reader: a file whose stored CRC does not match is refused, not decoded."""
    asm = synth.make_assembly(40, 1500, 5)
    batch = synth.simulate_library(asm, synth.LibrarySpec('fr', 500.0, 50.0), 3000, 6)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'x.bam')
        bam_writer.write_bam(path, batch, block_bytes=20000, align_records=True)
        assert len(bamio.read_bam(path, threads=2)) == len(batch)
        raw = bytearray(open(path, 'rb').read())
        # the third block's trailer: walk the BSIZE fields
        at = 0
        for _ in range(2):
            at += (raw[at + 16] | (raw[at + 17] << 8)) + 1
        end = at + (raw[at + 16] | (raw[at + 17] << 8)) + 1
        raw[end - 8] ^= 0x01                                 # one bit of its CRC-32
        with open(path, 'wb') as fh:
            fh.write(raw)
        with pytest.raises(IOError):
            bamio.read_bam(path, threads=2)
