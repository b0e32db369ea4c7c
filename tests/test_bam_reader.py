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
