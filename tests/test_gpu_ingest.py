"""BAM ingest on the GPU (csrc/bgzf_gpu.hip): the device inflate against zlib byte for byte, the device record decode
against the host reader column by column, and the hand-over to the host form for layouts the device form does not take."""
import os
import random
import struct
import tempfile
import zlib

import numpy as np
import pytest

from besst_amd import _lib, bamio, synth
from tests import bam_writer

pytestmark = pytest.mark.gpu

COLS = ('tid', 'mtid', 'pos', 'mpos', 'tlen', 'flag', 'mapq', 'qlen')


def _bgzf(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, extra=b''):
    """One BGZF block (a gzip member whose first extra subfield is BC); `extra`: further subfields behind it."""
    comp = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    payload = comp.compress(data) + comp.flush()
    bsize = len(payload) + 25 + len(extra)           # BSIZE: the block's size minus one
    assert bsize <= 65535
    head = struct.pack('<BBBBIBBHBBHH', 31, 139, 8, 4, 0, 0, 255, 6 + len(extra), ord('B'), ord('C'), 2, bsize)
    return head + extra + payload + struct.pack('<II', zlib.crc32(data) & 0xffffffff, len(data))


@pytest.fixture(params=['second', 'first'])
def inflate_form(request, monkeypatch):
    """Both forms of the inflate kernel (csrc/bgzf_gpu.hip): the second - the lanes decode a DEFLATE block's symbols side by
    side, the default - and the first (BESST_INFLATE=1: every lane decodes at one bit position)."""
    if request.param == 'first':
        monkeypatch.setenv('BESST_INFLATE', '1')
    else:
        monkeypatch.delenv('BESST_INFLATE', raising=False)
    return request.param


def _payloads():
    rnd = random.Random(11)
    out = {'empty': b'', 'one': b'x', 'zeros': bytes(65280), 'random': os.urandom(40000),
           'acgt': bytes(rnd.choice(b'ACGT') for _ in range(65280)),
           'period3': b'abc' * 20000, 'period70': bytes(range(70)) * 900}
    words = [os.urandom(rnd.randint(1, 14)) for _ in range(400)]
    out['words'] = b''.join(rnd.choice(words) for _ in range(12000))[:65280]
    out['skewed'] = bytes(int(rnd.expovariate(0.03)) & 255 for _ in range(50000))   # long Huffman codes
    far = os.urandom(2000)
    out['far_matches'] = (far + os.urandom(30700) + far + os.urandom(29000) + far)[:65280]   # distances near 32768
    out['bamlike'] = b''.join(struct.pack('<IiiBBHHHIiii', 180, 7, 1000 + 13 * i, 8, 60, 4680, 1, 99, 100, 7, 1400 + 13 * i, 500) +
                              (b'read%03d\0' % (i % 1000)) + struct.pack('<I', 100 << 4) + bytes([0x12, 0x48] * 25) + b'I' * 100
                              for i in range(330))[:65280]
    return out


@pytest.mark.parametrize('level,strategy', [(0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY),
                                            (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE),
                                            (9, zlib.Z_FILTERED)])
def test_inflate_equals_zlib(level, strategy, inflate_form):
    """Stored, fixed and dynamic blocks, codes longer than the primary table, overlapping and far matches, the empty block,
    blocks at every payload alignment (extra subfields of 0..3 bytes shift the DEFLATE stream)."""
    pay = _payloads()
    data = b''
    want = b''
    for k, (name, raw) in enumerate(sorted(pay.items())):
        if level == 0 and len(raw) > 65000:
            raw = raw[:65000]                   # (stored blocks add 5 bytes per 65535)
        extra = struct.pack('<BBH', ord('X'), ord('Y'), k % 4) + bytes(k % 4)
        data += _bgzf(raw, level, strategy, extra if k % 2 else b'')
        want += raw
    got = bamio.inflate_bgzf_device(data, out_cap=len(want) + 16)
    assert len(got) == len(want)
    assert got == want


def test_inflate_code_shapes_at_random(inflate_form):
    """Payloads that push the symbol loop's batches to their ends - one- and two-bit codes (more symbols in a window of 64 bit
    positions than a batch lays into lanes), long skewed codes (the second-level table and the codes beyond it), long
    matches, runs - under every strategy and level, 3000 blocks against the bytes that were compressed."""
    rnd = random.Random(77)
    for _ in range(15):
        blocks, want = [], []
        for i in range(200):
            kind = rnd.randrange(6)
            n = rnd.randint(1, 65000)
            if kind == 0:
                raw = bytes(rnd.choice(b'\x00\xff') for _ in range(min(n, 20000)))
            elif kind == 1:
                raw = bytes(rnd.choice(b'ACGT') for _ in range(min(n, 30000)))
            elif kind == 2:
                raw = os.urandom(min(n, 3000)) * rnd.randint(1, 20)
            elif kind == 3:
                raw = bytes(int(rnd.expovariate(0.02)) & 255 for _ in range(min(n, 20000)))
            elif kind == 4:
                raw = bytes([rnd.randrange(3)]) * n
            else:
                raw = b''.join(os.urandom(rnd.randint(1, 9)) * rnd.randint(1, 40) for _ in range(300))
            raw = raw[:65000]
            strategy = rnd.choice((zlib.Z_DEFAULT_STRATEGY, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED, zlib.Z_FILTERED))
            blocks.append(_bgzf(raw, rnd.choice((1, 1, 6, 9)), strategy))
            want.append(raw)
        got = bamio.inflate_bgzf_device(b''.join(blocks), out_cap=sum(map(len, want)) + 16)
        assert got == b''.join(want)


def test_inflate_many_deflate_blocks_in_one_bgzf_block(inflate_form):
    """A BGZF block is one DEFLATE stream of ANY number of blocks: flush points every few hundred bytes (dynamic blocks of a
    dozen symbols - shorter than the 64 ranges the second form cuts a block's bits into -, each followed by an empty
    stored block), full flushes (the window restarts), and levels that alternate stored and dynamic blocks."""
    rnd = random.Random(5)
    src = _payloads()
    blocks, want = [], []
    for i in range(120):
        raw = src[rnd.choice(('bamlike', 'words', 'acgt', 'skewed', 'random'))][rnd.randint(0, 3000):][:rnd.randint(1, 40000)]
        comp = zlib.compressobj(rnd.choice((1, 6, 9)), zlib.DEFLATED, -15, 8, rnd.choice((zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY)))
        payload, at = b'', 0
        while at < len(raw):
            step = rnd.choice((1, 7, 40, 300, 2500, 20000))
            payload += comp.compress(raw[at:at + step]) + comp.flush(rnd.choice((zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_SYNC_FLUSH)))
            at += step
        payload += comp.flush()
        if len(payload) + 25 > 65535:
            continue
        head = struct.pack('<BBBBIBBHBBHH', 31, 139, 8, 4, 0, 0, 255, 6, ord('B'), ord('C'), 2, len(payload) + 25)
        blocks.append(head + payload + struct.pack('<II', zlib.crc32(raw) & 0xffffffff, len(raw)))
        want.append(raw)
    assert len(blocks) > 60
    got = bamio.inflate_bgzf_device(b''.join(blocks), out_cap=sum(map(len, want)) + 16)
    assert got == b''.join(want)


def test_inflate_blocks_of_the_largest_size_and_of_symbols_only(inflate_form):
    """ISIZE = 65536 (the most a BGZF block holds) made of literals alone - as many symbols as bytes, the most the second
    form's symbol buffer has to take for a block - and of one-bit codes (a range of the block's bits holds as many symbols
    as bits); codes of one length, which a decoder that starts inside a symbol never falls into step with (the hand-overs
    take their 64 rounds)."""
    rnd = random.Random(9)
    cases = [(bytes(rnd.choice(b'ACGT') for _ in range(65536)), 6, zlib.Z_HUFFMAN_ONLY),
             (bytes(rnd.choice(b'AC') for _ in range(65536)), 6, zlib.Z_HUFFMAN_ONLY),
             (bytes(rnd.choice(b'ACGTNacg') for _ in range(65536)), 6, zlib.Z_HUFFMAN_ONLY),     # eight codes of three bits
             (bytes(rnd.choice(bytes(range(16))) for _ in range(65536)), 9, zlib.Z_HUFFMAN_ONLY),  # sixteen of four
             (b'\x00' * 65536, 1, zlib.Z_DEFAULT_STRATEGY), (b'ab' * 32768, 9, zlib.Z_DEFAULT_STRATEGY),
             (bytes(rnd.choice(b'ACGT') for _ in range(65536)), 1, zlib.Z_DEFAULT_STRATEGY)]
    data = b''.join(_bgzf(raw, level, strategy) for raw, level, strategy in cases)
    got = bamio.inflate_bgzf_device(data * 3, out_cap=3 * 65536 * len(cases) + 16)
    assert got == b''.join(raw for raw, _, _ in cases) * 3


def test_a_stream_that_claims_more_than_its_block_holds_is_refused(inflate_form):
    """ISIZE smaller than what the DEFLATE stream makes (a valid stream under a wrong trailer): the block is refused, the
    blocks around it - whose bytes and, in the second form, symbols lie next to its own - inflate to what they were."""
    src = _payloads()
    big, small = src['zeros'], src['bamlike'][:30000]
    for raw, claim in ((big, 300), (big, 65000), (src['acgt'], 100), (src['bamlike'], 64), (src['words'], 4000)):
        liar = bytearray(_bgzf(raw, 6))
        liar[-4:] = struct.pack('<I', claim)
        good = _bgzf(small, 6)
        with pytest.raises(_lib.BesstDeviceError) as e:
            bamio.inflate_bgzf_device(good + bytes(liar) + good, out_cap=3 * 65536)
        assert 'block 1' in str(e.value)
        assert bamio.inflate_bgzf_device(good * 2, out_cap=60016) == small * 2
    # ... and far more: 40 MB of zeros in 40 KB of payload - one lane's range of the block's bits makes more bytes than the
    # lanes' counters hold (2^17): the places the symbols and bytes go to are checked, not trusted
    comp = zlib.compressobj(9, zlib.DEFLATED, -15)
    payload = comp.compress(bytes(40 << 20)) + comp.flush()
    assert 30000 < len(payload) < 65000
    for claim in (65536, 1000):
        bomb = struct.pack('<BBBBIBBHBBHH', 31, 139, 8, 4, 0, 0, 255, 6, ord('B'), ord('C'), 2, len(payload) + 25) + payload + struct.pack('<II', 0, claim)
        good = _bgzf(small, 6)
        with pytest.raises(_lib.BesstDeviceError) as e:
            bamio.inflate_bgzf_device(good + bomb + good, out_cap=4 * 65536)
        assert 'block 1' in str(e.value)
        assert bamio.inflate_bgzf_device(good * 2, out_cap=60016) == small * 2


def test_inflate_streams_of_libdeflate(inflate_form):
    """BGZF blocks compressed by libdeflate - the compressor htslib (pysam, samtools, the aligners' writers) uses for them:
    another block splitting, other length-limited codes, near-optimal parsing at level 12 - levels 1 to 12, every payload."""
    from tests import libdeflate_util as LD
    if not LD.available():
        pytest.skip('no libdeflate in this image')
    rnd = random.Random(12)
    pay = _payloads()
    blocks, want = [], []
    for level in (1, 3, 6, 9, 12):
        for name, raw in sorted(pay.items()):
            raw = raw[:65000]
            payload = LD.deflate(raw, level)
            if len(payload) + 25 > 65535:
                continue                                     # (incompressible at this size: not a BGZF block)
            head = struct.pack('<BBBBIBBHBBHH', 31, 139, 8, 4, 0, 0, 255, 6, ord('B'), ord('C'), 2, len(payload) + 25)
            blocks.append(head + payload + struct.pack('<II', zlib.crc32(raw) & 0xffffffff, len(raw)))
            want.append(raw)
    for _ in range(300):                                     # ... and slices of the BAM-like payload at any length
        raw = pay['bamlike'][rnd.randint(0, 5000):][:rnd.randint(1, 60000)]
        payload = LD.deflate(raw, rnd.choice((1, 6, 6, 12)))
        head = struct.pack('<BBBBIBBHBBHH', 31, 139, 8, 4, 0, 0, 255, 6, ord('B'), ord('C'), 2, len(payload) + 25)
        blocks.append(head + payload + struct.pack('<II', zlib.crc32(raw) & 0xffffffff, len(raw)))
        want.append(raw)
    assert len(blocks) > 340
    got = bamio.inflate_bgzf_device(b''.join(blocks), out_cap=sum(map(len, want)) + 16)
    assert got == b''.join(want)


def test_inflate_many_blocks(inflate_form):
    """More blocks than waves fit the chip, in one call and across the hook's chunks."""
    rnd = random.Random(3)
    base = os.urandom(3000)
    blocks, want = [], []
    for i in range(5000):
        n = rnd.randint(0, 9000)
        raw = (base * 4)[rnd.randint(0, 2000):][:n] + bytes(rnd.randint(0, 50))
        blocks.append(_bgzf(raw, rnd.choice((1, 6))))
        want.append(raw)
    got = bamio.inflate_bgzf_device(b''.join(blocks), out_cap=sum(map(len, want)) + 16)
    assert got == b''.join(want)


def test_inflate_reports_a_corrupt_block(inflate_form):
    raw = os.urandom(500) * 20
    good = _bgzf(raw)
    bad = bytearray(_bgzf(raw))
    for i in range(40, 60):
        bad[i] ^= 0x5a
    with pytest.raises(_lib.BesstDeviceError) as e:
        bamio.inflate_bgzf_device(good + bytes(bad) + good, out_cap=3 * len(raw) + 16)
    assert 'block 1' in str(e.value)


def _library(n_pairs=9000, seed=32):
    asm = synth.make_assembly(120, 1500, seed - 1)
    batch = synth.simulate_library(asm, synth.LibrarySpec('rf', 1500.0, 150.0, contam_frac=0.2), n_pairs, seed)
    batch.rlen[::9] = 0
    return batch


def _check_against_host(path, bam, host):
    got = bam.ctx.fetch_records()
    assert len(bam) == len(host)
    for col in COLS:
        assert np.array_equal(got[col], getattr(host, col)), col
    k = min(1000, len(host))
    assert np.array_equal(bam.rlen, host.rlen[:k]) and np.array_equal(bam.alen, host.alen[:k])
    assert np.array_equal(bam.qlen, host.qlen[:k])
    assert list(bam.references) == list(host.references) and list(bam.lengths) == list(host.lengths)


@pytest.mark.parametrize('writer,level,block_bytes,chunk_blocks', [('native', 1, 0, 0), ('native', 6, 0, 64), ('python', 6, 60000, 64),
                                                                   ('python', 6, 3000, 100), ('python', 6, 700, 0)])
def test_device_ingest_equals_host_reader(writer, level, block_bytes, chunk_blocks):
    """htslib's layout (every block begins with a record): the device form's columns, head arrays and counts equal the
    host reader's; several chunks (64 blocks each), blocks of a few records, one chunk."""
    batch = _library(30000 if writer == 'native' else 6000)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'x.bam')
        if writer == 'native':
            bamio.write_bam(path, batch, threads=4, level=level)
        else:
            bam_writer.write_bam(path, batch, block_bytes=block_bytes, align_records=True)
        host = bamio.read_bam(path, threads=2)
        bam = bamio.ResidentBam(path, threads=3, mode='device', chunk_blocks=chunk_blocks)
        try:
            assert bam.ingest.on_device == 1 and bam.ingest.blocks > 0 and bam.ingest.starts_repaired == 0
            assert bam.ingest.bytes_h2d <= os.path.getsize(path)
            _check_against_host(path, bam, host)
        finally:
            bam.close()
    for col in COLS:
        assert np.array_equal(getattr(host, col), getattr(batch, col)), col


@pytest.mark.parametrize('level,align', [(6, True), (1, True), (12, True), (6, False)])
def test_a_file_whose_blocks_libdeflate_compressed(level, align, inflate_form, monkeypatch):
    """A BAM as htslib writes it where it is built with libdeflate (pysam, samtools, the aligners): blocks of 0xff00 bytes
    that begin with a record, compressed by libdeflate - and the same compressor under blocks cut at arbitrary bytes.  The
    device form's records equal the host reader's (zlib's inflate) and the batch the file was written from."""
    from tests import libdeflate_util as LD
    if not LD.available():
        pytest.skip('no libdeflate in this image')
    monkeypatch.setattr(bam_writer, 'LIBDEFLATE_LEVEL', level)
    batch = _library(12000)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'x.bam')
        bam_writer.write_bam(path, batch, block_bytes=0xff00, align_records=align)
        host = bamio.read_bam(path, threads=2)
        bam = bamio.ResidentBam(path, threads=3, mode='device', chunk_blocks=16)
        try:
            assert bam.ingest.on_device == 1 and bam.ingest.blocks > 30
            _check_against_host(path, bam, host)
        finally:
            bam.close()
    for col in COLS:
        assert np.array_equal(getattr(host, col), getattr(batch, col)), col


@pytest.mark.parametrize('block_bytes,chunk_blocks', [(5000, 0), (5000, 7), (65280, 3), (700, 64), (90, 0), (90, 50)])
def test_straddling_records_are_decoded_on_the_device(block_bytes, chunk_blocks):
    """Blocks cut at arbitrary bytes (htsjdk / Picard; htslib never does that): records run on into the next block, the
    next chunk (the unfinished record travels in front of the next chunk's first block), or - 90-byte blocks - cover blocks
    whole.  The device form locates every record start (guess per block, verified from block to block) and leaves the host
    reader's columns."""
    batch = _library(4000)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'x.bam')
        bam_writer.write_bam(path, batch, block_bytes=block_bytes, align_records=False)
        host = bamio.read_bam(path, threads=2)
        bam = bamio.ResidentBam(path, threads=2, mode='device', chunk_blocks=chunk_blocks)
        try:
            assert bam.ingest.on_device == 1
            _check_against_host(path, bam, host)
        finally:
            bam.close()
    for col in COLS:
        assert np.array_equal(getattr(host, col), getattr(batch, col)), col


@pytest.mark.parametrize('block_bytes,chunk_blocks', [(5000, 0), (300, 40)])
def test_bytes_that_look_like_a_record_do_not_mislead_the_device(block_bytes, chunk_blocks):
    """Every record's qualities spell the fixed fields of a record, so that most blocks of a straddling layout begin with a
    stretch that passes for a record start and is none: the block-to-block verification replaces each wrong guess by what
    the block before it says, and the columns are the host reader's."""
    batch = _library(3000)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'x.bam')
        bam_writer.write_bam(path, batch, block_bytes=block_bytes, align_records=False, decoys=True)
        host = bamio.read_bam(path, threads=2)
        bam = bamio.ResidentBam(path, threads=2, mode='device', chunk_blocks=chunk_blocks)
        try:
            assert bam.ingest.on_device == 1 and bam.ingest.starts_repaired > 10      # (the test tests something)
            _check_against_host(path, bam, host)
        finally:
            bam.close()
    for col in COLS:
        assert np.array_equal(getattr(host, col), getattr(batch, col)), col


def test_pinned_staging_is_pooled_and_can_be_released():
    """The ingest forms' pinned staging buffers go back to a process-wide pool: a second ingest finds them (same result), and
    besst_release_cached_memory() empties the pool without disturbing the next call."""
    batch = _library(8000)
    lib = _lib.load()
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'x.bam')
        bamio.write_bam(path, batch, threads=4, level=1)
        host = bamio.read_bam(path, threads=2)
        for mode in ('device', 'device', 'host', 'release', 'device', 'host'):
            if mode == 'release':
                lib.besst_release_cached_memory()
                continue
            bam = bamio.ResidentBam(path, threads=2, mode=mode)
            try:
                assert bam.ingest.on_device == (1 if mode == 'device' else 0)
                _check_against_host(path, bam, host)
            finally:
                bam.close()


def test_random_block_layouts():
    """Blocks cut at random sizes (30 bytes ... 64 KiB, changing from block to block; now and then an empty block), reads of
    random lengths incl. none, some with bytes in their qualities that pass for a record header, random chunk sizes: the device
    form leaves the host reader's columns.  BESST_FUZZ_ROUNDS (default 12) sets the number of files."""
    rounds = int(os.environ.get('BESST_FUZZ_ROUNDS', '12'))
    for seed in range(rounds):
        rng = np.random.default_rng(1000 + seed)
        batch = _library(int(rng.integers(200, 3000)), seed=40 + seed)
        n = len(batch)
        want = rng.choice([0, 36, 100, 151, 250, 5000, 70000], n, p=[.1, .2, .4, .2, .08, .015, .005])
        batch.rlen[:] = np.where(want == 0, 0, np.maximum(want, batch.qlen)).astype(batch.rlen.dtype)    # (soft clip = rlen - qlen)
        with tempfile.TemporaryDirectory() as tmp:
            raw_path, path = os.path.join(tmp, 'raw.bam'), os.path.join(tmp, 'x.bam')
            bam_writer.write_bam(raw_path, batch, block_bytes=65280, align_records=False, decoys=bool(seed & 1))
            # re-cut the same BAM bytes into blocks of random sizes
            data, at, raw = open(raw_path, 'rb').read(), 0, bytearray()
            while at < len(data):
                size = struct.unpack_from('<H', data, at + 16)[0] + 1
                raw += zlib.decompress(data[at + 18:at + size - 8], -15)
                at += size
            small = int(rng.choice([30, 90, 400, 3000, 20000, 65280]))
            with open(path, 'wb') as fh:
                at = 0
                while at < len(raw):
                    take = int(rng.integers(1, small + 1))
                    fh.write(bam_writer._bgzf_block(bytes(raw[at:at + take])))
                    at += take
                    if rng.random() < 0.02:
                        fh.write(bam_writer._bgzf_block(b''))
                fh.write(bam_writer._bgzf_block(b''))
            host = bamio.read_bam(path, threads=2)
            bam = bamio.ResidentBam(path, threads=2, mode='device', chunk_blocks=int(rng.choice([0, 64, 100, 1000])))
            try:
                assert bam.ingest.on_device == 1, seed
                _check_against_host(path, bam, host)
            finally:
                bam.close()
            # the same file in slices (multi-rank ingest): contiguous pieces of the stream, together the whole of it
            from besst_amd import distributed
            world = int(rng.integers(2, 5))
            try:
                slices, _ = distributed.ingest_all_slices(path, world, device_index=0, threads=2, chunk_blocks=int(rng.choice([0, 64])))
            except _lib.BesstDeviceError as e:              # (a slice that one long read covers whole)
                assert 'no record begins' in str(e) or 'behind its first chunk' in str(e), (seed, str(e))
                slices = None
            if slices is not None:
                try:
                    for col in COLS:
                        parts = [b.ctx.fetch_records()[col] for b, _ in slices]
                        assert np.array_equal(np.concatenate(parts), getattr(host, col)), (seed, world, col)
                finally:
                    for b, _ in slices:
                        b.close()
        for col in COLS:
            assert np.array_equal(getattr(host, col), getattr(batch, col)), (seed, col)


def test_columns_grow_again_while_chunks_are_in_flight():
    """The record columns are sized from the records-per-byte of the chunks seen so far; a file whose first part holds long
    reads and whose rest holds short ones (many more records per compressed byte) makes that estimate fall short again and
    again, so the columns move while other chunks' kernels are queued: the result is still the host reader's."""
    batch = _library(24000)
    n = len(batch)
    batch.rlen[:] = 0
    batch.rlen[:n // 8] = 4000
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'x.bam')
        bam_writer.write_bam(path, batch, block_bytes=60000, align_records=True)
        host = bamio.read_bam(path, threads=2)
        bam = bamio.ResidentBam(path, threads=2, mode='device', chunk_blocks=64)
        try:
            assert bam.ingest.on_device == 1 and bam.ingest.chunks >= 6
            _check_against_host(path, bam, host)
        finally:
            bam.close()
    for col in COLS:
        assert np.array_equal(getattr(host, col), getattr(batch, col)), col


def test_long_reads_cover_blocks_and_chunks_whole():
    """Reads of 150-400 kb (a record of up to 0.6 MB) in 64 KiB blocks and chunks of 64 blocks: records cover several blocks
    whole - blocks in which no record begins -, some run on into the next chunk as a tail of several blocks; the device form
    leaves the host reader's columns."""
    batch = _library(3000)
    rng = np.random.default_rng(5)
    pick = rng.choice(len(batch), 60, replace=False)
    batch.rlen[pick] = rng.integers(150000, 400000, 60).astype(batch.rlen.dtype)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'x.bam')
        bam_writer.write_bam(path, batch, block_bytes=65280, align_records=False)
        host = bamio.read_bam(path, threads=2)
        bam = bamio.ResidentBam(path, threads=2, mode='device', chunk_blocks=64)
        try:
            assert bam.ingest.on_device == 1 and bam.ingest.chunks >= 3
            _check_against_host(path, bam, host)
        finally:
            bam.close()
    for col in COLS:
        assert np.array_equal(getattr(host, col), getattr(batch, col)), col


def test_a_file_that_ends_inside_a_record_is_an_error():
    """A straddling file cut off behind a block in the middle of a record (and closed with an EOF marker): the device form
    reports it (BESST_ERR_ARG: corrupt input, not an unsupported form) and appends nothing."""
    batch = _library(2000)
    with tempfile.TemporaryDirectory() as tmp:
        path, cut = os.path.join(tmp, 'x.bam'), os.path.join(tmp, 'cut.bam')
        bam_writer.write_bam(path, batch, block_bytes=5000, align_records=False)
        data = open(path, 'rb').read()
        offs, at = [], 0
        while at < len(data):
            offs.append(at)
            at += struct.unpack_from('<H', data, at + 16)[0] + 1
        eof = data[offs[-1]:]
        with open(cut, 'wb') as fh:
            fh.write(data[:offs[len(offs) // 2]] + eof)
        with pytest.raises(_lib.BesstDeviceError) as e:
            bamio.ResidentBam(cut, threads=2, mode='device')
        assert 'ends inside a record' in str(e.value)


@pytest.mark.parametrize('block_bytes,world,decoys,chunk_blocks', [(5000, 2, False, 0), (5000, 5, False, 64), (700, 3, False, 64),
                                                                   (90, 4, False, 0), (5000, 6, True, 64), (65280, 3, False, 64)])
def test_slices_of_a_straddling_file_tile_its_records(block_bytes, world, decoys, chunk_blocks):
    """Multi-GPU ingest of a file whose records straddle BGZF blocks: a record belongs to the slice it begins in; every rank
    guesses where its slice's first record begins, the slices' (offset used, bytes of the last record in the next slice)
    are compared and a slice whose guess was wrong is read again (distributed.ingest_all_slices: the protocol of
    ingest_slice with simulated ranks).  The slices are contiguous pieces of the stream and together the whole of it; with
    bytes in the qualities that pass for a record header some guesses ARE wrong."""
    from besst_amd import distributed
    batch = _library(6000)
    if block_bytes == 65280:
        batch.rlen[::50] = 9000                              # (records long enough to straddle 64 KiB blocks often)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'x.bam')
        bam_writer.write_bam(path, batch, block_bytes=block_bytes, align_records=False, decoys=decoys)
        slices, rereads = distributed.ingest_all_slices(path, world, device_index=0, threads=2, chunk_blocks=chunk_blocks)
        try:
            got = {c: [] for c in COLS}
            for r, (bam, _) in enumerate(slices):
                assert bam.ingest.on_device == 1
                if r > 0:
                    assert bam.boundary[0] == slices[r - 1][0].boundary[1]
                cols = bam.ctx.fetch_records()
                for c in COLS:
                    got[c].append(cols[c])
            assert slices[-1][0].boundary[1] == 0
            assert any(b.boundary[0] > 0 for b, _ in slices[1:])          # (the layout does straddle)
        finally:
            for bam, _ in slices:
                bam.close()
    if decoys:
        assert rereads > 0                                   # (the test tests something)
    for c in COLS:
        assert np.array_equal(np.concatenate(got[c]), getattr(batch, c)), c


def test_a_part_of_a_straddling_file_is_refused():
    """The PART form (besst_ctx_push_bam_device_part) begins every part with its first block's first byte, which only
    htslib's layout allows: a part of a file whose records straddle blocks answers BESST_ERR_UNSUPPORTED (context and reader
    untouched) - such files take the slice form (test above)."""
    batch = _library(4000)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'x.bam')
        bam_writer.write_bam(path, batch, block_bytes=5000, align_records=False)
        for part in ((0, 2), (1, 2)):
            with pytest.raises(_lib.BesstDeviceError) as e:
                bamio.ResidentBam(path, threads=2, part=part)
            assert 'status 5' in str(e.value)


@pytest.mark.parametrize('name', ['handmade_a.bam', 'handmade_b.bam'])
def test_hand_assembled_bam(name):
    """The fixtures of tests/golden/bam (every CIGAR operation, records without CIGAR / sequence, straddling records, an
    empty block, no EOF marker) through 'auto': whichever form runs, the columns are the host reader's."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'bam')
    path = os.path.join(here, name)
    host = bamio.read_bam(path, threads=2)
    bam = bamio.ResidentBam(path, threads=2, mode='auto')
    try:
        _check_against_host(path, bam, host)
    finally:
        bam.close()


def test_every_cigar_operation_on_the_device():
    """The hand-made records re-packed into htslib's layout, so that the DEVICE decode sees them: CIGAR I/D/N/S/H/P/=/X,
    no CIGAR, no sequence, the CG:B,I placeholder, long and short names, auxiliary fields."""
    import gzip
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'bam')
    with gzip.open(os.path.join(here, 'handmade_a.bam'), 'rb') as fh:
        raw = fh.read()
    l_text = struct.unpack_from('<I', raw, 4)[0]
    at = 8 + l_text
    n_ref = struct.unpack_from('<I', raw, at)[0]
    at += 4
    for _ in range(n_ref):
        at += 8 + struct.unpack_from('<I', raw, at)[0]
    header, recs = raw[:at], []
    while at < len(raw):
        size = struct.unpack_from('<I', raw, at)[0]
        recs.append(raw[at:at + 4 + size])
        at += 4 + size
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'aligned.bam')
        with open(path, 'wb') as fh:
            fh.write(_bgzf(header))
            for i in range(0, len(recs), 3):                 # three records per block
                fh.write(_bgzf(b''.join(recs[i:i + 3]), level=(1, 6, 9)[i % 3]))
            fh.write(_bgzf(b''))
        host = bamio.read_bam(path, threads=2)
        bam = bamio.ResidentBam(path, threads=2, mode='device')
        try:
            assert bam.ingest.on_device == 1 and len(bam) == len(recs)
            _check_against_host(path, bam, host)
        finally:
            bam.close()


def test_header_and_records_share_a_block():
    """The first record begins in the middle of the header's block (in-block offset of the reader's position)."""
    batch = _library(800)
    with tempfile.TemporaryDirectory() as tmp:
        a, path = os.path.join(tmp, 'a.bam'), os.path.join(tmp, 'x.bam')
        bam_writer.write_bam(a, batch, block_bytes=60000, align_records=True)
        import gzip
        with gzip.open(a, 'rb') as fh:
            raw = fh.read()
        l_text = struct.unpack_from('<I', raw, 4)[0]
        at = 8 + l_text
        n_ref = struct.unpack_from('<I', raw, at)[0]
        at += 4
        for _ in range(n_ref):
            at += 8 + struct.unpack_from('<I', raw, at)[0]
        cuts = [0]
        pos = at
        while pos < len(raw):                                # header + the first records in one block, then record-aligned blocks
            size = struct.unpack_from('<I', raw, pos)[0]
            if pos + 4 + size - cuts[-1] > 30000:
                cuts.append(pos)
            pos += 4 + size
        cuts.append(len(raw))
        with open(path, 'wb') as fh:
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                fh.write(_bgzf(raw[lo:hi]))
            fh.write(_bgzf(b''))
        host = bamio.read_bam(path, threads=2)
        bam = bamio.ResidentBam(path, threads=2, mode='device')
        try:
            assert bam.ingest.on_device == 1
            _check_against_host(path, bam, host)
        finally:
            bam.close()


def test_second_library_appends():
    """push_bam on a context that already holds records: the new ones follow them."""
    from besst_amd import device
    batch = _library(3000)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'x.bam')
        bamio.write_bam(path, batch, threads=2, level=1)
        lib = _lib.load()
        with device.GraphContext(0) as ctx:
            zeros = [0] * len(batch.references)
            ctx.set_contigs(scaf_id=zeros, scaf_len=zeros, ctg_pos=zeros, ctg_len=zeros, direction=zeros, cls=zeros)
            for _ in range(2):
                handle, _, _ = bamio._open(lib, path, 2)
                try:
                    stats, _, _, _ = ctx.push_bam(handle, mode='device', chunk_blocks=64)
                    assert stats.on_device == 1 and stats.records == len(batch)
                finally:
                    lib.besst_bam_close(handle)
            got = ctx.fetch_records()
    for col in COLS:
        want = np.concatenate([getattr(batch, col), getattr(batch, col)])
        assert np.array_equal(got[col], want), col


def _ingest_parts(path, parts, **kw):
    got, counts = {c: [] for c in COLS}, []
    for r in range(parts):
        bam = bamio.ResidentBam(path, threads=2, part=(r, parts), **kw)
        try:
            assert bam.ingest.on_device == 1
            cols = bam.ctx.fetch_records()
            counts.append(len(bam))
            for c in COLS:
                got[c].append(cols[c])
        finally:
            bam.close()
    return {c: np.concatenate(v) for c, v in got.items()}, counts


@pytest.mark.parametrize('parts', [2, 3, 8])
def test_parts_of_a_file_tile_its_records(parts):
    """Multi-GPU ingest: rank r of W takes part (r, W) - the parts, cut at BGZF block boundaries every rank finds on its
    own, are contiguous slices of the stream and together the whole of it."""
    batch = _library(20000)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'x.bam')
        bamio.write_bam(path, batch, threads=3, level=1, realistic=True)
        whole, counts = _ingest_parts(path, parts, chunk_blocks=64)
    assert sum(counts) == len(batch) and min(counts) > 0
    assert max(counts) < 2 * len(batch) / parts            # (cut by bytes: about equal)
    for c in COLS:
        assert np.array_equal(whole[c], getattr(batch, c)), c


def test_parts_of_a_small_file_and_a_false_block_header():
    """More parts than blocks (most parts are empty), and payload bytes that spell a BGZF header (a stored block whose
    record carries them in an auxiliary field): a boundary is only where blocks chain behind it."""
    import gzip
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'bam')
    with gzip.open(os.path.join(here, 'handmade_a.bam'), 'rb') as fh:
        raw = fh.read()
    l_text = struct.unpack_from('<I', raw, 4)[0]
    at = 8 + l_text
    n_ref = struct.unpack_from('<I', raw, at)[0]
    at += 4
    for _ in range(n_ref):
        at += 8 + struct.unpack_from('<I', raw, at)[0]
    header, recs = raw[:at], []
    while at < len(raw):
        size = struct.unpack_from('<I', raw, at)[0]
        recs.append(raw[at:at + 4 + size])
        at += 4 + size
    fake = struct.pack('<BBBBIBBHBBHH', 31, 139, 8, 4, 0, 0, 255, 6, ord('B'), ord('C'), 2, 99) + bytes(range(60))
    body = recs[0][4:] + b'ZZB' + b'C' + struct.pack('<I', len(fake)) + fake       # an auxiliary B:C array holding it
    liar = struct.pack('<I', len(body)) + body
    recs = recs[:2] + [liar] * 40 + recs[2:]
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'aligned.bam')
        with open(path, 'wb') as fh:
            fh.write(_bgzf(header))
            for i in range(0, len(recs), 2):
                fh.write(_bgzf(b''.join(recs[i:i + 2]), level=0 if i % 4 == 0 else 6))
            fh.write(_bgzf(b''))
        host = bamio.read_bam(path, threads=2)
        assert len(host) == len(recs)
        for parts in (2, 5, 64):
            whole, counts = _ingest_parts(path, parts)
            assert sum(counts) == len(recs), (parts, counts)
            for c in COLS:
                assert np.array_equal(whole[c], getattr(host, c)), (parts, c)
        # the slice form on the same file: slices without a block pass on what the slice before them reports
        from besst_amd import distributed
        for world in (3, 64):
            slices, rereads = distributed.ingest_all_slices(path, world, device_index=0, threads=2)
            try:
                assert rereads == 0 and sum(len(b) for b, _ in slices) == len(recs)
                for c in COLS:
                    got = np.concatenate([b.ctx.fetch_records()[c] for b, _ in slices if len(b)])
                    assert np.array_equal(got, getattr(host, c)), (world, c)
            finally:
                for b, _ in slices:
                    b.close()


def test_endless_empty_blocks_end_at_the_payload():
    """A payload of empty fixed-Huffman blocks that never sets BFINAL - every bit valid, no output, no end: the kernel stops
    at the end of the payload instead of following the stream out of its buffer."""
    bits = '010' + '0000000'                                 # BFINAL 0, BTYPE 01 (LSB first: 1, 0), end-of-block code
    stream = ''.join(bits for _ in range(400))
    stream += '0' * (-len(stream) % 8)
    payload = bytes(int(stream[i:i + 8][::-1], 2) for i in range(0, len(stream), 8))
    assert zlib.decompressobj(-15).decompress(payload) == b''        # zlib agrees: valid so far, nothing produced
    bsize = len(payload) + 25
    head = struct.pack('<BBBBIBBHBBHH', 31, 139, 8, 4, 0, 0, 255, 6, ord('B'), ord('C'), 2, bsize)
    block = head + payload + struct.pack('<II', 0, 77)       # ISIZE claims 77 bytes
    good = _bgzf(b'abc' * 100)
    with pytest.raises(_lib.BesstDeviceError) as e:
        bamio.inflate_bgzf_device(good + block + good * 200, out_cap=1 << 20)
    assert 'block 1' in str(e.value)


def test_corrupt_payloads_are_refused_not_followed(inflate_form):
    """Random damage to the DEFLATE payloads of 300 blocks (bytes flipped, payloads cut short with ISIZE kept): the call
    reports the first block that does not inflate - or, where the damage happens to decode, still returns ISIZE bytes per
    block - and the device is fine afterwards."""
    rnd = random.Random(17)
    src = _payloads()['bamlike']
    blocks = []
    for i in range(300):
        raw = src[rnd.randint(0, 2000):][:rnd.randint(200, 60000)]
        b = bytearray(_bgzf(raw, rnd.choice((1, 6, 9))))
        kind = i % 3
        if kind == 0:                                        # a few flipped bytes inside the payload
            for _ in range(rnd.randint(1, 6)):
                b[rnd.randint(18, len(b) - 9)] ^= rnd.randint(1, 255)
        elif kind == 1:                                      # the payload's tail replaced by zeros
            cut = rnd.randint(18, len(b) - 9)
            b[cut:len(b) - 8] = bytes(len(b) - 8 - cut)
        else:                                                # garbage from some point on
            cut = rnd.randint(18, len(b) - 9)
            b[cut:len(b) - 8] = os.urandom(len(b) - 8 - cut)
        blocks.append(bytes(b))
    total = sum(struct.unpack('<I', b[-4:])[0] for b in blocks)
    try:
        out = bamio.inflate_bgzf_device(b''.join(blocks), out_cap=total + 16)
        assert len(out) == total                             # (every damaged stream happened to decode to its ISIZE)
    except _lib.BesstDeviceError as e:
        assert 'did not inflate' in str(e)
    good = _bgzf(src[:50000])
    assert bamio.inflate_bgzf_device(good * 3, out_cap=150016) == src[:50000] * 3


def test_a_wrong_block_crc_is_refused_by_both_forms():
    """The device checks every block's CRC-32 (bgzf_crc_kernel) against its gzip trailer: a payload that inflates to ISIZE
    bytes under a CRC that does not match is a refused block (status 10), in a file the device form answers
    BESST_ERR_UNSUPPORTED, and the host form - which checks too - refuses the file as well."""
    raw = _payloads()['bamlike'][:40000]
    good = _bgzf(raw)
    bad = bytearray(good)
    bad[-8] ^= 0x10
    assert bamio.inflate_bgzf_device(good * 2, out_cap=80016) == raw * 2
    with pytest.raises(_lib.BesstDeviceError) as e:
        bamio.inflate_bgzf_device(good + bytes(bad) + good, out_cap=120016)
    assert 'block 1' in str(e.value) and 'status 10' in str(e.value)
    batch = _library(30000)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'x.bam')
        bamio.write_bam(path, batch, threads=2, level=1)
        data = bytearray(open(path, 'rb').read())
        at = 0
        for _ in range(60):                                  # (a block the reader does not touch when it opens the file)
            at += (data[at + 16] | (data[at + 17] << 8)) + 1
        end = at + (data[at + 16] | (data[at + 17] << 8)) + 1
        data[end - 7] ^= 0x80
        with open(path, 'wb') as fh:
            fh.write(data)
        with pytest.raises(_lib.BesstDeviceError) as e:
            bamio.ResidentBam(path, threads=2, mode='device')
        assert 'status 5' in str(e.value)
        with pytest.raises((_lib.BesstDeviceError, IOError)):
            bamio.ResidentBam(path, threads=2, mode='auto')


_PROFILE_SCRIPT = r'''
import os, sys, tempfile
sys.path.insert(0, %(repo)r)
from besst_amd import bamio
from tests import test_gpu_ingest as T
batch = T._library(6000)
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, 'x.bam')
    bamio.write_bam(path, batch, threads=2, level=1)
    for mode in ('device', 'host'):
        bam = bamio.ResidentBam(path, threads=2, mode=mode)
        assert len(bam) == len(batch)
        bam.close()
print('DONE')
'''


def test_profile_knobs_print_their_phase_times():
    """BESST_INGEST_PROFILE=1 / BESST_BAM_PROFILE=1 (development aids of tools/ingest_probe.py and tools/bam_probe.py,
    read by the native code at call time): a line of phase times per call on stderr, the ingest itself unchanged."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, BESST_INGEST_PROFILE='1', BESST_BAM_PROFILE='1')
    out = subprocess.run([sys.executable, '-c', _PROFILE_SCRIPT % dict(repo=os.path.dirname(here))], env=env,
                         capture_output=True, text=True, timeout=600)
    assert 'DONE' in out.stdout, (out.stdout[-1000:], out.stderr[-2000:])
    assert '[push_bam_device] setup' in out.stderr and ' alloc ' in out.stderr and '[bam] read' in out.stderr, out.stderr[-2000:]


def _decode_bam_independently(path):
    """A BAM file read with Python's gzip + struct only (SAM specification section 4.2) - no code of this library - into the
    columns the path uses, with pysam 0.8.4's definitions restated from its documentation: qlen = query_alignment_length =
    l_seq (or, without a sequence, the CIGAR's M/I/S/=/X total) minus leading and trailing soft clips, hard clips skipped;
    rlen = query_length = l_seq; alen = reference_length = reference bases consumed by M/D/N/=/X (0 without a CIGAR)."""
    import gzip
    with gzip.open(path, 'rb') as fh:                        # (a BGZF file is a series of gzip members)
        raw = fh.read()
    assert raw[:4] == b'BAM\x01'
    at = 8 + struct.unpack_from('<i', raw, 4)[0]
    n_ref = struct.unpack_from('<i', raw, at)[0]
    at += 4
    refs = []
    for _ in range(n_ref):
        l_name = struct.unpack_from('<i', raw, at)[0]
        refs.append((raw[at + 4:at + 4 + l_name - 1].decode(), struct.unpack_from('<i', raw, at + 4 + l_name)[0]))
        at += 8 + l_name
    cols = {k: [] for k in COLS + ('rlen', 'alen', 'cigar')}
    while at < len(raw):
        size = struct.unpack_from('<i', raw, at)[0]
        tid, pos, l_name, mapq, _bin, n_cig, flag, l_seq, mtid, mpos, tlen = struct.unpack_from('<iiBBHHHiiii', raw, at + 4)
        ops = struct.unpack_from('<%dI' % n_cig, raw, at + 36 + l_name)
        kinds = [('MIDNSHP=X'[op & 15], op >> 4) for op in ops]
        soft = [n for k, n in kinds if k != 'H']             # hard clips are not part of the stored query
        lead = soft[0] if kinds and [k for k, _ in kinds if k != 'H'][0] == 'S' else 0
        trail = soft[-1] if len(soft) > 1 and [k for k, _ in kinds if k != 'H'][-1] == 'S' else 0
        q_total = l_seq if l_seq else sum(n for k, n in kinds if k in 'MIS=X')
        qlen = max(0, q_total - lead - trail) if kinds else l_seq
        alen = sum(n for k, n in kinds if k in 'MDN=X')
        for k, v in zip(COLS, (tid, mtid, pos, mpos, tlen, flag, mapq, min(qlen, 65535))):
            cols[k].append(v)
        cols['rlen'].append(l_seq)
        cols['alen'].append(alen)
        cols['cigar'].append(''.join('%d%s' % (n, k) for k, n in kinds) or '*')
        at += 4 + size
    return refs, cols


@pytest.mark.parametrize('name', ['handmade_a.bam', 'handmade_b.bam'])
def test_device_columns_equal_an_independent_decode(name):
    """The hand-assembled files against a decoder written in this test from the SAM specification and pysam 0.8.4's
    documented attribute semantics - nothing of besst_amd's readers or writers on the expected side: soft clips, hard
    clips, N skips, a CIGAR of nothing but clip and skip, no CIGAR, no sequence (the residual risk named in README: no
    byte of a real aligner's output has been through the path; these bytes at least are not the builder's own)."""
    import json
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'bam')
    refs, want = _decode_bam_independently(os.path.join(here, name))
    cigars = ' '.join(want['cigar'])
    assert 'S' in cigars and 'H' in cigars and 'N' in cigars and '*' in cigars and '100S150N' in cigars
    with open(os.path.join(here, 'handmade_bam.json')) as fh:
        doc = json.load(fh)                                  # the values stated by hand next to the bytes
    assert [r[0] for r in refs] == doc['references'] and [r[1] for r in refs] == doc['lengths']
    for k in COLS + ('rlen', 'alen'):
        assert want[k] == [r[k] for r in doc['records']], k
    bam = bamio.ResidentBam(os.path.join(here, name), threads=2, mode='auto')
    try:
        got = bam.ctx.fetch_records()
        assert len(bam) == len(want['tid'])
        for k in COLS:
            assert got[k].astype(np.int64).tolist() == want[k], k
        assert bam.rlen.tolist() == want['rlen'] and bam.alen.tolist() == want['alen']
        assert list(bam.references) == doc['references'] and list(bam.lengths) == doc['lengths']
    finally:
        bam.close()
    # and the same records in htslib's layout, so that the DEVICE decode is what is compared
    import gzip
    with gzip.open(os.path.join(here, name), 'rb') as fh:
        raw = fh.read()
    at = 8 + struct.unpack_from('<i', raw, 4)[0]
    n_ref = struct.unpack_from('<i', raw, at)[0]
    at += 4
    for _ in range(n_ref):
        at += 8 + struct.unpack_from('<i', raw, at)[0]
    header, recs = raw[:at], []
    while at < len(raw):
        size = struct.unpack_from('<i', raw, at)[0]
        recs.append(raw[at:at + 4 + size])
        at += 4 + size
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'aligned.bam')
        with open(path, 'wb') as fh:
            fh.write(_bgzf(header))
            for i in range(0, len(recs), 4):
                fh.write(_bgzf(b''.join(recs[i:i + 4])))
            fh.write(_bgzf(b''))
        bam = bamio.ResidentBam(path, threads=2, mode='device')
        try:
            assert bam.ingest.on_device == 1
            got = bam.ctx.fetch_records()
            for k in COLS:
                assert got[k].astype(np.int64).tolist() == want[k], k
        finally:
            bam.close()


@pytest.mark.parametrize('shift', [0, 1])
def test_empty_block_on_a_chunk_boundary_hands_the_known_start_on(shift):
    """htslib's layout, 64 blocks per chunk, and an EMPTY block (an interior EOF marker) as the first block of the second chunk;
    the record that begins the block behind it carries a name no heuristic would take for one (a control byte).  That block's
    start is known - offset 0, like the empty block's would have been - and must not be left to a guess that nobody in front
    of it can vouch for: a later offset would be accepted and the record dropped.  (Every second block around the boundary is
    empty and every block behind one is patched; `shift` moves the pattern by one block, so that in one of the two files an
    empty block is the chunk's first whether or not the header's block counts.)"""
    batch = _library(6000)
    with tempfile.TemporaryDirectory() as tmp:
        raw_path, path = os.path.join(tmp, 'raw.bam'), os.path.join(tmp, 'x.bam')
        bam_writer.write_bam(raw_path, batch, block_bytes=1500, align_records=True)
        data, at, payloads = open(raw_path, 'rb').read(), 0, []
        while at < len(data):
            size = struct.unpack_from('<H', data, at + 16)[0] + 1
            payloads.append(bytearray(zlib.decompress(data[at + 18:at + size - 8], -15)))
            at += size
        assert len(payloads) > 200 and len(payloads[-1]) == 0
        marked = range(58 + shift, 72 + shift)
        for k in marked:
            assert struct.unpack_from('<i', payloads[k], 0)[0] >= 32 and chr(payloads[k][36]) == 'r'
            payloads[k][36] = 1                              # the read name begins with a control byte
        with open(path, 'wb') as fh:
            for k, pl in enumerate(payloads):
                if k in marked:
                    fh.write(bam_writer._bgzf_block(b''))
                fh.write(bam_writer._bgzf_block(bytes(pl)))
        host = bamio.read_bam(path, threads=2)
        assert len(host) == len(batch)
        bam = bamio.ResidentBam(path, threads=2, mode='device', chunk_blocks=64)
        try:
            assert bam.ingest.on_device == 1 and len(bam) == len(batch)
            _check_against_host(path, bam, host)
        finally:
            bam.close()
