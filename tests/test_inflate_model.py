"""The scalar model of csrc/bgzf_gpu.hip's DEFLATE decoder (oracle/inflate_model.py: the table construction, the arithmetic
form of the length / distance codes and the long-code path the kernel mirrors) against zlib - the part of the device
inflate that can be checked without a GPU."""


def test_model_equals_zlib(capsys):
    from oracle import inflate_model
    inflate_model.main()
    assert 'equal to zlib' in capsys.readouterr().out


def test_ranges_model_equals_zlib(capsys):
    """The second form of the kernel (bgzf_inflate2_kernel) restated lane by lane: ranges of the block's bits, hand-overs,
    checkpoints, counts - equal to zlib, the hand-overs stable within 65 rounds even for codes of one length, and what the
    rounds counted for a lane is what its second decode makes (asserted inside the model)."""
    from oracle import inflate_model
    inflate_model.main_ranges()
    assert 'equal to zlib' in capsys.readouterr().out


def test_models_on_streams_of_libdeflate():
    """Streams of another compressor - libdeflate, what htslib writes BGZF blocks with: its own block splitting and
    length-limited codes, near-optimal parsing at level 12 - through both models (the tables; the ranges and hand-overs)."""
    import os
    import random
    import zlib
    import pytest
    from oracle import inflate_model
    from tests import libdeflate_util as LD
    if not LD.available():
        pytest.skip('no libdeflate in this image')
    rnd = random.Random(8)
    bam = b''.join(b'read%05d\0' % i + bytes([0x12, 0x48] * 20) + bytes(rnd.choice(b'FFFFF:,#') for _ in range(80)) for i in range(300))
    cases = [bam, bytes(rnd.choice(b'ACGT') for _ in range(12000)), bytes(30000), b'abcde' * 4000, os.urandom(2000) + bam[:8000],
             bytes(int(rnd.expovariate(0.03)) & 255 for _ in range(12000))]
    n = 0
    for raw in cases:
        for level in (1, 6, 12):
            comp = LD.deflate(raw, level)
            assert zlib.decompress(comp, -15) == raw
            assert inflate_model.inflate(comp) == raw
            assert inflate_model.inflate_ranges(comp) == raw
            n += 1
    assert n == 18


def test_ranges_model_refuses_a_block_without_its_end_code():
    import zlib
    import pytest
    from oracle import inflate_model
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(b'abcdefgh' * 500 + bytes(range(256))) + c.flush()
    with pytest.raises((ValueError, AssertionError)):
        inflate_model.inflate_ranges(comp[:len(comp) // 2])


def test_length_and_distance_codes_in_closed_form():
    """RFC 1951's tables against the shifts the kernel computes them with."""
    lb = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
    le = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
    db = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
          8193, 12289, 16385, 24577]
    de = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]
    for s in range(29):
        if s < 8:
            b, e = 3 + s, 0
        elif s == 28:
            b, e = 258, 0
        else:
            e = (s - 4) >> 2
            b = 3 + ((4 + (s & 3)) << e)
        assert (b, e) == (lb[s], le[s])
    for s in range(30):
        if s < 4:
            b, e = 1 + s, 0
        else:
            e = (s - 2) >> 1
            b = 1 + ((2 + (s & 1)) << e)
        assert (b, e) == (db[s], de[s])
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    for i in range(19):
        got = 16 + i if i < 3 else 0 if i == 3 else 8 - ((i - 3) >> 1) if i & 1 else 8 + ((i - 4) >> 1)
        assert got == order[i]
