"""TEST INFRASTRUCTURE (not part of the product path): scalar model of csrc/bgzf_gpu.hip's DEFLATE decoder - the same
table construction (canonical counts / first codes / offsets, primary table filled by decoding every slot's bit pattern),
the same second-level table for literal / length codes of 11-13 bits, the same closed forms for RFC 1951's length / distance codes, the same path for codes longer than the tables -
checked against zlib (the published algorithm's reference implementation; BGZF blocks are raw DEFLATE streams, SAM spec
section 4.1).  The kernel mirrors this step by step, so a logic error shows up here, without a GPU; `inflate_ranges` below restates the
SECOND form of the kernel - 64 lanes in 64 ranges of a block's bits, hand-overs, checkpoints - the same way
(tests/test_inflate_model.py); on the GPU the kernel itself is compared with zlib byte for byte (tests/test_gpu_ingest.py).

    python oracle/inflate_model.py
"""
import os
import random
import zlib

TAB = 10
CL_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


def bitrev(v, n):
    r = 0
    for _ in range(n):
        r = (r << 1) | (v & 1)
        v >>= 1
    return r


class Canon:
    """count / first code / offset per length + symbols sorted by (length, symbol) + primary table of 2^bits entries:
    entry = symbol << 4 | length, 0 where the code is longer than the table (or unused).  sub_bits > 0 (the literal / length
    code): the first sub_cap primary slots that are no short code, in slot order, point (LINK | index) to second-level tables
    of 2^sub_bits entries - the slot's pattern extended by sub_bits bits, decoded canonically - as build_code does in
    csrc/bgzf_gpu.hip; a code beyond bits + sub_bits (or a slot beyond the capacity) is left to slow()."""
    LINK = 1 << 20

    def __init__(self, lens, bits, sub_bits=0, sub_cap=64):
        self.bits = bits
        self.sub_bits = sub_bits
        self.cnt = [0] * 16
        for l in lens:
            if l:
                self.cnt[l] += 1
        self.first = [0] * 16
        self.offs = [0] * 16
        code = 0
        off = 0
        left = 1
        for L in range(1, 16):
            code = (code + self.cnt[L - 1]) << 1 if L > 1 else 0
            self.first[L] = code
            self.offs[L] = off
            off += self.cnt[L]
            left = (left << 1) - self.cnt[L]
            if left < 0:
                raise ValueError('oversubscribed')
        run = [0] * 16
        self.sorted = [0] * max(1, off)
        for s, l in enumerate(lens):
            if l:
                self.sorted[self.offs[l] + run[l]] = s
                run[l] += 1
        self.tab = [0] * (1 << bits)
        for slot in range(1 << bits):
            r = bitrev(slot, bits)
            for L in range(1, bits + 1):
                c = r >> (bits - L)
                d = c - self.first[L]
                if 0 <= d < self.cnt[L]:
                    self.tab[slot] = (self.sorted[self.offs[L] + d] << 4) | L
                    break
        self.sub = []
        if sub_bits:
            wide = bits + sub_bits
            n_sub = 0
            for slot in range(1 << bits):
                if self.tab[slot] != 0 or n_sub >= sub_cap:
                    continue
                self.tab[slot] = self.LINK | n_sub
                n_sub += 1
                for ext in range(1 << sub_bits):
                    r = bitrev(slot | (ext << bits), wide)
                    entry = 0
                    for L in range(bits + 1, wide + 1):
                        d = (r >> (wide - L)) - self.first[L]
                        if 0 <= d < self.cnt[L]:
                            entry = (self.sorted[self.offs[L] + d] << 4) | L
                            break
                    self.sub.append(entry)

    def slow(self, low15):
        r = bitrev(low15, 15)
        for L in range(self.bits + 1, 16):
            c = r >> (15 - L)
            d = c - self.first[L]
            if 0 <= d < self.cnt[L]:
                return self.sorted[self.offs[L] + d], L
        raise ValueError('bad code')


def inflate(data):
    bb = 0
    bc = 0
    ip = 0
    out = bytearray()

    def refill():
        nonlocal bb, bc, ip
        while bc <= 32:
            w = int.from_bytes(data[ip:ip + 4].ljust(4, b'\0'), 'little')
            bb |= w << bc
            bc += 32
            ip += 4

    def take(n):
        nonlocal bb, bc
        v = bb & ((1 << n) - 1)
        bb >>= n
        bc -= n
        return v

    def sym(c):
        nonlocal bb, bc
        e = c.tab[bb & ((1 << c.bits) - 1)]
        if e & Canon.LINK:                        # the second-level table: the next sub_bits bits pick the entry
            e = c.sub[((e & (Canon.LINK - 1)) << c.sub_bits) + ((bb >> c.bits) & ((1 << c.sub_bits) - 1))]
        l = e & 15
        s = e >> 4
        if l == 0:
            s, l = c.slow(bb & 0x7fff)
        bb >>= l
        bc -= l
        return s

    while True:
        refill()
        final = take(1)
        typ = take(2)
        if typ == 0:
            take(bc & 7)
            refill()
            ln = take(16)
            nl = take(16)
            assert ln == (~nl & 0xffff)
            pos = ip - bc // 8            # byte position of the next unread input byte
            out += data[pos:pos + ln]
            ip = pos + ln
            bb = 0
            bc = 0
        elif typ in (1, 2):
            if typ == 1:
                ll = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8
                dl = [5] * 30
            else:
                hlit = take(5) + 257
                hdist = take(5) + 1
                hclen = take(4) + 4
                cl = [0] * 19
                for i in range(hclen):
                    refill()
                    cl[CL_ORDER[i]] = take(3)
                cc = Canon(cl, 7)
                lens = []
                while len(lens) < hlit + hdist:
                    refill()
                    s = sym(cc)
                    if s < 16:
                        lens.append(s)
                    elif s == 16:
                        lens += [lens[-1]] * (3 + take(2))
                    elif s == 17:
                        lens += [0] * (3 + take(3))
                    else:
                        lens += [0] * (11 + take(7))
                assert len(lens) == hlit + hdist
                ll = lens[:hlit]
                dl = lens[hlit:]
            lc = Canon(ll, TAB, sub_bits=3, sub_cap=64)
            dc = Canon(dl, 9)
            while True:
                refill()
                s = sym(lc)
                if s < 256:
                    out.append(s)
                    continue
                if s == 256:
                    break
                s -= 257
                assert s < 29
                if s < 8:
                    base, ex = 3 + s, 0
                elif s == 28:
                    base, ex = 258, 0
                else:
                    ex = (s - 4) >> 2
                    base = 3 + ((4 + (s & 3)) << ex)
                length = base + take(ex)
                refill()
                d = sym(dc)
                assert d < 30
                if d < 4:
                    base, ex = 1 + d, 0
                else:
                    ex = (d - 2) >> 1
                    base = 1 + ((2 + (d & 1)) << ex)
                dist = base + take(ex)
                assert dist <= len(out)
                p = len(out)
                for i in range(length):
                    out.append(out[p - dist + (i % dist)])
        else:
            raise ValueError('bad block type')
        if final:
            return bytes(out)


# ---- the second form of the kernel (bgzf_inflate2_kernel): the symbols of a DEFLATE block decoded by 64 lanes side by side ----
# The same steps as the kernel, one lane after the other: the block's bits cut into 64 ranges; every lane decodes from a
# guessed start until it leaves its range and hands the bit where it did to the next lane; a lane whose start changed decodes
# again - until the first of three checkpoints of its range where it stands where the chain it decoded before stood (then
# the rest is what it was: only the counts in front of the checkpoint change); when no hand-over changes any more the lanes'
# counts place their symbols, a second decode lists them, and the bytes follow from the list.  What the model is for: the
# bookkeeping (counts across checkpoints, liveness of lanes behind the block's end, rounds until the hand-overs are stable)
# can be checked against zlib without a GPU.
GRID = (64, 192, 448)                                     # the checkpoints: bits behind a range's first (kGrid0 .. 2 of the kernel)
LIT, MATCH, END, BAD = 0, 1, 2, 3


def _len_base(s):
    if s < 8:
        return 3 + s, 0
    if s == 28:
        return 258, 0
    ex = (s - 4) >> 2
    return 3 + ((4 + (s & 3)) << ex), ex


def _dist_base(d):
    if d < 4:
        return 1 + d, 0
    ex = (d - 2) >> 1
    return 1 + ((2 + (d & 1)) << ex), ex


class _Bits:
    def __init__(self, data):
        self.data = bytes(data) + bytes(16)

    def at(self, p, n=48):                                   # n bits from bit p on
        i = p >> 3
        return (int.from_bytes(self.data[i:i + 8], 'little') >> (p & 7)) & ((1 << n) - 1)


def _symbol(bits, lc, dc, p):
    """One symbol at bit p -> (kind, bits it takes, bytes it makes, literal or distance) - decode_symbol of the kernel."""
    x = bits.at(p)

    def code(c, v):
        e = c.tab[v & ((1 << c.bits) - 1)]
        if e & Canon.LINK:
            e = c.sub[((e & (Canon.LINK - 1)) << c.sub_bits) + ((v >> c.bits) & ((1 << c.sub_bits) - 1))]
        if e & 15:
            return e >> 4, e & 15
        try:
            return c.slow(v & 0x7fff)
        except ValueError:
            return 0, 0

    sa, la = code(lc, x)
    if la == 0:
        return BAD, 0, 0, 0
    if sa < 256:
        return LIT, la, 1, sa
    if sa == 256:
        return END, la, 0, 0
    if sa >= 286:
        return BAD, 0, 0, 0
    base, xa = _len_base(sa - 257)
    length = base + ((x >> la) & ((1 << xa) - 1))
    xq = x >> (la + xa)
    sb, lb = code(dc, xq)
    if lb == 0 or sb >= 30:
        return BAD, 0, 0, 0
    dbase, xb = _dist_base(sb)
    return MATCH, la + xa + lb + xb, length, dbase + ((xq >> lb) & ((1 << xb) - 1))


def _ranges_block(bits, lc, dc, p0, end_bit, out, stats):
    """The symbols of one Huffman block that begin at bit p0: appended to `out` as bytes; -> the bit behind its end code."""
    rng = max(64, (end_bit - p0 + 63) >> 6)
    lo = [p0 + k * rng for k in range(64)]
    lim = [min(lo[k] + rng, end_bit) for k in range(64)]
    start = list(lo)
    live = [s < end_bit for s in start]
    done_for = [None] * 64
    r_end, r_kind, tot = [0] * 64, [BAD] * 64, [(0, 0)] * 64
    n_cp = len(GRID)
    grids = GRID
    cp = [[None] * n_cp for _ in range(64)]
    pre = [[(0, 0)] * n_cp for _ in range(64)]
    rounds = 0
    while True:
        rounds += 1
        assert rounds <= 130
        for k in range(64):
            if not (live[k] and start[k] != done_for[k]):
                continue
            p, nb, ns, kind, j, in_step = start[k], 0, 0, LIT, 0, False
            grid = lo[k] + grids[0]
            while p < lim[k]:
                kind, n, bytes_, _ = _symbol(bits, lc, dc, p)
                if kind >= END:
                    if kind == END:
                        p += n
                    break
                nb, ns, p = nb + bytes_, ns + 1, p + n
                stats['bits'] += n
                if p >= grid:
                    if p == cp[k][j]:
                        in_step = True
                        break
                    cp[k][j], pre[k][j] = p, (nb, ns)
                    j += 1
                    grid = lo[k] + grids[j] if j < n_cp else 1 << 62
            if in_step:                                      # the rest is what it was: the counts in front of the checkpoint changed
                db, ds = nb - pre[k][j][0], ns - pre[k][j][1]
                tot[k] = (tot[k][0] + db, tot[k][1] + ds)
                for i in range(j, n_cp):
                    pre[k][i] = (pre[k][i][0] + db, pre[k][i][1] + ds) if i > j else (nb, ns)
            else:
                if kind < END and p >= end_bit:
                    kind = BAD
                tot[k], r_end[k], r_kind[k] = (nb, ns), p, (kind if kind >= END else LIT)
                for i in range(j, n_cp):
                    cp[k][i] = None
            done_for[k] = start[k]
            if 'decodes' in stats:                           # (for a look at where the lanes fall into step: round, lane, bits, checkpoint)
                stats['decodes'].append((rounds, k, p - start[k], j if in_step else -1))
        changed = False                                      # the hand-over, every lane from the state BEFORE it
        new_live, new_start = list(live), list(start)
        for k in range(1, 64):
            now = live[k - 1] and r_kind[k - 1] == LIT and r_end[k - 1] < end_bit
            if now != live[k] or (now and r_end[k - 1] != start[k]):
                changed = True
            new_live[k] = now
            if now:
                new_start[k] = r_end[k - 1]
        live, start = new_live, new_start
        if not changed:
            break
    stats['rounds'] = max(stats['rounds'], rounds)
    stops = [k for k in range(64) if live[k] and r_kind[k] != LIT]
    if not stops or r_kind[stops[0]] != END:
        raise ValueError('no end-of-block code')
    last = stops[0]
    for k in range(last + 1):                                # the second decode: the lanes' symbols in the order of the stream
        assert live[k] and done_for[k] == start[k]
        p, nb, ns = start[k], 0, 0
        while p < r_end[k]:
            kind, n, bytes_, what = _symbol(bits, lc, dc, p)
            if kind >= END:
                break
            if 'symbols' in stats:                           # (for a look at the matches: place, bytes, distance - 0: a literal)
                stats['symbols'].append((len(out), bytes_, 0 if kind == LIT else what))
            if kind == LIT:
                out.append(what)
            else:
                assert what <= len(out)
                at = len(out)
                for i in range(bytes_):
                    out.append(out[at - what + (i % what)])
            nb, ns, p = nb + bytes_, ns + 1, p + n
        assert (nb, ns) == tot[k], (k, (nb, ns), tot[k])     # what the rounds counted is what the lane makes
    return r_end[last]


def inflate_ranges(data, stats=None):
    """A raw DEFLATE stream inflated the way bgzf_inflate2_kernel does; stats: {'rounds': most rounds of hand-overs a block
    took, 'bits': bits decoded in the rounds (the second decode not counted)}."""
    stats = stats if stats is not None else {}
    stats.setdefault('rounds', 0)
    stats.setdefault('bits', 0)
    bits = _Bits(data)
    end_bit = len(data) * 8
    out = bytearray()
    p = 0
    while True:
        final, typ = bits.at(p, 1), bits.at(p + 1, 2)
        p += 3
        if typ == 0:
            p = (p + 7) & ~7
            ln, nl = bits.at(p, 16), bits.at(p + 16, 16)
            assert ln == (~nl & 0xffff)
            at = (p >> 3) + 4
            out += data[at:at + ln]
            p = (at + ln) * 8
        elif typ in (1, 2):
            if typ == 1:
                ll = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8
                dl = [5] * 30
            else:
                hlit, hdist, hclen = bits.at(p, 5) + 257, bits.at(p + 5, 5) + 1, bits.at(p + 10, 4) + 4
                p += 14
                cl = [0] * 19
                for i in range(hclen):
                    cl[CL_ORDER[i]] = bits.at(p, 3)
                    p += 3
                cc = Canon(cl, 7)
                lens = []
                while len(lens) < hlit + hdist:
                    e = cc.tab[bits.at(p, 7)]
                    s, l = e >> 4, e & 15
                    assert l
                    p += l
                    if s < 16:
                        lens.append(s)
                    elif s == 16:
                        lens += [lens[-1]] * (3 + bits.at(p, 2))
                        p += 2
                    elif s == 17:
                        lens += [0] * (3 + bits.at(p, 3))
                        p += 3
                    else:
                        lens += [0] * (11 + bits.at(p, 7))
                        p += 7
                assert len(lens) == hlit + hdist
                ll, dl = lens[:hlit], lens[hlit:]
            p = _ranges_block(bits, Canon(ll, TAB, sub_bits=3, sub_cap=64), Canon(dl, 10), p, end_bit, out, stats)
        else:
            raise ValueError('bad block type')
        if final:
            return bytes(out)


def main_ranges():
    rnd = random.Random(6)
    words = [os.urandom(rnd.randint(1, 12)) for _ in range(300)]
    bam = b''.join(b'read%05d\0' % i + bytes([0x12, 0x48] * 20) + bytes(rnd.choice(b'FFFFF:,#') for _ in range(80)) for i in range(400))
    cases = [(b'', 6, 0), (b'a', 6, 0), (b'abc' * 3000, 6, 0), (bytes(40000), 1, 0), (bytes(40000), 9, 0), (bam, 1, 0), (bam, 6, 0),
             (bam, 6, zlib.Z_FIXED), (bam, 6, zlib.Z_HUFFMAN_ONLY), (b''.join(rnd.choice(words) for _ in range(3000)), 6, 0),
             (bytes(rnd.choice(b'ACGT') for _ in range(20000)), 6, zlib.Z_HUFFMAN_ONLY),        # codes of one length: no falling into step
             (bytes(int(rnd.expovariate(0.02)) & 255 for _ in range(20000)), 6, 0), (os.urandom(20000) + bam, 1, 0)]
    worst = 0
    for raw, level, strategy in cases:
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        comp = c.compress(raw) + c.flush()
        stats = {}
        assert inflate_ranges(comp, stats) == raw, (len(raw), level, strategy)
        assert stats['rounds'] <= 65                         # lane k's start is final once the k lanes in front of it are
        worst = max(worst, stats['rounds'])
    # many blocks in one stream: flush points every few hundred bytes (blocks shorter than 64 ranges of 64 bits)
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = b''.join(c.compress(bam[i:i + 700]) + c.flush(zlib.Z_SYNC_FLUSH if i % 2100 else zlib.Z_FULL_FLUSH) for i in range(0, len(bam), 700)) + c.flush()
    assert inflate_ranges(comp) == bam
    stats = {}
    c = zlib.compressobj(1, zlib.DEFLATED, -15)
    comp = c.compress(bam) + c.flush()
    inflate_ranges(comp, stats)
    print('inflate model, ranges: %d streams equal to zlib, at most %d rounds of hand-overs; a BAM-like stream of %d bits: %d bits '
          'decoded in %d rounds' % (len(cases) + 1, worst, len(comp) * 8, stats['bits'], stats['rounds']))


def main():
    rnd = random.Random(5)
    cases = [b'', b'a', b'abc' * 1000, bytes(60000), os.urandom(3000)]
    cases.append(bytes(rnd.choice(b'ACGT') for _ in range(50000)))
    words = [os.urandom(rnd.randint(1, 12)) for _ in range(300)]
    cases.append(b''.join(rnd.choice(words) for _ in range(9000))[:65000])
    cases.append(bytes(int(rnd.expovariate(0.02)) & 255 for _ in range(40000)))
    n = 0
    for raw in cases:
        for level in (0, 1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
                comp = c.compress(raw) + c.flush()
                assert inflate(comp) == raw, (len(raw), level, strategy)
                n += 1
    print('inflate model: %d streams equal to zlib' % n)


if __name__ == '__main__':
    main()
    main_ranges()
