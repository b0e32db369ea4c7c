"""TEST INFRASTRUCTURE (not part of the product path): scalar model of csrc/bgzf_gpu.hip's DEFLATE decoder - the same
table construction (canonical counts / first codes / offsets, primary table filled by decoding every slot's bit pattern),
the same second-level table for literal / length codes of 11-13 bits, the same closed forms for RFC 1951's length / distance codes, the same path for codes longer than the tables -
checked against zlib (the published algorithm's reference implementation; BGZF blocks are raw DEFLATE streams, SAM spec
section 4.1).  The kernel mirrors this step by step, so a logic error shows up here, without a GPU
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
