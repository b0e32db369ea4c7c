// BAM ingest on the GPU: BGZF inflate + record walk + record decode, the COMPRESSED file is what crosses PCIe.
// (the device form of bam_reader.hip's host path - `pysam.Samfile` iteration, runBESST:162,
// CreateGraph.py:111, libmetrics.py:63,257,293 - for files in htslib's block layout, where no record straddles a block.)
//
// A BGZF block is an independent DEFLATE stream of at most 64 KiB of output, and a BAM of C3's size holds 1.3 million of
// them: the parallelism is across blocks, so a block belongs to ONE WAVE; what is sequential about the stream - Huffman
// state, output position, control flow - is wave-uniform, what is data parallel uses the wave's 64 lanes:
//
//   bgzf_inflate_kernel   one single-wave workgroup per block.
//       input    256 compressed bytes per coalesced load, a dword per lane, handed to the 64-bit bit buffer with
//                v_readlane; the buffer itself lives in vector registers (the CU's one scalar ALU is this kernel's limit);
//       tables   canonical Huffman codes from the code lengths: symbols ranked by (length, symbol) with one ballot per
//                length and 64 symbols, then every lane fills the primary-table slots it owns by DECODING the slot's bit
//                pattern canonically (first code / count / offset per length) - balanced, no replication loops; codes
//                longer than the 10-bit primary table (rare) are decoded the same way on the spot; base and extra-bit
//                count of a length / distance code come from two per-lane registers (v_readlane);
//       window   the block's own output in HBM / L2 (default; 6 KB of LDS, seven waves per SIMD) - a wave's vector memory
//                instructions are processed in order, a load behind a store of the same wave returns the stored byte -
//                or a 32 KiB ring in LDS that leaves for HBM in 8 KiB granules (BESST_BGZF_WINDOW=lds; 38 KB of LDS, one
//                wave per SIMD: 2.3 x slower, kept for A/B runs);
//       output   in groups of 64 bytes: lane k of a group holds a literal or the place its byte is copied from; a full
//                group, a match that reads from the group, or the end of the block flushes it with one gather and one
//                store (an overlapping match - distance < length - is written directly: byte i is byte i mod dist of its
//                last dist bytes).
//   bgzf_crc_kernel       the CRC-32 of every block's inflated bytes against the block's gzip trailer (what htslib checks):
//                         a thread per slice, the slices' values combined as zlib's crc32_combine does
//   bam_walk_kernel       one lane per block: follows the records' length prefixes from the block's first byte
//                         (u16 offsets per record, count, and whether the walk ended exactly at the block's end)
//   bam_scan_kernel       exclusive scan of the blocks' record counts + the chunk's verdict (all blocks inflated, all walks
//                         closed: htslib's layout) in one workgroup
//   bam_decode_kernel     one workgroup per block, a thread per record: the 36 fixed bytes as ten aligned dwords, the
//                         CIGAR walk of pysam 0.8.4's qlen / alen (bam_reader.hip has the semantics), coalesced stores
//                         into the record columns at the block's place in the stream
// DESIGN.md section 8 has the measurements that led here.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace besst {

namespace {

constexpr int kRing = 32768;
constexpr int kTabBits = 10;
constexpr int kTabSize = 1 << kTabBits;
constexpr int kClBits = 7;
constexpr uint32_t kFlushGranule = 8192;
constexpr uint32_t kGroupLit = 0x80000000u;   // output group: the lane holds a literal (else a source position, < 2^31)
constexpr uint32_t kNoEntry = 0xFFF0u;     // table entry of a pattern that is no short code: length nibble 0, and not below 0x1000 (a literal)

// status of a block (0 = inflated)
enum : uint32_t {
    kInfOk = 0, kInfBadBlockType, kInfBadStored, kInfBadLengths, kInfOversubscribed, kInfBadCode, kInfBadDistance,
    kInfOutputOverrun, kInfInputOverrun, kInfSizeMismatch, kInfCrcMismatch
};

struct CanonLds {               // per code: count / first code / offset per length, symbols sorted by (length, symbol)
    uint16_t cnt[16], first[16], offs[16];
};

template <int kRingBytes>
struct InflateLds {
    __attribute__((aligned(16))) uint8_t ring[kRingBytes ? kRingBytes : 16];
    uint16_t lit_tab[kTabSize];
    uint16_t dist_tab[kTabSize];
    uint16_t cl_tab[1 << kClBits];
    uint16_t lit_sorted[288];
    uint16_t dist_sorted[32];
    uint16_t cl_sorted[32];
    CanonLds lit_c, dist_c, cl_c;
    uint8_t lens[288 + 32 + 16];
    uint8_t cl_lens[32];
};

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// Canonical code of `n` symbols with lengths lens[0..n) (0 = unused): per-length counts, first codes and offsets into
// `sorted` (symbols by length, then by value), then the primary table of 2^bits entries (symbol << 4 | length; kNoEntry:
// the code is longer than the table, or the pattern is not a code).  Returns false when the lengths oversubscribe the
// code space.  All lanes take part; everything returned in LDS.
__device__ __forceinline__ bool build_code(const uint8_t* lens, int n, int bits, uint16_t* tab, uint16_t* sorted, CanonLds* c, int lane) {
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t cnt[16];
#pragma unroll
    for (int L = 0; L < 16; ++L) cnt[L] = 0;
    for (int base = 0; base < n; base += 64) {
        const int s = base + lane;
        const uint32_t l = s < n ? lens[s] : 0u;
#pragma unroll
        for (int L = 1; L < 16; ++L) cnt[L] += (uint32_t)__popcll(__ballot(l == (uint32_t)L));
    }
    uint32_t first[16], offs[16];
    uint32_t code = 0, off = 0;
    int left = 1;
    bool ok = true;
    first[0] = 0; offs[0] = 0;
#pragma unroll
    for (int L = 1; L < 16; ++L) {
        code = L > 1 ? (code + cnt[L - 1]) << 1 : 0u;
        first[L] = code;
        offs[L] = off;
        off += cnt[L];
        left = (left << 1) - (int)cnt[L];
        if (left < 0) ok = false;
    }
    if (!ok) return false;                                   // uniform
    if (lane < 16) {
        uint32_t cv = 0, fv = 0, ov = 0;
#pragma unroll
        for (int L = 1; L < 16; ++L)
            if (lane == L) { cv = cnt[L]; fv = first[L]; ov = offs[L]; }
        c->cnt[lane] = (uint16_t)cv;
        c->first[lane] = (uint16_t)fv;
        c->offs[lane] = (uint16_t)ov;
    }
    uint32_t run[16];
#pragma unroll
    for (int L = 0; L < 16; ++L) run[L] = offs[L];
    for (int base = 0; base < n; base += 64) {
        const int s = base + lane;
        const uint32_t l = s < n ? lens[s] : 0u;
        uint32_t at = 0;
#pragma unroll
        for (int L = 1; L < 16; ++L) {
            const unsigned long long m = __ballot(l == (uint32_t)L);
            if (l == (uint32_t)L) at = run[L] + (uint32_t)__popcll(m & lt);
            run[L] += (uint32_t)__popcll(m);
        }
        if (l) sorted[at] = (uint16_t)s;
    }
    __builtin_amdgcn_wave_barrier();
    for (int slot = lane; slot < (1 << bits); slot += 64) {
        const uint32_t r = __brev((uint32_t)slot) >> (32 - bits);          // the slot's bits as an MSB-first code prefix
        uint32_t e = kNoEntry;
#pragma unroll
        for (int L = 1; L <= kTabBits; ++L) {
            if (L <= bits) {
                const uint32_t d = (r >> (bits - L)) - first[L];
                if (e == kNoEntry && d < cnt[L]) e = ((uint32_t)sorted[offs[L] + d] << 4) | (uint32_t)L;
            }
        }
        tab[slot] = (uint16_t)e;
    }
    __builtin_amdgcn_wave_barrier();
    return true;
}

// a code longer than the primary table: canonical decode of the next 15 bits (uniform); 0 = not a code
__device__ __forceinline__ uint32_t slow_code(const CanonLds* c, const uint16_t* sorted, uint32_t low15, int bits) {
    const uint32_t r = __brev(low15) >> 17;
    for (int L = bits + 1; L < 16; ++L) {
        const uint32_t d = (r >> (15 - L)) - (uint32_t)c->first[L];
        if (d < (uint32_t)c->cnt[L]) return ((uint32_t)sorted[(uint32_t)c->offs[L] + d] << 4) | (uint32_t)L;
    }
    return 0u;
}

// The bit buffer is wave-uniform but lives in VECTOR registers (an empty asm with a "+v" operand pins it there): a CU has
// ONE scalar ALU for all its waves, and with the whole symbol loop in scalar code that unit was the kernel's limit
// (4.9 G scalar against 1.1 G vector instructions per launch by the counters, 43 + 10 per symbol).  The buffer's shifts,
// masks and table indexes cost the vector units - idle otherwise - the same single issue; what steers control flow comes
// back to a scalar register with v_readfirstlane.
struct BitReader {
    const uint32_t* words;      // 4-byte aligned start of the block's payload (uniform)
    uint32_t in;                // the current window of 64 dwords of input: one per lane
    uint32_t widx;              // next dword of the current window (uniform)
    uint32_t wcount;            // dwords handed to the bit buffer so far, counted from `words` (uniform)
    uint64_t bb;                // bit buffer (the same in every lane)
    uint32_t bc;                // valid bits in it (uniform)
    int lane;

    __device__ __forceinline__ void pin() { asm volatile("" : "+v"(bb)); }
    __device__ __forceinline__ void seek(uint32_t byte_pos) {
        wcount = byte_pos >> 2;
        in = words[wcount + (uint32_t)lane];
        widx = 0;
        bb = 0;
        bc = 0;
        pin();
        refill();
        const uint32_t skip = (byte_pos & 3u) * 8u;
        bb >>= skip;
        bc -= skip;
        refill();
    }
    // at least 32 valid bits afterwards
    __device__ __forceinline__ void refill() {
        if (bc <= 32u) {                                     // uniform
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)in, (int)widx);
            bb |= (uint64_t)w << bc;
            pin();
            bc += 32u;
            ++wcount;
            ++widx;
            // (the next window is loaded when this one is used up, not ahead: a second register handed on at every
            // refill site made the compiler wait for ALL outstanding memory operations - the group's stores - at each
            // of them; the six other waves of the SIMD cover the load)
            if (widx == 64u) {                               // uniform
                in = words[wcount + (uint32_t)lane];
                widx = 0;
            }
        }
    }
    // the low bits of the buffer, not consumed (table indexes)
    __device__ __forceinline__ uint32_t peek(uint32_t mask) const { return (uint32_t)bb & mask; }
    __device__ __forceinline__ void drop(uint32_t n) {
        bb >>= n;
        bc -= n;
    }
    __device__ __forceinline__ uint32_t take(uint32_t n) {   // (the value back in a scalar register)
        const uint32_t v = uni((uint32_t)bb & ((1u << n) - 1u));
        bb >>= n;
        bc -= n;
        return v;
    }
    // bytes of input consumed so far (whole bytes: call on a byte boundary)
    __device__ __forceinline__ uint32_t byte_pos() const { return wcount * 4u - (bc >> 3); }
};

}  // namespace

// kRingBytes = kRing: the window is an LDS ring (38 KB of LDS: four waves per CU); 0: the window is the block's own output
// in HBM / L2 (6 KB of LDS: the registers allow six waves per SIMD).
template <int kRingBytes>
__global__ __launch_bounds__(64) void bgzf_inflate_kernel(const uint8_t* __restrict__ src, const BgzfBlock* __restrict__ blocks,
                                                          uint32_t n_blocks, uint8_t* dst, uint32_t* __restrict__ status) {
    __shared__ InflateLds<kRingBytes> s;
    constexpr bool kLds = kRingBytes != 0;
    constexpr uint32_t kRingMask = kLds ? (uint32_t)kRingBytes - 1u : 0u;
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) return;
    const uint32_t src_off = uni(blocks[b].src_off), src_len = uni(blocks[b].src_len);
    const uint32_t dst_len = uni(blocks[b].dst_len);
    uint8_t* out = dst + (size_t)uni(blocks[b].dst_off_lo) + ((size_t)uni(blocks[b].dst_off_hi) << 32);
    if (dst_len == 0) {                                      // the EOF marker block
        if (lane == 0) status[b] = kInfOk;
        return;
    }
    BitReader br;
    br.lane = lane;
    {
        // (src is 4-byte aligned; no detour through an integer, so that the window loads are global loads - a flat load
        // also counts as an LDS operation, and every table look-up behind one would wait for it)
        br.words = reinterpret_cast<const uint32_t*>(src + (src_off & ~3u));
        br.seek(src_off & 3u);
    }
    const uint32_t in_base = src_off & 3u;
    // RFC 1951's length and distance codes in closed form, one code per lane: base | extra bits << 9 (<< 16)
    uint32_t len_info, dist_info;
    {
        const uint32_t k = (uint32_t)lane;
        uint32_t base, extra = 0;
        if (k < 8u) base = 3u + k;
        else if (k == 28u) base = 258u;
        else {
            extra = (k - 4u) >> 2;
            base = 3u + ((4u + (k & 3u)) << extra);
        }
        len_info = k < 29u ? base | (extra << 9) : 0u;
        extra = 0;
        if (k < 4u) base = 1u + k;
        else {
            extra = (k - 2u) >> 1;
            base = 1u + ((2u + (k & 1u)) << extra);
        }
        dist_info = k < 30u ? base | (extra << 16) : 0u;
    }
    uint32_t pos = 0, flushed = 0;
    uint32_t err = kInfOk;
    // ---- the window.  LDS form: bytes go to the ring and leave for HBM in granules.  Global form: bytes go straight to the
    // block's output and a match reads its source there: a wave's vector memory instructions are processed in order, a
    // load behind a store of the same wave to the same address returns the stored byte (the loads skip the CU's L1).
    auto put_byte = [&](uint32_t at, uint32_t v) {
        if constexpr (kLds) s.ring[at & kRingMask] = (uint8_t)v;
        else out[at] = (uint8_t)v;
    };
    auto get_byte = [&](uint32_t at) -> uint32_t {
        if constexpr (kLds) return s.ring[at & kRingMask];
        else return __hip_atomic_load(out + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto flush_granules = [&]() {
        if constexpr (!kLds) return;
        while (pos - flushed >= kFlushGranule) {             // uniform
#pragma unroll
            for (int k = 0; k < (int)(kFlushGranule / 1024u); ++k) {
                const uint32_t o = flushed + (uint32_t)k * 1024u + (uint32_t)lane * 16u;
                const uint4 v = *reinterpret_cast<const uint4*>(&s.ring[o & kRingMask]);
                *reinterpret_cast<uint4*>(out + o) = v;
            }
            flushed += kFlushGranule;
        }
    };
    for (;;) {
        // (a DEFLATE block may be empty: without this test a payload of nothing but empty blocks - corrupt, but every bit of
        // it valid - would be followed out of the chunk's buffer)
        if (br.byte_pos() - in_base > src_len + 8u) { err = kInfInputOverrun; break; }
        br.refill();
        const uint32_t final_block = br.take(1);
        const uint32_t type = br.take(2);
        if (type == 0u) {
            // ---- stored: LEN bytes straight from the input
            br.drop(br.bc & 7u);
            br.refill();
            const uint32_t len = br.take(16), nlen = br.take(16);
            if (len != (~nlen & 0xffffu)) { err = kInfBadStored; break; }
            const uint32_t at = br.byte_pos();               // relative to br.words
            if (at - in_base + len > src_len) { err = kInfInputOverrun; break; }
            if (pos + len > dst_len) { err = kInfOutputOverrun; break; }
            const uint8_t* from = reinterpret_cast<const uint8_t*>(br.words) + at;
            for (uint32_t i0 = 0; i0 < len; i0 += 2048u) {   // uniform; a granule's worth at a time
                const uint32_t part = len - i0 < 2048u ? len - i0 : 2048u;
                for (uint32_t i = (uint32_t)lane; i < part; i += 64u) put_byte(pos + i, from[i0 + i]);
                pos += part;
                __builtin_amdgcn_wave_barrier();
                flush_granules();
            }
            br.seek(at + len);
        } else if (type == 1u || type == 2u) {
            int n_lit = 288, n_dist = 30;
            if (type == 1u) {
                for (int i = lane; i < 288; i += 64) s.lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
                if (lane < 32) s.lens[288 + lane] = 5;
            } else {
                const uint32_t hlit = br.take(5) + 257u, hdist = br.take(5) + 1u, hclen = br.take(4) + 4u;
                if (hlit > 286u || hdist > 30u) { err = kInfBadLengths; break; }
                if (lane < 32) s.cl_lens[lane] = 0;
                __builtin_amdgcn_wave_barrier();
                for (uint32_t i = 0; i < hclen; ++i) {       // uniform
                    br.refill();
                    const uint32_t v = br.take(3);
                    // 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
                    const uint32_t order = i < 3u ? 16u + i : i == 3u ? 0u : (i & 1u) ? 8u - ((i - 3u) >> 1) : 8u + ((i - 4u) >> 1);
                    s.cl_lens[order] = (uint8_t)v;
                }
                __builtin_amdgcn_wave_barrier();
                if (!build_code(s.cl_lens, 19, kClBits, s.cl_tab, s.cl_sorted, &s.cl_c, lane)) { err = kInfOversubscribed; break; }
                const uint32_t total = hlit + hdist;
                uint32_t have = 0, prev = 0;
                bool bad = false;
                while (have < total) {                        // uniform
                    br.refill();
                    const uint32_t e = uni(s.cl_tab[br.peek((1u << kClBits) - 1u)]);
                    const uint32_t l = e & 15u, sym = e >> 4;
                    if (l == 0u) { bad = true; break; }
                    br.drop(l);
                    if (sym < 16u) {
                        s.lens[have++] = (uint8_t)sym;
                        prev = sym;
                        continue;
                    }
                    uint32_t rep, val = 0;
                    if (sym == 16u) {
                        if (have == 0u) { bad = true; break; }
                        rep = 3u + br.take(2);
                        val = prev;
                    } else if (sym == 17u) {
                        rep = 3u + br.take(3);
                    } else {
                        rep = 11u + br.take(7);
                    }
                    if (have + rep > total) { bad = true; break; }
                    for (uint32_t i = (uint32_t)lane; i < rep; i += 64u) s.lens[have + i] = (uint8_t)val;
                    have += rep;
                    prev = val;
                }
                if (bad) { err = kInfBadLengths; break; }
                __builtin_amdgcn_wave_barrier();
                // the distance lengths follow the literal / length ones: move them to their own place
                if (lane < 32) {
                    const uint8_t v = (uint32_t)lane < hdist ? s.lens[hlit + (uint32_t)lane] : (uint8_t)0;
                    __builtin_amdgcn_wave_barrier();
                    s.lens[288 + lane] = v;
                }
                __builtin_amdgcn_wave_barrier();
                for (uint32_t i = hlit + (uint32_t)lane; i < 288u; i += 64u) s.lens[i] = 0;
                n_lit = (int)hlit;
                n_dist = (int)hdist;
                if (uni(s.lens[256]) == 0u) { err = kInfBadLengths; break; }     // no end-of-block code
            }
            __builtin_amdgcn_wave_barrier();
            if (!build_code(s.lens, n_lit, kTabBits, s.lit_tab, s.lit_sorted, &s.lit_c, lane)) { err = kInfOversubscribed; break; }
            if (!build_code(s.lens + 288, n_dist, kTabBits, s.dist_tab, s.dist_sorted, &s.dist_c, lane)) { err = kInfOversubscribed; break; }
            // ---- the symbols.  Output leaves in GROUPS of up to 64 bytes: lane k of the group stands for the byte at
            // pos + k and holds either a literal (flagged by the top bit) or the place its byte is copied from; a literal
            // is a compare and a select, a match two compares and a select per piece.  A full group -
            // or a match whose source reaches into the group, or the end of the block - flushes it: ONE gather of the
            // lanes that copy, ONE contiguous store.  The memory round trip of a match (~1 us with the window in HBM / L2,
            // and a wave has nothing else to do meanwhile) is paid once per 64 bytes of output instead of once per match
            // (a sequencer's file: 5241 matches of 9.7 bytes and 8865 literals in a block of 60 KB).
            uint32_t g = 0;                                  // lane k: kGroupLit | its literal, or where its byte is copied from
            uint32_t filled = 0;                             // uniform: lanes of the group in use
            auto flush_group = [&]() {
                if (filled != 0u) {
                    if ((uint32_t)lane < filled) {
                        uint32_t v = g;
                        if (!(g & kGroupLit)) v = get_byte(g);
                        put_byte(pos + (uint32_t)lane, v);
                    }
                    pos = uni(pos + filled);                 // (kept scalar by force: without it the compiler turns this
                    filled = 0;                              // branch into selects and the whole symbol loop into vector code)
                    if (kLds && pos - flushed >= kFlushGranule) flush_granules();
                }
            };
            for (;;) {
                // (at least once per 64 literals: a corrupt stream is not followed more than a few hundred bytes past its
                // payload - the chunk's buffer has 4 KB behind its last block)
                if (br.byte_pos() - in_base > src_len + 8u) { err = kInfInputOverrun; break; }
                br.refill();
                uint32_t e = uni(s.lit_tab[br.peek((uint32_t)(kTabSize - 1))]);
                // literals whose code fits the table (other entries are >= 0x1000), until the group is full: ONE way out
                // of this loop - a second exit costs every iteration the flag registers the compiler threads through it
                while (((e >> 12) | (filled >> 6)) == 0u) {    // e < 0x1000 (a literal) and filled < 64, as one test
                    br.drop(e & 15u);
                    g = (uint32_t)lane == filled ? kGroupLit | (e >> 4) : g;
                    ++filled;
                    br.refill();
                    e = uni(s.lit_tab[br.peek((uint32_t)(kTabSize - 1))]);
                }
                if (filled == 64u) {
                    if (pos + 64u > dst_len) { err = kInfOutputOverrun; break; }
                    flush_group();
                    if (e < 0x1000u) continue;               // (the entry is looked up again: nothing of it was consumed)
                }
                if ((e & 15u) == 0u) {
                    e = uni(slow_code(&s.lit_c, s.lit_sorted, uni(br.peek(0x7fffu)), kTabBits));
                    if (e == 0u) { err = kInfBadCode; break; }
                }
                br.drop(e & 15u);
                uint32_t sym = e >> 4;
                if (sym < 256u) {                            // a literal with a long code
                    g = (uint32_t)lane == filled ? kGroupLit | sym : g;
                    if (++filled == 64u) {
                        if (pos + 64u > dst_len) { err = kInfOutputOverrun; break; }
                        flush_group();
                    }
                    continue;
                }
                if (sym == 256u) {
                    if (pos + filled > dst_len) err = kInfOutputOverrun;
                    else flush_group();
                    break;
                }
                // base and extra-bit count of the length / distance code: lane k of two registers holds them for code k
                // (filled in once per wave), a v_readlane fetches them - the arithmetic on the symbol and its three
                // branches were a fifth of a match's scalar instructions
                sym -= 257u;
                if (sym >= 29u) { err = kInfBadCode; break; }
                const uint32_t li = (uint32_t)__builtin_amdgcn_readlane((int)len_info, (int)sym);
                const uint32_t length = (li & 0x1ffu) + br.take(li >> 9);
                br.refill();
                uint32_t d = uni(s.dist_tab[br.peek((uint32_t)(kTabSize - 1))]);
                if ((d & 15u) == 0u) {
                    d = uni(slow_code(&s.dist_c, s.dist_sorted, uni(br.peek(0x7fffu)), kTabBits));
                    if (d == 0u) { err = kInfBadCode; break; }
                }
                br.drop(d & 15u);
                const uint32_t dsym = d >> 4;
                if (dsym >= 30u) { err = kInfBadCode; break; }
                const uint32_t di = (uint32_t)__builtin_amdgcn_readlane((int)dist_info, (int)dsym);
                const uint32_t dist = (di & 0xffffu) + br.take(di >> 16);
                const uint32_t at = pos + filled;            // where the match begins
                if (dist > at) { err = kInfBadDistance; break; }
                if (at + length > dst_len) { err = kInfOutputOverrun; break; }
                if (dist >= length) {
                    // every byte's source is finished output - once the group is out of the way where the source reaches
                    // into it.  The match joins the group piece by piece: the byte at position P comes from P - dist.
                    if (at - dist + length > pos) flush_group();
                    uint32_t left = length;
                    while (left != 0u) {                     // uniform
                        const uint32_t room = 64u - filled;
                        const uint32_t take = left < room ? left : room;
                        const bool in = (uint32_t)lane >= filled && (uint32_t)lane < filled + take;
                        g = in ? pos + (uint32_t)lane - dist : g;
                        filled = uni(filled + take);
                        left -= take;
                        if (filled == 64u) flush_group();
                    }
                } else {
                    // an overlapping match repeats its last `dist` bytes: byte i is byte i mod dist of them
                    flush_group();
                    if (dist >= 64u) {
                        for (uint32_t i = (uint32_t)lane; i < length; i += 64u)     // (rounds complete in order)
                            put_byte(pos + i, get_byte(pos + i - dist));
                    } else {
                        const float rcp = __frcp_rn((float)dist);
                        for (uint32_t i = (uint32_t)lane; i < length; i += 64u) {
                            int q = (int)((float)i * rcp);
                            int r = (int)i - q * (int)dist;
                            if (r < 0) r += (int)dist;
                            else if (r >= (int)dist) r -= (int)dist;
                            put_byte(pos + i, get_byte(pos - dist + (uint32_t)r));
                        }
                    }
                    pos += length;
                    if (kLds && pos - flushed >= kFlushGranule) flush_granules();
                }
            }
            if (err) break;
        } else {
            err = kInfBadBlockType;
            break;
        }
        if (final_block) break;
    }
    if (!err) {
        if (pos != dst_len) err = kInfSizeMismatch;
        else if (br.byte_pos() - in_base > src_len + 8u) err = kInfInputOverrun;   // (the bit buffer reads ahead of its use)
    }
    if (!err && kLds) {
        // what is left in the ring: whole 16-byte units, then bytes
        __builtin_amdgcn_wave_barrier();
        const uint32_t rest = pos - flushed;
        const uint32_t n16 = rest >> 4;
        for (uint32_t j = (uint32_t)lane; j < n16; j += 64u) {
            const uint32_t o = flushed + j * 16u;
            *reinterpret_cast<uint4*>(out + o) = *reinterpret_cast<const uint4*>(&s.ring[o & kRingMask]);
        }
        const uint32_t tail0 = flushed + n16 * 16u;
        if (tail0 + (uint32_t)lane < pos) out[tail0 + (uint32_t)lane] = s.ring[(tail0 + (uint32_t)lane) & kRingMask];
    }
    if (lane == 0) status[b] = err;
}

// ---- the blocks' CRC32 ---------------------------------------------------------------------------------------------------
// BGZF stores the CRC-32 of every block's inflated bytes (the gzip trailer); htslib - what the reference reads its files
// through - checks it, and so does this path: a damaged payload that still decodes, or a byte the window logic got wrong,
// is a refused block, not a wrong record.  One workgroup per block, a thread per slice of <= 256 bytes: byte-table CRC
// of the slice, then the slice's CRC is carried over the bytes behind it - multiplication by x^(8n) modulo the CRC polynomial, zlib's
// crc32_combine - and the 256 values are XORed.
namespace {

constexpr uint32_t kCrcPoly = 0xedb88320u;
__device__ const uint32_t kCrcX2n[32] = {           // x^(2^k) mod the polynomial, reflected (zlib's x2n_table)
    0x40000000u, 0x20000000u, 0x08000000u, 0x00800000u, 0x00008000u, 0xedb88320u, 0xb1e6b092u, 0xa06a2517u,
    0xed627daeu, 0x88d14467u, 0xd7bbfe6au, 0xec447f11u, 0x8e7ea170u, 0x6427800eu, 0x4d47bae0u, 0x09fe548fu,
    0x83852d0fu, 0x30362f1au, 0x7b5a9cc3u, 0x31fec169u, 0x9fec022au, 0x6c8dedc4u, 0x15d6874du, 0x5fde7a4eu,
    0xbad90e37u, 0x2e4e5eefu, 0x4eaba214u, 0xa8a472c0u, 0x429a969eu, 0x148d302au, 0xc40ba6d0u, 0xc4e22c3cu};

// a(x) * b(x) modulo the polynomial (both reflected: bit 31 is x^0); `a` is the same in every lane
__device__ __forceinline__ uint32_t crc_mul(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (uint32_t m = 1u << 31; m; m >>= 1) {                // uniform
        if (a & m) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return p;
}

}  // namespace

__global__ __launch_bounds__(256) void bgzf_crc_kernel(const uint8_t* __restrict__ inflated, const BgzfBlock* __restrict__ blocks,
                                                       uint32_t n_blocks, uint32_t* __restrict__ status) {
    __shared__ uint32_t s_tab[256], s_part[4];
    const uint32_t b = blockIdx.x, t = threadIdx.x;
    if (b >= n_blocks) return;
    const uint32_t len = blocks[b].dst_len;
    if (len == 0u || status[b] != kInfOk) return;            // uniform
    {   // the byte table: entry t is t carried through eight steps of the polynomial division
        uint32_t c = t;
#pragma unroll
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ kCrcPoly : c >> 1;
        s_tab[t] = c;
    }
    __syncthreads();
    const uint8_t* base = inflated + (size_t)blocks[b].dst_off_lo + ((size_t)blocks[b].dst_off_hi << 32);
    // slice t = [t S, (t + 1) S) cut at len; S a multiple of 16, so that every slice is read as aligned 16-byte words
    // (the next word is requested before the current one's 32 table look-ups - a byte at a time the loads alone, each
    // waited for, took longer than the inflate)
    const uint32_t S = (((len + 255u) >> 8) + 15u) & ~15u;
    const uint32_t lo = t * S < len ? t * S : len;
    const uint32_t hi = lo + S < len ? lo + S : len;
    const uint32_t n_slice = hi - lo;
    const uint4* src = reinterpret_cast<const uint4*>(base + lo);
    uint32_t crc = 0xffffffffu;
    auto eat = [&](uint32_t w, uint32_t count) {             // the low `count` bytes of a word
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            if (k < count) {
                crc = s_tab[(crc ^ (w >> (8u * k))) & 0xffu] ^ (crc >> 8);
            }
        }
    };
    uint4 cur = n_slice ? src[0] : make_uint4(0, 0, 0, 0);
    for (uint32_t q = 0; q * 16u < n_slice; ++q) {
        const uint4 next = (q + 1u) * 16u < n_slice ? src[q + 1u] : make_uint4(0, 0, 0, 0);
        const uint32_t rem = n_slice - q * 16u;
        if (rem >= 16u) {
            eat(cur.x, 4); eat(cur.y, 4); eat(cur.z, 4); eat(cur.w, 4);
        } else {
            eat(cur.x, rem); eat(cur.y, rem > 4u ? rem - 4u : 0u); eat(cur.z, rem > 8u ? rem - 8u : 0u);
            eat(cur.w, rem > 12u ? rem - 12u : 0u);
        }
        cur = next;
    }
    crc = n_slice ? ~crc : 0u;
    // carry it over the bytes behind the slice: multiply by x^(8 n)
    uint32_t n = len - hi;
    uint32_t p = 1u << 31;                                   // x^0
    for (uint32_t k = 3; n; n >>= 1, ++k)
        if (n & 1u) p = crc_mul(kCrcX2n[k & 31u], p);
    // (p differs per lane: crc_mul's first argument must be uniform, so the product is formed bit by bit of `crc`... the
    // multiplication is commutative: run it with the per-lane operands swapped into its lane-wise form)
    uint32_t prod = 0;
    {
        uint32_t a = crc, bb = p;
        for (int j = 0; j < 32; ++j) {
            if (a & (1u << 31)) prod ^= bb;
            a <<= 1;
            bb = (bb & 1u) ? (bb >> 1) ^ kCrcPoly : bb >> 1;
        }
    }
    uint32_t x = prod;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x ^= (uint32_t)__shfl_xor((int)x, d, 64);
    if ((t & 63u) == 0u) s_part[t >> 6] = x;
    __syncthreads();
    if (t == 0u) {
        const uint32_t got = s_part[0] ^ s_part[1] ^ s_part[2] ^ s_part[3];
        if (got != blocks[b].crc) status[b] = kInfCrcMismatch;
    }
}

namespace {

__device__ __forceinline__ uint32_t ld32u(const uint8_t* p) {     // little-endian dword at any alignment
    const uintptr_t a = (uintptr_t)p;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(a & 3u) * 8u;
    const uint32_t lo = w[0];
    if (sh == 0u) return lo;
    return (lo >> sh) | (w[1] << (32u - sh));
}

}  // namespace

// one lane per block: the chain of length prefixes from `first_off` (block 0 of the file's record stream) or 0
__global__ __launch_bounds__(64) void bam_walk_kernel(const uint8_t* __restrict__ inflated, const BgzfBlock* __restrict__ blocks,
                                                      uint32_t n_blocks, uint32_t first_off, const uint32_t* __restrict__ status,
                                                      uint16_t* __restrict__ offs, uint32_t* __restrict__ count,
                                                      uint32_t* __restrict__ closed) {
    const uint32_t b = blockIdx.x * 64u + threadIdx.x;
    if (b >= n_blocks) return;
    const uint32_t len = blocks[b].dst_len;
    const uint8_t* base = inflated + (size_t)blocks[b].dst_off_lo + ((size_t)blocks[b].dst_off_hi << 32);
    uint32_t cur = b == 0u ? first_off : 0u, cnt = 0;
    if (status[b] != kInfOk || cur > len) {
        count[b] = 0;
        closed[b] = 0;
        return;
    }
    uint16_t* o = offs + (size_t)b * kBamBlockRecs;
    while (len - cur >= 4u && cnt < (uint32_t)kBamBlockRecs) {
        const uint32_t block_size = ld32u(base + cur);
        if (block_size < 32u || len - cur - 4u < block_size) break;
        o[cnt++] = (uint16_t)cur;
        cur += 4u + block_size;
    }
    count[b] = cnt;
    closed[b] = cur == len ? 1u : 0u;
}

// exclusive scan of the blocks' record counts; summary[0] = records of the chunk, [1] = 1 when every block inflated and
// every walk ended at its block's end, [2] = first block that did not, [3] = that block's inflate status
__global__ __launch_bounds__(1024) void bam_scan_kernel(const uint32_t* __restrict__ count, const uint32_t* __restrict__ closed,
                                                        const uint32_t* __restrict__ status, uint32_t n_blocks,
                                                        uint32_t* __restrict__ rec_base, uint32_t* __restrict__ summary) {
    __shared__ uint32_t s_w[16], s_bad[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t per = (n_blocks + 1023u) / 1024u;
    const uint32_t b0 = (uint32_t)t * per, b1 = b0 + per < n_blocks ? b0 + per : n_blocks;
    uint32_t sum = 0, bad = 0xffffffffu;
    for (uint32_t b = b0; b < b1; ++b) {
        sum += count[b];
        if ((status[b] != kInfOk || !closed[b]) && bad == 0xffffffffu) bad = b;
    }
    uint32_t x = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)x, d, 64);
        if (lane >= d) x += v;
    }
    uint32_t mb = bad;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint32_t v = (uint32_t)__shfl_xor((int)mb, d, 64);
        mb = v < mb ? v : mb;
    }
    if (lane == 63) s_w[wave] = x;
    if (lane == 0) s_bad[wave] = mb;
    __syncthreads();
    uint32_t off = x - sum, total = 0, first_bad = 0xffffffffu;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        if (q < wave) off += s_w[q];
        total += s_w[q];
        first_bad = s_bad[q] < first_bad ? s_bad[q] : first_bad;
    }
    for (uint32_t b = b0; b < b1; ++b) {
        rec_base[b] = off;
        off += count[b];
    }
    if (t == 0) {
        summary[0] = total;
        summary[1] = first_bad == 0xffffffffu ? 1u : 0u;
        summary[2] = first_bad;
        summary[3] = first_bad == 0xffffffffu ? 0u : status[first_bad];
    }
}

// one workgroup per block, a thread per record
__global__ __launch_bounds__(256) void bam_decode_kernel(const uint8_t* __restrict__ inflated, const BgzfBlock* __restrict__ blocks,
                                                         const uint16_t* __restrict__ offs, const uint32_t* __restrict__ count,
                                                         const uint32_t* __restrict__ rec_base, BamColumns col, int64_t out_base,
                                                         int64_t rel_base, int64_t head_records, uint32_t* __restrict__ flags) {
    const uint32_t b = blockIdx.x;
    const uint32_t cnt = count[b];
    const uint8_t* base = inflated + (size_t)blocks[b].dst_off_lo + ((size_t)blocks[b].dst_off_hi << 32);
    const uint16_t* o = offs + (size_t)b * kBamBlockRecs;
    const int64_t first = out_base + (int64_t)rec_base[b];
    for (uint32_t i = threadIdx.x; i < cnt; i += 256u) {
        const uint8_t* rec = base + o[i];
        const uintptr_t a = (uintptr_t)rec;
        const uint32_t* w = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
        const uint32_t sh = (uint32_t)(a & 3u) * 8u;
        uint32_t x[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) x[k] = w[k];
        uint32_t f[9];                                       // the record's first nine dwords: block_size, then 32 fixed bytes
#pragma unroll
        for (int k = 0; k < 9; ++k) f[k] = sh ? (x[k] >> sh) | (x[k + 1] << (32u - sh)) : x[k];
        const uint32_t block_size = f[0];
        const uint32_t l_read_name = f[3] & 0xffu, mapq = (f[3] >> 8) & 0xffu;
        const uint32_t n_cigar = f[4] & 0xffffu, flag = f[4] >> 16;
        const uint32_t l_seq = f[5];
        const int64_t r = first + (int64_t)i;
        col.tid[r] = (int32_t)f[1];
        col.pos[r] = (int32_t)f[2];
        col.mapq[r] = (uint8_t)mapq;
        col.flag[r] = (uint16_t)flag;
        col.mtid[r] = (int32_t)f[6];
        col.mpos[r] = (int32_t)f[7];
        col.tlen[r] = (int32_t)f[8];
        long long q_total = 0, ref_len = 0, lead = 0, trail = 0;
        if (32ull + l_read_name + 4ull * n_cigar > (unsigned long long)block_size) {
            atomicOr(&flags[0], 1u);                         // corrupt record
        } else {
            // pysam 0.8.4's qlen / alen (bam_reader.hip's decode has the reasoning)
            const uint8_t* cg = rec + 36 + l_read_name;
            bool in_lead = true;
            for (uint32_t c = 0; c < n_cigar; ++c) {
                const uint32_t v = ld32u(cg + 4u * c);
                const uint32_t op = v & 15u, len = v >> 4;
                if (op == 0u || op == 1u || op == 4u || op == 7u || op == 8u) q_total += len;
                if (op == 0u || op == 2u || op == 3u || op == 7u || op == 8u) ref_len += len;
                if (in_lead) {
                    if (op == 4u) lead += len;
                    else if (op != 5u) in_lead = false;
                }
            }
            for (uint32_t c = n_cigar; c-- > 1u;) {
                const uint32_t v = ld32u(cg + 4u * c);
                const uint32_t op = v & 15u, len = v >> 4;
                if (op == 4u) trail += len;
                else if (op != 5u) break;
            }
        }
        long long q_aln = (l_seq ? (long long)l_seq : q_total) - lead - trail;
        if (q_aln < 0) q_aln = 0;
        if (q_aln > 65535) {
            q_aln = 65535;
            atomicAdd(&flags[1], 1u);                        // saturated qlen
        }
        col.qlen[r] = (uint16_t)q_aln;
        const int64_t rel = rel_base + (int64_t)rec_base[b] + (int64_t)i;     // index among the records of this call
        if (rel < head_records) {
            col.head_rlen[rel] = (int32_t)l_seq;
            col.head_alen[rel] = (int32_t)ref_len;
            col.head_qlen[rel] = (uint16_t)q_aln;
        }
    }
}

int launch_bgzf_inflate(hipStream_t s, const uint8_t* src, const BgzfBlock* blocks, uint32_t n_blocks, uint8_t* dst,
                        uint32_t* status) {
    if (n_blocks == 0) return BESST_OK;
    // BESST_BGZF_WINDOW=lds: the LDS-ring form (A/B runs)
    static const bool lds_ring = [] { const char* e = getenv("BESST_BGZF_WINDOW"); return e && strcmp(e, "lds") == 0; }();
    if (lds_ring) hipLaunchKernelGGL((bgzf_inflate_kernel<kRing>), dim3(n_blocks), dim3(64), 0, s, src, blocks, n_blocks, dst, status);
    else hipLaunchKernelGGL((bgzf_inflate_kernel<0>), dim3(n_blocks), dim3(64), 0, s, src, blocks, n_blocks, dst, status);
    // BESST_BGZF_CRC=0: skip the check (timing runs)
    static const bool check_crc = [] { const char* e = getenv("BESST_BGZF_CRC"); return !(e && atoi(e) == 0); }();
    if (check_crc) hipLaunchKernelGGL(bgzf_crc_kernel, dim3(n_blocks), dim3(256), 0, s, dst, blocks, n_blocks, status);
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

int launch_bam_walk_scan(hipStream_t s, const uint8_t* inflated, const BgzfBlock* blocks, uint32_t n_blocks, uint32_t first_off,
                         const uint32_t* status, uint16_t* offs, uint32_t* count, uint32_t* closed, uint32_t* rec_base,
                         uint32_t* summary) {
    if (n_blocks == 0) return BESST_OK;
    hipLaunchKernelGGL(bam_walk_kernel, dim3((n_blocks + 63u) / 64u), dim3(64), 0, s, inflated, blocks, n_blocks, first_off,
                       status, offs, count, closed);
    hipLaunchKernelGGL(bam_scan_kernel, dim3(1), dim3(1024), 0, s, count, closed, status, n_blocks, rec_base, summary);
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

int launch_bam_decode(hipStream_t s, const uint8_t* inflated, const BgzfBlock* blocks, uint32_t n_blocks, const uint16_t* offs,
                      const uint32_t* count, const uint32_t* rec_base, const BamColumns& col, int64_t out_base,
                      int64_t rel_base, int64_t head_records, uint32_t* flags) {
    if (n_blocks == 0) return BESST_OK;
    hipLaunchKernelGGL(bam_decode_kernel, dim3(n_blocks), dim3(256), 0, s, inflated, blocks, offs, count, rec_base, col, out_base,
                       rel_base, head_records, flags);
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

}  // namespace besst
