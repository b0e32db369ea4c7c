// BAM ingest on the GPU: BGZF inflate + record walk + record decode, the COMPRESSED file is what crosses PCIe.
// (the device form of bam_reader.hip's host path - `pysam.Samfile` iteration, runBESST:162,
// CreateGraph.py:111, libmetrics.py:63,257,293 - for any BGZF block layout: records may straddle blocks and chunks.)
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
//                pattern canonically (first code / count / offset per length) - balanced, no replication loops; a slot
//                that is the prefix of literal / length codes of 11 - 13 bits points to a second-level table of eight
//                entries, filled the same way (codes beyond that, rare, are decoded canonically on the spot); base and
//                extra-bit count of a length / distance code come from two per-lane registers (v_readlane);
//       symbols  A DOZEN per turn of the loop: lane i decodes - speculatively - the literal / length symbol AND the distance
//                symbol that would begin at bit i, 64 + i and 128 + i of the next 192 bits of input (three windows of 64
//                positions; per position two table look-ups, base and extra bits: vector work, the same for every lane),
//                and the chain of symbols that really begin there is then walked window by window with one v_readlane
//                per symbol and compacted into lanes; what the walk cannot take (a code longer than the tables, the end
//                of a block) is left to the one-symbol-at-a-time loop.  (One window per turn, round 4: the per-turn part -
//                input window, scans, the lay-in of the output bytes, loop control - was a third of a turn's ~300
//                instructions for ~4 symbols; three windows: 8.3 -> 6.4 ms per 6144 blocks of a sequencer-like file.)
//       window   the block's own output in HBM / L2 (6.4 KB of LDS, five waves per SIMD) - a wave's vector memory
//                instructions are processed in order, a load behind a store of the same wave returns the stored byte
//                (round 3's other form, a 32 KiB ring in LDS at one wave per SIMD, was 2.3 x slower and is gone);
//       output   in groups of 64 bytes: lane k of a group holds a literal or the place its byte is copied from; a full
//                group, a match that reads from the group, or the end of the block flushes it with one gather and one
//                store (an overlapping match - distance < length - is written directly: byte i is byte i mod dist of its
//                last dist bytes).
//   bgzf_crc_kernel       the CRC-32 of every block's inflated bytes against the block's gzip trailer (what htslib checks):
//                         a thread per slice, the slices' values combined as zlib's crc32_combine does
//   bam_entry_kernel      one wave per block guesses the block's first record start (offset 0 in htslib's layout)
//   bam_walk_kernel       one lane per block: follows the records' length prefixes from the guess (u16 offsets per record,
//                         count, and where the walk leaves the block)
//   bam_scan_kernel       one workgroup: verifies the guesses from block to block and walks a block again where one was
//                         wrong, finds the record the chunk leaves unfinished, exclusive scan of the record counts
//                         (the comment in front of bam_entry_kernel has the scheme)
//   bam_decode_kernel     one workgroup per block, a thread per record: the 36 fixed bytes as ten aligned dwords, the
//                         CIGAR walk of pysam 0.8.4's qlen / alen (bam_reader.hip has the semantics), coalesced stores
//                         into the record columns at the block's place in the stream
// DESIGN.md section 8 has the measurements that led here.
#include <stdlib.h>
#include <mutex>
#include <string.h>

#include "common.h"

namespace besst {

namespace {

constexpr int kTabBits = 10;
constexpr int kTabSize = 1 << kTabBits;
#ifndef BESST_INF_DISTBITS
#define BESST_INF_DISTBITS 9
#endif
#ifndef BESST_INF_WINDOWS
#define BESST_INF_WINDOWS 3
#endif
constexpr int kWin = BESST_INF_WINDOWS;            // a turn of the symbol loop looks at kWin windows of 64 bit positions (kWin per lane):
                                                 // per 6144 blocks of a sequencer-like file 8.3 ms with one window, 6.9 with two, 6.4 with three, 6.3 with four
constexpr int kDistBits = BESST_INF_DISTBITS;                     // the distance code's primary table (10 bits: the same speed; the KB went to the second-level table)
#ifndef BESST_INF_WAVES
#define BESST_INF_WAVES 5
#endif
constexpr int kInfWaves = BESST_INF_WAVES;        // per SIMD (96 VGPRs, 6.4 KB of LDS; six waves at 80 VGPRs spill and read 4 % slower, four 14 %)
constexpr int kDistSize = 1 << kDistBits;
constexpr int kClBits = 7;
constexpr int kLaneLongBits = 3;                 // literal / length codes of up to kTabBits + 3 bits are decoded by the lanes too:
constexpr int kSubCap = 64;                      // through a SECOND-LEVEL table of 2^kLaneLongBits entries per primary slot that is the prefix of longer codes
constexpr uint32_t kLinkFlag = 0x8000u, kLinkMask = 0xC000u;   // primary entry of such a slot: kLinkFlag | first entry of its second-level table << 4
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

struct InflateLds {
    uint16_t lit_tab[kTabSize];
    uint16_t dist_tab[kDistSize];
    uint16_t cl_tab[1 << kClBits];
    uint16_t lit_sub[kSubCap << kLaneLongBits];   // second-level entries: symbol << 4 | length (kTabBits + 1 ..), or kNoEntry
    uint16_t lit_sorted[288];
    uint16_t dist_sorted[32];
    uint16_t cl_sorted[32];
    CanonLds lit_c, dist_c, cl_c;
    uint8_t lens[288 + 32 + 16];
    uint8_t cl_lens[32];
    uint32_t mark[64];          // emission: which symbol of the batch begins at a lane of the output group
    unsigned long long sym[64]; // a batch of several windows: its chain's symbols (bytes | literal or distance << 32), one per lane
};

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// Canonical code of `n` symbols with lengths lens[0..n) (0 = unused): per-length counts, first codes and offsets into
// `sorted` (symbols by length, then by value), then the primary table of 2^bits entries (symbol << 4 | length; kNoEntry:
// the code is longer than the table, or the pattern is not a code).  Returns false when the lengths oversubscribe the
// code space.  All lanes take part; everything returned in LDS.
// sub / owner (literal / length code only): the second-level table.  Every primary slot that is no short code is the prefix
// of longer ones (or of nothing, in an incomplete code); the first kSubCap of them, in slot order, get 2^kLaneLongBits entries
// each - the slot's pattern extended by that many bits, decoded canonically like the primary slots - and a primary entry that
// points there; what lies beyond (more such slots, longer codes) keeps kNoEntry and is left to the one-symbol path.
__device__ __forceinline__ bool build_code(const uint8_t* lens, int n, int bits, uint16_t* tab, uint16_t* sorted, CanonLds* c, int lane,
                                           uint16_t* sub = nullptr, uint32_t* owner = nullptr) {
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
    if (sub != nullptr) {
        uint32_t n_sub = 0;                                  // uniform
        for (int base = 0; base < (1 << bits); base += 64) {
            const int slot = base + lane;
            const bool open = tab[slot] == (uint16_t)kNoEntry;
            const unsigned long long m = __ballot(open);
            const uint32_t r = n_sub + (uint32_t)__popcll(m & lt);
            if (open && r < (uint32_t)kSubCap) {
                owner[r] = (uint32_t)slot;
                tab[slot] = (uint16_t)(kLinkFlag | (r << (kLaneLongBits + 4)));
            }
            n_sub += (uint32_t)__popcll(m);
        }
        if (n_sub > (uint32_t)kSubCap) n_sub = (uint32_t)kSubCap;
        __builtin_amdgcn_wave_barrier();
        const int wide = bits + kLaneLongBits;
        for (uint32_t e = (uint32_t)lane; e < (n_sub << kLaneLongBits); e += 64u) {
            const uint32_t pat = owner[e >> kLaneLongBits] | ((e & ((1u << kLaneLongBits) - 1u)) << bits);   // LSB first, as the input arrives
            const uint32_t r = __brev(pat) >> (32 - wide);
            uint32_t v = kNoEntry;
#pragma unroll
            for (int L = kTabBits + 1; L <= kTabBits + kLaneLongBits; ++L) {
                const uint32_t d = (r >> (wide - L)) - first[L];
                if (v == kNoEntry && d < cnt[L]) v = ((uint32_t)sorted[offs[L] + d] << 4) | (uint32_t)L;
            }
            sub[e] = (uint16_t)v;
        }
        __builtin_amdgcn_wave_barrier();
    }
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
// inclusive scans over the wave (DPP row shifts inside the rows of 16, then row_bcast 15 and 31): sums and maxima of
// values that are zero / non-negative where a lane takes no part
__device__ __forceinline__ uint32_t scan_add(uint32_t v) {
#define BESST_SCAN_STEP(ctrl, rows) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xf, false);
    BESST_SCAN_STEP(0x111, 0xf) BESST_SCAN_STEP(0x112, 0xf) BESST_SCAN_STEP(0x114, 0xf) BESST_SCAN_STEP(0x118, 0xf)
    BESST_SCAN_STEP(0x142, 0xa) BESST_SCAN_STEP(0x143, 0xc)
#undef BESST_SCAN_STEP
    return v;
}
__device__ __forceinline__ uint32_t scan_max(uint32_t v) {
#define BESST_SCAN_STEP(ctrl, rows) { const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xf, false); v = o > v ? o : v; }
    BESST_SCAN_STEP(0x111, 0xf) BESST_SCAN_STEP(0x112, 0xf) BESST_SCAN_STEP(0x114, 0xf) BESST_SCAN_STEP(0x118, 0xf)
    BESST_SCAN_STEP(0x142, 0xa) BESST_SCAN_STEP(0x143, 0xc)
#undef BESST_SCAN_STEP
    return v;
}

// The input as the symbol loop wants it: a POSITION in a window of 64 dwords that the lanes hold (a dword each), not a
// bit buffer.  What a SIMD runs short of in this kernel is its scalar issue slot - one scalar instruction or branch per four
// cycles for all its waves, ~90 % in use by the counters -, and a buffer that is shifted, topped up and tested after every
// symbol spends that slot on bookkeeping: consuming bits is ONE addition here, and the window moves on by half its length
// (two lane permutes and a load, every 1024 bits) so that the four dwords behind the position always lie in it.
struct BitReader {
    const uint32_t* words;      // 4-byte aligned start of the block's payload (uniform)
    uint32_t in0, in1;          // dwords [wbase, wbase + 64) and [wbase + 64, wbase + 128) of it, a dword per lane
    uint32_t wbase;             // (uniform)
    uint32_t bitpos;            // bits of in0 consumed (uniform; < 1024 + what is consumed between two calls of roll())
    int lane;

    __device__ __forceinline__ void seek(uint32_t byte_pos) {
        wbase = byte_pos >> 2;
        in0 = words[wbase + (uint32_t)lane];
        in1 = words[wbase + 64u + (uint32_t)lane];
        bitpos = (byte_pos & 3u) * 8u;
    }
    __device__ __forceinline__ void roll() {
        if (bitpos >= 1024u) {                               // uniform
            const uint32_t a = (uint32_t)__shfl((int)in0, lane + 32, 64), b = (uint32_t)__shfl((int)in1, lane - 32, 64);
            in0 = lane < 32 ? a : b;
            wbase += 32u;
            in1 = words[wbase + 64u + (uint32_t)lane];
            bitpos -= 1024u;
        }
    }
    // the next 32 bits (uniform, in a scalar register)
    __device__ __forceinline__ uint32_t peek32() const {
        const uint32_t i = bitpos >> 5;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)in0, (int)i);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)in0, (int)i + 1);
        return (uint32_t)((((uint64_t)hi << 32) | lo) >> (bitpos & 31u));
    }
    __device__ __forceinline__ void drop(uint32_t n) {
        bitpos += n;
        roll();
    }
    __device__ __forceinline__ uint32_t take(uint32_t n) {
        const uint32_t v = peek32() & ((1u << n) - 1u);
        drop(n);
        return v;
    }
    __device__ __forceinline__ void align_to_byte() { bitpos = (bitpos + 7u) & ~7u; }
    // the four dwords the next 64 bit positions read from (uniform), and the position's offset in the first
    __device__ __forceinline__ uint32_t ahead(uint32_t (&v)[4]) const {
        const uint32_t i = bitpos >> 5;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (uint32_t)__builtin_amdgcn_readlane((int)in0, (int)i + j);
        return bitpos & 31u;
    }
    // ... and the 2 k + 2 dwords that k windows of 64 positions read from
    template <int N>
    __device__ __forceinline__ uint32_t ahead_n(uint32_t (&v)[N]) const {
        const uint32_t i = bitpos >> 5;
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = (uint32_t)__builtin_amdgcn_readlane((int)in0, (int)i + j);
        return bitpos & 31u;
    }
    // bytes of input consumed so far (rounded up)
    __device__ __forceinline__ uint32_t byte_pos() const { return wbase * 4u + ((bitpos + 7u) >> 3); }
};

}  // namespace

// what lane i found at bit i of the window, as the walk reads it: the bits of the whole symbol that would begin there (a
// literal, or a length with its distance), or kWalkStop where that is nothing the batch can take (the end of the block, a
// code longer than the table, an invalid symbol, a match whose distance code begins beyond the 64 positions looked at)
constexpr uint32_t kWalkStop = 0x40u;
constexpr uint32_t kBadDist = 0x80u;                    // lane's distance symbol: bits | distance << 8, or this

__global__ __launch_bounds__(64, kInfWaves) void bgzf_inflate_kernel(const uint8_t* __restrict__ src, const BgzfBlock* __restrict__ blocks,
                                                          uint32_t n_blocks, uint8_t* dst, uint32_t* __restrict__ status) {
    __shared__ InflateLds s;
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
    uint32_t pos = 0;
    uint32_t err = kInfOk;
    // ---- the window: bytes go straight to the block's output and a match reads its source there: a wave's vector memory
    // instructions are processed in order, a load behind a store of the same wave to the same address returns the stored
    // byte (the loads skip the CU's L1).
    auto put_byte = [&](uint32_t at, uint32_t v) { out[at] = (uint8_t)v; };
    auto get_byte = [&](uint32_t at) -> uint32_t {
        return __hip_atomic_load(out + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // ---- output leaves in GROUPS of up to 64 bytes: lane k of the group stands for the byte at pos + k and holds either a
    // literal (flagged by the top bit) or the place its byte is copied from.  A full group (or the end of the block)
    // flushes it: ONE gather of the lanes that copy from finished output, ONE contiguous store.  A lane that copies from a
    // lane of its own group (a match close behind its source, or one that overlaps it) is resolved in registers: every
    // lane follows its source to a lane that holds a byte (six doubling steps of a lane permute, only when there is such
    // a lane).  The group's gather stays in flight while the next group is decoded: it is waited for, and the group
    // stored, at the next flush - before that group's gather is issued, so that every load sees the bytes in front of it.
    uint32_t g = 0;                                          // lane k: kGroupLit | its literal, or where its byte is copied from
    uint32_t filled = 0;                                     // uniform: lanes of the group in use
    uint32_t pend_n = 0, pend_pos = 0;                       // uniform: bytes of the group in flight (0: none), where they go
    uint32_t pend_g = 0, pend_v = 0, pend_s = 0;             // its lanes: literal / the gather's byte / the lane to take the byte from
    bool pend_res = false;                                   // uniform: some lane takes its byte from another lane
    auto retire = [&]() {
        if (pend_n != 0u) {                                  // uniform
            uint32_t val = (pend_g & kGroupLit) ? pend_g : pend_v;
            if (pend_res) val = (uint32_t)__shfl((int)val, (int)pend_s, 64);
            if ((uint32_t)lane < pend_n) put_byte(pend_pos + (uint32_t)lane, val);
            pend_n = 0;
        }
    };
    auto flush_group = [&]() {
        if (filled != 0u) {                                  // uniform
            retire();                                        // (its bytes may be what this group copies from)
            const bool copy = (uint32_t)lane < filled && !(g & kGroupLit);
            const bool own = copy && g >= pos;               // the source is a lane of this very group
            // (every lane loads - one that copies nothing reads the block's first byte -, so that no branch stands between
            // the load and its use at the next flush: behind a branch the compiler waits for it at once)
            pend_v = get_byte((copy && !own) ? g : 0u);
            uint32_t src = own ? g - pos : (uint32_t)lane;
            pend_res = __ballot(own) != 0ull;
            if (pend_res) {
#pragma unroll
                for (int k = 0; k < 6; ++k) src = (uint32_t)__shfl((int)src, (int)src, 64);
            }
            pend_s = src;
            pend_g = g;
            pend_pos = pos;
            pend_n = filled;
            pos = uni(pos + filled);
            filled = 0;
        }
    };
    for (;;) {
        // (a DEFLATE block may be empty: without this test a payload of nothing but empty blocks - corrupt, but every bit of
        // it valid - would be followed out of the chunk's buffer)
        if (br.byte_pos() - in_base > src_len + 8u) { err = kInfInputOverrun; break; }
        const uint32_t final_block = br.take(1);
        const uint32_t type = br.take(2);
        if (type == 0u) {
            // ---- stored: LEN bytes straight from the input
            br.align_to_byte();
            const uint32_t len = br.take(16), nlen = br.take(16);
            if (len != (~nlen & 0xffffu)) { err = kInfBadStored; break; }
            const uint32_t at = br.byte_pos();               // relative to br.words
            if (at - in_base + len > src_len) { err = kInfInputOverrun; break; }
            if (pos + len > dst_len) { err = kInfOutputOverrun; break; }
            const uint8_t* from = reinterpret_cast<const uint8_t*>(br.words) + at;
            flush_group();                                   // (what a Huffman block in front of this one left)
            retire();
            for (uint32_t i = (uint32_t)lane; i < len; i += 64u) put_byte(pos + i, from[i]);
            pos += len;
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
                    const uint32_t e = uni(s.cl_tab[br.peek32() & ((1u << kClBits) - 1u)]);
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
            if (!build_code(s.lens, n_lit, kTabBits, s.lit_tab, s.lit_sorted, &s.lit_c, lane, s.lit_sub, s.mark)) { err = kInfOversubscribed; break; }
            if (!build_code(s.lens + 288, n_dist, kDistBits, s.dist_tab, s.dist_sorted, &s.dist_c, lane)) { err = kInfOversubscribed; break; }
            // ---- the symbols, a batch per turn.  One symbol at a time cost ~50 scalar instructions and branches per symbol,
            // and a SIMD issues ONE of those per four cycles for all its waves: by the counters that slot was 90 % in use
            // and the vector slot 30 % (a sequencer's file: 14 000 symbols per block, more than half of them matches of ~7
            // bytes).  So the work is moved to the vector side:
            //   1. the 64 lanes decode what WOULD begin at each of the next 64 bit positions - literal / length symbol, the
            //      distance symbol a length would be followed by (fetched from the lane where it begins), bits and output
            //      bytes of the whole symbol: the same vector instructions for one lane or for all;
            //   2. the symbols that really begin there are found by following the bit counts from position 0 - a
            //      v_readlane, a bit set and an addition per symbol, the only per-symbol scalar work left;
            //   3. those symbols' bytes are laid into the output group by the lanes: an exclusive scan of their lengths says
            //      where each begins, a marker per symbol and a max-scan say which symbol a byte belongs to, a lane permute
            //      fetches its literal or distance.
            // What a batch cannot take - the end of the block, a code longer than the table - is decoded on its own.
            for (;;) {
                // (once per turn: a corrupt stream is not followed more than a few hundred bytes past its payload - the
                // chunk's buffer has 4 KB behind its last block)
                if (br.byte_pos() - in_base > src_len + 8u) { err = kInfInputOverrun; break; }
                // the speculative decode of ONE window of 64 positions: `x` = the 32 bits of input from the lane's position on.
                // -> bits of the literal / length code + its extra bits (0: nothing the batch can take), the match length,
                // the literal, and what the lane found as a DISTANCE symbol at its position (bits | distance << 8, or kBadDist)
                auto decode = [&](uint32_t x, uint32_t& la_xa, uint32_t& mlen, uint32_t& lit, bool& is_lit, bool& is_len, uint32_t& wb) {
                    const uint32_t ea = s.lit_tab[x & (uint32_t)(kTabSize - 1)];
                    const uint32_t eb = s.dist_tab[x & (uint32_t)(kDistSize - 1)];
                    // codes one, two or three bits longer than the table (a seventh of a sequencer file's literals) are decoded by
                    // the lanes as well: the primary entry of their first kTabBits bits points to a second-level table, indexed by
                    // the next kLaneLongBits bits (every lane looks - a lane with a short code at entry 0 - so that no branch
                    // stands between the two look-ups)
                    const bool link = (ea & kLinkMask) == kLinkFlag;
                    const uint32_t e2 = s.lit_sub[link ? ((ea >> 4) & (uint32_t)((kSubCap << kLaneLongBits) - 1)) + ((x >> kTabBits) & ((1u << kLaneLongBits) - 1u)) : 0u];
                    const uint32_t ec = link ? e2 : ea;
                    const uint32_t la = ec & 15u, sa = ec >> 4;
                    const uint32_t li = (uint32_t)__shfl((int)len_info, (int)(sa - 257u), 64);
                    is_lit = la != 0u && sa < 256u;
                    is_len = la != 0u && sa > 256u && sa < 286u;
                    const uint32_t xa = is_len ? li >> 9 : 0u;
                    mlen = (li & 0x1ffu) + ((x >> la) & ((1u << xa) - 1u));
                    lit = kGroupLit | sa;
                    la_xa = la + xa;
                    const uint32_t lb = eb & 15u, sb = eb >> 4;
                    const uint32_t di = (uint32_t)__shfl((int)dist_info, (int)sb, 64);
                    const uint32_t xb = di >> 16;
                    const uint32_t mdist = (di & 0xffffu) + ((x >> lb) & ((1u << xb) - 1u));
                    wb = (lb != 0u && sb < 30u) ? (lb + xb) | (mdist << 8) : kBadDist;
                };
                // kWin windows of 64 positions per turn: window k's lane l looks at bit 64 k + l
                uint32_t bits_w[kWin], bytes_w[kWin], what_w[kWin], x;
                {
                    uint32_t v[2 * kWin + 2];
                    const uint32_t t = br.ahead_n<2 * kWin + 2>(v) + (uint32_t)lane;     // the lane's bit in window 0, counted from v[0]
                    uint32_t la_xa[kWin], mlen[kWin], lit[kWin], wb[kWin];
                    bool is_lit[kWin], is_len[kWin];
#pragma unroll
                    for (int k = 0; k < kWin; ++k) {
                        const uint32_t lo = t < 32u ? v[2 * k] : t < 64u ? v[2 * k + 1] : v[2 * k + 2];
                        const uint32_t hi = t < 32u ? v[2 * k + 1] : t < 64u ? v[2 * k + 2] : v[2 * k + 3];
                        const uint32_t xk = __builtin_amdgcn_alignbit(hi, lo, t & 31u);   // 32 bits of input from that bit on
                        if (k == 0) x = xk;
                        decode(xk, la_xa[k], mlen[k], lit[k], is_lit[k], is_len[k], wb[k]);
                    }
#pragma unroll
                    for (int k = 0; k < kWin; ++k) {
                        // a length's distance symbol begins la + xa (<= 20) bits on: in this window or in the next one - what the
                        // lane there found (beyond the last window: not this batch's)
                        const uint32_t q = (uint32_t)lane + la_xa[k];
                        uint32_t bq = (uint32_t)__shfl((int)wb[k], (int)q, 64);
                        if (k + 1 < kWin) {
                            const uint32_t bn = (uint32_t)__shfl((int)wb[k + 1], (int)q, 64);
                            bq = q < 64u ? bq : bn;
                        } else {
                            bq = q < 64u ? bq : kBadDist;
                        }
                        const bool is_match = is_len[k] && !(bq & kBadDist);
                        bits_w[k] = is_lit[k] ? la_xa[k] : is_match ? la_xa[k] + (bq & 31u) : kWalkStop;
                        bytes_w[k] = is_lit[k] ? 1u : is_match ? mlen[k] : 0u;
                        what_w[k] = is_lit[k] ? lit[k] : bq >> 8;                          // the literal, or the match's distance
                    }
                }
                // ---- 2. the chain from position 0, window by window
                // (a stop is worth 64 bits: a window's loop has ONE test and ends on it, and what it added is taken back behind
                // the loop - with a test of its own inside, the compiler turned the loop body into sixteen selects)
                unsigned long long chain_w[kWin];
                uint32_t p = 0, n_sym = 0;
                bool stopped = false;
#pragma unroll
                for (int k = 0; k < kWin; ++k) {
                    chain_w[k] = 0ull;
                    if (!stopped) {                          // uniform
                        uint32_t pk = p - 64u * (uint32_t)k, w, from_p;
                        unsigned long long c = 0ull;
                        do {
                            w = (uint32_t)__builtin_amdgcn_readlane((int)bits_w[k], (int)pk);
                            from_p = pk;
                            c |= 1ull << pk;
                            pk += w;
                        } while (pk < 64u);
                        if (w & kWalkStop) {
                            c ^= 1ull << from_p;
                            pk = from_p;
                            stopped = true;
                        }
                        // (a batch lays at most 64 symbols into lanes: a window that would not fit is the next turn's)
                        const uint32_t n_k = (uint32_t)__popcll(c);
                        if (k > 0 && n_sym + n_k > 64u) {
                            stopped = true;
                        } else {
                            chain_w[k] = c;
                            n_sym += n_k;
                            p = 64u * (uint32_t)k + pk;
                        }
                    }
                }
                if (n_sym != 0u) {                           // uniform
                    // ---- 3. the bytes of the chain's symbols
                    bool on = (chain_w[0] >> lane) & 1ull;
                    uint32_t mine = on ? bytes_w[0] : 0u, what = what_w[0];
                    if (kWin > 1 && n_sym != (uint32_t)__popcll(chain_w[0])) {           // uniform
                        // one symbol per lane for what follows: the chain's symbols of all windows, in order, in lanes 0 ..
                        uint32_t before = 0;
#pragma unroll
                        for (int k = 0; k < kWin; ++k) {
                            const unsigned long long c = chain_w[k];
                            const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(c >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)c, before));
                            if ((c >> lane) & 1ull) s.sym[r] = (unsigned long long)bytes_w[k] | ((unsigned long long)what_w[k] << 32);
                            before += (uint32_t)__popcll(c);
                        }
                        __builtin_amdgcn_wave_barrier();
                        const unsigned long long sy = s.sym[lane];
                        __builtin_amdgcn_wave_barrier();
                        on = (uint32_t)lane < n_sym;
                        mine = on ? (uint32_t)sy : 0u;
                        what = (uint32_t)(sy >> 32);
                    }
                    const uint32_t incl = scan_add(mine);
                    const uint32_t begin = incl - mine;      // the symbol's first byte, counted from the batch's first
                    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    if (pos + filled + total > dst_len) { err = kInfOutputOverrun; break; }
                    uint32_t done = 0;
                    bool bad = false;
                    while (done < total) {                   // uniform: once, unless the batch overflows the group
                        const uint32_t room = 64u - filled;
                        const uint32_t take = total - done < room ? total - done : room;
                        // group lane filled + j takes byte done + j of the batch; the symbol that byte belongs to is the
                        // last one beginning at or before it
                        const unsigned long long before = __ballot(on && begin <= done);
                        const uint32_t first = 64u - (uint32_t)__clzll((long long)before);   // (that symbol's lane + 1: never 0)
                        // (the barriers: to the compiler a lane's LDS words are its own between synchronisation points)
                        s.mark[lane] = 0u;
                        __builtin_amdgcn_wave_barrier();
                        if (on && begin > done && begin < done + take) s.mark[filled + begin - done] = (uint32_t)lane + 1u;
                        __builtin_amdgcn_wave_barrier();
                        uint32_t m = s.mark[lane];
                        if ((uint32_t)lane == filled) m = first;
                        m = scan_max(m);
                        const uint32_t from = (uint32_t)__shfl((int)what, (int)(m - 1u), 64);
                        const bool in = (uint32_t)lane >= filled && (uint32_t)lane < filled + take;
                        const uint32_t at = pos + (uint32_t)lane;
                        bad = bad || (in && !(from & kGroupLit) && from > at);
                        g = in ? ((from & kGroupLit) ? from : at - from) : g;
                        filled = uni(filled + take);
                        done += take;
                        if (filled == 64u) flush_group();
                    }
                    if (__ballot(bad) != 0ull) { err = kInfBadDistance; break; }
                    br.drop(p);
                    continue;
                }
                // ---- one symbol on its own: the end of the block, a code longer than the table, an invalid symbol
                const uint32_t xs = (uint32_t)__builtin_amdgcn_readlane((int)x, 0);
                uint32_t e = uni(s.lit_tab[xs & (uint32_t)(kTabSize - 1)]);
                if ((e & 15u) == 0u) {
                    e = uni(slow_code(&s.lit_c, s.lit_sorted, xs & 0x7fffu, kTabBits));
                    if (e == 0u) { err = kInfBadCode; break; }
                }
                uint32_t used = e & 15u;
                const uint32_t sym = e >> 4;
                if (sym < 256u) {
                    g = (uint32_t)lane == filled ? kGroupLit | sym : g;
                    if (pos + filled + 1u > dst_len) { err = kInfOutputOverrun; break; }
                    if (++filled == 64u) flush_group();
                    br.drop(used);
                    continue;
                }
                if (sym == 256u) {
                    br.drop(used);
                    break;
                }
                if (sym >= 286u) { err = kInfBadCode; break; }
                const uint32_t li2 = (uint32_t)__builtin_amdgcn_readlane((int)len_info, (int)(sym - 257u));
                const uint32_t length = (li2 & 0x1ffu) + ((xs >> used) & ((1u << (li2 >> 9)) - 1u));
                used += li2 >> 9;                            // <= 20: the distance symbol begins inside the window
                const uint32_t xq = (uint32_t)__builtin_amdgcn_readlane((int)x, (int)used);
                uint32_t d = uni(s.dist_tab[xq & (uint32_t)(kDistSize - 1)]);
                if ((d & 15u) == 0u) {
                    d = uni(slow_code(&s.dist_c, s.dist_sorted, xq & 0x7fffu, kDistBits));
                    if (d == 0u) { err = kInfBadCode; break; }
                }
                if ((d >> 4) >= 30u) { err = kInfBadCode; break; }
                const uint32_t di2 = (uint32_t)__builtin_amdgcn_readlane((int)dist_info, (int)(d >> 4));
                const uint32_t dist = (di2 & 0xffffu) + ((xq >> (d & 15u)) & ((1u << (di2 >> 16)) - 1u));
                used += (d & 15u) + (di2 >> 16);
                if (dist > pos + filled) { err = kInfBadDistance; break; }
                if (pos + filled + length > dst_len) { err = kInfOutputOverrun; break; }
                uint32_t left = length;
                while (left != 0u) {                         // uniform: the match joins the group piece by piece
                    const uint32_t room = 64u - filled;
                    const uint32_t take = left < room ? left : room;
                    const bool in = (uint32_t)lane >= filled && (uint32_t)lane < filled + take;
                    g = in ? pos + (uint32_t)lane - dist : g;
                    filled = uni(filled + take);
                    left -= take;
                    if (filled == 64u) flush_group();
                }
                br.drop(used);
            }
            if (err) break;
        } else {
            err = kInfBadBlockType;
            break;
        }
        if (final_block) break;
    }
    if (!err) flush_group();
    retire();
    if (!err) {
        if (pos != dst_len) err = kInfSizeMismatch;
        else if (br.byte_pos() - in_base > src_len + 8u) err = kInfInputOverrun;
    }
    if (lane == 0) status[b] = err;
}

// ---- the second form of the inflate (round 6): the symbols of a DEFLATE block decoded by the 64 lanes SIDE BY SIDE ------------
// bgzf_inflate_kernel above finds a dozen symbols per turn of ~300 vector instructions: every lane decodes speculatively
// at one BIT position and one lane in five holds a real symbol.  It is bound by vector issue alone (351 k instructions per
// block x 4 cycles x 5 waves per SIMD = the 3.4 ms a chunk of 5120 blocks takes).  Here every lane decodes REAL symbols, one
// after the other, in its own 1/64 of the block's bits:
//   1. the block's bits are cut into 64 equal ranges; lane 0 begins where the block's symbols begin, the others at a guess
//      (their range's first bit).  Every lane decodes until it leaves its range and hands the position where it did to its
//      neighbour; a lane whose start changed decodes again - but only until it is in step with what it decoded before: a
//      Huffman decoder that starts inside a symbol falls into step with the true chain after a few dozen symbols (~250 lie
//      in a range), and a lane remembers where its chain stood at three CHECKPOINTS of its range and what it had counted
//      until there.  The hand-overs are stable after two or three rounds - and exact after at most 64: lane k's start is
//      final once the k lanes in front of it are;
//   2. the symbols and bytes each lane's chain makes are known then: exclusive scans place the lanes' symbols in the
//      block's symbol buffer (four bytes per symbol: bytes | literal or distance) and their bytes in the output;
//   3. the lanes decode once more and store their symbols, four at a time;
//   4. the symbols are read back 64 at a time, in order, and their bytes laid into output groups of 64 bytes - the flush of
//      the first form: a scan of the lengths, a marker per symbol and a max-scan, one gather of the lanes that copy from
//      finished output, six doubling steps of a lane permute for those that copy from the group itself, one store.
// Same tables, same statuses, same CRC kernel behind it.  BESST_INFLATE=1 selects the first form.
// Measured (a sequencer-like file, 4096 blocks per launch, the kernel alone): 1.89 ms against 4.04 ms; by the counters
// 121 k vector + 68 k scalar instructions per block against 351 k + 250 k.  A quarter of the time is the rounds of
// hand-overs, a quarter the second decode, nearly half the flush (54 k + 25 k instructions per block: ~80 per output
// group).  What it costs: four bytes of symbol buffer per inflated byte of a chunk (touched: ~56 KB of a block's 260 KB).
constexpr int kInf2Waves = 5;                    // per SIMD (96 VGPRs; six at 80 spill and gain nothing, four lose 7 %)
struct Inflate2Lds {
    uint16_t lit_tab[kTabSize];
    uint16_t dist_tab[kDistSize];
    uint16_t cl_tab[1 << kClBits];
    uint16_t lit_sub[kSubCap << kLaneLongBits];
    uint16_t lit_sorted[288];
    uint16_t dist_sorted[32];
    uint16_t cl_sorted[32];
    CanonLds lit_c, dist_c, cl_c;
    uint8_t lens[288 + 32 + 16];
    uint8_t cl_lens[32];
    uint32_t mark[64];
    uint32_t len_info[32], dist_info[32];      // RFC 1951's length / distance codes: base | extra bits << 9 (<< 16)
};

// what a lane reads its symbols from: three dwords of the payload from word `w` on (96 bits: a length with its distance and
// all extra bits is at most 48) - and the dword behind them, loaded when the window moved on last: a lane's stream is its
// own (64 lanes, 64 cache lines; twenty waves' lines do not stay in a CU's L1), and a load whose dword the very next symbol
// needs costs the way to the L2 and back once per symbol
struct LaneBits {
    const uint32_t* words;
    uint32_t w, d0, d1, d2, d3;
    __device__ __forceinline__ void seek(uint32_t p) {
        w = p >> 5;
        d0 = words[w]; d1 = words[w + 1u]; d2 = words[w + 2u]; d3 = words[w + 3u];
    }
    __device__ __forceinline__ void advance(uint32_t p) {    // p is at most two words on
        if ((p >> 5) > w) { d0 = d1; d1 = d2; d2 = d3; ++w; d3 = words[w + 3u]; }
        if ((p >> 5) > w) { d0 = d1; d1 = d2; d2 = d3; ++w; d3 = words[w + 3u]; }
    }
    // 32 bits from bit q of d0 on, q < 64
    __device__ __forceinline__ uint32_t at(uint32_t q) const {
        const bool far = q >= 32u;
        return __builtin_amdgcn_alignbit(far ? d2 : d1, far ? d1 : d0, q & 31u);
    }
};

enum : uint32_t { kSymLit = 0, kSymMatch = 1, kSymEnd = 2, kSymBad = 3 };
constexpr uint32_t kSymIsLit = 0x8000u;                  // a stored symbol: bytes (9 bits) | this | literal or distance << 16
constexpr uint32_t kCountSym = 1u << 17;                 // a lane's count: bytes + symbols << 17 (modulo 2^32 while it speculates)
constexpr uint32_t kGrid0 = 64u, kGrid1 = 192u, kGrid2 = 448u;   // the checkpoints: bits behind the first bit of the lane's range
constexpr uint32_t kSymSlack = 256u;                     // symbol places a block has beyond one per byte (lanes are padded to 4)

// one symbol at bit p of the lane's window: kind, bits it takes, bytes it makes, literal / distance.  No branch but for codes
// longer than the tables: the lanes of a wave sit in different symbols, and what is a branch to one lane is a detour for all
// (by the counters the loop issued more scalar instructions - execution masks, jumps - than vector ones): a literal looks a
// distance up like a length does and throws it away.
__device__ __forceinline__ uint32_t decode_symbol(const Inflate2Lds& s, const LaneBits& lb, uint32_t p, uint32_t& bits, uint32_t& bytes,
                                                  uint32_t& what) {
    const uint32_t q0 = p & 31u;
    const uint32_t x = lb.at(q0);
    uint32_t e = s.lit_tab[x & (uint32_t)(kTabSize - 1)];
    const bool link = (e & kLinkMask) == kLinkFlag;
    const uint32_t e2 = s.lit_sub[link ? ((e >> 4) & (uint32_t)((kSubCap << kLaneLongBits) - 1)) + ((x >> kTabBits) & ((1u << kLaneLongBits) - 1u)) : 0u];
    e = link ? e2 : e;
    if ((e & 15u) == 0u) e = slow_code(&s.lit_c, s.lit_sorted, x & 0x7fffu, kTabBits);   // (0: not a code)
    const uint32_t la = e & 15u, sa = e >> 4;
    const bool is_len = sa > 256u;
    const uint32_t li = s.len_info[is_len ? (sa - 257u) & 31u : 0u];       // (codes 286, 287: no length - entries of zero)
    const uint32_t xa = is_len ? li >> 9 : 0u;
    const uint32_t xq = lb.at(q0 + la + xa);                 // (la + xa <= 20: inside the window)
    uint32_t d = s.dist_tab[xq & (uint32_t)(kDistSize - 1)];
    if (is_len && (d & 15u) == 0u) d = slow_code(&s.dist_c, s.dist_sorted, xq & 0x7fffu, kDistBits);
    const uint32_t lb_ = d & 15u, sb = d >> 4;
    const uint32_t di = s.dist_info[sb & 31u];               // (codes 30, 31: no distance - entries of zero)
    const uint32_t xb = di >> 16;
    const uint32_t dist = (di & 0xffffu) + __builtin_amdgcn_ubfe(xq, lb_, xb);
    bits = is_len ? la + xa + lb_ + xb : la;
    bytes = is_len ? (li & 0x1ffu) + __builtin_amdgcn_ubfe(x, la, xa) : 1u;
    what = is_len ? dist : sa;
    const bool bad = la == 0u || (is_len && (li == 0u || lb_ == 0u || di == 0u));
    return bad ? (uint32_t)kSymBad : sa < 256u ? (uint32_t)kSymLit : sa == 256u ? (uint32_t)kSymEnd : (uint32_t)kSymMatch;
}

__global__ __launch_bounds__(64, kInf2Waves) void bgzf_inflate2_kernel(const uint8_t* __restrict__ src, const BgzfBlock* __restrict__ blocks,
                                                                       uint32_t n_blocks, uint8_t* dst, uint32_t* sym_all,
                                                                       uint32_t* __restrict__ status) {
    __shared__ Inflate2Lds s;
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) return;
    const uint32_t src_off = uni(blocks[b].src_off), src_len = uni(blocks[b].src_len);
    const uint32_t dst_len = uni(blocks[b].dst_len);
    const size_t out_at = (size_t)uni(blocks[b].dst_off_lo) + ((size_t)uni(blocks[b].dst_off_hi) << 32);
    uint8_t* out = dst + out_at;
    // the block's symbols: dst_len + kSymSlack places (a symbol makes a byte at least; 64 lanes are padded to four symbols),
    // 16-byte aligned, behind those of the blocks before
    uint32_t* sym = sym_all + ((out_at + 3u) & ~(size_t)3u) + (size_t)kSymSlack * b;
    if (dst_len == 0) {                                      // the EOF marker block
        if (lane == 0) status[b] = kInfOk;
        return;
    }
    BitReader br;
    br.lane = lane;
    br.words = reinterpret_cast<const uint32_t*>(src + (src_off & ~3u));
    br.seek(src_off & 3u);
    const uint32_t in_base = src_off & 3u;
    const uint32_t end_bit = (in_base + src_len) * 8u;       // the payload's last bit + 1, counted like the lanes' positions
    {
        const uint32_t k = (uint32_t)lane;
        uint32_t base, extra = 0;
        if (k < 8u) base = 3u + k;
        else if (k == 28u) base = 258u;
        else {
            extra = (k - 4u) >> 2;
            base = 3u + ((4u + (k & 3u)) << extra);
        }
        if (lane < 32) s.len_info[lane] = k < 29u ? base | (extra << 9) : 0u;
        extra = 0;
        if (k < 4u) base = 1u + k;
        else {
            extra = (k - 2u) >> 1;
            base = 1u + ((2u + (k & 1u)) << extra);
        }
        if (lane < 32) s.dist_info[lane] = k < 30u ? base | (extra << 16) : 0u;
    }
    uint32_t pos = 0;
    uint32_t err = kInfOk;
    // ---- output groups, as in the first form (see there): lane k of the group stands for the byte at pos + k
    auto put_byte = [&](uint32_t at, uint32_t v) { out[at] = (uint8_t)v; };
    auto get_byte = [&](uint32_t at) -> uint32_t { return __hip_atomic_load(out + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    uint32_t g = 0, filled = 0, pend_n = 0, pend_pos = 0, pend_g = 0, pend_v = 0, pend_s = 0;
    bool pend_res = false;
    auto retire = [&]() {
        if (pend_n != 0u) {                                  // uniform
            uint32_t val = (pend_g & kGroupLit) ? pend_g : pend_v;
            if (pend_res) val = (uint32_t)__shfl((int)val, (int)pend_s, 64);
            if ((uint32_t)lane < pend_n) put_byte(pend_pos + (uint32_t)lane, val);
            pend_n = 0;
        }
    };
    auto flush_group = [&]() {
        if (filled != 0u) {                                  // uniform
            retire();
            const bool copy = (uint32_t)lane < filled && !(g & kGroupLit);
            const bool own = copy && g >= pos;
            pend_v = get_byte((copy && !own) ? g : 0u);
            uint32_t from = own ? g - pos : (uint32_t)lane;
            pend_res = __ballot(own) != 0ull;
            if (pend_res) {
#pragma unroll
                for (int k = 0; k < 6; ++k) from = (uint32_t)__shfl((int)from, (int)from, 64);
            }
            pend_s = from;
            pend_g = g;
            pend_pos = pos;
            pend_n = filled;
            pos = uni(pos + filled);
            filled = 0;
        }
    };
    for (;;) {
        if (br.byte_pos() - in_base > src_len + 8u) { err = kInfInputOverrun; break; }
        const uint32_t final_block = br.take(1);
        const uint32_t type = br.take(2);
        if (type == 0u) {
            br.align_to_byte();
            const uint32_t len = br.take(16), nlen = br.take(16);
            if (len != (~nlen & 0xffffu)) { err = kInfBadStored; break; }
            const uint32_t at = br.byte_pos();
            if (at - in_base + len > src_len) { err = kInfInputOverrun; break; }
            if (pos + len > dst_len) { err = kInfOutputOverrun; break; }
            const uint8_t* from = reinterpret_cast<const uint8_t*>(br.words) + at;
            for (uint32_t i = (uint32_t)lane; i < len; i += 64u) out[pos + i] = from[i];
            pos += len;
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
                for (uint32_t i = 0; i < hclen; ++i) {
                    const uint32_t v = br.take(3);
                    const uint32_t order = i < 3u ? 16u + i : i == 3u ? 0u : (i & 1u) ? 8u - ((i - 3u) >> 1) : 8u + ((i - 4u) >> 1);
                    s.cl_lens[order] = (uint8_t)v;
                }
                __builtin_amdgcn_wave_barrier();
                if (!build_code(s.cl_lens, 19, kClBits, s.cl_tab, s.cl_sorted, &s.cl_c, lane)) { err = kInfOversubscribed; break; }
                const uint32_t total = hlit + hdist;
                uint32_t have = 0, prev = 0;
                bool bad = false;
                while (have < total) {
                    const uint32_t e = uni(s.cl_tab[br.peek32() & ((1u << kClBits) - 1u)]);
                    const uint32_t l = e & 15u, sym_ = e >> 4;
                    if (l == 0u) { bad = true; break; }
                    br.drop(l);
                    if (sym_ < 16u) {
                        s.lens[have++] = (uint8_t)sym_;
                        prev = sym_;
                        continue;
                    }
                    uint32_t rep, val = 0;
                    if (sym_ == 16u) {
                        if (have == 0u) { bad = true; break; }
                        rep = 3u + br.take(2);
                        val = prev;
                    } else if (sym_ == 17u) {
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
                if (lane < 32) {
                    const uint8_t v = (uint32_t)lane < hdist ? s.lens[hlit + (uint32_t)lane] : (uint8_t)0;
                    __builtin_amdgcn_wave_barrier();
                    s.lens[288 + lane] = v;
                }
                __builtin_amdgcn_wave_barrier();
                for (uint32_t i = hlit + (uint32_t)lane; i < 288u; i += 64u) s.lens[i] = 0;
                n_lit = (int)hlit;
                n_dist = (int)hdist;
                if (uni(s.lens[256]) == 0u) { err = kInfBadLengths; break; }
            }
            __builtin_amdgcn_wave_barrier();
            if (!build_code(s.lens, n_lit, kTabBits, s.lit_tab, s.lit_sorted, &s.lit_c, lane, s.lit_sub, s.mark)) { err = kInfOversubscribed; break; }
            if (!build_code(s.lens + 288, n_dist, kDistBits, s.dist_tab, s.dist_sorted, &s.dist_c, lane)) { err = kInfOversubscribed; break; }
            __builtin_amdgcn_wave_barrier();
            // ---- 1. the lanes' ranges and the rounds of hand-overs
            const uint32_t p0 = br.wbase * 32u + br.bitpos;  // where the block's symbols begin (uniform)
            if (p0 >= end_bit) { err = kInfInputOverrun; break; }
            uint32_t range = (end_bit - p0 + 63u) >> 6;
            if (range < 64u) range = 64u;                    // (longer than any symbol: a hand-over lies in the next lane's range)
            const uint32_t lo = p0 + (uint32_t)lane * range; // the lane's first bit, and the first that is not its own
            const uint32_t bound = lo + range;
            const uint32_t lim = bound < end_bit ? bound : end_bit;
            uint32_t start = lo;
            bool live = start < end_bit;
            uint32_t done_for = 0xffffffffu;                 // the start the lane's results below belong to
            uint32_t r_end = 0, r_kind = kSymBad, tot = 0;   // of its chain: where it left the range, how, what it counted
            uint32_t cp0 = 0xffffffffu, cp1 = 0xffffffffu, cp2 = 0xffffffffu, pre0 = 0, pre1 = 0, pre2 = 0;   // ... and where it stood at
            LaneBits lb;                                     // the checkpoints, what it had counted until there
            lb.words = br.words;
            for (int round = 0; round < 130; ++round) {
                const bool need = live && start != done_for;
                if (need) {
                    uint32_t p = start, cnt = 0, kind = kSymLit, j = 0, grid = lo + kGrid0;
                    bool in_step = false;
                    lb.seek(p);
                    bool go = p < lim;
                    while (go) {                             // (one condition, computed: every `break` is a mask to keep per turn)
                        uint32_t bits, bytes, what;
                        kind = decode_symbol(s, lb, p, bits, bytes, what);
                        const bool sym = kind < (uint32_t)kSymEnd;
                        cnt += sym ? bytes + kCountSym : 0u;
                        p += (sym || kind == (uint32_t)kSymEnd) ? bits : 0u;
                        lb.advance(p);
                        if (sym && p >= grid) {              // the first symbol behind a checkpoint: in step with the chain before?
                            const uint32_t was = j == 0u ? cp0 : j == 1u ? cp1 : cp2;
                            in_step = p == was;
                            if (!in_step) {
                                if (j == 0u) { cp0 = p; pre0 = cnt; grid = lo + kGrid1; }
                                else if (j == 1u) { cp1 = p; pre1 = cnt; grid = lo + kGrid2; }
                                else { cp2 = p; pre2 = cnt; grid = 0xffffffffu; }
                                ++j;
                            }
                        }
                        go = sym && !in_step && p < lim;
                    }
                    if (in_step) {                           // the rest is what it was: only what lies in front of the checkpoint changed
                        const uint32_t delta = cnt - (j == 0u ? pre0 : j == 1u ? pre1 : pre2);
                        tot += delta;
                        if (j == 0u) { pre0 = cnt; pre1 += delta; pre2 += delta; }
                        else if (j == 1u) { pre1 = cnt; pre2 += delta; }
                        else pre2 = cnt;
                    } else {
                        if (kind < (uint32_t)kSymEnd && p >= end_bit) kind = kSymBad;      // the payload ends inside the block
                        tot = cnt;
                        r_end = p;
                        r_kind = kind >= (uint32_t)kSymEnd ? kind : (uint32_t)kSymLit;
                        if (j <= 0u) cp0 = 0xffffffffu;      // (checkpoints this chain did not reach are no longer its own)
                        if (j <= 1u) cp1 = 0xffffffffu;
                        if (j <= 2u) cp2 = 0xffffffffu;
                    }
                    done_for = start;
                }
                // the hand-over: a lane lives when the lane before it lives and left its range in the middle of the stream
                const uint32_t up_end = (uint32_t)__shfl_up((int)r_end, 1, 64);
                const uint32_t up_kind = (uint32_t)__shfl_up((int)r_kind, 1, 64);
                const int up_live = __shfl_up((int)live, 1, 64);
                bool changed = false;
                if (lane > 0) {
                    const bool now = up_live != 0 && up_kind == (uint32_t)kSymLit && up_end < end_bit;
                    changed = now != live || (now && up_end != start);
                    live = now;
                    if (live) start = up_end;
                }
                if (__ballot(changed) == 0ull) break;        // uniform: every hand-over is what it was, every lane has decoded its own
            }
            // the stream ends in the first lane whose symbols did not leave its range
            const unsigned long long m_stop = __ballot(live && r_kind != (uint32_t)kSymLit);
            if (m_stop == 0ull) { err = kInfInputOverrun; break; }          // no end-of-block code inside the payload
            const int last = __ffsll((long long)m_stop) - 1;
            if ((uint32_t)__builtin_amdgcn_readlane((int)r_kind, last) != (uint32_t)kSymEnd) { err = kInfBadCode; break; }
            const bool mine = live && lane <= last;
            // ---- 2. where the lanes' symbols and bytes go
            const uint32_t nb = mine ? tot & (kCountSym - 1u) : 0u;
            const uint32_t ns = mine ? tot >> 17 : 0u;
            const uint32_t np = (ns + 3u) & ~3u;
            const uint32_t incl_b = scan_add(nb), incl_s = scan_add(np);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl_b, 63);
            const uint32_t n_sym = (uint32_t)__builtin_amdgcn_readlane((int)incl_s, 63);
            // (the counts are exact modulo 2^32 only: a corrupt stream may claim anything - the places are checked, not trusted)
            if (__ballot(mine && (nb > dst_len || ns > dst_len)) != 0ull || pos + total > dst_len || n_sym > dst_len + kSymSlack - 4u) {
                err = kInfOutputOverrun;
                break;
            }
            // ---- 3. the symbols, four at a time
            bool bad_dist = false, overrun = false;
            if (mine) {
                uint32_t o = pos + incl_b - nb, p = start, at = incl_s - np, have = 0;
                const uint32_t o_end = pos + incl_b;
                uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
                lb.seek(p);
                bool go = p < r_end;
                while (go) {
                    uint32_t bits, bytes, what;
                    const uint32_t kind = decode_symbol(s, lb, p, bits, bytes, what);
                    const bool far = kind == (uint32_t)kSymMatch && what > o;
                    // (a lane's counts are exact modulo 2^32: a corrupt stream whose chain makes more than 2^17 bytes in one
                    // range would write past the places its lanes were given - into other blocks' symbols and bytes)
                    const bool over = kind < (uint32_t)kSymEnd && (have >= ns || o + bytes > o_end);
                    const bool is_sym = kind < (uint32_t)kSymEnd && !far && !over;   // (else: the end-of-block code - the last lane's last symbol)
                    bad_dist = bad_dist || far;
                    overrun = overrun || over;
                    if (is_sym) {
                        o += bytes;
                        q0 = q1; q1 = q2; q2 = q3;
                        q3 = bytes | (kind == (uint32_t)kSymLit ? kSymIsLit : 0u) | (what << 16);
                        if ((++have & 3u) == 0u) {
                            *reinterpret_cast<uint4*>(sym + at) = make_uint4(q0, q1, q2, q3);
                            at += 4u;
                        }
                        p += bits;
                        lb.advance(p);
                    }
                    go = is_sym && p < r_end;
                }
                const uint32_t rest = have & 3u;             // (the padding: symbols of no bytes)
                if (rest != 0u && !bad_dist && !overrun)
                    *reinterpret_cast<uint4*>(sym + at) = rest == 1u ? make_uint4(q3, 0u, 0u, 0u) : rest == 2u ? make_uint4(q2, q3, 0u, 0u)
                                                                                                       : make_uint4(q1, q2, q3, 0u);
            }
            if (__ballot(bad_dist) != 0ull) { err = kInfBadDistance; break; }
            if (__ballot(overrun) != 0ull) { err = kInfOutputOverrun; break; }
            // ---- 4. their bytes, 64 symbols at a time
            uint32_t sy_next = (uint32_t)lane < n_sym ? __hip_atomic_load(sym + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            for (uint32_t i = 0; i < n_sym; i += 64u) {      // uniform
                const uint32_t sy = sy_next;                 // (the batch behind this one is on its way while this one is laid out)
                sy_next = i + 64u + (uint32_t)lane < n_sym ? __hip_atomic_load(sym + i + 64u + (uint32_t)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                const uint32_t bytes = sy & 0x1ffu;
                const bool on = bytes != 0u;
                const uint32_t what = (sy & kSymIsLit) ? kGroupLit | ((sy >> 16) & 0xffu) : sy >> 16;
                const uint32_t incl = scan_add(bytes);
                const uint32_t begin = incl - bytes;         // the symbol's first byte, counted from the batch's first
                const uint32_t n_bytes = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                uint32_t done = 0;
                while (done < n_bytes) {                     // uniform
                    const uint32_t room = 64u - filled;
                    const uint32_t take = n_bytes - done < room ? n_bytes - done : room;
                    // group lane filled + j takes byte done + j of the batch; the symbol that byte belongs to is the last one
                    // beginning at or before it
                    const unsigned long long before = __ballot(on && begin <= done);
                    const uint32_t first = 64u - (uint32_t)__clzll((long long)before);   // (that symbol's lane + 1: never 0)
                    s.mark[lane] = 0u;
                    __builtin_amdgcn_wave_barrier();
                    if (on && begin > done && begin < done + take) s.mark[filled + begin - done] = (uint32_t)lane + 1u;
                    __builtin_amdgcn_wave_barrier();
                    uint32_t m = s.mark[lane];
                    if ((uint32_t)lane == filled) m = first;
                    m = scan_max(m);
                    const uint32_t from = (uint32_t)__shfl((int)what, (int)(m - 1u), 64);
                    const bool in = (uint32_t)lane >= filled && (uint32_t)lane < filled + take;
                    const uint32_t at = pos + (uint32_t)lane;
                    g = in ? ((from & kGroupLit) ? from : at - from) : g;
                    filled = uni(filled + take);
                    done += take;
                    if (filled == 64u) flush_group();
                }
            }
            flush_group();                                   // (a stored block writes its bytes itself)
            const uint32_t after = (uint32_t)__builtin_amdgcn_readlane((int)r_end, last);   // the bit behind the end-of-block code
            br.seek(after >> 3);
            br.bitpos += after & 7u;
        } else {
            err = kInfBadBlockType;
            break;
        }
        if (final_block) break;
    }
    retire();
    if (!err) {
        if (pos != dst_len) err = kInfSizeMismatch;
        else if (br.byte_pos() - in_base > src_len + 8u) err = kInfInputOverrun;
    }
    if (lane == 0) status[b] = err;
}

// symbol places (four bytes each) the second form needs for n_blocks blocks that inflate to inflated_bytes bytes
size_t bgzf_inflate_symbol_places(size_t inflated_bytes, size_t n_blocks) { return inflated_bytes + 8 + (size_t)kSymSlack * (n_blocks + 1); }

// ---- the blocks' CRC32 ---------------------------------------------------------------------------------------------------
// BGZF stores the CRC-32 of every block's inflated bytes (the gzip trailer); htslib - what the reference reads its files
// through - checks it, and so does this path: a damaged payload that still decodes, or a byte the window logic got wrong,
// is a refused block, not a wrong record.  One workgroup per block, a thread per slice of <= 272 bytes: byte-table CRC
// of the slice, then the slice's CRC is carried over the bytes behind it - multiplication by x^(8n) modulo the CRC polynomial,
// zlib's crc32_combine, the powers from two tables - and the 256 values are XORed.
namespace {

constexpr uint32_t kCrcPoly = 0xedb88320u;
__device__ const uint32_t kCrcX2n[32] = {           // x^(2^k) mod the polynomial, reflected (zlib's x2n_table)
    0x40000000u, 0x20000000u, 0x08000000u, 0x00800000u, 0x00008000u, 0xedb88320u, 0xb1e6b092u, 0xa06a2517u,
    0xed627daeu, 0x88d14467u, 0xd7bbfe6au, 0xec447f11u, 0x8e7ea170u, 0x6427800eu, 0x4d47bae0u, 0x09fe548fu,
    0x83852d0fu, 0x30362f1au, 0x7b5a9cc3u, 0x31fec169u, 0x9fec022au, 0x6c8dedc4u, 0x15d6874du, 0x5fde7a4eu,
    0xbad90e37u, 0x2e4e5eefu, 0x4eaba214u, 0xa8a472c0u, 0x429a969eu, 0x148d302au, 0xc40ba6d0u, 0xc4e22c3cu};

// a(x) * b(x) modulo the polynomial (both reflected: bit 31 is x^0), any operands per lane
__device__ __forceinline__ uint32_t crc_mul_lanes(uint32_t a, uint32_t b) {
    uint32_t p = 0;
#pragma unroll 8
    for (int j = 0; j < 32; ++j) {
        p ^= (a & (1u << 31)) ? b : 0u;
        a <<= 1;
        b = (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return p;
}

// x^(8 n) modulo the polynomial for the n a slice's CRC has to be carried over: kCrcPow16[j] = x^(128 j) (whole slices behind
// it: slices are multiples of 16 bytes), kCrcPow8[r] = x^(8 r) (the last slice's bytes).  Filled once per device by
// crc_pow_kernel (binary exponentiation over zlib's x2n table); the per-thread exponentiation they replace - up to sixteen
// 32-step multiplications per slice - was 60 % of the CRC kernel's instructions.
constexpr int kCrcPow16N = 4352, kCrcPow8N = 288;
__device__ uint32_t kCrcPow16[kCrcPow16N];
__device__ uint32_t kCrcPow8[kCrcPow8N];

__global__ __launch_bounds__(256) void crc_pow_kernel() {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= (uint32_t)(kCrcPow16N + kCrcPow8N)) return;
    const bool small = i >= (uint32_t)kCrcPow16N;
    uint32_t n = small ? i - (uint32_t)kCrcPow16N : i;       // the exponent, in units of 8 bits (small) or 128 bits
    uint32_t p = 1u << 31;                                   // x^0
    for (uint32_t k = small ? 3u : 7u; n; n >>= 1, ++k)
        if (n & 1u) p = crc_mul_lanes(kCrcX2n[k & 31u], p);
    if (small) kCrcPow8[i - (uint32_t)kCrcPow16N] = p;
    else kCrcPow16[i] = p;
}

}  // namespace

__global__ __launch_bounds__(256) void bgzf_crc_kernel(const uint8_t* __restrict__ inflated, const BgzfBlock* __restrict__ blocks,
                                                       uint32_t n_blocks, uint32_t* __restrict__ status) {
    // four tables (slicing by four): entry [j][t] is byte t carried over j further zero bytes, so that a dword of input costs
    // four INDEPENDENT look-ups instead of a chain of four (a thread's 256-byte slice was a chain of 256 LDS round trips)
    __shared__ uint32_t s_tab[4][256], s_part[5];
    const uint32_t b = blockIdx.x, t = threadIdx.x;
    if (b >= n_blocks) return;
    const uint32_t len = blocks[b].dst_len;
    if (len == 0u || status[b] != kInfOk) return;            // uniform
    {   // the byte table: entry t is t carried through eight steps of the polynomial division
        uint32_t c = t;
#pragma unroll
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ kCrcPoly : c >> 1;
        s_tab[0][t] = c;
        __syncthreads();
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            c = s_tab[0][c & 0xffu] ^ (c >> 8);
            s_tab[j][t] = c;
        }
    }
    __syncthreads();
    const uint8_t* first = inflated + (size_t)blocks[b].dst_off_lo + ((size_t)blocks[b].dst_off_hi << 32);
    // The blocks of a chunk lie back to back (a record may run on into the next block), so a block begins at any byte: the
    // slices are cut from the 16-byte boundary in front of it - slice t = [t S, (t + 1) S) of `pad` foreign bytes followed by
    // the block's, S a multiple of 16, every slice read as aligned 16-byte words (the next word requested before the
    // current one's look-ups; a byte at a time the loads alone, each waited for, took longer than the inflate) - and
    // the first slice skips the foreign bytes.
    const uint32_t pad = (uint32_t)((uintptr_t)first & 15u);
    const uint8_t* base = first - pad;
    const uint32_t vlen = len + pad;
    const uint32_t S = (((vlen + 255u) >> 8) + 15u) & ~15u;
    const uint32_t lo = t * S < vlen ? t * S : vlen;
    const uint32_t hi = lo + S < vlen ? lo + S : vlen;
    uint32_t skip = lo < pad ? (pad - lo < hi - lo ? pad - lo : hi - lo) : 0u;   // (only slice 0: S >= 16 > pad)
    const uint32_t n_slice = hi - lo - skip;
    const uint4* src = reinterpret_cast<const uint4*>(base + lo);
    uint32_t crc = 0xffffffffu;
    auto eat = [&](uint32_t w, uint32_t from, uint32_t count) {   // bytes [from, count) of a word
        if (from == 0u && count >= 4u) {
            const uint32_t x = crc ^ w;
            crc = s_tab[3][x & 0xffu] ^ s_tab[2][(x >> 8) & 0xffu] ^ s_tab[1][(x >> 16) & 0xffu] ^ s_tab[0][x >> 24];
            return;
        }
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            if (k >= from && k < count) {
                crc = s_tab[0][(crc ^ (w >> (8u * k))) & 0xffu] ^ (crc >> 8);
            }
        }
    };
    auto part = [](uint32_t v, uint32_t at) { return v > at ? (v - at < 4u ? v - at : 4u) : 0u; };   // of bytes [0, v): how many lie in dword `at / 4`
    uint4 cur = hi > lo ? src[0] : make_uint4(0, 0, 0, 0);
    for (uint32_t q = 0; q * 16u < hi - lo; ++q) {
        const uint4 next = (q + 1u) * 16u < hi - lo ? src[q + 1u] : make_uint4(0, 0, 0, 0);
        const uint32_t rem = hi - lo - q * 16u;                  // bytes of the slice from this word on (>= 16: all of it)
        const uint32_t sk = q == 0u ? skip : 0u;
        eat(cur.x, part(sk, 0), part(rem, 0)); eat(cur.y, part(sk, 4), part(rem, 4));
        eat(cur.z, part(sk, 8), part(rem, 8)); eat(cur.w, part(sk, 12), part(rem, 12));
        cur = next;
    }
    crc = n_slice ? ~crc : 0u;
    // The block's CRC = XOR over the slices of (slice's CRC) x^(8 n), n = the bytes behind the slice (zlib's crc32_combine):
    // the last slice holds r bytes and every slice in front of it is followed by whole slices and those r bytes, so
    // x^(8 n_t) = x^(8 r) x^(8 S (L - 2 - t)): the second factor from the table, per lane; the first once, behind the XOR.
    const uint32_t L = (vlen + S - 1u) / S;                  // slices that hold bytes (uniform)
    const uint32_t r = vlen - (L - 1u) * S;
    uint32_t x = t + 1u < L ? crc_mul_lanes(crc, kCrcPow16[(L - 2u - t) * (S >> 4)]) : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x ^= (uint32_t)__shfl_xor((int)x, d, 64);
    if ((t & 63u) == 0u) s_part[t >> 6] = x;
    if (t + 1u == L) s_part[4] = crc;                         // the last slice's own CRC
    __syncthreads();
    if (t == 0u) {
        const uint32_t front = s_part[0] ^ s_part[1] ^ s_part[2] ^ s_part[3];
        const uint32_t got = (L > 1u ? crc_mul_lanes(front, kCrcPow8[r]) : 0u) ^ s_part[4];
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

// ---- where the records begin, for ANY block layout ------------------------------------------------------------------------
// htslib never lets a record straddle a BGZF block; htsjdk / Picard cut their blocks at 64 KiB of data wherever that falls.
// A chunk's blocks are inflated back to back, so a record is contiguous whatever the layout; what is sequential is the chain
// of length prefixes - which record start lies where depends on every record before it.  It is taken apart per block:
//   bam_entry_kernel   one wave per block GUESSES the block's first record start: the first offset whose 36 fixed bytes
//                      look like a record (reference ids inside the header's count, positions >= -1, a name, a body that
//                      holds name + CIGAR + sequence + qualities) and whose successor looks like one too - offset 0 in
//                      htslib's layout, found by the first lane;
//   bam_walk_kernel    one lane per block follows the length prefixes from the guess to the first start that lies in a later
//                      block - its EXIT, counted from the block's end - or to the record that the chunk's bytes do not
//                      finish (the chunk's TAIL: it is moved in front of the next chunk's first block);
//   bam_scan_kernel    makes the guesses exact: every block checks that its exit - carried over blocks that a long record
//                      covers whole, which must have guessed "no start" - IS the guess of the block it lands in.  The first
//                      block's entry is known (the end of the header, 0 behind a chunk that ended with a record, or the tail
//                      in the slot in front of it), so by induction every start of the chunk is a true start; any
//                      mismatch and the chunk is refused (the caller falls back to the host reader).  Then the exclusive
//                      scan of the record counts, and the chunk's summary.
// Descriptor 0 of a chunk is the slot of the previous chunk's tail (dst_len 0: none); the file's blocks follow from 1.
constexpr uint32_t kNoStart = 0xffffffffu;               // guess: no record begins in the block
constexpr uint32_t kExitBad = 0xffffffffu, kExitTail = 0xfffffffeu, kExitNone = 0xfffffffdu;   // exit: corrupt / chunk's tail / nothing walked

// do the bytes at p begin a record?  (`avail` of them are the chunk's, >= 36: what lies beyond is not looked at)
__device__ __forceinline__ bool bam_plausible(const uint8_t* p, unsigned long long avail, int32_t n_ref, uint32_t* size_out) {
    const uint32_t size = ld32u(p);
    const int32_t ref = (int32_t)ld32u(p + 4), pos = (int32_t)ld32u(p + 8);
    const uint32_t w3 = ld32u(p + 12), w4 = ld32u(p + 16);
    const int32_t l_seq = (int32_t)ld32u(p + 20), mref = (int32_t)ld32u(p + 24), mpos = (int32_t)ld32u(p + 28);
    const uint32_t l_name = w3 & 0xffu, n_cigar = w4 & 0xffffu;
    *size_out = size;
    if (size < 32u || size >= (1u << 28)) return false;
    if (ref < -1 || ref >= n_ref || mref < -1 || mref >= n_ref || pos < -1 || mpos < -1) return false;
    if (l_name == 0u || l_seq < 0) return false;
    const unsigned long long body = 32ull + l_name + 4ull * n_cigar + ((unsigned long long)l_seq + 1ull) / 2ull + (unsigned long long)l_seq;
    if (body > (unsigned long long)size) return false;
    // the name: printable, and closed by the NUL that l_name counts
    if (36ull + l_name <= avail) {
        if (p[36u + l_name - 1u] != 0u) return false;
        if (l_name > 1u && (p[36] < 0x21u || p[36] > 0x7eu)) return false;
    }
    return true;
}

__global__ __launch_bounds__(64) void bam_entry_kernel(const uint8_t* __restrict__ inflated, const BgzfBlock* __restrict__ blocks,
                                                       uint32_t n_blocks, unsigned long long chunk_end, int32_t n_ref,
                                                       uint32_t forced_block, uint32_t forced_entry, uint32_t mode, uint32_t* __restrict__ guess) {
    const uint32_t b = blockIdx.x;
    const int lane = threadIdx.x;
    if (b >= n_blocks) return;
    const uint32_t len = blocks[b].dst_len;
    const unsigned long long at0 = (unsigned long long)blocks[b].dst_off_lo | ((unsigned long long)blocks[b].dst_off_hi << 32);
    uint32_t found = kNoStart;
    if (b == 0u) found = len ? 0u : kNoStart;                // the tail of the chunk before begins with a record
    else if (mode & kWalkOverhang) found = kNoStart;        // (only the record in the tail slot is this chunk's)
    else if (b == forced_block) found = forced_entry < len ? forced_entry : kNoStart;
    else if (forced_block != 0xffffffffu && b < forced_block) found = kNoStart;   // (the bytes of a record of the part before)
    else {
        // an EMPTY forced block (an interior EOF marker on a chunk boundary) hands its entry on to the first block behind it
        // that holds bytes: that block's start is KNOWN too, not guessed - nobody in front of it could vouch for a guess
        bool inherits = forced_block != 0xffffffffu && len != 0u;
        for (uint32_t k = forced_block; inherits && k < b; ++k) inherits = blocks[k].dst_len == 0u;     // (stops at the first block with bytes)
        if (inherits) {
            if (lane == 0) guess[b] = forced_entry < len ? forced_entry : kNoStart;
            return;
        }
        for (uint32_t o0 = 0; o0 < len; o0 += 64u) {          // uniform
            const uint32_t o = o0 + (uint32_t)lane;
            bool ok = false;
            if (o < len && at0 + o + 36ull <= chunk_end) {
                uint32_t size = 0, size2 = 0;
                ok = bam_plausible(inflated + at0 + o, chunk_end - (at0 + o), n_ref, &size);
                const unsigned long long next = at0 + o + 4ull + size;
                if (ok && next + 36ull <= chunk_end) ok = bam_plausible(inflated + next, chunk_end - next, n_ref, &size2);
            }
            const unsigned long long m = __ballot(ok);
            if (m) {
                found = o0 + (uint32_t)__ffsll((long long)m) - 1u;
                break;
            }
        }
    }
    if (lane == 0) guess[b] = found;
}

// follow the length prefixes of block b from `entry` (the walk proper: one lane, a chain of dependent loads)
__device__ __forceinline__ void bam_walk_block(const uint8_t* __restrict__ inflated, const BgzfBlock* __restrict__ blocks, uint32_t b,
                                               unsigned long long chunk_end, const uint32_t* __restrict__ status, uint32_t entry,
                                               uint16_t* __restrict__ offs, uint32_t* count, uint32_t* exits, uint32_t* tail_at) {
    const uint32_t len = blocks[b].dst_len;
    const unsigned long long at0 = (unsigned long long)blocks[b].dst_off_lo | ((unsigned long long)blocks[b].dst_off_hi << 32);
    uint32_t cnt = 0, out = kExitNone;
    unsigned long long cur = entry;
    if (len != 0u && status[b] != kInfOk) out = kExitBad;
    else if (entry != kNoStart) {
        uint16_t* o = offs + (size_t)b * kBamBlockRecs;
        for (;;) {
            if (cur >= len) { out = (uint32_t)(cur - len); break; }       // the next start lies in a later block
            if (at0 + cur + 4ull > chunk_end) { out = kExitTail; break; }
            const uint32_t size = ld32u(inflated + at0 + cur);
            if (size < 32u || size >= (1u << 28) || cnt >= (uint32_t)kBamBlockRecs) { out = kExitBad; break; }
            if (at0 + cur + 4ull + size > chunk_end) { out = kExitTail; break; }
            o[cnt++] = (uint16_t)cur;
            cur += 4ull + size;
        }
    }
    count[b] = cnt;
    exits[b] = out;
    tail_at[b] = (uint32_t)cur;
}

__global__ __launch_bounds__(64) void bam_walk_kernel(const uint8_t* __restrict__ inflated, const BgzfBlock* __restrict__ blocks,
                                                      uint32_t n_blocks, unsigned long long chunk_end, const uint32_t* __restrict__ status,
                                                      const uint32_t* __restrict__ guess, uint16_t* __restrict__ offs,
                                                      uint32_t* __restrict__ count, uint32_t* __restrict__ exits,
                                                      uint32_t* __restrict__ tail_at) {
    const uint32_t b = blockIdx.x * 64u + threadIdx.x;
    if (b >= n_blocks) return;
    bam_walk_block(inflated, blocks, b, chunk_end, status, guess[b], offs, count, exits, tail_at);
}

// summary (12 words): [0] records, [1] 1 = every start verified, [2] first block that is not (verified: guesses replaced), [3] its inflate status, [4] bytes of the
// chunk's tail, [5] [6] where it begins in the chunk's buffer, [7] 1 = some block begins inside a record (not htslib's layout)
//
// A verdict is the smallest (block << 32 | what its predecessor says its entry is): kVerdictBad in the low word when there is
// nothing to repair (an inflate / CRC failure, a length that cannot be a record's on the true chain).
constexpr uint32_t kVerdictNoStart = 0x7fffffffu, kVerdictBad = 0xffffffffu;
constexpr unsigned long long kNoVerdict = ~0ull;
constexpr uint32_t kMaxRepairs = 4096;
// four waves: the workgroup has to find room on ONE compute unit next to the other chunks' inflate waves (sixteen waves waited
// 1-3 ms for that, on the path to the chunk's verdict)
constexpr uint32_t kScanThreads = 256;

__global__ __launch_bounds__(kScanThreads) void bam_scan_kernel(const uint8_t* __restrict__ inflated, const BgzfBlock* __restrict__ blocks,
                                                        uint16_t* __restrict__ offs, uint32_t* count, uint32_t* exits, uint32_t* guess,
                                                        uint32_t* tail_at, const uint32_t* __restrict__ status,
                                                        uint32_t n_blocks, unsigned long long chunk_end, uint32_t forced_block, uint32_t mode,
                                                        uint32_t* __restrict__ rec_base, uint32_t* __restrict__ summary) {
    __shared__ uint32_t s_w[kScanThreads / 64], s_tail, s_straddle;
    __shared__ unsigned long long s_bad;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t tail_b = 0xffffffffu, round = 0;
    unsigned long long bad = kNoVerdict;
    for (;; ++round) {
        if (t == 0) { s_bad = kNoVerdict; s_tail = 0xffffffffu; s_straddle = 0u; }
        __syncthreads();
        // ---- every walked block vouches for the block its exit lands in
        for (uint32_t b = (uint32_t)t; b < n_blocks; b += kScanThreads) {
            const uint32_t ex = exits[b], g = guess[b];
            const unsigned long long here = (unsigned long long)b << 32;
            if (blocks[b].dst_len != 0u && status[b] != kInfOk) { atomicMin(&s_bad, here | kVerdictBad); continue; }
            if (mode & kWalkOverhang) {                      // only the tail slot's record (unfinished still: the tail again)
                if (b == 0u && (ex == kExitBad || ex == kExitNone)) atomicMin(&s_bad, here | kVerdictBad);
                if (b == 0u && ex == kExitTail) atomicMin(&s_tail, 0u);
                continue;
            }
            if (b > 0u && b != forced_block && g != kNoStart && g != 0u) s_straddle = 1u;
            if (ex == kExitNone) continue;                   // guessed "no start": a block before it vouches for that (or fails to)
            if (ex == kExitBad) { atomicMin(&s_bad, here | kVerdictBad); continue; }
            if (ex == kExitTail) { atomicMin(&s_tail, b); continue; }
            uint32_t carry = ex, nb = b + 1u;
            while (nb < n_blocks && carry >= blocks[nb].dst_len) {       // blocks the record covers whole
                if (guess[nb] != kNoStart && blocks[nb].dst_len != 0u) break;
                carry -= blocks[nb].dst_len;
                ++nb;
            }
            if (nb < n_blocks) {
                if (guess[nb] != carry)
                    atomicMin(&s_bad, ((unsigned long long)nb << 32) | (carry >= blocks[nb].dst_len ? kVerdictNoStart : carry));
            } else if (carry != 0u) {
                atomicMin(&s_bad, here | kVerdictBad);       // (a record that ends beyond the chunk would have been its tail)
            }
        }
        if (t == 0 && !(mode & (kWalkFirstGuessed | kWalkOverhang))) {
            // the first block that holds bytes must hold a start: block 0 (a tail) or the forced block (in front of it lie
            // the bytes of a record of the part before) - nobody vouches for a block in front of the first walked one
            uint32_t f = 0;
            while (f < n_blocks && exits[f] == kExitNone && (blocks[f].dst_len == 0u || (forced_block != 0xffffffffu && f < forced_block))) ++f;
            if (f < n_blocks && exits[f] == kExitNone && f != 0u) atomicMin(&s_bad, ((unsigned long long)f << 32) | kVerdictBad);
        }
        __syncthreads();
        bad = s_bad;
        tail_b = s_tail;
        const uint32_t bad_b = (uint32_t)(bad >> 32), want = (uint32_t)bad;
        // settled: nothing wrong in front of the chunk's tail (behind it lie the unfinished record's bytes, whatever they look like)
        if (bad == kNoVerdict || (tail_b != 0xffffffffu && bad_b > tail_b)) break;
        if (want == kVerdictBad || round >= kMaxRepairs) break;
        // ---- a guess was wrong - a stretch of bytes that looks like a record and is none, or a record that looks like none.
        // Every block before it has been vouched for, so what its predecessor says IS its entry: walk it again from there.
        if (t == 0) {
            const uint32_t entry = want == kVerdictNoStart ? kNoStart : want;
            guess[bad_b] = entry;
            bam_walk_block(inflated, blocks, bad_b, chunk_end, status, entry, offs, count, exits, tail_at);
            __threadfence_block();
        }
        __syncthreads();
    }
    const uint32_t bad_b = (uint32_t)(bad >> 32);
    const uint32_t per = (n_blocks + kScanThreads - 1u) / kScanThreads;
    const uint32_t b0 = (uint32_t)t * per, b1 = b0 + per < n_blocks ? b0 + per : n_blocks;
    uint32_t sum = 0;
    for (uint32_t b = b0; b < b1; ++b) {
        if (b > tail_b) count[b] = 0u;                       // behind the tail's block: the unfinished record's bytes
        sum += count[b];
    }
    uint32_t x = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)x, d, 64);
        if (lane >= d) x += v;
    }
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    uint32_t off = x - sum, total = 0;
#pragma unroll
    for (int q = 0; q < (int)(kScanThreads / 64); ++q) {
        if (q < wave) off += s_w[q];
        total += s_w[q];
    }
    for (uint32_t b = b0; b < b1; ++b) {
        rec_base[b] = off;
        off += count[b];
    }
    if (t == 0) {
        const bool ok = bad == kNoVerdict || (tail_b != 0xffffffffu && bad_b > tail_b);
        summary[0] = total;
        summary[1] = ok ? 1u : 0u;
        summary[2] = ok ? round : bad_b;
        summary[3] = ok ? 0u : (bad_b < n_blocks ? status[bad_b] : 0u);
        unsigned long long at = chunk_end;
        if (tail_b != 0xffffffffu)
            at = ((unsigned long long)blocks[tail_b].dst_off_lo | ((unsigned long long)blocks[tail_b].dst_off_hi << 32)) + tail_at[tail_b];
        summary[4] = (uint32_t)(chunk_end - at);
        summary[5] = (uint32_t)at;
        summary[6] = (uint32_t)(at >> 32);
        summary[7] = s_straddle;
        // [8] [9]: where the chunk's first record begins in the chunk's buffer (~0: no record begins in it); for the blocks
        // behind a part's end: how many bytes of the tail slot's record lie in them (i.e. in the next part)
        unsigned long long first = ~0ull;
        if (mode & kWalkOverhang) {
            first = exits[0];
        } else {
            for (uint32_t f = 0; f < n_blocks; ++f)
                if (exits[f] != kExitNone) {
                    first = ((unsigned long long)blocks[f].dst_off_lo | ((unsigned long long)blocks[f].dst_off_hi << 32)) + guess[f];
                    break;
                }
        }
        summary[8] = (uint32_t)first;
        summary[9] = (uint32_t)(first >> 32);
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
                        uint32_t* status, uint32_t* symbols) {
    if (n_blocks == 0) return BESST_OK;
    // symbols (bgzf_inflate_symbol_places() of four bytes): the second form of the kernel; nullptr or BESST_INFLATE=1: the first
    const char* form = getenv("BESST_INFLATE");              // (read per call: the tests run both forms in one process)
    const bool first_form = form && form[0] == '1' && form[1] == 0;
    if (symbols && !first_form)
        hipLaunchKernelGGL(bgzf_inflate2_kernel, dim3(n_blocks), dim3(64), 0, s, src, blocks, n_blocks, dst, symbols, status);
    else
        hipLaunchKernelGGL(bgzf_inflate_kernel, dim3(n_blocks), dim3(64), 0, s, src, blocks, n_blocks, dst, status);
    {   // the CRC kernel's power tables, once per device (on this stream, in front of the first CRC launch; a second
        // thread's first launch on the same device waits for the one that fills them)
        static std::mutex mu;
        static bool filled[64];
        int device = 0;
        BESST_HIP_TRY(hipGetDevice(&device));
        std::lock_guard<std::mutex> g(mu);
        if (device >= 0 && device < 64 && !filled[device]) {
            hipLaunchKernelGGL(crc_pow_kernel, dim3((kCrcPow16N + kCrcPow8N + 255) / 256), dim3(256), 0, s);
            BESST_HIP_TRY(hipStreamSynchronize(s));
            filled[device] = true;
        }
    }
    hipLaunchKernelGGL(bgzf_crc_kernel, dim3(n_blocks), dim3(256), 0, s, dst, blocks, n_blocks, status);
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

int launch_bam_walk_scan(hipStream_t s, const uint8_t* inflated, const BgzfBlock* blocks, uint32_t n_blocks, uint64_t chunk_end,
                         int32_t n_ref, uint32_t forced_block, uint32_t forced_entry, uint32_t mode, const uint32_t* status, uint16_t* offs,
                         uint32_t* count, uint32_t* exits, uint32_t* rec_base, uint32_t* guess, uint32_t* tail_at, uint32_t* summary) {
    if (n_blocks == 0) return BESST_OK;
    hipLaunchKernelGGL(bam_entry_kernel, dim3(n_blocks), dim3(64), 0, s, inflated, blocks, n_blocks, (unsigned long long)chunk_end,
                       n_ref, forced_block, forced_entry, mode, guess);
    hipLaunchKernelGGL(bam_walk_kernel, dim3((n_blocks + 63u) / 64u), dim3(64), 0, s, inflated, blocks, n_blocks,
                       (unsigned long long)chunk_end, status, guess, offs, count, exits, tail_at);
    hipLaunchKernelGGL(bam_scan_kernel, dim3(1), dim3(kScanThreads), 0, s, inflated, blocks, offs, count, exits, guess, tail_at, status, n_blocks,
                       (unsigned long long)chunk_end, forced_block, mode, rec_base, summary);
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
