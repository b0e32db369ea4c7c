// Large tuple streams (mate-pair libraries: tens of millions of link tuples per GPU): edge table from the ordered
// tuple stream with ONE read of the keys per radix pass and no per-tile count tables.
//
//   1. os_hist_kernel      one read of the raw keys: digit histograms of EVERY pass (per-block rows in a small table);
//                          also clears the look-back descriptors of this call
//   2. os_offsets_kernel   column sums of the table + exclusive scan -> where every digit of every pass starts
//   3. os_scatter_kernel   one launch per digit: stable partition of 8192-key tiles by chained scan (decoupled
//                          look-back): a tile publishes its per-digit counts, then the running totals, as 8-byte
//                          {tag, value} granules; tiles are numbered by an arrival ticket, so every predecessor of a
//                          tile is running or done and the look-back cannot wait on a workgroup that is not resident
//   3b. packed keys of four digits or more: only the TOP 16 bits take stream-wide passes (two); that leaves up to
//                          65 536 buckets in stream order, each finished on chip - os_bucket_start_kernel (binary
//                          search of the bucket borders), os_bucket_wave_kernel (one wave per bucket, a handful of
//                          distinct keys: sorts the bucket in registers AND reduces it - observations to their
//                          sorted places, the bucket's edge rows staged; C3 0.23 ms for what three more stream
//                          passes and os_reduce_kernel did in 1.1), os_bucket_sort_kernel (the listed rest: LDS
//                          digit passes, global-memory passes for a hub, then the same outputs),
//                          os_bucket_rows_kernel (rows numbered and moved to the table);
//                          steps 4 and 5 are not run on this path
//   4. os_reduce_kernel    segmented reduction of the sorted stream into edge rows: head counts chained the same way,
//                          every row's nr_links / sum obs / sum obs^2 WRITTEN once by the tile that holds its head
//                          (segmented scan over the tile's threads, no atomics, no zero-initialised accumulators);
//   5. os_fixup_kernel     a row that runs on into later tiles gets their leading partial sums added (one thread
//                          per tile)
//
// Semantics: CreateGraph.py:842-862 - an edge is the unordered node pair, its observations stay in BAM order (the
// sort is stable, so the first tuple of a row is its first occurrence) and nr_links / obs / obs_sq are exact integers.
//
// Inter-workgroup traffic follows the granule form of the CDNA4 hand-off rules: every shared word is one naturally
// aligned 8-byte {tag, value} written by ONE relaxed agent-scope store and polled with relaxed agent-scope loads;
// tag = (phase << 2) | status with phase counted inside the call (never 0), all granules zeroed by the first kernel
// of the call.  Spins are bounded: a tile that gives up sets the workspace's error word instead of hanging the GPU.
#include "common.h"

namespace besst {

namespace {

typedef __attribute__((address_space(1))) unsigned long long gu64;
#define BESST_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// Tile geometry of the partition passes (build knobs).  Measured on full C3 (42.7 M tuples, 5 passes): larger tiles
// win - 256 x 16 keys 1.55 ms, 512 x 12 1.38, 512 x 16 1.12, 1024 x 16 1.02 for the five passes - as long as the
// keys per thread stay at 16 (24 / 32 keys per thread: 1.41 / 1.69 ms, the ranking rounds of a wave are serial).
#ifndef BESST_OS_THREADS
#define BESST_OS_THREADS 1024
#endif
#ifndef BESST_OS_ITEMS
#define BESST_OS_ITEMS 16
#endif
constexpr int kOsThreads = BESST_OS_THREADS;
constexpr int kOsWaves = kOsThreads / 64;
constexpr int kOsItems = BESST_OS_ITEMS;
constexpr int kOsTile = kOsThreads * kOsItems;          // 8192 keys per scatter tile
#ifndef BESST_OS_HIST_THREADS
#define BESST_OS_HIST_THREADS 1024
#endif
#ifndef BESST_OS_HIST_BLOCKS
#define BESST_OS_HIST_BLOCKS 512
#endif
#ifndef BESST_OS_HIST_COPIES
#define BESST_OS_HIST_COPIES 8
#endif
constexpr int kOsHistThreads = BESST_OS_HIST_THREADS;
constexpr int kOsHistTile = kOsHistThreads * 16;
constexpr int kOsHistBlocks = BESST_OS_HIST_BLOCKS;
constexpr int kOsHistCopies = BESST_OS_HIST_COPIES;
#ifndef BESST_OS_BITS
#define BESST_OS_BITS 8
#endif
constexpr int kOsBits = BESST_OS_BITS;
constexpr int kOsMaxPasses = (64 + kOsBits - 1) / kOsBits;
constexpr int kOsRedThreads = 256;
constexpr int kOsRedItems = 16;
constexpr int kOsRedTile = kOsRedThreads * kOsRedItems; // 4096 tuples per reduce tile
constexpr uint32_t kStAgg = 1u, kStPrefix = 2u;
constexpr uint32_t kSpinLimit = 1u << 22;               // ~ seconds; a healthy look-back waits microseconds
// BESST_OS_SPIN_LIMIT (tests): a limit of 0 makes the first unanswered poll give up, so that the error path - the
// workspace's error word, BESST_ROWS_SORT_FAILED in *n_rows - can be exercised on a healthy GPU
static uint32_t os_spin_limit() {
    static const uint32_t v = [] { const char* e = getenv("BESST_OS_SPIN_LIMIT"); return e ? (uint32_t)strtoul(e, nullptr, 10) : kSpinLimit; }();
    return v;
}

__device__ __forceinline__ void granule_store(unsigned long long* p, uint32_t tag, uint32_t value) {
    __hip_atomic_store((gu64*)p, ((unsigned long long)tag << 32) | value, BESST_RLX_AGENT);
}
__device__ __forceinline__ unsigned long long granule_load(const unsigned long long* p) {
    return __hip_atomic_load((gu64*)p, BESST_RLX_AGENT);
}
__device__ __forceinline__ uint32_t os_nblocks(uint32_t n, uint32_t tile) { return (n + tile - 1) / tile; }

// ---------------------------------------------------------------------------------------------------
// 1. histograms of every pass, one read of the keys
// ---------------------------------------------------------------------------------------------------
template <int BITS>
__global__ __launch_bounds__(kOsHistThreads) void os_hist_kernel(const uint64_t* __restrict__ keys,
                                                                 const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                                 int passes, int shift0, uint64_t key_base,
                                                                 uint32_t* __restrict__ table,
                                                                 unsigned long long* __restrict__ granules,
                                                                 size_t granule_words) {
    constexpr int RADIX = 1 << BITS;
    // 64 consecutive tuples of a (tid,pos)-ordered stream touch one or two contigs, so every digit takes a handful of
    // values inside a wave and per-lane LDS atomics pile up on a few counters.  kOsHistCopies copies of the counters,
    // picked by lane and shifted by one bank each, spread those lanes over different banks (C3: 0.21 -> 0.12 ms;
    // peeling off the distinct values with ballots instead cost the same 0.2 ms in scalar-chain latency).
    __shared__ uint32_t s_hist[kOsHistCopies * (kOsMaxPasses * RADIX + 1)];
    const int t = threadIdx.x, lane = t & 63;
    const int stride = passes * RADIX + 1;
    for (int d = t; d < kOsHistCopies * stride; d += kOsHistThreads) s_hist[d] = 0;
    // every granule of this call starts as "nothing published" (plain stores: first polled after a kernel boundary)
    {
        const size_t gsz = (size_t)gridDim.x * kOsHistThreads;
        for (size_t i = (size_t)blockIdx.x * kOsHistThreads + t; i < granule_words; i += gsz) granules[i] = 0ull;
    }
    uint32_t n = *n_ptr;
    n = n < cap ? n : cap;
    uint32_t* mine = s_hist + (lane & (kOsHistCopies - 1)) * stride;
    __syncthreads();
    for (uint32_t base = blockIdx.x * (uint32_t)kOsHistTile; base < n; base += gridDim.x * (uint32_t)kOsHistTile) {
        if (base + kOsHistTile <= n) {
            uint64_t k[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) k[r] = keys[base + r * kOsHistThreads + t] - key_base;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                for (int p = 0; p < passes; ++p) {
                    const uint32_t d = (uint32_t)(k[r] >> (shift0 + p * BITS)) & (uint32_t)(RADIX - 1);
                    const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
                    if (__all(d == f)) {                    // the whole wave on one value (high digits): one add
                        if (lane == 0) atomicAdd(&mine[p * RADIX + f], 64u);
                    } else {
                        atomicAdd(&mine[p * RADIX + d], 1u);
                    }
                }
            }
        } else {
            for (int r = 0; r < 16; ++r) {
                const uint32_t i = base + r * kOsHistThreads + t;
                if (i < n) {
                    const uint64_t k = keys[i] - key_base;
                    for (int p = 0; p < passes; ++p)
                        atomicAdd(&mine[p * RADIX + ((uint32_t)(k >> (shift0 + p * BITS)) & (uint32_t)(RADIX - 1))], 1u);
                }
            }
        }
    }
    __syncthreads();
    uint32_t* row = table + (size_t)blockIdx.x * passes * RADIX;
    for (int d = t; d < passes * RADIX; d += kOsHistThreads) {
        uint32_t v = 0;
#pragma unroll
        for (int c = 0; c < kOsHistCopies; ++c) v += s_hist[c * stride + d];
        row[d] = v;
    }
}

// 2. block p: start of every digit of pass p; clears the arrival tickets
template <int BITS>
__global__ __launch_bounds__(256) void os_offsets_kernel(const uint32_t* __restrict__ table, int hist_blocks, int passes,
                                                         uint32_t* __restrict__ digit_base, uint32_t* __restrict__ tickets) {
    constexpr int RADIX = 1 << BITS;
    constexpr int DPT = RADIX / 256;
    static_assert(DPT >= 1, "at least 8-bit digits");
    __shared__ uint32_t s_w[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int p = blockIdx.x;
    uint32_t tot[DPT];
#pragma unroll
    for (int q = 0; q < DPT; ++q) tot[q] = 0;
    for (int g0 = 0; g0 < hist_blocks; g0 += 16) {          // 16 rows in flight per round trip
        uint32_t v[16][DPT];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int g = g0 + u < hist_blocks ? g0 + u : hist_blocks - 1;
            const uint32_t* row = table + ((size_t)g * passes + p) * RADIX + (size_t)t * DPT;
#pragma unroll
            for (int q = 0; q < DPT; ++q) v[u][q] = row[q];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int q = 0; q < DPT; ++q) tot[q] += g0 + u < hist_blocks ? v[u][q] : 0u;
    }
    uint32_t run = 0;
#pragma unroll
    for (int q = 0; q < DPT; ++q) run += tot[q];
    uint32_t x = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(x, d, 64);
        if (lane >= d) x += o;
    }
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    uint32_t start = x - run;
    for (int w = 0; w < wave; ++w) start += s_w[w];
#pragma unroll
    for (int q = 0; q < DPT; ++q) {
        digit_base[(size_t)p * RADIX + t * DPT + q] = start;
        start += tot[q];
    }
    if (t == 0) tickets[p] = 0;
    if (p == 0 && t == 1) tickets[kOsMaxPasses] = 0;        // the reduce stage's ticket
}

// ---------------------------------------------------------------------------------------------------
// 3. one radix pass: stable partition by chained scan
// ---------------------------------------------------------------------------------------------------
// kFirst: the input is the raw key stream (the stream index is the position); kPacked: the words written (and read
// by later passes) are key << packed_bits | stream index; otherwise keys and indexes travel as two arrays (keys
// wider than 64 - index bits).
//
// Ranking (a key's position among the keys of its wave with the same digit, in lane order: the partition is stable)
// is where the instructions of this kernel go - with a plain BITS-ballot match-any per key the passes were bound by
// VALU issue, not by memory (~190 instructions per key round, 160 of the 223 us of a C3 pass).  The digits of 64
// consecutive tuples of a (tid,pos)-ordered stream take a handful of values, so the distinct values are peeled off
// one at a time - lowest lane still unranked -> its digit (readlane) -> one compare = the group's lane mask (a
// scalar) -> rank inside the group (mbcnt) and group size (scalar popcount) - at ~6 vector + ~6 scalar instructions
// per distinct value; only what is left after kPeel values (digits of an unordered stream) takes the match-any.
#ifndef BESST_OS_MIN_WAVES
#define BESST_OS_MIN_WAVES 4
#endif
#ifndef BESST_OS_PEEL
#define BESST_OS_PEEL 8
#endif
constexpr int kPeel = BESST_OS_PEEL;

constexpr int kSegWin = 255;            // block offsets a tile keeps in LDS: a tile that spans more blocks searches in memory

template <int BITS, bool kFirst, bool kPacked, bool kSeg = false>
__global__ __launch_bounds__(kOsThreads, BESST_OS_MIN_WAVES) void os_scatter_kernel(
    const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ idx_in, const uint32_t* __restrict__ n_ptr,
    uint32_t cap, int shift, int pass, int packed_bits, uint64_t key_base, const uint32_t* __restrict__ digit_base,
    unsigned long long* __restrict__ desc, uint32_t* __restrict__ ticket, uint64_t* __restrict__ keys_out,
    uint32_t* __restrict__ idx_out, uint32_t* __restrict__ err, uint32_t spin_limit, SegSource seg = SegSource{},
    const uint32_t* __restrict__ tile_first = nullptr, int seg_win = kSegWin) {
    constexpr int RADIX = 1 << BITS;
    // the raw key stream (first pass) and unpacked keys carry key_base; packed words hold key - key_base already
    const uint64_t sub = (kFirst || !kPacked) ? key_base : 0ull;
    constexpr int kIdxRegs = kPacked ? 1 : kOsItems;
    __shared__ uint32_t s_whist[kOsWaves][RADIX];
    __shared__ uint32_t s_base[RADIX];
    __shared__ uint32_t s_tile;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_tile = atomicAdd(ticket, 1u);
    uint32_t n = *n_ptr;
    n = n < cap ? n : cap;
#pragma unroll
    for (int w = 0; w < kOsWaves; ++w)
        for (int d = t; d < RADIX; d += kOsThreads) s_whist[w][d] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    if (tile >= os_nblocks(n, kOsTile)) return;             // uniform
    const uint32_t wbase = tile * (uint32_t)kOsTile + (uint32_t)wave * (kOsItems * 64);
    uint64_t key[kOsItems];
    uint32_t idx[kIdxRegs];
    if constexpr (kSeg) {
        // The stream still lies in the record loop's block segments (SegSource): dense position i belongs to the last
        // block whose first-tuple offset is <= i (empty blocks share their successor's offset), slot i - offset, one
        // further when the stitch dropped that block's head tuple at or before it.  The offsets of the blocks a tile
        // touches sit in LDS (the tile's first block was looked up by os_seg_tiles_kernel); the payload goes to its
        // dense place on the way - what compact_kernel would have done with a pass of its own.
        __shared__ uint32_t s_woff[kSegWin + 1], s_wskip[kSegWin + 1];
        const uint32_t b0 = tile_first[tile];
        const uint32_t n_all = *n_ptr;
        for (int q = t; q <= kSegWin; q += kOsThreads) {
            const uint32_t b = b0 + (uint32_t)q;
            // (seg_win < kSegWin, a test knob: the window ends early, so small streams take the search in memory too)
            const uint32_t bb = q <= seg_win ? b : b0 + (uint32_t)seg_win;
            s_woff[q] = bb < seg.nblocks ? seg.offsets[bb] : n_all;
            s_wskip[q] = b < seg.nblocks ? seg.skip[b] : 0xffffffffu;
        }
        __syncthreads();
        const uint32_t tile_end = (tile + 1u) * (uint32_t)kOsTile < n ? (tile + 1u) * (uint32_t)kOsTile : n;
        const bool in_window = s_woff[kSegWin] >= tile_end;  // uniform: every position of the tile lies in a windowed block
        int wq = 0;                                          // window index of the block of the lane's last position
        if (in_window) {                                     // the wave's first position: one search; 64 further on is
            int lo = 0, hi = kSegWin;                        // nearly always the same block or the next one
            const uint32_t i = wbase + lane;
#pragma unroll
            for (int step = 0; step < 8; ++step) {
                const int mid = (lo + hi) >> 1;
                if (s_woff[mid] <= i) lo = mid; else hi = mid;
            }
            wq = lo;
        }
#pragma unroll
        for (int r = 0; r < kOsItems; ++r) {
            const uint32_t i = wbase + r * 64 + lane;
            key[r] = ~0ull;
            if (i < n) {
                uint32_t b, off, skip;
                if (in_window) {
                    while (s_woff[wq + 1] <= i) ++wq;        // (s_woff[kSegWin] > i: ends inside the window)
                    b = b0 + (uint32_t)wq; off = s_woff[wq]; skip = s_wskip[wq];
                } else {
                    uint32_t lo = b0, hi = seg.nblocks;      // largest block in [b0, nblocks) with offsets[block] <= i
                    while (hi - lo > 1u) {
                        const uint32_t mid = lo + ((hi - lo) >> 1);
                        if (seg.offsets[mid] <= i) lo = mid; else hi = mid;
                    }
                    b = lo; off = seg.offsets[lo]; skip = seg.skip[lo];
                }
                uint32_t j = i - off;
                j += j >= skip ? 1u : 0u;
                const size_t src = (size_t)b * seg.tile + j;
                key[r] = seg.seg_keys[src];
                seg.payload_out[i] = seg.seg_payload[src];
            }
        }
    } else {
#pragma unroll
    for (int r = 0; r < kOsItems; ++r) {
        const uint32_t i = wbase + r * 64 + lane;
        key[r] = i < n ? keys_in[i] : ~0ull;
        if (!kPacked) idx[r] = kFirst ? i : (i < n ? idx_in[i] : 0u);
    }
    }
    uint32_t* my_hist = &s_whist[wave][0];
    uint32_t dig_rank[kOsItems];                            // digit | rank inside the wave's share << BITS
#pragma unroll
    for (int r = 0; r < kOsItems; ++r) {
        const uint32_t i = wbase + r * 64 + lane;
        const bool valid = i < n;
        const uint32_t d = (uint32_t)((key[r] - sub) >> shift) & (uint32_t)(RADIX - 1);
        uint32_t info = 0;                                   // rank inside my group | group size << 8
        bool todo = valid;
        unsigned long long rest = __ballot(todo);
        for (int it = 0; it < kPeel && rest; ++it) {         // uniform loop
            const int src = __ffsll((long long)rest) - 1;
            const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)d, src);
            const bool hit = todo && d == f;
            const unsigned long long m = __ballot(hit);
            const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            const uint32_t c8 = (uint32_t)__popcll(m) << 8;
            if (hit) { info = rk | c8; todo = false; }
            rest &= ~m;
        }
        if (rest) {                                          // uniform: many distinct digits in this wave
            unsigned long long pm = rest;
#pragma unroll
            for (int bit = 0; bit < BITS; ++bit) {
                const bool one = (d >> bit) & 1u;
                const unsigned long long bal = __ballot(one);
                pm &= one ? bal : ~bal;
            }
            if (todo) {
                const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                info = rk | ((uint32_t)__popcll(pm) << 8);
            }
        }
        // every lane reads its digit's running count, then the first lane of each group adds the group's size: LDS
        // operations of a wave execute in order, so the reads see the counts of the earlier rounds only
        uint32_t pre = 0;
        if (valid) {
            volatile uint32_t* slot = my_hist + d;
            pre = *slot;
            if ((info & 0xffu) == 0u) *slot = pre + (info >> 8);
        }
        dig_rank[r] = d | ((pre + (info & 0xffu)) << BITS);
    }
    __syncthreads();
    // per digit: the tile's count, published; the count of all earlier tiles, looked back; the waves' shares
    const uint32_t tag_agg = ((uint32_t)(pass + 1) << 2) | kStAgg, tag_pre = ((uint32_t)(pass + 1) << 2) | kStPrefix;
    for (int d = t; d < RADIX; d += kOsThreads) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < kOsWaves; ++w) {
            const uint32_t c = s_whist[w][d];
            s_whist[w][d] = run;
            run += c;
        }
        unsigned long long* mine = desc + (size_t)tile * RADIX + d;
        uint32_t excl = 0;
        if (tile == 0) {
            granule_store(mine, tag_pre, run);
        } else {
            granule_store(mine, tag_agg, run);
            uint32_t k = tile - 1, spins = 0;
            for (;;) {
                const unsigned long long g = granule_load(desc + (size_t)k * RADIX + d);
                const uint32_t tg = (uint32_t)(g >> 32);
                if (tg == tag_pre) { excl += (uint32_t)g; break; }
                if (tg == tag_agg) { excl += (uint32_t)g; --k; continue; }   // tile 0 always publishes a prefix
                if (++spins > spin_limit) { *err = 1u; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            granule_store(mine, tag_pre, excl + run);
        }
        s_base[d] = digit_base[d] + excl;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kOsItems; ++r) {
        const uint32_t i = wbase + r * 64 + lane;
        if (i < n) {
            const uint32_t d = dig_rank[r] & (uint32_t)(RADIX - 1);
            const uint32_t dst = s_base[d] + s_whist[wave][d] + (dig_rank[r] >> BITS);
            if (dst >= cap) continue;                        // (only if the histograms counted more tuples than fit: the
                                                             // caller's capacity check reports that pass as overflowed)
            if (kPacked) {
                keys_out[dst] = kFirst ? (((key[r] - key_base) << packed_bits) | i) : key[r];
            } else {
                keys_out[dst] = key[r];
                idx_out[dst] = idx[r];
            }
        }
    }
}

// first block of every scatter tile of a segmented stream: the last block whose first-tuple offset is <= the tile's
// first position
__global__ __launch_bounds__(256) void os_seg_tiles_kernel(SegSource seg, const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                           uint32_t* __restrict__ tile_first) {
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n = *n_ptr;
    n = n < cap ? n : cap;
    if (tile >= os_nblocks(n, kOsTile)) return;
    const uint32_t p0 = tile * (uint32_t)kOsTile;
    uint32_t lo = 0, hi = seg.nblocks;
    while (hi - lo > 1u) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (seg.offsets[mid] <= p0) lo = mid; else hi = mid;
    }
    tile_first[tile] = lo;
}

// ---------------------------------------------------------------------------------------------------
// 3b. the low key bits, bucket by bucket in LDS
// ---------------------------------------------------------------------------------------------------
// Keys of four or more digits: only the TOP two digits are sorted with stream-wide passes (LSD order on those two
// digits).  That leaves the stream partitioned into up to 65 536 buckets of equal top-16 bits - a few hundred to a
// few thousand tuples, each bucket still in stream order - and every bucket is finished by one workgroup: stable
// 7-bit LSD passes over the remaining low bits with the words in registers and ONE LDS buffer in between (the same
// peel-off ranking as the stream-wide passes), no look-back, no HBM traffic between the passes.  On C3 this replaces
// three stream-wide passes (16 B/tuple each plus their chained scans) by one read and one write of the words.
// A bucket that does not fit the LDS buffer (a hub scaffold end with thousands of links) is sorted by its workgroup
// in global memory, tile after tile, with the other ping-pong buffer as scratch: exact, not fast.
#ifndef BESST_BK_BITS
#define BESST_BK_BITS 7
#endif
#ifndef BESST_BK_PEEL
#define BESST_BK_PEEL 2
#endif
constexpr int kBkThreads = 256;
constexpr int kBkWaves = 4;
constexpr int kBkItems = 16;
constexpr int kBkCap = kBkThreads * kBkItems;          // 4096 words in LDS
constexpr int kBkBits = BESST_BK_BITS;
constexpr int kBkRadix = 1 << kBkBits;
constexpr int kBkPeel = BESST_BK_PEEL;
constexpr int kTopBuckets = 1 << 16;
static_assert(kBkBits >= 6 && kBkBits <= 10, "bucket digit width");

// start[b] = first position of the sorted-by-top-digits stream whose bucket number is >= b (binary search)
__global__ __launch_bounds__(256) void os_bucket_start_kernel(const uint64_t* __restrict__ words,
                                                              const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                              int bucket_shift, uint32_t* __restrict__ start) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > (uint32_t)kTopBuckets) return;
    uint32_t n = *n_ptr;
    n = n < cap ? n : cap;
    uint32_t lo = 0, hi = n;                                 // first index with (word >> shift) >= b
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((words[mid] >> bucket_shift) < (uint64_t)b) lo = mid + 1; else hi = mid;
    }
    start[b] = lo;
}

// rank of a word among the words of equal digit in its round of 64 (see os_scatter_kernel): rank | group size << 8
__device__ __forceinline__ uint32_t bk_wave_rank(uint32_t d, bool valid) {
    uint32_t info = 0;
    bool todo = valid;
    unsigned long long rest = __ballot(todo);
    for (int it = 0; it < kBkPeel && rest; ++it) {
        const int src = __ffsll((long long)rest) - 1;
        const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)d, src);
        const bool hit = todo && d == f;
        const unsigned long long m = __ballot(hit);
        const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (hit) { info = rk | ((uint32_t)__popcll(m) << 8); todo = false; }
        rest &= ~m;
    }
    if (rest) {
        unsigned long long pm = rest;
#pragma unroll
        for (int bit = 0; bit < kBkBits; ++bit) {
            const bool one = (d >> bit) & 1u;
            const unsigned long long bal = __ballot(one);
            pm &= one ? bal : ~bal;
        }
        if (todo) {
            const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
            info = rk | ((uint32_t)__popcll(pm) << 8);
        }
    }
    return info;
}

// vals[0 .. kBkRadix): totals in, exclusive prefix sums out (whole workgroup; ends with a barrier)
__device__ __forceinline__ void bk_excl_scan(uint32_t* vals, uint32_t* s_wtot, int t) {
    constexpr int per = kBkRadix > kBkThreads ? kBkRadix / kBkThreads : 1;
    constexpr int active = kBkRadix / per;
    const int lane = t & 63, wave = t >> 6;
    uint32_t loc[per];
    uint32_t tot = 0;
#pragma unroll
    for (int j = 0; j < per; ++j) {
        loc[j] = t < active ? vals[t * per + j] : 0u;
        tot += loc[j];
    }
    uint32_t x = tot;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
        const uint32_t o = __shfl_up(x, dd, 64);
        if (lane >= dd) x += o;
    }
    if (lane == 63) s_wtot[wave] = x;
    __syncthreads();
    uint32_t off = x - tot;
    for (int q = 0; q < wave; ++q) off += s_wtot[q];
    if (t < active) {
#pragma unroll
        for (int j = 0; j < per; ++j) {
            vals[t * per + j] = off;
            off += loc[j];
        }
    }
    __syncthreads();
}

// The usual bucket holds a handful of distinct keys (a few scaffold ends, each with a few partners and MANY links per
// partner: C3 has a median of 815 tuples and 4 keys per bucket, at most 15).  ONE WAVE per bucket, no LDS: find the
// smallest key not yet placed (a min over the lane's words, then over the wave), match it against every word of the
// bucket (one compare + mbcnt per round of 64) - the words of that key go, in stream order, right behind the ones
// placed so far - and repeat.  A bucket of more than kBwKeys distinct keys, or whose eight smallest keys cover less than
// a sixth of it, goes on the list of os_bucket_wave_lds_kernel (digit passes, still one wave per bucket), one of more
// than kBwCap words on the list of os_bucket_sort_kernel (a workgroup per bucket).
#ifndef BESST_BW_ITEMS
#define BESST_BW_ITEMS 24
#endif
#ifndef BESST_BW_DPP
#define BESST_BW_DPP 1
#endif
constexpr int kBwItems = BESST_BW_ITEMS;
constexpr int kBwCap = 64 * kBwItems;
#ifndef BESST_BW_KEYS
#define BESST_BW_KEYS 48
#endif
// a key costs ~280 wave instructions against ~4400 for the digit passes of a bucket, but those run at three waves per
// SIMD and a fixed ~20 us per bucket: measured, the peel-off is the faster one up to ~50 keys (30 M random tuples, 30
// keys per bucket: 0.81 ms against 1.3)
constexpr int kBwKeys = BESST_BW_KEYS;
static_assert(kBwItems % 4 == 0 && kBwItems <= 32, "rounds are dispatched in steps of four");

__device__ __forceinline__ uint32_t bw_wave_min(uint32_t v) {
#if BESST_BW_DPP
    // row_shr 1, 2, 4, 8 inside the rows of 16, then row_bcast 15 and 31: lane 63 holds the minimum
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x118, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x142, 0xa, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
#else
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, d, 64));
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#endif
}

// What the bucket kernels leave behind: every tuple's observations at its sorted place, and the bucket's edge rows
// STAGED at [bucket start + k] - a row's number needs the row counts of all buckets before it, so
// os_bucket_rows_kernel numbers and moves them afterwards.  (The sorted words themselves are
// nobody's input any more: the wave kernel does not write them.)
struct StagedRow {
    uint64_t key;
    unsigned long long sum, sq;
    uint32_t n, first, off, mask;
};

struct BwOut {
    const uint64_t* payload;
    const uint32_t* first_map;
    uint64_t key_base;
    int32_t* obs_lo;
    int32_t* obs_hi;
    struct StagedRow* staged;           // one 40-byte record per row: a row is read and written as a whole
    uint32_t* bucket_rows;
};

__device__ __forceinline__ unsigned long long bw_wave_sum64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// one bucket of at most 64 R words: sorted AND reduced; false: too many distinct keys, nothing written
template <int R>
__device__ __forceinline__ bool bw_bucket(const uint64_t* __restrict__ words, uint32_t s0, uint32_t n, int low_shift,
                                          int low_bits, int lane, const BwOut& o) {
    const uint32_t lowmask = (1u << low_bits) - 1u;          // low_bits <= 31 (launcher): a key is below 2^31
    const uint64_t idx_mask = (1ull << low_shift) - 1ull;
    uint32_t kk[R], idx[R], pos[R];
    uint64_t top = 0;                                        // the bits above the low key bits: the same for the whole bucket
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t p = r * 64 + lane;
        const bool valid = p < n;
        const uint64_t w = valid ? words[s0 + p] : ~0ull;
        kk[r] = valid ? ((uint32_t)(w >> low_shift) & lowmask) : 0xffffffffu;
        idx[r] = (uint32_t)(w & idx_mask);
        if (r == 0) top = w >> (low_shift + low_bits);       // lane 0 of round 0 is always valid
        pos[r] = 0;
    }
    top = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(top >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)top);
    uint32_t lo = 0, covered = 0;                            // keys below lo are placed; covered: how many words that is
    uint32_t my_key = 0, my_start = 0, my_cnt = 0, my_first = 0;   // lane k: the k-th smallest key and its row
    int K = 0;
    for (;; ++K) {
        // smallest key >= lo: kk - lo wraps around for the placed ones and is >= 2^31 for the padding
        uint32_t acc = 0xffffffffu;
#pragma unroll
        for (int r = 0; r < R; ++r) acc = min(acc, kk[r] - lo);
        const uint32_t dmin = bw_wave_min(acc);
        if (dmin >= 0x80000000u) break;                      // nothing left to place
        if (K == kBwKeys || (K == 8 && covered * 6u < n)) return false;
        const uint32_t f = lo + dmin;
        uint32_t run = covered, first = 0;
        bool found = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool hit = kk[r] == f;
            const unsigned long long m = __ballot(hit);
            const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, run));
            if (hit) pos[r] = rk;
            if (!found && m) {                               // the row's first tuple in stream order
                found = true;
                first = (uint32_t)__builtin_amdgcn_readlane((int)idx[r], __ffsll((long long)m) - 1);
            }
            run += (uint32_t)__popcll(m);
        }
        if (lane == K) { my_key = f; my_start = covered; my_cnt = run - covered; my_first = first; }
        covered = run;
        lo = f + 1u;
        if (covered == n) { ++K; break; }
    }
    // ---- observations to their sorted places; obs1 + obs2 per word stays in registers for the row sums
    uint32_t ov[R];
#pragma unroll
    for (int r0 = 0; r0 < R; r0 += 4) {
        uint64_t pl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pl[q] = (r0 + q) * 64 + lane < n ? o.payload[idx[r0 + q]] : 0ull;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + q;
            const uint32_t l = (uint32_t)pl[q], h = (uint32_t)(pl[q] >> 32) & 0x3fffffffu;
            if (r * 64 + lane < n) {
                o.obs_lo[s0 + pos[r]] = (int32_t)l;
                o.obs_hi[s0 + pos[r]] = (int32_t)h;
            }
            ov[r] = l + h;                                   // 0 for the padding
        }
    }
    // ---- row sums, key by key
    unsigned long long my_s = 0, my_s2 = 0;
    for (int k = 0; k < K; ++k) {
        const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)my_key, k);
        unsigned long long sum = 0, sq = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t m = kk[r] == f ? ov[r] : 0u;
            sum += m;
            sq += (unsigned long long)m * m;
        }
        sum = bw_wave_sum64(sum);
        sq = bw_wave_sum64(sq);
        if (lane == k) { my_s = sum; my_s2 = sq; }
    }
    if (lane < K) {
        const uint32_t row = s0 + (uint32_t)lane;
        StagedRow sr;
        sr.key = ((top << low_bits) | (uint64_t)my_key) + o.key_base;
        sr.sum = my_s;
        sr.sq = my_s2;
        sr.n = my_cnt;
        sr.first = o.first_map ? o.first_map[my_first] : my_first;
        sr.off = s0 + my_start;
        sr.mask = (uint32_t)(o.payload[my_first] >> 62);
        o.staged[row] = sr;
    }
    if (lane == 0) o.bucket_rows[0] = (uint32_t)K;
    return true;
}

__global__ __launch_bounds__(kBkThreads) void os_bucket_wave_kernel(const uint64_t* __restrict__ words,
                                                                    const uint32_t* __restrict__ start,
                                                                    int low_shift, int low_bits,
                                                                    uint32_t* __restrict__ bucket_class, BwOut o) {
    const int lane = threadIdx.x & 63;
    // wave-uniform values, and the compiler is told so: the round and key loops are scalar control flow
    const uint32_t bucket = blockIdx.x * kBkWaves + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)start[bucket]);
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)start[bucket + 1]) - s0;
    o.bucket_rows += bucket;
    if (n == 0) {
        if (lane == 0) { o.bucket_rows[0] = 0; bucket_class[bucket] = 0u; }
        return;
    }
    bool placed = false;
    if (n <= (uint32_t)kBwCap) {
        switch ((n + 255u) >> 8) {                           // rounds of 64, in steps of four
            case 1: placed = bw_bucket<4>(words, s0, n, low_shift, low_bits, lane, o); break;
            case 2: placed = bw_bucket<8>(words, s0, n, low_shift, low_bits, lane, o); break;
            case 3: placed = bw_bucket<12>(words, s0, n, low_shift, low_bits, lane, o); break;
            case 4: placed = bw_bucket<16>(words, s0, n, low_shift, low_bits, lane, o); break;
#if BESST_BW_ITEMS >= 20
            case 5: placed = bw_bucket<20>(words, s0, n, low_shift, low_bits, lane, o); break;
#endif
#if BESST_BW_ITEMS >= 24
            case 6: placed = bw_bucket<24>(words, s0, n, low_shift, low_bits, lane, o); break;
#endif
#if BESST_BW_ITEMS >= 28
            case 7: placed = bw_bucket<28>(words, s0, n, low_shift, low_bits, lane, o); break;
#endif
#if BESST_BW_ITEMS >= 32
            case 8: placed = bw_bucket<32>(words, s0, n, low_shift, low_bits, lane, o); break;
#endif
            default: break;
        }
    }
    // who finishes the bucket (a plain store per bucket: 65 536 appends to one list counter took 0.7 ms when every
    // bucket had to be handed on): 0 done, 1 many keys -> os_bucket_wave_lds_kernel, 2 many words -> os_bucket_sort_kernel
    if (lane == 0) bucket_class[bucket] = placed ? 0u : (n <= (uint32_t)kBwCap ? 1u : 2u);
}

// The buckets of the wave kernel's size class that hold MANY distinct keys (a library whose edges carry one or two links
// each, or a stream of unrelated keys): still one wave per bucket, four buckets per workgroup, but digit passes - words
// in registers, 7-bit stable passes through the wave's own LDS buffer with the peel-off ranking of the stream passes, no
// workgroup barrier (a wave's LDS operations complete in order) - then the reduction of the sorted words: observations
// gathered and written coalesced, head flags by comparing neighbours, row sums by a segmented scan per round of 64 with
// the open tail of a round carried into the next one as wave-uniform values.
constexpr int kBlPer = kBkRadix / 64;
static_assert(kBkRadix % 64 == 0, "digits per lane of the wave-level scan");

__global__ __launch_bounds__(kBkThreads) void os_bucket_wave_lds_kernel(const uint64_t* __restrict__ words,
                                                                        const uint32_t* __restrict__ start,
                                                                        int low_shift, int low_bits,
                                                                        const uint32_t* __restrict__ bucket_class, BwOut o) {
    __shared__ uint64_t s_words_all[kBkWaves][kBwCap];
    __shared__ uint32_t s_hist_all[kBkWaves][kBkRadix];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint64_t* s_words = s_words_all[wave];
    uint32_t* s_hist = s_hist_all[wave];
    const uint64_t idx_mask = (1ull << low_shift) - 1ull;
    const unsigned long long le_mask = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    const int n_pass = (low_bits + kBkBits - 1) / kBkBits;
    // the wave's buckets: wave_id, wave_id + waves, ... - their classes are fetched together (one round trip), then only
    // the marked ones are visited
    const uint32_t n_waves = gridDim.x * kBkWaves, wave_id = blockIdx.x * kBkWaves + (uint32_t)wave;
    const uint32_t mine = wave_id + (uint32_t)lane * n_waves;
    unsigned long long todo = __ballot(mine < (uint32_t)kTopBuckets && bucket_class[mine] == 1u);
    static_assert(kTopBuckets <= 64 * 2048 * kBkWaves, "one class per lane covers the wave's buckets (grid of 2048)");
    while (todo) {
        const int j = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const uint32_t bucket = wave_id + (uint32_t)j * n_waves;
        const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)start[bucket]);
        const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)start[bucket + 1]) - s0;
        const int rounds = (int)((n + 63) >> 6);
        uint64_t w[kBwItems];
#pragma unroll
        for (int r = 0; r < kBwItems; ++r) {
            const uint32_t p = r * 64 + lane;
            w[r] = (r < rounds && p < n) ? words[s0 + p] : ~0ull;
        }
        for (int pass = 0; pass < n_pass; ++pass) {
            const int shift = low_shift + pass * kBkBits;
            const int bits = low_bits - pass * kBkBits < kBkBits ? low_bits - pass * kBkBits : kBkBits;
            const uint32_t dmask = (1u << bits) - 1u;
#pragma unroll
            for (int j = 0; j < kBlPer; ++j) s_hist[j * 64 + lane] = 0;
            __builtin_amdgcn_wave_barrier();
            uint32_t dig_rank[kBwItems];
#pragma unroll
            for (int r = 0; r < kBwItems; ++r) {
                dig_rank[r] = 0;
                if (r < rounds) {
                    const uint32_t p = r * 64 + lane;
                    const bool valid = p < n;
                    const uint32_t d = (uint32_t)(w[r] >> shift) & dmask;
                    const uint32_t info = bk_wave_rank(d, valid);
                    uint32_t pre = 0;
                    if (valid) {
                        volatile uint32_t* slot = &s_hist[d];
                        pre = *slot;
                        if ((info & 0xffu) == 0u) *slot = pre + (info >> 8);
                    }
                    dig_rank[r] = d | ((pre + (info & 0xffu)) << kBkBits);
                    __builtin_amdgcn_wave_barrier();
                }
            }
            {   // digit counts -> exclusive starts, lane l owns digits [l * per, l * per + per)
                uint32_t c[kBlPer];
                uint32_t tot = 0;
#pragma unroll
                for (int j = 0; j < kBlPer; ++j) {
                    c[j] = s_hist[lane * kBlPer + j];
                    tot += c[j];
                }
                uint32_t x = tot;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) {
                    const uint32_t v = (uint32_t)__shfl_up((int)x, dd, 64);
                    if (lane >= dd) x += v;
                }
                uint32_t off = x - tot;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < kBlPer; ++j) {
                    s_hist[lane * kBlPer + j] = off;
                    off += c[j];
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < kBwItems; ++r) {
                if (r < rounds) {
                    const uint32_t p = r * 64 + lane;
                    if (p < n) s_words[s_hist[dig_rank[r] & (kBkRadix - 1)] + (dig_rank[r] >> kBkBits)] = w[r];
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < kBwItems; ++r) {
                if (r < rounds) {
                    const uint32_t p = r * 64 + lane;
                    w[r] = p < n ? s_words[p] : ~0ull;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // ---- the sorted bucket -> observations (all gathers of the bucket in flight together) ...
        uint32_t ov[kBwItems], mk[kBwItems];
#pragma unroll
        for (int r = 0; r < kBwItems; ++r) {
            ov[r] = 0; mk[r] = 0;
            if (r < rounds) {
                const uint32_t p = r * 64 + lane;
                if (p < n) {
                    const uint64_t pl = o.payload[(uint32_t)(w[r] & idx_mask)];
                    const uint32_t l = (uint32_t)pl, h = (uint32_t)(pl >> 32) & 0x3fffffffu;
                    o.obs_lo[s0 + p] = (int32_t)l;
                    o.obs_hi[s0 + p] = (int32_t)h;
                    ov[r] = l + h;
                    mk[r] = (uint32_t)(pl >> 62);
                }
            }
        }
        // ... and staged rows (carry_*: the row open at the end of the round before)
        uint32_t rows_before = 0, carry_c = 0;
        unsigned long long carry_s = 0, carry_q = 0;
        uint64_t prev_last = ~0ull;                          // key of the last word of the round before
#pragma unroll
        for (int r = 0; r < kBwItems; ++r) {
            if (r < rounds) {
                const uint32_t p = r * 64 + lane;
                const bool valid = p < n;
                const uint64_t key = w[r] >> low_shift;
                uint64_t before = __shfl_up(key, 1, 64);
                if (lane == 0) before = prev_last;
                const bool head = valid && (p == 0 || key != before);
                const unsigned long long hm = __ballot(head);
                const uint32_t row = s0 + rows_before + (uint32_t)__popcll(hm & le_mask) - 1u;   // never below s0: word 0 is a head
                if (r > 0 && (hm & 1ull) && lane == 0) {     // the carried row ended with the round before
                    StagedRow* sr = o.staged + (s0 + rows_before - 1u);
                    sr->n = carry_c;
                    sr->sum = carry_s;
                    sr->sq = carry_q;
                }
                if (head) {
                    const uint32_t src = (uint32_t)(w[r] & idx_mask);
                    StagedRow* sr = o.staged + row;
                    sr->key = key + o.key_base;
                    sr->first = o.first_map ? o.first_map[src] : src;
                    sr->off = s0 + p;
                    sr->mask = mk[r];
                }
                uint32_t c = valid ? 1u : 0u;
                unsigned long long sm = (unsigned long long)ov[r], sq = sm * sm;
                bool fl = head || lane == 0;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t oc = (uint32_t)__shfl_up((int)c, d, 64);
                    const unsigned long long os = __shfl_up(sm, d, 64), oq = __shfl_up(sq, d, 64);
                    const int of = __shfl_up((int)fl, d, 64);
                    if (lane >= d && !fl) { c += oc; sm += os; sq += oq; fl = of != 0; }
                }
                if ((hm & le_mask) == 0ull) { c += carry_c; sm += carry_s; sq += carry_q; }   // still the carried row
                const int tail_lane = (int)(n - (uint32_t)r * 64u > 64u ? 63u : n - (uint32_t)r * 64u - 1u);
                const bool seg_end = valid && (lane == tail_lane || ((hm >> (lane + 1)) & 1ull));
                const bool bucket_end = p + 1u == n;
                if (seg_end && (lane != tail_lane || bucket_end)) {
                    StagedRow* sr = o.staged + row;
                    sr->n = c;
                    sr->sum = sm;
                    sr->sq = sq;
                }
                // the open tail travels on
                carry_c = (uint32_t)__builtin_amdgcn_readlane((int)c, tail_lane);
                carry_s = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(sm >> 32), tail_lane) << 32) |
                          (uint32_t)__builtin_amdgcn_readlane((int)sm, tail_lane);
                carry_q = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(sq >> 32), tail_lane) << 32) |
                          (uint32_t)__builtin_amdgcn_readlane((int)sq, tail_lane);
                prev_last = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(key >> 32), tail_lane) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)key, tail_lane);
                rows_before += (uint32_t)__popcll(hm);
            }
        }
        if (lane == 0) o.bucket_rows[bucket] = rows_before;
        __builtin_amdgcn_wave_barrier();
    }
}

__device__ void bk_sort_bucket(uint64_t* words, uint64_t* scratch, uint32_t s0, uint32_t n, int low_shift, int low_bits) {
    __shared__ uint64_t s_words[kBkCap];
    __shared__ uint32_t s_whist[kBkWaves][kBkRadix];
    __shared__ uint32_t s_base[kBkRadix];
    __shared__ uint32_t s_wtot[kBkWaves];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n_pass = (low_bits + kBkBits - 1) / kBkBits;
    if (n <= (uint32_t)kBkCap) {
        // ---- the whole bucket in registers: word p = wave * (rounds * 64) + r * 64 + lane (stream order = position)
        const int rounds = (int)((n + kBkThreads - 1) / kBkThreads);
        const uint32_t wbase = (uint32_t)wave * (uint32_t)(rounds * 64);
        uint64_t w[kBkItems];
#pragma unroll
        for (int r = 0; r < kBkItems; ++r) {
            const uint32_t p = wbase + r * 64 + lane;
            w[r] = (r < rounds && p < n) ? words[s0 + p] : ~0ull;
        }
        for (int pass = 0; pass < n_pass; ++pass) {
            const int shift = low_shift + pass * kBkBits;
            const int bits = low_bits - pass * kBkBits < kBkBits ? low_bits - pass * kBkBits : kBkBits;
            const uint32_t dmask = (1u << bits) - 1u;
            for (int d = t; d < kBkWaves * kBkRadix; d += kBkThreads) (&s_whist[0][0])[d] = 0;
            __syncthreads();
            uint32_t dig_rank[kBkItems];
#pragma unroll
            for (int r = 0; r < kBkItems; ++r) {
                dig_rank[r] = 0;
                if (r < rounds) {                            // uniform
                    const uint32_t p = wbase + r * 64 + lane;
                    const bool valid = p < n;
                    const uint32_t d = (uint32_t)(w[r] >> shift) & dmask;
                    const uint32_t info = bk_wave_rank(d, valid);
                    uint32_t pre = 0;
                    if (valid) {
                        volatile uint32_t* slot = &s_whist[wave][d];
                        pre = *slot;
                        if ((info & 0xffu) == 0u) *slot = pre + (info >> 8);
                    }
                    dig_rank[r] = d | ((pre + (info & 0xffu)) << kBkBits);
                }
            }
            __syncthreads();
            for (int d = t; d < kBkRadix; d += kBkThreads) {
                uint32_t tot = 0;
#pragma unroll
                for (int q = 0; q < kBkWaves; ++q) tot += s_whist[q][d];
                s_base[d] = tot;
            }
            __syncthreads();
            bk_excl_scan(s_base, s_wtot, t);
            for (int d = t; d < kBkRadix; d += kBkThreads) {     // every wave's share of a digit, waves in order
                uint32_t run = s_base[d];
#pragma unroll
                for (int q = 0; q < kBkWaves; ++q) {
                    const uint32_t c = s_whist[q][d];
                    s_whist[q][d] = run;
                    run += c;
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < kBkItems; ++r) {
                if (r < rounds) {
                    const uint32_t p = wbase + r * 64 + lane;
                    if (p < n) s_words[s_whist[wave][dig_rank[r] & (kBkRadix - 1)] + (dig_rank[r] >> kBkBits)] = w[r];
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < kBkItems; ++r) {
                if (r < rounds) {
                    const uint32_t p = wbase + r * 64 + lane;
                    w[r] = p < n ? s_words[p] : ~0ull;
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < kBkItems; ++r) {
            if (r < rounds) {
                const uint32_t p = wbase + r * 64 + lane;
                if (p < n) words[s0 + p] = w[r];
            }
        }
        return;
    }
    // ---- a bucket larger than the LDS buffer: the same passes over tiles of 4096 words in global memory
    uint64_t* src = words + s0;
    uint64_t* dst = scratch + s0;
    for (int pass = 0; pass < n_pass; ++pass) {
        const int shift = low_shift + pass * kBkBits;
        const int bits = low_bits - pass * kBkBits < kBkBits ? low_bits - pass * kBkBits : kBkBits;
        const uint32_t dmask = (1u << bits) - 1u;
        for (int d = t; d < kBkRadix; d += kBkThreads) s_base[d] = 0;
        __syncthreads();
        for (uint32_t i = t; i < n; i += kBkThreads)
            atomicAdd(&s_base[(uint32_t)(__hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> shift) & dmask], 1u);
        __syncthreads();
        bk_excl_scan(s_base, s_wtot, t);
        for (uint32_t t0 = 0; t0 < n; t0 += kBkCap) {
            const uint32_t tn = n - t0 < (uint32_t)kBkCap ? n - t0 : (uint32_t)kBkCap;
            for (int d = t; d < kBkWaves * kBkRadix; d += kBkThreads) (&s_whist[0][0])[d] = 0;
            __syncthreads();
            const uint32_t wbase = (uint32_t)wave * (kBkItems * 64);
            uint64_t w[kBkItems];
            uint32_t dig_rank[kBkItems];
#pragma unroll
            for (int r = 0; r < kBkItems; ++r) {
                const uint32_t p = wbase + r * 64 + lane;
                const bool valid = p < tn;
                w[r] = valid ? __hip_atomic_load(&src[t0 + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : ~0ull;
                const uint32_t d = (uint32_t)(w[r] >> shift) & dmask;
                const uint32_t info = bk_wave_rank(d, valid);
                uint32_t pre = 0;
                if (valid) {
                    volatile uint32_t* slot = &s_whist[wave][d];
                    pre = *slot;
                    if ((info & 0xffu) == 0u) *slot = pre + (info >> 8);
                }
                dig_rank[r] = d | ((pre + (info & 0xffu)) << kBkBits);
            }
            __syncthreads();
            for (int d = t; d < kBkRadix; d += kBkThreads) {     // this tile's shares: digit base, then the waves in order
                uint32_t run = s_base[d];
#pragma unroll
                for (int q = 0; q < kBkWaves; ++q) {
                    const uint32_t c = s_whist[q][d];
                    s_whist[q][d] = run;
                    run += c;
                }
                s_base[d] = run;                             // where the next tile continues
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < kBkItems; ++r) {
                const uint32_t p = wbase + r * 64 + lane;
                if (p < tn) dst[s_whist[wave][dig_rank[r] & (kBkRadix - 1)] + (dig_rank[r] >> kBkBits)] = w[r];
            }
            __syncthreads();
        }
        // the next pass of THIS workgroup reads what it has just written
        __threadfence();
        __syncthreads();
        uint64_t* tmp = src; src = dst; dst = tmp;
    }
    if (src != words + s0) {                                 // an odd number of passes left the result in the scratch
        for (uint32_t i = t; i < n; i += kBkThreads) words[s0 + i] = src[i];
    }
}

// A sorted bucket of any size -> observations + staged rows, by one workgroup: 256 words at a time, run sums by a
// segmented scan inside each wave, one atomic triple per (run, wave) into the staged row (zeroed first).
__device__ void bk_reduce_bucket(const uint64_t* words, uint32_t s0, uint32_t n, int packed_bits, const BwOut& o) {
    __shared__ uint32_t s_cnt[kBkWaves];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint64_t idx_mask = (1ull << packed_bits) - 1ull;
    // rows of the bucket
    uint32_t heads = 0;
    for (uint32_t i = t; i < n; i += kBkThreads) {
        const uint64_t key = words[s0 + i] >> packed_bits;
        heads += (i == 0 || (words[s0 + i - 1] >> packed_bits) != key) ? 1u : 0u;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) heads += (uint32_t)__shfl_xor((int)heads, d, 64);
    if (lane == 0) s_cnt[wave] = heads;
    __syncthreads();
    const uint32_t rows = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    for (uint32_t k = t; k < rows; k += kBkThreads) {
        o.staged[s0 + k].n = 0;
        o.staged[s0 + k].sum = 0ull;
        o.staged[s0 + k].sq = 0ull;
    }
    __threadfence();
    __syncthreads();
    uint32_t rows_before = 0;
    for (uint32_t c0 = 0; c0 < n; c0 += kBkThreads) {
        const uint32_t i = c0 + t;
        const bool valid = i < n;
        const uint64_t w = valid ? words[s0 + i] : 0ull;
        const uint64_t key = w >> packed_bits;
        const bool head = valid && (i == 0 || (words[s0 + i - 1] >> packed_bits) != key);
        const uint32_t src = (uint32_t)(w & idx_mask);
        const uint64_t pl = valid ? o.payload[src] : 0ull;
        const uint32_t l = (uint32_t)pl, h = (uint32_t)(pl >> 32) & 0x3fffffffu;
        if (valid) {
            o.obs_lo[s0 + i] = (int32_t)l;
            o.obs_hi[s0 + i] = (int32_t)h;
        }
        const unsigned long long hm = __ballot(head);
        if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(hm);
        __syncthreads();
        uint32_t incl = (uint32_t)__popcll(hm & ((2ull << lane) - 1ull));
        uint32_t chunk_heads = 0;
#pragma unroll
        for (int q = 0; q < kBkWaves; ++q) {
            if (q < wave) incl += s_cnt[q];
            chunk_heads += s_cnt[q];
        }
        const uint32_t row = s0 + rows_before + incl - 1u;   // the bucket's first word is a head: never below s0
        if (head) {
            o.staged[row].key = key + o.key_base;
            o.staged[row].off = s0 + i;
            o.staged[row].first = o.first_map ? o.first_map[src] : src;
            o.staged[row].mask = (uint32_t)(pl >> 62);
        }
        // run sums inside the wave
        uint32_t c = valid ? 1u : 0u;
        unsigned long long sm = (unsigned long long)(l + h), sq = sm * sm;
        bool fl = head || lane == 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t oc = (uint32_t)__shfl_up((int)c, d, 64);
            const unsigned long long os = __shfl_up(sm, d, 64), oq = __shfl_up(sq, d, 64);
            const int of = __shfl_up((int)fl, d, 64);
            if (lane >= d && !fl) { c += oc; sm += os; sq += oq; fl = of != 0; }
        }
        const bool next_head = lane == 63 || ((hm >> (lane + 1)) & 1ull) || i + 1 >= n;
        if (valid && next_head) {
            atomicAdd(&o.staged[row].n, c);
            atomicAdd(&o.staged[row].sum, sm);
            atomicAdd(&o.staged[row].sq, sq);
        }
        rows_before += chunk_heads;
        __syncthreads();
    }
    if (t == 0) o.bucket_rows[0] = rows;
}

// the buckets os_bucket_wave_kernel has put on the list, one workgroup each: digit passes, then the rows
__global__ __launch_bounds__(kBkThreads) void os_bucket_sort_kernel(uint64_t* words, uint64_t* scratch,
                                                                    const uint32_t* __restrict__ start,
                                                                    int low_shift, int low_bits,
                                                                    const uint32_t* __restrict__ bucket_class, BwOut o) {
    // the workgroup's buckets: blockIdx, blockIdx + grid, ... - classes fetched together, only the marked ones visited
    __shared__ unsigned long long s_todo;
    {
        const uint32_t mine = blockIdx.x + (threadIdx.x & 63u) * gridDim.x;
        const unsigned long long m = __ballot(mine < (uint32_t)kTopBuckets && bucket_class[mine] == 2u);
        if (threadIdx.x == 0) s_todo = m;
    }
    static_assert(kTopBuckets <= 64 * 1024, "one class per lane of the first wave covers the workgroup's buckets (grid of 1024)");
    __syncthreads();
    unsigned long long todo = s_todo;
    while (todo) {
        const int j = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const uint32_t b = blockIdx.x + (uint32_t)j * gridDim.x;
        const uint32_t s0 = start[b];
        const uint32_t n = start[b + 1] - s0;
        if (n > 1) bk_sort_bucket(words, scratch, s0, n, low_shift, low_bits);
        __threadfence();
        __syncthreads();
        BwOut ob = o;
        ob.bucket_rows += b;
        bk_reduce_bucket(words, s0, n, low_shift, ob);
        __syncthreads();
    }
}

// staged rows -> the edge table.  Workgroup g owns buckets [1024 g, 1024 g + 1024): it sums the row counts of all
// buckets before them (coalesced, at most 256 KB from L2), scans its own, and moves its rows one thread per row - the
// row's bucket by binary search in the scanned counts (LDS) - so the table is written in order whatever the buckets'
// sizes.  The last workgroup writes the table's row count.
constexpr int kRowsThreads = 1024;

__global__ __launch_bounds__(kRowsThreads) void os_bucket_rows_kernel(const uint32_t* __restrict__ start, BwOut o,
                                                                      const uint32_t* __restrict__ err,
                                                                      uint32_t* __restrict__ n_rows,
                                                                      uint64_t* __restrict__ row_key, uint32_t* __restrict__ row_mask,
                                                                      uint32_t* __restrict__ row_n,
                                                                      unsigned long long* __restrict__ row_sum,
                                                                      unsigned long long* __restrict__ row_sum_sq,
                                                                      uint32_t* __restrict__ row_first,
                                                                      uint32_t* __restrict__ row_offset) {
    __shared__ uint32_t s_excl[kRowsThreads + 1];
    __shared__ uint32_t s_w[kRowsThreads / 64], s_b[kRowsThreads / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t b0 = blockIdx.x * kRowsThreads;
    uint32_t before = 0;
    for (uint32_t i = t; i < b0; i += kRowsThreads) before += o.bucket_rows[i];
    const uint32_t mine = o.bucket_rows[b0 + t];
    uint32_t x = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)x, d, 64);
        if (lane >= d) x += v;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) before += (uint32_t)__shfl_xor((int)before, d, 64);
    if (lane == 63) s_w[wave] = x;
    if (lane == 0) s_b[wave] = before;
    __syncthreads();
    uint32_t off = x - mine, base = 0, total = 0;
#pragma unroll
    for (int q = 0; q < kRowsThreads / 64; ++q) {
        if (q < wave) off += s_w[q];
        total += s_w[q];
        base += s_b[q];
    }
    s_excl[t] = off;
    if (t == 0) s_excl[kRowsThreads] = total;
    // (a stream pass whose look-back gave up has left a wrong partition behind: the caller must not trust the table)
    if (blockIdx.x == gridDim.x - 1 && t == 0) *n_rows = *err ? BESST_ROWS_SORT_FAILED : base + total;
    __syncthreads();
    for (uint32_t i = t; i < total; i += kRowsThreads) {
        int lo = 0, hi = kRowsThreads;                       // last bucket whose exclusive count is <= i
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_excl[mid] <= i) lo = mid; else hi = mid;
        }
        const uint32_t src = start[b0 + lo] + (i - s_excl[lo]);
        const uint32_t dst = base + i;
        const StagedRow sr = o.staged[src];
        row_key[dst] = sr.key;
        row_mask[dst] = sr.mask;
        row_n[dst] = sr.n;
        row_sum[dst] = sr.sum;
        row_sum_sq[dst] = sr.sq;
        row_first[dst] = sr.first;
        row_offset[dst] = sr.off;
    }
}

// ---------------------------------------------------------------------------------------------------
// 4. segmented reduction into edge rows
// ---------------------------------------------------------------------------------------------------
struct Part {           // partial sums of one run of tuples
    uint32_t n;
    unsigned long long s, s2;
};
__device__ __forceinline__ Part part_zero() { return Part{0u, 0ull, 0ull}; }
__device__ __forceinline__ void part_add(Part& a, const Part& b) { a.n += b.n; a.s += b.s; a.s2 += b.s2; }
__device__ __forceinline__ Part part_shfl_up(const Part& a, int d) {
    Part o;
    o.n = __shfl_up(a.n, d, 64);
    o.s = __shfl_up(a.s, d, 64);
    o.s2 = __shfl_up(a.s2, d, 64);
    return o;
}

// Two phases per 4096-tuple tile.  Phase 1 is a streaming pass in lane-contiguous layout: sorted words in (coalesced),
// observations gathered through the stream index and written out (coalesced), per tuple only obs1 + obs2 and a head
// flag are kept, in LDS.  Phase 2 re-reads them thread-contiguous (16 consecutive tuples per thread, padded rows:
// conflict free), so that the row sums are sequential adds plus ONE segmented scan over the tile's threads; the few
// tuples that start a row fetch their word again for the row's key / first index / graph mask.
// (Loading the words thread-contiguous straight from memory - 128 bytes per lane - thrashed the 32 KB L1 and ran the
// stage at 0.71 ms on C3; this form takes 0.40.)
__device__ __forceinline__ int os_pad(int i) { return i + (i >> 4); }

__global__ __launch_bounds__(kOsRedThreads) void os_reduce_kernel(
    const uint64_t* __restrict__ words, const uint32_t* __restrict__ idx, const uint64_t* __restrict__ payload,
    const uint32_t* __restrict__ n_ptr, uint32_t cap, int packed_bits, uint64_t key_base,
    unsigned long long* __restrict__ rdesc, uint32_t* __restrict__ ticket, uint32_t* __restrict__ tile_base, uint32_t* __restrict__ lead_n,
    unsigned long long* __restrict__ lead_s, unsigned long long* __restrict__ lead_s2, uint32_t* __restrict__ n_rows,
    uint64_t* __restrict__ row_key, uint32_t* __restrict__ row_mask, uint32_t* __restrict__ row_n,
    unsigned long long* __restrict__ row_sum, unsigned long long* __restrict__ row_sum_sq,
    uint32_t* __restrict__ row_first, uint32_t* __restrict__ row_offset, int32_t* __restrict__ obs_lo,
    int32_t* __restrict__ obs_hi, const uint32_t* __restrict__ first_map, uint32_t* __restrict__ err, uint32_t spin_limit) {
    __shared__ uint32_t s_o[kOsRedTile + kOsRedTile / 16];          // obs1 + obs2 per tuple, padded rows of 16
    __shared__ unsigned long long s_heads[kOsRedTile / 64];         // head flags, one bit per tuple
    __shared__ uint32_t s_tile, s_base;
    __shared__ int s_wcnt[4];
    __shared__ uint32_t s_pn[4], s_pf[4];
    __shared__ unsigned long long s_ps[4], s_ps2[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_tile = atomicAdd(ticket, 1u);
    uint32_t n = *n_ptr;
    n = n < cap ? n : cap;
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t ntiles = os_nblocks(n, kOsRedTile);
    if (n == 0) {
        if (tile == 0 && t == 0) *n_rows = 0;
        return;
    }
    if (tile >= ntiles) return;
    const uint32_t tbase = tile * (uint32_t)kOsRedTile;
    const uint64_t idx_mask = packed_bits ? ((1ull << packed_bits) - 1ull) : 0ull;
    // ---- phase 1, two halves of 8 rounds (all loads of a half in flight together)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint64_t w[8];
        uint32_t src[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t i = tbase + (uint32_t)(h * 8 + q) * kOsRedThreads + t;
            w[q] = i < n ? words[i] : 0ull;
            src[q] = packed_bits ? (uint32_t)(w[q] & idx_mask) : (i < n ? idx[i] : 0u);
        }
        uint64_t pl[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t i = tbase + (uint32_t)(h * 8 + q) * kOsRedThreads + t;
            pl[q] = i < n ? payload[src[q]] : 0ull;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = h * 8 + q;
            const uint32_t i = tbase + (uint32_t)r * kOsRedThreads + t;
            uint64_t wp = __shfl_up(w[q], 1, 64);
            if (lane == 0) wp = (i > 0 && i < n) ? words[i - 1] : ~w[q];
            const bool head = i < n && (i == 0 || (w[q] >> packed_bits) != (wp >> packed_bits));
            const unsigned long long hm = __ballot(head);
            if (lane == 0) s_heads[r * 4 + wave] = hm;
            const uint32_t lo = (uint32_t)pl[q], hi = (uint32_t)(pl[q] >> 32) & 0x3fffffffu;
            if (i < n) {
                obs_lo[i] = (int32_t)lo;
                obs_hi[i] = (int32_t)hi;
            }
            s_o[os_pad(r * kOsRedThreads + t)] = i < n ? lo + hi : 0u;
        }
    }
    __syncthreads();
    // ---- phase 2: thread t owns tuples [16 t, 16 t + 16) of the tile
    const uint32_t i0 = tbase + (uint32_t)t * kOsRedItems;
    const uint32_t headbits = (uint32_t)(s_heads[t >> 2] >> ((t & 3) * 16)) & 0xffffu;
    const int cnt = __popc(headbits);
    int x = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(x, d, 64);
        if (lane >= d) x += o;
    }
    if (lane == 63) s_wcnt[wave] = x;
    __syncthreads();
    int pre = x - cnt;
    for (int w = 0; w < wave; ++w) pre += s_wcnt[w];
    const uint32_t tile_heads = (uint32_t)(s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3]);
    // rows before this tile: chained scan of the head counts, wave 0, 64 predecessors per round
    if (wave == 0) {
        const uint32_t tag_agg = (1u << 2) | kStAgg, tag_pre = (1u << 2) | kStPrefix;
        uint32_t excl = 0;
        if (tile == 0) {
            if (lane == 0) granule_store(rdesc, tag_pre, tile_heads);
        } else {
            if (lane == 0) granule_store(rdesc + tile, tag_agg, tile_heads);
            int64_t k = (int64_t)tile - 1;
            uint32_t spins = 0;
            for (;;) {
                const int64_t j = k - lane;
                unsigned long long g = ((unsigned long long)tag_pre << 32);          // before tile 0: prefix 0
                if (j >= 0) g = granule_load(rdesc + j);
                const uint32_t tg = (uint32_t)(g >> 32);
                const unsigned long long m_pre = __ballot(tg == tag_pre);
                const unsigned long long m_bad = __ballot(tg != tag_pre && tg != tag_agg);
                const int first_pre = m_pre ? __ffsll((long long)m_pre) - 1 : 64;
                const int first_bad = m_bad ? __ffsll((long long)m_bad) - 1 : 64;
                const int take = first_pre < first_bad ? first_pre + 1 : first_bad;  // lanes 0 .. take-1 are usable
                uint32_t v = lane < take ? (uint32_t)g : 0u;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
                excl += v;
                if (first_pre < first_bad) break;
                k -= take;
                if (take == 0) {
                    if (++spins > spin_limit) { if (lane == 0) *err = 1u; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if (lane == 0) granule_store(rdesc + tile, tag_pre, excl + tile_heads);
        }
        if (lane == 0) {
            s_base = excl;
            tile_base[tile] = excl;
            if (tile == ntiles - 1) *n_rows = excl + tile_heads;
        }
    }
    __syncthreads();
    const uint32_t base = s_base;
    // ---- the thread's own runs
    int64_t row = (int64_t)base + pre - 1;                  // row open when the thread starts
    Part run = part_zero(), lead = part_zero();
    bool seen = false;
#pragma unroll
    for (int k = 0; k < kOsRedItems; ++k) {
        const uint32_t i = i0 + k;
        if (i < n) {
            if ((headbits >> k) & 1u) {
                if (!seen) {
                    lead = run;                              // closes the row that was open at the thread's start
                } else {                                     // a row that starts and ends inside the thread
                    row_n[row] = run.n;
                    row_sum[row] = run.s;
                    row_sum_sq[row] = run.s2;
                }
                seen = true;
                run = part_zero();
                ++row;
                const uint64_t w = words[i];
                const uint32_t src = packed_bits ? (uint32_t)(w & idx_mask) : idx[i];
                row_key[row] = packed_bits ? (w >> packed_bits) + key_base : w;
                row_mask[row] = (uint32_t)(payload[src] >> 62);
                row_first[row] = first_map ? first_map[src] : src;
                row_offset[row] = i;
            }
            const unsigned long long o = (unsigned long long)s_o[os_pad(t * kOsRedItems + k)];
            run.n += 1;
            run.s += o;
            run.s2 += o * o;
        }
    }
    // ---- segmented scan over the threads: value = the thread's open tail (its whole share when it holds no head)
    Part sc = run;
    bool fl = seen;                                          // a head inside the scanned range
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const Part o = part_shfl_up(sc, d);
        const int of = __shfl_up((int)fl, d, 64);
        if (lane >= d && !fl) {
            part_add(sc, o);
            fl = of != 0;
        }
    }
    if (lane == 63) { s_pn[wave] = sc.n; s_ps[wave] = sc.s; s_ps2[wave] = sc.s2; s_pf[wave] = fl ? 1u : 0u; }
    __syncthreads();
    Part carry = part_zero();                                // open run entering this wave
    bool carry_f = false;                                    // a head in an earlier wave of the tile
    for (int w = 0; w < wave; ++w) {
        const Part o = Part{s_pn[w], s_ps[w], s_ps2[w]};
        if (s_pf[w]) { carry = o; carry_f = true; } else { part_add(carry, o); }
    }
    Part incl = sc;
    if (!fl) part_add(incl, carry);
    // exclusive value: what is open just before this thread
    Part before = part_shfl_up(incl, 1);
    int before_f = __shfl_up((int)fl, 1, 64);
    if (lane == 0) { before = carry; before_f = 0; }
    const bool head_before = (before_f != 0) || carry_f;     // a head earlier in the TILE
    if (seen) {
        Part total = before;
        part_add(total, lead);
        if (head_before) {                                   // the row it closes began in this tile: final value
            const int64_t r = (int64_t)base + pre - 1;
            row_n[r] = total.n;
            row_sum[r] = total.s;
            row_sum_sq[r] = total.s2;
        } else {                                             // it began in an earlier tile: hand the share over
            lead_n[tile] = total.n;
            lead_s[tile] = total.s;
            lead_s2[tile] = total.s2;
        }
    }
    if (t == kOsRedThreads - 1) {
        const bool any = fl || carry_f;
        if (any) {                                           // the row open at the end of the tile (so far)
            const int64_t r = (int64_t)base + tile_heads - 1;
            row_n[r] = incl.n;
            row_sum[r] = incl.s;
            row_sum_sq[r] = incl.s2;
        } else {                                             // no head in the whole tile
            lead_n[tile] = incl.n;
            lead_s[tile] = incl.s;
            lead_s2[tile] = incl.s2;
        }
    }
}

// 5. a tile that begins inside a row adds its leading share to that row (the last row before the tile)
__global__ __launch_bounds__(256) void os_fixup_kernel(const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                       const uint32_t* __restrict__ tile_base,
                                                       const uint32_t* __restrict__ lead_n,
                                                       const unsigned long long* __restrict__ lead_s,
                                                       const unsigned long long* __restrict__ lead_s2,
                                                       uint32_t* __restrict__ row_n,
                                                       unsigned long long* __restrict__ row_sum,
                                                       unsigned long long* __restrict__ row_sum_sq,
                                                       const uint32_t* __restrict__ err, uint32_t* __restrict__ n_rows) {
    uint32_t n = *n_ptr;
    n = n < cap ? n : cap;
    const uint32_t ntiles = os_nblocks(n, kOsRedTile);
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    // (the last kernel of this form: a look-back that gave up - in a stream pass or in the row numbering - is reported)
    if (tile == 0 && *err) *n_rows = BESST_ROWS_SORT_FAILED;
    if (tile == 0 || tile >= ntiles) return;
    const uint32_t c = lead_n[tile];
    if (c == 0) return;
    const uint32_t row = tile_base[tile] - 1;                // tile 0 starts with a head, so a base of 0 cannot occur here
    atomicAdd(&row_n[row], c);
    atomicAdd(&row_sum[row], lead_s[tile]);
    atomicAdd(&row_sum_sq[row], lead_s2[tile]);
}

struct OsWorkspace {
    uint32_t* table;
    uint32_t* digit_base;
    uint32_t* tickets;          // kOsMaxPasses + 1
    uint32_t* err;
    unsigned long long* granules;   // scatter descriptors, then the reduce stage's
    size_t desc_words, granule_words;
    uint32_t* tile_base;
    uint32_t* lead_n;
    unsigned long long* lead_s;
    unsigned long long* lead_s2;
    uint32_t* bucket_start;     // kTopBuckets + 1
    uint32_t* bucket_class;     // kTopBuckets: which kernel finishes the bucket
    uint32_t* tile_first;       // per scatter tile: its first block of a segmented stream
    uint32_t* bucket_rows;      // kTopBuckets
    char* staged;               // 40 bytes per tuple of capacity: the buckets' rows before they are numbered
    size_t total;
};

OsWorkspace os_carve(void* ws, int64_t cap, int bits) {
    OsWorkspace w;
    char* p = static_cast<char*>(ws);
    size_t off = 0;
    const size_t radix = (size_t)1 << bits;
    const size_t nt_sort = (size_t)((cap + kOsTile - 1) / kOsTile);
    const size_t nt_red = (size_t)((cap + kOsRedTile - 1) / kOsRedTile);
    w.table = reinterpret_cast<uint32_t*>(p + off); off += align_up((size_t)kOsHistBlocks * kOsMaxPasses * radix * 4, 256);
    w.digit_base = reinterpret_cast<uint32_t*>(p + off); off += align_up((size_t)kOsMaxPasses * radix * 4, 256);
    w.tickets = reinterpret_cast<uint32_t*>(p + off); off += 256;
    w.err = reinterpret_cast<uint32_t*>(p + off); off += 256;
    w.desc_words = nt_sort * radix;
    w.granule_words = w.desc_words + nt_red;
    w.granules = reinterpret_cast<unsigned long long*>(p + off); off += align_up(w.granule_words * 8, 256);
    w.tile_base = reinterpret_cast<uint32_t*>(p + off); off += align_up(nt_red * 4, 256);
    w.lead_n = reinterpret_cast<uint32_t*>(p + off); off += align_up(nt_red * 4, 256);
    w.lead_s = reinterpret_cast<unsigned long long*>(p + off); off += align_up(nt_red * 8, 256);
    w.lead_s2 = reinterpret_cast<unsigned long long*>(p + off); off += align_up(nt_red * 8, 256);
    w.bucket_start = reinterpret_cast<uint32_t*>(p + off); off += align_up(((size_t)kTopBuckets + 1) * 4, 256);
    w.bucket_class = reinterpret_cast<uint32_t*>(p + off); off += align_up((size_t)kTopBuckets * 4, 256);
    w.tile_first = reinterpret_cast<uint32_t*>(p + off); off += align_up(nt_sort * 4, 256);
    w.bucket_rows = reinterpret_cast<uint32_t*>(p + off); off += align_up((size_t)kTopBuckets * 4, 256);
    w.staged = p + off; off += align_up((size_t)cap * 40, 256);
    w.total = off;
    return w;
}

}  // namespace

// rows of the table when the histograms come with the tuples (PresortSpec): every row costs os_offsets_kernel a load per
// digit, and 24 k compacting workgroups spread over 64 rows do not contend
constexpr int kOsPresortRows = 64;
static_assert(kOsPresortRows <= kOsHistBlocks, "the table is sized for kOsHistBlocks rows");

// the two-pass + buckets form (3b) serves packed keys of four digits or more
static bool os_hybrid(int64_t cap, int key_bits) {
    const int passes = (key_bits + kOsBits - 1) / kOsBits;
    int idx_bits = 1;
    while (((int64_t)1 << idx_bits) < cap) ++idx_bits;
    const bool packed = key_bits + idx_bits <= 64;
    return packed && kOsBits == 8 && passes >= 4 && key_bits - 16 <= 31 && cap <= ((int64_t)1 << 30);
}

bool onesweep_presort_spec(int64_t cap, int key_bits, uint64_t key_base, void* ws, PresortSpec* out) {
    if (!os_hybrid(cap, key_bits)) return false;
    const OsWorkspace w = os_carve(ws, cap, kOsBits);
    out->table = w.table;
    out->rows = kOsPresortRows;
    out->shift = key_bits - 16;
    out->key_base = key_base;
    out->cap = (uint32_t)cap;
    return true;
}

void* onesweep_staged_rows(void* ws, int64_t cap) { return os_carve(ws, cap < 1 ? 1 : cap, kOsBits).staged; }

size_t onesweep_workspace_bytes(int64_t cap) {
    if (cap < 1) cap = 1;
    return os_carve(nullptr, cap, kOsBits).total;
}

int launch_onesweep_sort_reduce(hipStream_t s, int64_t cap, const uint32_t* n_tuples, int key_bits,
                                const uint64_t* keys, const uint64_t* payload, uint64_t* buf_keys[2],
                                uint32_t* buf_idx[2], uint64_t* row_key, uint32_t* row_mask, uint32_t* row_n,
                                int64_t* row_sum, int64_t* row_sum_sq, uint32_t* row_first, uint32_t* row_offset,
                                int32_t* obs_lo, int32_t* obs_hi, uint32_t* n_rows, void* ws, size_t ws_bytes,
                                const uint32_t* first_map, uint64_t key_base, bool hist_ready, const SegSource* seg) {
    const OsWorkspace w = os_carve(ws, cap, kOsBits);
    BESST_REQUIRE(ws != nullptr && ws_bytes >= w.total, "reduce: chained-scan workspace too small");
    int passes = (key_bits + kOsBits - 1) / kOsBits;
    BESST_REQUIRE(passes >= 1 && passes <= kOsMaxPasses, "reduce: too many radix passes");
    int idx_bits = 1;
    while (((int64_t)1 << idx_bits) < cap) ++idx_bits;
    const int packed_bits = key_bits + idx_bits <= 64 ? idx_bits : 0;
    // keys of four digits or more: two stream-wide passes on the top 16 bits, the rest bucket by bucket in LDS (3b)
    const bool hybrid = os_hybrid(cap, key_bits);
    const int shift0 = hybrid ? key_bits - 16 : 0;
    if (hybrid) passes = 2;
    const uint32_t nt_sort = (uint32_t)((cap + kOsTile - 1) / kOsTile);
    const uint32_t nt_red = (uint32_t)((cap + kOsRedTile - 1) / kOsRedTile);
    constexpr int RADIX = 1 << kOsBits;
    const uint32_t spin_limit = os_spin_limit();
    BESST_HIP_TRY(hipMemsetAsync(w.err, 0, 4, s));
    BESST_REQUIRE(!seg || (hist_ready && hybrid), "reduce: a segmented stream needs the bucket form and its histograms");
    if (hist_ready && hybrid) {
        // the histograms came with the tuples (PresortSpec: compact_kernel); only the descriptors are left to clear
        BESST_HIP_TRY(hipMemsetAsync(w.granules, 0, w.granule_words * 8, s));
    } else {
        ProfScope ps(s, kProfOsHist);
        hipLaunchKernelGGL((os_hist_kernel<kOsBits>), dim3(kOsHistBlocks), dim3(kOsHistThreads), 0, s, keys, n_tuples,
                           (uint32_t)cap, passes, shift0, key_base, w.table, w.granules, w.granule_words);
    }
    {
        ProfScope ps(s, kProfOsOffsets);
        hipLaunchKernelGGL((os_offsets_kernel<kOsBits>), dim3(passes), dim3(256), 0, s, w.table,
                           hist_ready && hybrid ? kOsPresortRows : kOsHistBlocks, passes, w.digit_base, w.tickets);
    }
    const uint64_t* kin = keys;
    const uint32_t* iin = nullptr;
    for (int p = 0; p < passes; ++p) {
        ProfScope ps(s, kProfOsScatter);
        uint64_t* kout = buf_keys[p & 1];
        uint32_t* iout = buf_idx[p & 1];
        const int shift = shift0 + p * kOsBits + (p > 0 ? packed_bits : 0);
#define BESST_OS_LAUNCH(FIRST, PACKED)                                                                                   \
    hipLaunchKernelGGL((os_scatter_kernel<kOsBits, FIRST, PACKED>), dim3(nt_sort), dim3(kOsThreads), 0, s, kin, iin,      \
                       n_tuples, (uint32_t)cap, shift, p, packed_bits, key_base, w.digit_base + (size_t)p * RADIX, w.granules, \
                       w.tickets + p, kout, iout, w.err, spin_limit)
        if (seg && p == 0) {
            int seg_win = kSegWin;                           // BESST_SEG_WINDOW (tests): see os_scatter_kernel
            if (const char* e = getenv("BESST_SEG_WINDOW")) {
                const int v = atoi(e);
                if (v >= 1 && v < kSegWin) seg_win = v;
            }
            hipLaunchKernelGGL(os_seg_tiles_kernel, dim3((nt_sort + 255) / 256), dim3(256), 0, s, *seg, n_tuples, (uint32_t)cap,
                               w.tile_first);
            hipLaunchKernelGGL((os_scatter_kernel<kOsBits, true, true, true>), dim3(nt_sort), dim3(kOsThreads), 0, s, kin, iin,
                               n_tuples, (uint32_t)cap, shift, p, packed_bits, key_base, w.digit_base + (size_t)p * RADIX,
                               w.granules, w.tickets + p, kout, iout, w.err, spin_limit, *seg, w.tile_first, seg_win);
        } else if (packed_bits) {
            if (p == 0) BESST_OS_LAUNCH(true, true); else BESST_OS_LAUNCH(false, true);
        } else {
            if (p == 0) BESST_OS_LAUNCH(true, false); else BESST_OS_LAUNCH(false, false);
        }
#undef BESST_OS_LAUNCH
        kin = kout;
        iin = iout;
    }
    if (hybrid) {
        BwOut o;
        o.payload = payload; o.first_map = first_map; o.key_base = key_base; o.obs_lo = obs_lo; o.obs_hi = obs_hi;
        o.staged = reinterpret_cast<StagedRow*>(w.staged);
        o.bucket_rows = w.bucket_rows;
        {
            ProfScope ps(s, kProfOsBucket);
            hipLaunchKernelGGL(os_bucket_start_kernel, dim3((kTopBuckets + 1 + 255) / 256), dim3(256), 0, s, kin, n_tuples,
                               (uint32_t)cap, shift0 + packed_bits, w.bucket_start);
            hipLaunchKernelGGL(os_bucket_wave_kernel, dim3(kTopBuckets / kBkWaves), dim3(kBkThreads), 0, s, buf_keys[1],
                               w.bucket_start, packed_bits, shift0, w.bucket_class, o);
            hipLaunchKernelGGL(os_bucket_wave_lds_kernel, dim3(2048), dim3(kBkThreads), 0, s, buf_keys[1], w.bucket_start,
                               packed_bits, shift0, w.bucket_class, o);
            hipLaunchKernelGGL(os_bucket_sort_kernel, dim3(1024), dim3(kBkThreads), 0, s, buf_keys[1], buf_keys[0],
                               w.bucket_start, packed_bits, shift0, w.bucket_class, o);
        }
        {
            ProfScope ps(s, kProfOsBucketRows);
            hipLaunchKernelGGL(os_bucket_rows_kernel, dim3(kTopBuckets / kRowsThreads), dim3(kRowsThreads), 0, s,
                               w.bucket_start, o, w.err, n_rows, row_key, row_mask, row_n, reinterpret_cast<unsigned long long*>(row_sum),
                               reinterpret_cast<unsigned long long*>(row_sum_sq), row_first, row_offset);
        }
        BESST_HIP_TRY(hipGetLastError());
        return BESST_OK;
    }
    {
        ProfScope ps(s, kProfOsReduce);
        hipLaunchKernelGGL(os_reduce_kernel, dim3(nt_red), dim3(kOsRedThreads), 0, s, kin, iin, payload, n_tuples,
                           (uint32_t)cap, packed_bits, key_base, w.granules + w.desc_words, w.tickets + kOsMaxPasses, w.tile_base,
                           w.lead_n, w.lead_s, w.lead_s2, n_rows, row_key, row_mask, row_n,
                           reinterpret_cast<unsigned long long*>(row_sum), reinterpret_cast<unsigned long long*>(row_sum_sq),
                           row_first, row_offset, obs_lo, obs_hi, first_map, w.err, spin_limit);
    }
    {
        ProfScope ps(s, kProfOsFixup);
        hipLaunchKernelGGL(os_fixup_kernel, dim3((nt_red + 255) / 256), dim3(256), 0, s, n_tuples, (uint32_t)cap, w.tile_base,
                           w.lead_n, w.lead_s, w.lead_s2, row_n, reinterpret_cast<unsigned long long*>(row_sum),
                           reinterpret_cast<unsigned long long*>(row_sum_sq), w.err, n_rows);
    }
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

}  // namespace besst
