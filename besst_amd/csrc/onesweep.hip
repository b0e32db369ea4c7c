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

constexpr int kOsThreads = 512;
constexpr int kOsWaves = kOsThreads / 64;
constexpr int kOsItems = 16;
constexpr int kOsTile = kOsThreads * kOsItems;          // 8192 keys per scatter tile
constexpr int kOsHistThreads = 256;
constexpr int kOsHistTile = kOsHistThreads * 16;
constexpr int kOsHistBlocks = 512;
constexpr int kOsMaxPasses = 8;
constexpr int kOsRedThreads = 256;
constexpr int kOsRedItems = 16;
constexpr int kOsRedTile = kOsRedThreads * kOsRedItems; // 4096 tuples per reduce tile
constexpr uint32_t kStAgg = 1u, kStPrefix = 2u;
constexpr uint32_t kSpinLimit = 1u << 22;               // ~ seconds; a healthy look-back waits microseconds

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
                                                                 int passes, uint32_t* __restrict__ table,
                                                                 unsigned long long* __restrict__ granules,
                                                                 size_t granule_words) {
    constexpr int RADIX = 1 << BITS;
    __shared__ uint32_t s_hist[kOsMaxPasses * RADIX];
    const int t = threadIdx.x, lane = t & 63;
    for (int d = t; d < passes * RADIX; d += kOsHistThreads) s_hist[d] = 0;
    // every granule of this call starts as "nothing published" (plain stores: first polled after a kernel boundary)
    {
        const size_t gsz = (size_t)gridDim.x * kOsHistThreads;
        for (size_t i = (size_t)blockIdx.x * kOsHistThreads + t; i < granule_words; i += gsz) granules[i] = 0ull;
    }
    uint32_t n = *n_ptr;
    n = n < cap ? n : cap;
    __syncthreads();
    for (uint32_t base = blockIdx.x * (uint32_t)kOsHistTile; base < n; base += gridDim.x * (uint32_t)kOsHistTile) {
        if (base + kOsHistTile <= n) {
            uint64_t k[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) k[r] = keys[base + r * kOsHistThreads + t];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                for (int p = 0; p < passes; ++p) {
                    const uint32_t d = (uint32_t)(k[r] >> (p * BITS)) & (uint32_t)(RADIX - 1);
                    // high digits of a (tid,pos)-ordered stream: the whole wave shares one digit, which as 64 LDS
                    // atomics on one counter would serialise
                    const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
                    if (__all(d == f)) {
                        if (lane == 0) atomicAdd(&s_hist[p * RADIX + f], 64u);
                    } else {
                        atomicAdd(&s_hist[p * RADIX + d], 1u);
                    }
                }
            }
        } else {
            for (int r = 0; r < 16; ++r) {
                const uint32_t i = base + r * kOsHistThreads + t;
                if (i < n) {
                    const uint64_t k = keys[i];
                    for (int p = 0; p < passes; ++p)
                        atomicAdd(&s_hist[p * RADIX + ((uint32_t)(k >> (p * BITS)) & (uint32_t)(RADIX - 1))], 1u);
                }
            }
        }
    }
    __syncthreads();
    uint32_t* row = table + (size_t)blockIdx.x * passes * RADIX;
    for (int d = t; d < passes * RADIX; d += kOsHistThreads) row[d] = s_hist[d];
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
    for (int g = 0; g < hist_blocks; ++g) {
        const uint32_t* row = table + ((size_t)g * passes + p) * RADIX + (size_t)t * DPT;
#pragma unroll
        for (int q = 0; q < DPT; ++q) tot[q] += row[q];
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
// kFirst: the input is the raw key stream (the stream index is the position); packed_bits > 0: the words written
// (and read by later passes) are key << packed_bits | stream index; packed_bits == 0: keys and indexes travel as
// two arrays (keys wider than 64 - index bits).
template <int BITS, bool kFirst>
__global__ __launch_bounds__(kOsThreads) void os_scatter_kernel(
    const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ idx_in, const uint32_t* __restrict__ n_ptr,
    uint32_t cap, int shift, int pass, int packed_bits, const uint32_t* __restrict__ digit_base,
    unsigned long long* __restrict__ desc, uint32_t* __restrict__ ticket, uint64_t* __restrict__ keys_out,
    uint32_t* __restrict__ idx_out, uint32_t* __restrict__ err) {
    constexpr int RADIX = 1 << BITS;
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
    uint32_t idx[kOsItems];
#pragma unroll
    for (int r = 0; r < kOsItems; ++r) {
        const uint32_t i = wbase + r * 64 + lane;
        key[r] = i < n ? keys_in[i] : ~0ull;
        idx[r] = kFirst ? i : ((i < n && !packed_bits) ? idx_in[i] : 0u);
    }
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t dig_rank[kOsItems];                            // digit | rank inside the wave's share << BITS
#pragma unroll
    for (int r = 0; r < kOsItems; ++r) {
        const uint32_t i = wbase + r * 64 + lane;
        const bool valid = i < n;
        const uint32_t d = valid ? ((uint32_t)(key[r] >> shift) & (uint32_t)(RADIX - 1)) : (uint32_t)(RADIX - 1);
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < BITS; ++bit) {
            const bool one = (d >> bit) & 1u;
            const unsigned long long bal = __ballot(one);
            peers &= one ? bal : ~bal;
        }
        uint32_t pre = 0;
        const int leader = __ffsll((long long)peers) - 1;
        if (valid && lane == leader) {
            volatile uint32_t* slot = &s_whist[wave][d];     // an earlier round of this wave may have updated it
            pre = *slot;
            *slot = pre + (uint32_t)__popcll(peers);
        }
        pre = __shfl(pre, leader < 0 ? 0 : leader, 64);
        dig_rank[r] = d | ((pre + (uint32_t)__popcll(peers & lt_mask)) << BITS);
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
                if (++spins > kSpinLimit) { *err = 1u; break; }
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
            if (packed_bits) {
                keys_out[dst] = kFirst ? ((key[r] << packed_bits) | idx[r]) : key[r];
            } else {
                keys_out[dst] = key[r];
                idx_out[dst] = idx[r];
            }
        }
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
__device__ __forceinline__ Part part_shfl(const Part& a, int src) {
    Part o;
    o.n = __shfl(a.n, src, 64);
    o.s = __shfl(a.s, src, 64);
    o.s2 = __shfl(a.s2, src, 64);
    return o;
}

__global__ __launch_bounds__(kOsRedThreads) void os_reduce_kernel(
    const uint64_t* __restrict__ words, const uint32_t* __restrict__ idx, const uint64_t* __restrict__ payload,
    const uint32_t* __restrict__ n_ptr, uint32_t cap, int packed_bits, unsigned long long* __restrict__ rdesc,
    uint32_t* __restrict__ ticket, uint32_t* __restrict__ tile_base, uint32_t* __restrict__ lead_n,
    unsigned long long* __restrict__ lead_s, unsigned long long* __restrict__ lead_s2, uint32_t* __restrict__ n_rows,
    uint64_t* __restrict__ row_key, uint32_t* __restrict__ row_mask, uint32_t* __restrict__ row_n,
    unsigned long long* __restrict__ row_sum, unsigned long long* __restrict__ row_sum_sq,
    uint32_t* __restrict__ row_first, uint32_t* __restrict__ row_offset, int32_t* __restrict__ obs_lo,
    int32_t* __restrict__ obs_hi, const uint32_t* __restrict__ first_map, uint32_t* __restrict__ err) {
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
    const uint32_t i0 = tile * (uint32_t)kOsRedTile + (uint32_t)t * kOsRedItems;
    const uint64_t idx_mask = packed_bits ? ((1ull << packed_bits) - 1ull) : 0ull;
    uint64_t key[kOsRedItems];
    uint32_t src[kOsRedItems];
    uint64_t prev = (i0 > 0 && i0 - 1 < n) ? (words[i0 - 1] >> packed_bits) : 0ull;
    if (i0 + kOsRedItems <= n) {
#pragma unroll
        for (int k = 0; k < kOsRedItems; k += 2) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(words + i0 + k);
            key[k] = v.x >> packed_bits;
            key[k + 1] = v.y >> packed_bits;
            src[k] = (uint32_t)(v.x & idx_mask);
            src[k + 1] = (uint32_t)(v.y & idx_mask);
        }
        if (!packed_bits) {
#pragma unroll
            for (int k = 0; k < kOsRedItems; k += 4) {
                const uint4 v = *reinterpret_cast<const uint4*>(idx + i0 + k);
                src[k] = v.x; src[k + 1] = v.y; src[k + 2] = v.z; src[k + 3] = v.w;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < kOsRedItems; ++k) {
            const uint32_t i = i0 + k;
            const uint64_t v = i < n ? words[i] : 0ull;
            key[k] = v >> packed_bits;
            src[k] = packed_bits ? (uint32_t)(v & idx_mask) : (i < n ? idx[i] : 0u);
        }
    }
    // the observations are gathered through the stream index; nothing below depends on the look-back yet
    uint64_t pl[kOsRedItems];
#pragma unroll
    for (int k = 0; k < kOsRedItems; ++k) pl[k] = (i0 + k) < n ? payload[src[k]] : 0ull;
    uint32_t headbits = 0;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < kOsRedItems; ++k) {
        const uint32_t i = i0 + k;
        const bool h = i < n && (i == 0 || key[k] != prev);
        headbits |= h ? (1u << k) : 0u;
        cnt += h ? 1 : 0;
        prev = key[k];
    }
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
    // ---- rows before this tile: chained scan of the head counts, wave 0, 64 predecessors per round
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
                    if (++spins > kSpinLimit) { if (lane == 0) *err = 1u; break; }
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
            const uint32_t lo = (uint32_t)pl[k], hi = (uint32_t)(pl[k] >> 32);
            const int32_t o_lo = (int32_t)lo, o_hi = (int32_t)(hi & 0x3fffffffu);
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
                row_key[row] = key[k];
                row_mask[row] = hi >> 30;
                row_first[row] = first_map ? first_map[src[k]] : src[k];
                row_offset[row] = i;
            }
            obs_lo[i] = o_lo;
            obs_hi[i] = o_hi;
            const unsigned long long o = (unsigned long long)((long long)o_lo + o_hi);
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
    (void)part_shfl;
}

// 5. a tile that begins inside a row adds its leading share to that row (the last row before the tile)
__global__ __launch_bounds__(256) void os_fixup_kernel(const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                       const uint32_t* __restrict__ tile_base,
                                                       const uint32_t* __restrict__ lead_n,
                                                       const unsigned long long* __restrict__ lead_s,
                                                       const unsigned long long* __restrict__ lead_s2,
                                                       uint32_t* __restrict__ row_n,
                                                       unsigned long long* __restrict__ row_sum,
                                                       unsigned long long* __restrict__ row_sum_sq) {
    uint32_t n = *n_ptr;
    n = n < cap ? n : cap;
    const uint32_t ntiles = os_nblocks(n, kOsRedTile);
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
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
    w.total = off;
    return w;
}

#ifndef BESST_OS_BITS
#define BESST_OS_BITS 8
#endif
constexpr int kOsBits = BESST_OS_BITS;

}  // namespace

size_t onesweep_workspace_bytes(int64_t cap) {
    if (cap < 1) cap = 1;
    return os_carve(nullptr, cap, kOsBits).total;
}

int launch_onesweep_sort_reduce(hipStream_t s, int64_t cap, const uint32_t* n_tuples, int key_bits,
                                const uint64_t* keys, const uint64_t* payload, uint64_t* buf_keys[2],
                                uint32_t* buf_idx[2], uint64_t* row_key, uint32_t* row_mask, uint32_t* row_n,
                                int64_t* row_sum, int64_t* row_sum_sq, uint32_t* row_first, uint32_t* row_offset,
                                int32_t* obs_lo, int32_t* obs_hi, uint32_t* n_rows, void* ws, size_t ws_bytes,
                                const uint32_t* first_map) {
    const OsWorkspace w = os_carve(ws, cap, kOsBits);
    BESST_REQUIRE(ws != nullptr && ws_bytes >= w.total, "reduce: chained-scan workspace too small");
    const int passes = (key_bits + kOsBits - 1) / kOsBits;
    BESST_REQUIRE(passes >= 1 && passes <= kOsMaxPasses, "reduce: too many radix passes");
    int idx_bits = 1;
    while (((int64_t)1 << idx_bits) < cap) ++idx_bits;
    const int packed_bits = key_bits + idx_bits <= 64 ? idx_bits : 0;
    const uint32_t nt_sort = (uint32_t)((cap + kOsTile - 1) / kOsTile);
    const uint32_t nt_red = (uint32_t)((cap + kOsRedTile - 1) / kOsRedTile);
    constexpr int RADIX = 1 << kOsBits;
    BESST_HIP_TRY(hipMemsetAsync(w.err, 0, 4, s));
    {
        ProfScope ps(s, kProfSortHist);
        const uint32_t hist_tiles = (uint32_t)((cap + kOsHistTile - 1) / kOsHistTile);
        (void)hist_tiles;
        hipLaunchKernelGGL((os_hist_kernel<kOsBits>), dim3(kOsHistBlocks), dim3(kOsHistThreads), 0, s, keys, n_tuples,
                           (uint32_t)cap, passes, w.table, w.granules, w.granule_words);
    }
    {
        ProfScope ps(s, kProfSortScan);
        hipLaunchKernelGGL((os_offsets_kernel<kOsBits>), dim3(passes), dim3(256), 0, s, w.table, kOsHistBlocks, passes,
                           w.digit_base, w.tickets);
    }
    const uint64_t* kin = keys;
    const uint32_t* iin = nullptr;
    for (int p = 0; p < passes; ++p) {
        ProfScope ps(s, kProfSortScatter);
        uint64_t* kout = buf_keys[p & 1];
        uint32_t* iout = buf_idx[p & 1];
        const int shift = p * kOsBits + (p > 0 ? packed_bits : 0);
        if (p == 0)
            hipLaunchKernelGGL((os_scatter_kernel<kOsBits, true>), dim3(nt_sort), dim3(kOsThreads), 0, s, kin, iin, n_tuples,
                               (uint32_t)cap, shift, p, packed_bits, w.digit_base + (size_t)p * RADIX, w.granules,
                               w.tickets + p, kout, iout, w.err);
        else
            hipLaunchKernelGGL((os_scatter_kernel<kOsBits, false>), dim3(nt_sort), dim3(kOsThreads), 0, s, kin, iin, n_tuples,
                               (uint32_t)cap, shift, p, packed_bits, w.digit_base + (size_t)p * RADIX, w.granules,
                               w.tickets + p, kout, iout, w.err);
        kin = kout;
        iin = iout;
    }
    {
        ProfScope ps(s, kProfRowReduce);
        hipLaunchKernelGGL(os_reduce_kernel, dim3(nt_red), dim3(kOsRedThreads), 0, s, kin, iin, payload, n_tuples,
                           (uint32_t)cap, packed_bits, w.granules + w.desc_words, w.tickets + kOsMaxPasses, w.tile_base,
                           w.lead_n, w.lead_s, w.lead_s2, n_rows, row_key, row_mask, row_n,
                           reinterpret_cast<unsigned long long*>(row_sum), reinterpret_cast<unsigned long long*>(row_sum_sq),
                           row_first, row_offset, obs_lo, obs_hi, first_map, w.err);
    }
    {
        ProfScope ps(s, kProfRowScan);
        hipLaunchKernelGGL(os_fixup_kernel, dim3((nt_red + 255) / 256), dim3(256), 0, s, n_tuples, (uint32_t)cap, w.tile_base,
                           w.lead_n, w.lead_s, w.lead_s2, row_n, reinterpret_cast<unsigned long long*>(row_sum),
                           reinterpret_cast<unsigned long long*>(row_sum_sq));
    }
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

}  // namespace besst
