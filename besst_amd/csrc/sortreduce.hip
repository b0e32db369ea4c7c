// Stage 2 of the scaffold-graph build: edge table from the ordered tuple stream.
//
//   keys[i]    = ((min_node << node_bits) | max_node) << 1 | is_fishy      (CreateGraph.py:842-843 keys
//                the edge by the unordered node pair; fishy_edges :161-162 likewise)
//   payload[i] = obs_of_min_node | (obs_of_max_node | mask << 30) << 32
//
// 1. Stable LSD radix sort (only the significant key bits; 11-bit digits while the tuple count is small and
//    the stage is launch-latency bound, 8-bit digits for large streams where scatter locality matters) of
//    (key, stream index).
//    Stability keeps every edge's observations in BAM order, which is the order the reference appends
//    them in (CreateGraph.py:845,856,862), and makes the first tuple of a row its first occurrence.
//    Ranking is wave-native: a 64-lane match-any built from 8 ballots gives each key its rank among
//    equal digits of the same wave; waves are stitched through a 4 x 256 LDS table.
// 2. Segmented reduction of the sorted stream into edge rows: nr_links, sum obs, sum obs^2
//    (int64, exact), first stream index, slice offset; observations are gathered once through the
//    sorted index and written grouped by row.
//
// All sizes are read from device memory (n_tuples), so the whole stage is enqueued without a host
// round trip; grids are sized by the caller's capacity bound and surplus workgroups exit at once.
#include <unistd.h>

#include <atomic>
#include <random>
#include <type_traits>

#include "common.h"

namespace besst {

namespace {

__device__ __forceinline__ uint32_t nblocks_of(uint32_t n, uint32_t tile) { return (n + tile - 1) / tile; }

// digit of a key: a radix digit (mode 0) or the owning rank of the key's min scaffold (mode 1)
__device__ __forceinline__ uint32_t digit_of(uint64_t key, const DigitSel& ds) {
    if (ds.mode == 0) return (uint32_t)((key - ds.base) >> ds.shift) & ((1u << ds.bits) - 1u);
    const uint32_t scaf = (uint32_t)(key >> (2 + ds.node_bits));   // key = ((min_node << nb) | max_node) << 1 | f
    return owner_of_scaffold(scaf, ds.world);
}

// ---------------------------------------------------------------------------------------------------
// radix sort
// ---------------------------------------------------------------------------------------------------
// Per-block digit counts live in `table`.  Two layouts:
//   scan-free (few blocks): table[block][digit]; every scatter workgroup sums the columns of the blocks
//                           before it and the digit totals itself - no scan kernel, coalesced reads
//   scanned   (many blocks): table[digit][block], exclusive-scanned along blocks by radix_rowscan_kernel
// Every kernel loads its keys speculatively (guarded by the caller's capacity, not by the device-side count)
// so that the count, the table and the keys arrive after ONE memory latency instead of three.
// (limit of the scan-free layout, a build knob: at 128 - 256 tiles the column sums of 2048-digit rows cost more
// than the scan launch, C2 step 113 -> 118 us)
#ifndef BESST_SCAN_FREE_MAX_BLOCKS
#define BESST_SCAN_FREE_MAX_BLOCKS 64
#endif
constexpr int kScanFreeMaxBlocks = BESST_SCAN_FREE_MAX_BLOCKS;
// Largest stream (in sort tiles) that still gets 11-bit LSD digits.  Measured on C3 slices of 0.5 - 8.5 M tuples:
// the per-tile table of 2048 counters then holds as many entries as a quarter to a half of the keys, written
// and read as scattered 4-byte words, and 8-bit digits win at every size (8.5 M tuples: 0.45 vs 0.63 ms for the
// sort) although they need one more pass.  Streams of <= 64 tiles take the MSD + bucket path below, so the
// wide LSD variant is effectively retired; the knob stays for experiments.
#ifndef BESST_WIDE_DIGIT_MAX_BLOCKS
#define BESST_WIDE_DIGIT_MAX_BLOCKS 64
#endif
constexpr int kWideDigitMaxBlocks = BESST_WIDE_DIGIT_MAX_BLOCKS;

template <int BITS, int ITEMS = kSortItems>
__global__ __launch_bounds__(kSortThreads) void radix_hist_kernel(
    const uint64_t* __restrict__ keys, const uint32_t* __restrict__ n_ptr, uint32_t cap, DigitSel ds,
    uint32_t* __restrict__ table, uint32_t stride, int scanned, uint32_t* __restrict__ zero_n,
    unsigned long long* __restrict__ zero_sum, unsigned long long* __restrict__ zero_sum_sq) {
    constexpr int RADIX = 1 << BITS;
    __shared__ uint32_t s_hist[RADIX];
    const int t = threadIdx.x;
    const uint32_t b = blockIdx.x;
    const uint32_t base = b * (kSortThreads * ITEMS);
    uint64_t k[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t i = base + r * kSortThreads + t;
        k[r] = i < cap ? keys[i] : 0ull;
    }
    uint32_t n = *n_ptr;
    n = n < cap ? n : cap;
    if (b >= nblocks_of(n, (kSortThreads * ITEMS))) return;
    for (int d = t; d < RADIX; d += kSortThreads) s_hist[d] = 0;
    __syncthreads();
    // Later passes see nearly sorted keys: whole waves share one digit and per-lane LDS atomics on one bin
    // serialise.  Lanes are consecutive keys, so count RUNS: only the first lane of a run of equal digits adds,
    // with the run length.
    const int lane = t & 63;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t i = base + r * kSortThreads + t;
        const bool valid = i < n;
        const uint32_t d = valid ? digit_of(k[r], ds) : 0xffffffffu;
        const uint32_t up = (uint32_t)__shfl_up((int)d, 1, 64);
        const bool head = valid && (lane == 0 || d != up);
        const unsigned long long heads = __ballot(head);
        const unsigned long long vmask = __ballot(valid);
        if (head) {
            const unsigned long long later = lane == 63 ? 0ull : (heads >> (lane + 1));
            const int next = later ? lane + 1 + (__ffsll((long long)later) - 1) : (int)__popcll(vmask);
            atomicAdd(&s_hist[d], (uint32_t)(next - lane));
        }
    }
    __syncthreads();
    for (int d = t; d < RADIX; d += kSortThreads) {
        if (scanned) table[(uint32_t)d * stride + b] = s_hist[d];
        else table[b * RADIX + d] = s_hist[d];
    }
    if (zero_n) {   // last pass: clear the edge-row accumulators this tile can reach (rows <= tuples)
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const uint32_t i = base + r * kSortThreads + t;
            if (i < n) { zero_n[i] = 0; zero_sum[i] = 0; zero_sum_sq[i] = 0; }
        }
    }
}

// one workgroup per digit: exclusive scan of that digit's per-block counts, total to row_total[d]
// (A tile-major table - coalesced rows from the histogram kernel, a workgroup per 64 tiles x 256 digits scanning its
// columns in place, the last workgroup scanning the chunk totals - was built to get rid of the one-word-per-cache-line
// accesses of this layout: the histogram launches got 16 % faster, the scan three times slower (few, serial
// workgroups), a C3 slice went from 1.24 to 1.34 ms and 1 M tuples from 78 to 82 us.)
__global__ __launch_bounds__(256) void radix_rowscan_kernel(const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                            uint32_t* __restrict__ table, uint32_t stride,
                                                            uint32_t* __restrict__ row_total, uint32_t tile = kSortTile) {
    uint32_t n = *n_ptr;
    n = n < cap ? n : cap;
    const uint32_t nb = nblocks_of(n, tile);
    uint32_t* row = table + (size_t)blockIdx.x * stride;
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_carry;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < nb; c0 += 256) {
        const uint32_t i = c0 + t;
        const uint32_t v = i < nb ? row[i] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(x, d, 64);
            if (lane >= d) x += o;
        }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        uint32_t pre = s_carry;
        for (int w = 0; w < wave; ++w) pre += s_w[w];
        if (i < nb) row[i] = pre + x - v;
        __syncthreads();
        if (t == 255) s_carry = pre + x;
        __syncthreads();
    }
    if (t == 0) row_total[blockIdx.x] = s_carry;
}

template <int BITS, bool kFirst, int ITEMS = kSortItems>
__global__ __launch_bounds__(kSortThreads) void radix_scatter_kernel(
    const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ idx_in,
    const uint32_t* __restrict__ n_ptr, uint32_t cap, DigitSel ds, const uint32_t* __restrict__ table,
    uint32_t stride, const uint32_t* __restrict__ row_total, int scanned, uint64_t* __restrict__ keys_out,
    uint32_t* __restrict__ idx_out, uint32_t* __restrict__ bucket_start, int packed_bits) {
    // packed_bits > 0: the sorted value is key << packed_bits | stream index in ONE 64-bit word (formed by the first
    // pass), no index arrays are read or written: 16 instead of 24 bytes of traffic per tuple and pass.
    constexpr int RADIX = 1 << BITS;
    constexpr int DPT = RADIX / kSortThreads;     // digits per thread (contiguous)
    __shared__ uint32_t s_whist[4][RADIX];
    __shared__ uint32_t s_base[RADIX];
    __shared__ uint32_t s_w[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t b = blockIdx.x;
    const uint32_t wbase = b * (kSortThreads * ITEMS) + wave * (ITEMS * 64);
    uint64_t key[ITEMS];
    uint32_t idx[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t i = wbase + r * 64 + lane;
        key[r] = i < cap ? keys_in[i] : ~0ull;
        idx[r] = kFirst ? i : ((i < cap && !packed_bits) ? idx_in[i] : 0u);
    }
    uint32_t n = *n_ptr;
    n = n < cap ? n : cap;
    const uint32_t nb = nblocks_of(n, (kSortThreads * ITEMS));
    if (b >= nb) return;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        for (int d = t; d < RADIX; d += kSortThreads) s_whist[w][d] = 0;
    {   // global start of this block's share of every digit
        uint32_t pre[DPT], tot[DPT];
#pragma unroll
        for (int q = 0; q < DPT; ++q) { pre[q] = 0; tot[q] = 0; }
        if (scanned) {
#pragma unroll
            for (int q = 0; q < DPT; ++q) {
                const uint32_t d = (uint32_t)t * DPT + q;
                tot[q] = row_total[d];
                pre[q] = table[d * stride + b];
            }
        } else {
            // 8 blocks' rows in flight per round trip (the loads of one group are issued before any is used)
            constexpr int kGroupBlocks = 8;
            for (uint32_t bb0 = 0; bb0 < nb; bb0 += kGroupBlocks) {
                uint32_t v[kGroupBlocks][DPT];
#pragma unroll
                for (int u = 0; u < kGroupBlocks; ++u) {
                    const uint32_t bb = bb0 + u;
                    const uint32_t* row = table + (size_t)(bb < nb ? bb : 0) * RADIX + (size_t)t * DPT;
#pragma unroll
                    for (int q = 0; q < DPT; ++q) v[u][q] = row[q];
                }
#pragma unroll
                for (int u = 0; u < kGroupBlocks; ++u) {
                    const uint32_t bb = bb0 + u;
#pragma unroll
                    for (int q = 0; q < DPT; ++q) {
                        const uint32_t x = bb < nb ? v[u][q] : 0u;
                        tot[q] += x;
                        if (bb < b) pre[q] += x;
                    }
                }
            }
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
            s_base[t * DPT + q] = start + pre[q];
            if (bucket_start && b == 0) bucket_start[t * DPT + q] = start;     // where digit d begins (MSD buckets)
            start += tot[q];
        }
        if (bucket_start && b == 0 && t == kSortThreads - 1) bucket_start[RADIX] = start;
    }
    __syncthreads();

    // The MSD partition of packed words need not be stable (the words are unique and every bucket is fully sorted
    // afterwards): an LDS atomic per key replaces the match-any ranking and the per-wave offset pass.
    const bool unstable = packed_bits != 0 && bucket_start != nullptr;
    if (unstable) {
        const uint64_t low_mask = ds.shift > 0 ? ((1ull << ds.shift) - 1ull) : 0ull;
        uint32_t dig_rank[ITEMS];
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const uint32_t i = wbase + r * 64 + lane;
            dig_rank[r] = 0;
            if (i < n) {
                const uint32_t d = digit_of(key[r], ds);
                dig_rank[r] = d | (atomicAdd(&s_whist[0][d], 1u) << BITS);
            }
        }
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const uint32_t i = wbase + r * 64 + lane;
            if (i < n) {
                const uint32_t dst = s_base[dig_rank[r] & (RADIX - 1)] + (dig_rank[r] >> BITS);
                // the bucket number IS the digit: only the key bits below it travel in the word, so that keys of
                // up to 64 - index bits + 11 bits still pack (bucket_reduce_kernel puts the digit back)
                keys_out[dst] = (((key[r] - ds.base) & low_mask) << packed_bits) | idx[r];
            }
        }
        return;
    }
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t dig_rank[ITEMS];   // digit | rank << BITS
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t i = wbase + r * 64 + lane;
        const bool valid = i < n;
        const uint32_t d = valid ? digit_of(key[r], ds) : (uint32_t)(RADIX - 1);
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
            // volatile: another lane of this wave may have updated the counter in an earlier round
            volatile uint32_t* slot = &s_whist[wave][d];
            pre = *slot;
            *slot = pre + (uint32_t)__popcll(peers);
        }
        pre = __shfl(pre, leader < 0 ? 0 : leader, 64);
        dig_rank[r] = d | ((pre + (uint32_t)__popcll(peers & lt_mask)) << BITS);
    }
    __syncthreads();
    for (int d = t; d < RADIX; d += kSortThreads) {   // per-digit start of each wave inside the block's range
        uint32_t run = s_base[d];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t c = s_whist[w][d];
            s_whist[w][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t i = wbase + r * 64 + lane;
        if (i < n) {
            const uint32_t dst = s_whist[wave][dig_rank[r] & (RADIX - 1)] + (dig_rank[r] >> BITS);
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
// small streams (<= 262 144 tuples): ONE most-significant-digit pass + a per-bucket sort
// ---------------------------------------------------------------------------------------------------
// The LSD sort needs one dependent hist+scatter pair per 11 key bits; at this size every launch costs more than
// its work.  Instead the stream is partitioned once by the TOP 11 significant key bits (2048 buckets of a few
// dozen tuples; the stable scatter keeps stream order inside a bucket) and every bucket is finished by its own
// workgroup with a bitonic sort on (key, stream index) - equivalent to a stable sort by key because the index is
// unique.  Buckets up to 512 tuples sort in LDS (6 KB, so every bucket of the launch is resident at once); larger
// ones (a hub scaffold) sort in place in global scratch.
// Digit width of the partition: wider digits make the scatter's table walk longer, narrower ones make the
// O(n^2) bucket sort explode (C2: scatter + bucket sort = 37.5 / 44 / 73 / 165 us with 11 / 10 / 9 / 8 bits).
#ifndef BESST_MSD_BITS
#define BESST_MSD_BITS 11
#endif
constexpr int kMsdBits = BESST_MSD_BITS;
#ifndef BESST_BUCKET_LDS
#define BESST_BUCKET_LDS 512
#endif
constexpr int kBucketLds = BESST_BUCKET_LDS;
#ifndef BESST_BUCKET_THREADS
#define BESST_BUCKET_THREADS 256
#endif
constexpr int kBucketThreads = BESST_BUCKET_THREADS;

// The partition of the packed words in ONE launch (histogram + scatter were two, with a scan launch between them
// beyond 64 tiles): the partition need not be stable, so a tile's share of a bucket can begin wherever the tile's
// count ARRIVES - one returning atomic per (tile, digit) on the digit totals replaces the per-tile table, its scan
// and the scatter's walk over the rows of the tiles before it.  The bucket starts need the final totals, i.e. every
// tile's counts: the tiles wait for each other inside the launch.  Two forms:
//   * up to kMsdMutualTiles tiles, every workgroup does both halves and they wait for EACH OTHER - safe only because
//     that few workgroups (16 KB of LDS, 256 threads) are on the chip together whatever else runs there, even a
//     dozen launches of this kind on other streams (what else runs finishes without waiting for anything here);
//   * beyond that (kSplit), a tile is served by two workgroups: one of the first half of the grid counts, leaves the
//     tile's shares in a table row (written and read once, by row - no walk) and ends; one of the second half waits
//     for the counting workgroups - all of them started before it, so the wait cannot deadlock however little of
//     the grid fits on the chip (three streams sorting 1.8 M tuples each would otherwise hold each other's slots) -
//     ranks its keys again and scatters.
// The waits are bounded all the same: a tile that gives up leaves the
// call's nonce in the status word, the bucket kernels then do nothing and *n_rows reads BESST_ROWS_SORT_FAILED.
// No state is assumed in the workspace: workgroup 0 clears the totals and says so, flags carry the call's 64-bit
// nonce (a random base per process + a counter), which no stale word equals.
//   flags[0] = totals cleared, flags[1] = a tile gave up, flags[2 + b] = tile b's counts are in
typedef __attribute__((address_space(1))) unsigned long long msd_gu64;
typedef __attribute__((address_space(1))) uint32_t msd_gu32;
#define BESST_MSD_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr uint32_t kMsdSpinLimit = 1u << 22;
// keys per thread of a partition tile (build knob).  Without a per-tile table smaller tiles cost nothing but atomics:
// 16 / 8 / 4 keys per thread = C2's step 100.1 / 97.7 / 97.9 us, the partition of C3's 0.5 M runs 22.5 / 21.6 / 23.8 us
#ifndef BESST_MSD_ITEMS
#define BESST_MSD_ITEMS 8
#endif
constexpr int kMsdItems = BESST_MSD_ITEMS;
static uint32_t msd_spin_limit() {      // BESST_MSD_SPIN_LIMIT (tests): 0 = the first unanswered poll gives up
    static const uint32_t v = [] { const char* e = getenv("BESST_MSD_SPIN_LIMIT"); return e ? (uint32_t)strtoul(e, nullptr, 10) : kMsdSpinLimit; }();
    return v;
}
static int msd_one_launch() {           // BESST_MSD_ONE_LAUNCH=0: the histogram / (scan) / scatter launches
    static const int v = [] { const char* e = getenv("BESST_MSD_ONE_LAUNCH"); return e ? atoi(e) : 1; }();
    return v;
}
static unsigned long long msd_next_nonce() {
    static std::atomic<unsigned long long> ctr{[] {
        std::random_device rd;
        return (((unsigned long long)rd() << 32) ^ (unsigned long long)rd() ^ ((unsigned long long)getpid() << 20)) | 1ull;
    }()};
    return ctr.fetch_add(0x9E3779B97F4A7C15ull) | 1ull;   // (odd: never 0, the value of a zero-filled workspace)
}

#ifndef BESST_MSD_MUTUAL_TILES
#define BESST_MSD_MUTUAL_TILES 128
#endif
constexpr uint32_t kMsdMutualTiles = BESST_MSD_MUTUAL_TILES;

template <int ITEMS, bool kSplit>
__global__ __launch_bounds__(kSortThreads) void msd_partition_kernel(
    const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ n_ptr, uint32_t cap, DigitSel ds,
    uint32_t* total, unsigned long long* flags, unsigned long long nonce, uint32_t spin_limit,
    uint64_t* __restrict__ keys_out, uint32_t* __restrict__ bucket_start, int packed_bits,
    uint32_t* __restrict__ zero_n, unsigned long long* __restrict__ zero_sum,
    unsigned long long* __restrict__ zero_sum_sq, uint32_t* table) {
    constexpr int BITS = kMsdBits;
    constexpr int RADIX = 1 << BITS;
    constexpr int DPT = RADIX / kSortThreads;
    __shared__ uint32_t s_hist[RADIX];
    __shared__ uint32_t s_base[RADIX];
    __shared__ uint32_t s_w[4];
    __shared__ int s_fail;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // kSplit: the first half of the grid counts, the second half scatters; else every workgroup does both
    const uint32_t tiles = kSplit ? gridDim.x / 2u : gridDim.x;
    const bool counts = !kSplit || blockIdx.x < tiles, scatters = !kSplit || blockIdx.x >= tiles;
    const uint32_t b = blockIdx.x < tiles ? blockIdx.x : blockIdx.x - tiles;
    const uint32_t wbase = b * (kSortThreads * ITEMS) + wave * (ITEMS * 64);
    uint64_t key[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t i = wbase + r * 64 + lane;
        key[r] = i < cap ? keys_in[i] : ~0ull;
    }
    uint32_t n = *n_ptr;
    n = n < cap ? n : cap;
    const uint32_t nb = nblocks_of(n, (kSortThreads * ITEMS));
    if (b == 0 && counts) {
#pragma unroll
        for (int q = 0; q < DPT; ++q) __hip_atomic_store((msd_gu32*)&total[q * kSortThreads + t], 0u, BESST_MSD_RLX);
        __threadfence();                                     // the clears have landed ...
        __syncthreads();
        if (t == 0) __hip_atomic_store((msd_gu64*)&flags[0], nonce, BESST_MSD_RLX);   // ... before anyone is told
    }
    if (b >= nb) return;
    if (t == 0) s_fail = 0;
#pragma unroll
    for (int q = 0; q < DPT; ++q) s_hist[q * kSortThreads + t] = 0;
    __syncthreads();
    const uint64_t low_mask = ds.shift > 0 ? ((1ull << ds.shift) - 1ull) : 0ull;
    uint32_t dig_rank[ITEMS];                                // digit | rank inside the tile's share of it << BITS
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t i = wbase + r * 64 + lane;
        dig_rank[r] = 0;
        if (i < n) {
            const uint32_t d = digit_of(key[r], ds);
            dig_rank[r] = d | (atomicAdd(&s_hist[d], 1u) << BITS);
        }
    }
    if (b != 0 && counts && t == 0) {                        // the totals are clear (workgroup 0 starts first: long done)
        uint32_t spins = 0;
        while (__hip_atomic_load((msd_gu64*)&flags[0], BESST_MSD_RLX) != nonce) {
            if (spins++ >= spin_limit) { s_fail = 1; break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    if (s_fail) {                                            // uniform
        if (t == 0) __hip_atomic_store((msd_gu64*)&flags[1], nonce, BESST_MSD_RLX);
        return;
    }
    // where the tile's share of every digit begins inside the digit's bucket: arrival order
    if (counts) {
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            const int d = q * kSortThreads + t;              // (a wave's 64 lanes on 64 consecutive counters)
            const uint32_t c = s_hist[d];
            const uint32_t base = c ? __hip_atomic_fetch_add((msd_gu32*)&total[d], c, BESST_MSD_RLX) : 0u;
            if (kSplit) __hip_atomic_store((msd_gu32*)&table[(size_t)b * RADIX + d], base, BESST_MSD_RLX);
            else s_base[d] = base;
        }
        if (kSplit) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the row is written through: the stores themselves)
        __syncthreads();                                     // every returning atomic of the tile has returned
        if (t == 0) __hip_atomic_store((msd_gu64*)&flags[2 + b], nonce, BESST_MSD_RLX);
        if (!scatters) return;
    }
    if (wave == 0) {                                         // wait for the counts of all tiles
        uint32_t spins = 0;
        for (;;) {
            bool ok = true;
            for (uint32_t j = lane; j < nb; j += 64)
                ok = ok && __hip_atomic_load((msd_gu64*)&flags[2 + j], BESST_MSD_RLX) == nonce;
            if (__all(ok)) break;
            if (__hip_atomic_load((msd_gu64*)&flags[1], BESST_MSD_RLX) == nonce || spins++ >= spin_limit) {
                if (lane == 0) s_fail = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    if (s_fail) {
        if (t == 0) __hip_atomic_store((msd_gu64*)&flags[1], nonce, BESST_MSD_RLX);
        return;
    }
    {   // bucket starts: exclusive scan of the totals (thread t: digits t * DPT ..)
        uint32_t tot[DPT];
        uint32_t run = 0;
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            tot[q] = __hip_atomic_load((msd_gu32*)&total[t * DPT + q], BESST_MSD_RLX);
            run += tot[q];
        }
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
            if (kSplit) s_base[t * DPT + q] = start + __hip_atomic_load((msd_gu32*)&table[(size_t)b * RADIX + t * DPT + q], BESST_MSD_RLX);
            else s_base[t * DPT + q] += start;
            if (b == 0) bucket_start[t * DPT + q] = start;
            start += tot[q];
        }
        if (b == 0 && t == kSortThreads - 1) bucket_start[RADIX] = start;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t i = wbase + r * 64 + lane;
        if (i < n) {
            const uint32_t dst = s_base[dig_rank[r] & (RADIX - 1)] + (dig_rank[r] >> BITS);
            // only the key bits below the digit travel in the word (bucket_reduce_kernel puts the digit back)
            keys_out[dst] = (((key[r] - ds.base) & low_mask) << packed_bits) | i;
            if (zero_n) { zero_n[i] = 0; zero_sum[i] = 0; zero_sum_sq[i] = 0; }   // rows <= tuples: the accumulators
        }
    }
}

__device__ __forceinline__ bool pair_less(uint64_t ka, uint32_t ia, uint64_t kb, uint32_t ib) {
    return ka < kb || (ka == kb && ia < ib);
}

template <typename KP, typename IP>
__device__ __forceinline__ void bitonic_pairs(KP k, IP x, int np) {
    for (int size = 2; size <= np; size <<= 1) {
        for (int j = size >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < np; i += blockDim.x) {
                const int p = i ^ j;
                if (p > i) {
                    const uint64_t ka = k[i], kb = k[p];
                    const uint32_t ia = x[i], ib = x[p];
                    const bool up = (i & size) == 0;
                    if (pair_less(kb, ib, ka, ia) == up) { k[i] = kb; k[p] = ka; x[i] = ib; x[p] = ia; }
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ void bitonic_words(uint64_t* k, int np) {          // unique 64-bit words, in LDS
    for (int size = 2; size <= np; size <<= 1) {
        for (int j = size >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < np; i += blockDim.x) {
                const int p = i ^ j;
                if (p > i) {
                    const uint64_t ka = k[i], kb = k[p];
                    const bool up = (i & size) == 0;
                    if ((kb < ka) == up) { k[i] = kb; k[p] = ka; }
                }
            }
            __syncthreads();
        }
    }
}

// kPacked: the words are key << idx_bits | stream index (unique), one compare per pair and no index array.
// kCap: LDS capacity in tuples.  512 serves the rare streams whose (key, index) pairs do not fit one word (rank sort
// only); 4096 serves every packed stream up to 4 M tuples (on C2's ~100-word buckets it is as fast as a dedicated
// 512-word variant was, 11 vs 12.5 us for the launch), whose larger buckets of several hundred
// to a few thousand words are first split in LDS by the next 8 key bits (counting sort: histogram, scan, grouped
// copy) and then rank-sorted inside each group - linear instead of the n^2 of a plain rank sort or the n log^2 n and
// ~50 barriers of a bitonic network (2 M tuples: 150 us with the network, 206 us with the plain rank sort).
#ifndef BESST_RANK_COST_LIMIT
#define BESST_RANK_COST_LIMIT 256
#endif
constexpr uint32_t kRankCostLimit = BESST_RANK_COST_LIMIT;   // LDS reads per word above which a bucket takes the network

template <bool kPacked, int kCap>
__global__ __launch_bounds__(kBucketThreads) void bucket_sort_kernel(uint64_t* __restrict__ keys, uint32_t* __restrict__ idx,
                                                          const uint32_t* __restrict__ bucket_start,
                                                          const uint32_t* __restrict__ n_ptr,
                                                          uint64_t* __restrict__ big_keys,
                                                          uint32_t* __restrict__ big_idx,
                                                          uint32_t* __restrict__ bucket_rows, int packed_bits,
                                                          int sub_bits /* key bits below the MSD digit */,
                                                          const unsigned long long* __restrict__ msd_flags,
                                                          unsigned long long nonce) {
    constexpr int kRank = kCap <= 512 ? kCap : 256;     // largest bucket the plain rank sort takes
    static_assert(kPacked || kCap <= 512, "unpacked pairs are only sorted in the small-stream configuration");
    static_assert(kBucketThreads == 256 || kCap <= 512, "the group scan of the two-level sort uses one thread per group");
    __shared__ uint64_t s_k[kCap <= 512 ? kCap : 256];   // rank-sort input (buckets up to kRank words)
    __shared__ uint32_t s_x[kPacked ? 1 : kCap];
    // packed: sorted copy, to count the bucket's distinct keys (and the grouped copy of the two-level sort)
    __shared__ uint64_t s_sorted[kPacked ? (kCap <= 512 ? kRank : kCap) : 1];
    __shared__ uint32_t s_cnt[kCap <= 512 ? 1 : 256], s_cur[kCap <= 512 ? 1 : 256];
    __shared__ uint32_t s_heads;
    // the three loads are issued together (one memory round trip); bucket_start is stale when nothing was partitioned
    const uint32_t n_all = *n_ptr;
    const uint32_t s0 = bucket_start[blockIdx.x], e0 = bucket_start[blockIdx.x + 1];
    if (msd_flags && msd_flags[1] == nonce) return;   // the partition gave up: bucket_start is not this call's
    const int n = n_all == 0 ? 0 : (int)(e0 - s0);
    if (n <= 1) {                     // nothing to sort; a single tuple is a single edge row
        if (kPacked && threadIdx.x == 0) bucket_rows[blockIdx.x] = (uint32_t)n;
        return;
    }
    if (threadIdx.x == 0) s_heads = 0;
    int np = 2;
    while (np < n) np <<= 1;
    uint32_t heads = 0;               // packed: distinct keys of the bucket = its edge rows (bucket_reduce_kernel)
    if (n <= kRank) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            s_k[i] = keys[s0 + i];
            if (!kPacked) s_x[i] = idx[s0 + i];
        }
        __syncthreads();
        // rank sort: element i goes to position #{j : (key_j, idx_j) < (key_i, idx_i)}.  O(n^2) LDS broadcast reads
        // (every lane reads the same j: conflict free) but a single barrier - for the ~100-tuple buckets of this path
        // that beats a bitonic network's 28-45 barrier-separated stages.
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const uint64_t ki = s_k[i];
            int rank = 0;
            if (kPacked) {
#pragma unroll 8
                for (int j = 0; j < n; ++j) rank += s_k[j] < ki ? 1 : 0;
                keys[s0 + rank] = ki;
                s_sorted[rank] = ki;
            } else {
                const uint32_t xi = s_x[i];
                for (int j = 0; j < n; ++j) rank += pair_less(s_k[j], s_x[j], ki, xi) ? 1 : 0;
                keys[s0 + rank] = ki;
                idx[s0 + rank] = xi;
            }
        }
        if (kPacked) {
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += blockDim.x)
                heads += (i == 0 || (s_sorted[i] >> packed_bits) != (s_sorted[i - 1] >> packed_bits)) ? 1u : 0u;
        }
    } else if (kPacked && kCap > 512 && n <= kCap) {
        // two levels: group by the next 8 key bits (the grouped copy is the only large LDS array; the words are read
        // from memory twice, the second time from cache), then rank inside the group
        const int sub_shift = packed_bits + (sub_bits > 8 ? sub_bits - 8 : 0);
        const uint32_t sub_mask = sub_bits >= 8 ? 255u : ((1u << sub_bits) - 1u);
        if (threadIdx.x < 256) s_cnt[threadIdx.x] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x)
            atomicAdd(&s_cnt[(uint32_t)(keys[s0 + i] >> sub_shift) & sub_mask], 1u);
        __syncthreads();
        {   // exclusive scan of the 256 group sizes (one per thread; kBucketThreads == 256)
            const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
            const uint32_t c = s_cnt[t];
            uint32_t x = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)x, d, 64);
                if (lane >= d) x += o;
            }
            __shared__ uint32_t s_wt[4];
            if (lane == 63) s_wt[wave] = x;
            __syncthreads();
            uint32_t pre = 0;
            for (int w = 0; w < wave; ++w) pre += s_wt[w];
            s_cur[t] = pre + x - c;                      // group start, then the running cursor of the grouped copy
            __syncthreads();
        }
        // Ranking inside a group costs its size per word: fine for the ~10-100 words that share 19 key bits on a
        // paired-end library, ruinous when one scaffold end carries hundreds of links (mate pairs at high
        // coverage: 0.6 ms for 1.7 M tuples).  Such a bucket takes a bitonic network over all its words instead.
        // The choice is made per bucket on the rank sort's cost, the sum of the squared group sizes, against ~256
        // LDS operations per word for the network.
        __shared__ uint32_t s_sumsq;
        if (threadIdx.x == 0) s_sumsq = 0;
        __syncthreads();
        if (s_cnt[threadIdx.x] > 1u) atomicAdd(&s_sumsq, s_cnt[threadIdx.x] * s_cnt[threadIdx.x]);
        __syncthreads();
        if (s_sumsq > kRankCostLimit * (uint32_t)n) {
            for (int i = threadIdx.x; i < np; i += blockDim.x) s_sorted[i] = i < n ? keys[s0 + i] : ~0ull;
            __syncthreads();
            bitonic_words(s_sorted, np);
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                keys[s0 + i] = s_sorted[i];
                heads += (i == 0 || (s_sorted[i] >> packed_bits) != (s_sorted[i - 1] >> packed_bits)) ? 1u : 0u;
            }
        } else {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const uint64_t w = keys[s0 + i];
            s_sorted[atomicAdd(&s_cur[(uint32_t)(w >> sub_shift) & sub_mask], 1u)] = w;
        }
        __syncthreads();                                 // now s_cur[d] = END of group d, s_cnt[d] its size
        for (int p = threadIdx.x; p < n; p += blockDim.x) {
            const uint64_t w = s_sorted[p];
            const uint64_t key_floor = (w >> packed_bits) << packed_bits;    // smallest word with the same key
            const uint32_t d = (uint32_t)(w >> sub_shift) & sub_mask;
            const int hi = (int)s_cur[d], lo = hi - (int)s_cnt[d];
            int rank = lo, below_key = lo;
#pragma unroll 8
            for (int q = lo; q < hi; ++q) {
                const uint64_t v = s_sorted[q];
                rank += v < w ? 1 : 0;
                below_key += v < key_floor ? 1 : 0;
            }
            keys[s0 + rank] = w;                         // every read of keys[] happened before the barrier above
            heads += rank == below_key ? 1u : 0u;        // no word of the same key precedes it: first of its edge row
        }
        }
    } else {
        // rare: bucket larger than LDS; padded copy at offset 2*s0 of a 2*capacity scratch (disjoint per bucket)
        uint64_t* gk = big_keys + 2 * (size_t)s0;
        uint32_t* gx = big_idx + 2 * (size_t)s0;
        for (int i = threadIdx.x; i < np; i += blockDim.x) {
            gk[i] = i < n ? keys[s0 + i] : ~0ull;
            gx[i] = kPacked ? 0u : (i < n ? idx[s0 + i] : ~0u);
        }
        __syncthreads();
        bitonic_pairs(gk, gx, np);
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            keys[s0 + i] = gk[i];
            if (!kPacked) idx[s0 + i] = gx[i];
            if (kPacked) heads += (i == 0 || (gk[i] >> packed_bits) != (gk[i - 1] >> packed_bits)) ? 1u : 0u;
        }
    }
    if (kPacked) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) heads += (uint32_t)__shfl_xor((int)heads, d, 64);
        if ((threadIdx.x & 63) == 0 && heads) atomicAdd(&s_heads, heads);
        __syncthreads();
        if (threadIdx.x == 0) bucket_rows[blockIdx.x] = s_heads;
    }
}

// One workgroup per bucket of the packed MSD path: the bucket's tuples are sorted and complete (all tuples of a key
// share the bucket), so its edge rows are local.  Row numbering needs only the row counts of the buckets before
// it (bucket_rows, written by bucket_sort_kernel): no head-count launch, no cross-workgroup row accumulation.
// Per chunk of 256 tuples: head flags -> row index (ballot scan), observation gather, run-segmented wave sums
// (one atomic triple per row and wave, on accumulators cleared by the histogram pass).
__global__ __launch_bounds__(256) void bucket_reduce_kernel(
    const uint64_t* __restrict__ words, const uint64_t* __restrict__ payload, const uint32_t* __restrict__ n_ptr,
    const uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ bucket_rows, int packed_bits, int msd_shift,
    uint32_t* __restrict__ n_rows, uint64_t* __restrict__ row_key, uint32_t* __restrict__ row_mask,
    uint32_t* __restrict__ row_n, unsigned long long* __restrict__ row_sum,
    unsigned long long* __restrict__ row_sum_sq, uint32_t* __restrict__ row_first, uint32_t* __restrict__ row_offset,
    int32_t* __restrict__ obs_lo, int32_t* __restrict__ obs_hi, const uint32_t* __restrict__ first_map,
    uint64_t key_base, const unsigned long long* __restrict__ msd_flags, unsigned long long nonce) {
    __shared__ uint32_t s_part[4];
    __shared__ int s_wheads[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t b = blockIdx.x;
    const uint32_t n_all = *n_ptr;
    const uint32_t s0 = bucket_start[b], e0 = bucket_start[b + 1];
    if (msd_flags && msd_flags[1] == nonce) {         // the partition gave up: say so instead of a row count
        if (b == 0 && t == 0) *n_rows = BESST_ROWS_SORT_FAILED;
        return;
    }
    if (n_all == 0) {
        if (b == 0 && t == 0) *n_rows = 0;
        return;
    }
    uint32_t pre = 0;
    for (uint32_t bb = t; bb < b; bb += 256) pre += bucket_rows[bb];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) pre += (uint32_t)__shfl_xor((int)pre, d, 64);
    if (lane == 0) s_part[wave] = pre;
    __syncthreads();
    const uint32_t base = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    if (b == gridDim.x - 1 && t == 0) *n_rows = base + bucket_rows[b];
    const int n = (int)(e0 - s0);
    if (n == 0) return;
    const uint64_t idx_mask = (1ull << packed_bits) - 1ull;
    const unsigned long long le_mask = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    int64_t rows_done = (int64_t)base;            // rows that start before the current chunk
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int j = c0 + t;
        const bool live = j < n;
        const uint32_t i = s0 + (uint32_t)j;
        const uint64_t w = live ? words[i] : ~0ull;
        const uint64_t key = ((uint64_t)b << msd_shift) | (w >> packed_bits);   // digit (= bucket) | low key bits
        const uint32_t src = (uint32_t)(w & idx_mask);
        const uint64_t pkey = (live && j > 0) ? (((uint64_t)b << msd_shift) | (words[i - 1] >> packed_bits)) : ~key;
        const bool head = live && (j == 0 || key != pkey);
        const unsigned long long hmask = __ballot(head);
        if (lane == 0) s_wheads[wave] = __popcll(hmask);
        __syncthreads();
        int wpre = 0, ctot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < wave) wpre += s_wheads[q];
            ctot += s_wheads[q];
        }
        // row of this item (rows are numbered by their heads; an item before the chunk's first head continues the
        // last row of the previous chunk)
        const int64_t row = rows_done + wpre + __popcll(hmask & le_mask) - 1;
        uint64_t pl = 0;
        if (live) pl = payload[src];
        const uint32_t lo = (uint32_t)pl, hi = (uint32_t)(pl >> 32);
        const int32_t o_lo = (int32_t)lo, o_hi = (int32_t)(hi & 0x3fffffffu);
        if (live) {
            obs_lo[i] = o_lo;
            obs_hi[i] = o_hi;
            if (head) {
                row_key[row] = key + key_base;
                row_mask[row] = hi >> 30;
                row_first[row] = first_map ? first_map[src] : src;
                row_offset[row] = i;
            }
        }
        // run-segmented wave sums: a run = the lanes of one row inside this wave
        {
            const unsigned long long starts = hmask | 1ull;                 // lane 0 starts a run (maybe a continued row)
            const int start = 63 - __clzll((long long)(starts & le_mask));
            uint32_t cn = live ? 1u : 0u;
            unsigned long long cs = live ? (unsigned long long)((long long)o_lo + o_hi) : 0ull;
            unsigned long long cq = cs * cs;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t on = (uint32_t)__shfl_up((int)cn, d, 64);
                const unsigned long long os = __shfl_up(cs, d, 64), oq = __shfl_up(cq, d, 64);
                if (lane - d >= start) { cn += on; cs += os; cq += oq; }
            }
            const unsigned long long lmask = __ballot(live);
            const bool tail = live && (lane == 63 || !((lmask >> (lane + 1)) & 1ull) || ((hmask >> (lane + 1)) & 1ull));
            if (tail) {
                atomicAdd(&row_n[row], cn);
                atomicAdd(&row_sum[row], cs);
                atomicAdd(&row_sum_sq[row], cq);
            }
        }
        rows_done += ctot;
        __syncthreads();                                                    // s_wheads is reused by the next chunk
    }
}

// ---------------------------------------------------------------------------------------------------
// segmented reduction into edge rows
// ---------------------------------------------------------------------------------------------------
constexpr int kRowScanFreeMaxBlocks = 2048;

__global__ __launch_bounds__(kRedThreads) void row_heads_kernel(const uint64_t* __restrict__ keys,
                                                                const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                                uint32_t* __restrict__ blk_heads, int packed_bits) {
    __shared__ int s_w[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t b = blockIdx.x;
    const uint32_t i0 = b * kRedTile + t * kRedItems;
    uint64_t k[kRedItems + 1];
    k[0] = (i0 > 0 && i0 - 1 < cap) ? keys[i0 - 1] >> packed_bits : 0ull;
#pragma unroll
    for (int j = 0; j < kRedItems; ++j) k[j + 1] = (i0 + j) < cap ? keys[i0 + j] >> packed_bits : 0ull;
    uint32_t n = *n_ptr;
    n = n < cap ? n : cap;
    if (b >= nblocks_of(n, kRedTile)) return;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < kRedItems; ++j) {
        const uint32_t i = i0 + j;
        if (i < n) cnt += (i == 0 || k[j + 1] != k[j]) ? 1 : 0;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
    if (lane == 0) s_w[wave] = cnt;
    __syncthreads();
    if (t == 0) blk_heads[b] = (uint32_t)(s_w[0] + s_w[1] + s_w[2] + s_w[3]);
}

__global__ __launch_bounds__(1024) void row_scan_kernel(const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                        const uint32_t* __restrict__ blk_heads,
                                                        uint32_t* __restrict__ blk_base,
                                                        uint32_t* __restrict__ n_rows) {
    uint32_t n = *n_ptr;
    n = n < cap ? n : cap;
    const uint32_t nb = nblocks_of(n, kRedTile);
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < nb; c0 += 1024) {
        const uint32_t i = c0 + t;
        const uint32_t v = i < nb ? blk_heads[i] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(x, d, 64);
            if (lane >= d) x += o;
        }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        uint32_t pre = s_carry;
        for (int w = 0; w < wave; ++w) pre += s_w[w];
        if (i < nb) blk_base[i] = pre + x - v;
        __syncthreads();
        if (t == 1023) s_carry = pre + x;
        __syncthreads();
    }
    if (t == 0) *n_rows = s_carry;
}

__global__ __launch_bounds__(kRedThreads) void row_reduce_kernel(
    const uint64_t* __restrict__ keys, const uint32_t* __restrict__ idx,
    const uint64_t* __restrict__ payload, const uint32_t* __restrict__ n_ptr, uint32_t cap,
    const uint32_t* __restrict__ blk_heads, const uint32_t* __restrict__ blk_base, int scanned,
    uint32_t* __restrict__ n_rows, uint64_t* __restrict__ row_key, uint32_t* __restrict__ row_mask,
    uint32_t* __restrict__ row_n, unsigned long long* __restrict__ row_sum,
    unsigned long long* __restrict__ row_sum_sq, uint32_t* __restrict__ row_first,
    uint32_t* __restrict__ row_offset, int32_t* __restrict__ obs_lo, int32_t* __restrict__ obs_hi,
    const uint32_t* __restrict__ first_map, int packed_bits) {
    __shared__ int s_w[4];
    __shared__ uint32_t s_pre[4], s_tot[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t b = blockIdx.x;
    const uint32_t i0 = b * kRedTile + t * kRedItems;
    uint64_t key[kRedItems];
    uint32_t src[kRedItems];
    // packed_bits > 0: the sorted words are key << packed_bits | stream index (no separate index array)
    const uint64_t idx_mask = packed_bits ? ((1ull << packed_bits) - 1ull) : 0ull;
    uint64_t prev = (i0 > 0 && i0 - 1 < cap) ? keys[i0 - 1] >> packed_bits : 0ull;
#pragma unroll
    for (int k = 0; k < kRedItems; ++k) {
        const uint32_t i = i0 + k;
        const uint64_t v = i < cap ? keys[i] : 0ull;
        key[k] = v >> packed_bits;
        src[k] = packed_bits ? (uint32_t)(v & idx_mask) : (i < cap ? idx[i] : 0u);
    }
    uint32_t n = *n_ptr;
    n = n < cap ? n : cap;
    const uint32_t nb = nblocks_of(n, kRedTile);
    if (b >= nb) return;
    uint32_t base;
    if (scanned) {
        base = blk_base[b];
    } else {   // few blocks: every workgroup adds up the head counts before it (block 0 also publishes the total)
        uint32_t pre = 0, tot = 0;
        for (uint32_t bb = t; bb < nb; bb += kRedThreads) {
            const uint32_t v = blk_heads[bb];
            tot += v;
            if (bb < b) pre += v;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { pre += __shfl_xor(pre, d, 64); tot += __shfl_xor(tot, d, 64); }
        if (lane == 0) { s_pre[wave] = pre; s_tot[wave] = tot; }
        __syncthreads();
        base = s_pre[0] + s_pre[1] + s_pre[2] + s_pre[3];
        if (b == 0 && t == 0) *n_rows = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
    }
    bool head[kRedItems];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < kRedItems; ++k) {
        const uint32_t i = i0 + k;
        head[k] = i < n && (i == 0 || key[k] != prev);
        cnt += head[k] ? 1 : 0;
        prev = key[k];
    }
    int x = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(x, d, 64);
        if (lane >= d) x += o;
    }
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    int pre = 0;
    for (int w = 0; w < wave; ++w) pre += s_w[w];
    // index of the row the thread's first item belongs to (rows are numbered by their heads)
    int64_t row = (int64_t)base + pre + x - cnt - 1;
    uint64_t pl[kRedItems];
#pragma unroll
    for (int k = 0; k < kRedItems; ++k) pl[k] = (i0 + k) < n ? payload[src[k]] : 0ull;
    // Per-thread runs: only a thread's LAST run can continue into the next thread, and only its FIRST run can
    // continue from the previous one.  Runs that start and end inside the thread are flushed directly; the open
    // run of every lane is then combined across the wave by row id (one atomic triple per distinct row of the
    // wave instead of one per lane: the per-lane version spent most of its time in device-scope atomics).
    uint32_t run_n = 0;
    unsigned long long run_s = 0, run_s2 = 0;
#pragma unroll
    for (int k = 0; k < kRedItems; ++k) {
        const uint32_t i = i0 + k;
        if (i >= n) break;
        if (head[k]) {
            if (run_n) {
                atomicAdd(&row_n[row], run_n);
                atomicAdd(&row_sum[row], run_s);
                atomicAdd(&row_sum_sq[row], run_s2);
                run_n = 0; run_s = 0; run_s2 = 0;
            }
            row++;
        }
        const uint32_t lo = (uint32_t)pl[k], hi = (uint32_t)(pl[k] >> 32);
        const int32_t o_lo = (int32_t)lo, o_hi = (int32_t)(hi & 0x3fffffffu);
        obs_lo[i] = o_lo;
        obs_hi[i] = o_hi;
        if (head[k]) {
            row_key[row] = key[k];
            row_mask[row] = hi >> 30;
            row_first[row] = first_map ? first_map[src[k]] : src[k];
            row_offset[row] = i;
        }
        const unsigned long long o = (unsigned long long)((long long)o_lo + o_hi);
        run_n += 1;
        run_s += o;
        run_s2 += o * o;
    }
    {
        bool active = run_n != 0;
        unsigned long long mask = __ballot(active);
        // rows are numbered in key order, so the open rows of a wave span [row of first active lane, row of last]:
        // with many short rows (PE libraries: ~15 links per edge) the combine loop would cost more than it saves
        int iter = 0;
        if (mask) {
            const int first_l = __ffsll((long long)mask) - 1, last_l = 63 - __clzll((long long)mask);
            const int span = __shfl((int)(row & 0xffffffff), last_l, 64) - __shfl((int)(row & 0xffffffff), first_l, 64);
            if (span >= 8) iter = 48;
        }
        while (mask) {
            if (iter++ >= 48) {              // many rows per wave: per-lane atomics
                if (active) {
                    atomicAdd(&row_n[row], run_n);
                    atomicAdd(&row_sum[row], run_s);
                    atomicAdd(&row_sum_sq[row], run_s2);
                }
                break;
            }
            const int leader = __ffsll((long long)mask) - 1;
            const long long r0 = __shfl((int)(row & 0xffffffff), leader, 64);
            const bool match = active && (int)(row & 0xffffffff) == (int)r0;
            uint32_t tn = match ? run_n : 0u;
            unsigned long long ts = match ? run_s : 0ull, ts2 = match ? run_s2 : 0ull;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                tn += (uint32_t)__shfl_xor((int)tn, d, 64);
                ts += ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(ts >> 32), d, 64) << 32) |
                      (uint32_t)__shfl_xor((int)(uint32_t)ts, d, 64);
                ts2 += ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(ts2 >> 32), d, 64) << 32) |
                       (uint32_t)__shfl_xor((int)(uint32_t)ts2, d, 64);
            }
            if (lane == leader) {
                atomicAdd(&row_n[row], tn);
                atomicAdd(&row_sum[row], ts);
                atomicAdd(&row_sum_sq[row], ts2);
            }
            active = active && !match;
            mask = __ballot(active);
        }
    }
}

struct RedWorkspace {
    uint64_t* keys[2];
    uint32_t* idx[2];
    uint32_t* table;
    uint32_t* row_total;
    uint32_t* blk_heads;
    uint32_t* blk_base;
    uint32_t* bucket_start;
    uint32_t* bucket_rows;
    unsigned long long* msd_flags;   // one-launch partition: cleared / failed / per-tile arrival words
    uint64_t* big_keys;
    uint32_t* big_idx;
    uint32_t stride;
    char* os_ws;          // chained-scan path (onesweep.hip)
    size_t os_bytes;
    char* rg_ws;          // run-grouped path (runs.hip)
    size_t rg_bytes;
    size_t total;
};

constexpr int kMaxRadix = 1 << 11;
// Largest stream (in sort tiles) that takes the MSD + bucket path when its words can be packed: 4 M tuples, i.e.
// buckets of ~2000 words, which still fit the two-level in-LDS sort (measured against the LSD passes: 300 k tuples
// 132 -> 47 us, 1 M 170 -> 78 us, 2 M 246 -> 132 us, 4 M 487 -> 249 us).  Up to 64 tiles the partition uses the
// scan-free table, beyond that the row scan.  (The same idea one size up - a wide partition of 32 768-tuple tiles and
// one workgroup per ~20 k-word bucket, counting sort into the ping-pong buffer + LDS rank batches - was built and is
// exact, but on real mate-pair data a group of equal 19-bit prefixes is a whole scaffold end with hundreds of
// links, and ranking inside such groups made full C3 10.5 ms against 5.3 ms with the LSD passes.)
#ifndef BESST_MSD_MAX_BLOCKS
#define BESST_MSD_MAX_BLOCKS 1024
#endif
constexpr int kMsdMaxBlocks = BESST_MSD_MAX_BLOCKS;
// Streams of up to this many sort tiles take the MSD pass with kMsdSmallItems keys per thread (build knob, 0 =
// never).  Measured on C2 (34 sort tiles) with 64 / 8: the scatter drops from 15 to 10 us, the row scan the 68 small
// tiles need adds 6.5: step 112.6 vs 113.8 - 115.1 us.  Inside the noise of a default, so it stays off.
#ifndef BESST_MSD_SMALL_MAX_BLOCKS
#define BESST_MSD_SMALL_MAX_BLOCKS 0
#endif
#ifndef BESST_MSD_SMALL_ITEMS
#define BESST_MSD_SMALL_ITEMS 8
#endif
constexpr int kMsdSmallMaxBlocks = BESST_MSD_SMALL_MAX_BLOCKS;
constexpr int kMsdSmallItems = BESST_MSD_SMALL_ITEMS;
// Digit width of the LSD passes large streams take (build knob).  C3 slice of 8.5 M tuples (37-bit keys): 10 bits
// = 4 passes instead of 5, but the per-tile table of 1024 counters is written as scattered 4-byte words and the
// histogram launches go from 0.127 to 0.174 ms in total, the scatter stays at 0.22: step 1.24 -> 1.29 ms (11 bits:
// 1.44 ms).  Fewer passes only pay once the per-tile table goes (single-pass chained scan).
#ifndef BESST_LSD_BITS
#define BESST_LSD_BITS 8
#endif
constexpr int kLsdBits = BESST_LSD_BITS;

RedWorkspace carve(void* ws, int64_t cap) {
    RedWorkspace w;
    char* p = static_cast<char*>(ws);
    size_t off = 0;
    const size_t nb_sort = (size_t)((cap + kSortTile - 1) / kSortTile);
    const size_t nb_red = (size_t)((cap + kRedTile - 1) / kRedTile);
    for (int j = 0; j < 2; ++j) { w.keys[j] = reinterpret_cast<uint64_t*>(p + off); off += align_up((size_t)cap * 8, 256); }
    for (int j = 0; j < 2; ++j) { w.idx[j] = reinterpret_cast<uint32_t*>(p + off); off += align_up((size_t)cap * 4, 256); }
    w.stride = (uint32_t)nb_sort;
    // the wide-digit path is only taken for small streams, the 8-bit path for any size
    const size_t wide_max = (size_t)(kWideDigitMaxBlocks > kMsdMaxBlocks ? kWideDigitMaxBlocks : kMsdMaxBlocks);
    // (rows: the tiles of the MSD pass, which may be smaller than a sort tile - kMsdSmallItems or kMsdItems keys per thread)
    constexpr size_t kRowsPerSortTile = kSortItems / (kMsdSmallItems < kMsdItems ? kMsdSmallItems : kMsdItems);
    const size_t table_entries = nb_sort <= wide_max
                                     ? (nb_sort * kRowsPerSortTile > (size_t)kScanFreeMaxBlocks
                                            ? nb_sort * kRowsPerSortTile : (size_t)kScanFreeMaxBlocks) * kMaxRadix
                                     : nb_sort * (size_t)(1 << kLsdBits);
    w.table = reinterpret_cast<uint32_t*>(p + off); off += align_up(table_entries * 4, 256);
    w.row_total = reinterpret_cast<uint32_t*>(p + off); off += align_up(kMaxRadix * 4, 256);
    w.blk_heads = reinterpret_cast<uint32_t*>(p + off); off += align_up(nb_red * 4, 256);
    w.blk_base = reinterpret_cast<uint32_t*>(p + off); off += align_up(nb_red * 4, 256);
    w.bucket_start = reinterpret_cast<uint32_t*>(p + off); off += align_up((kMaxRadix + 1) * 4, 256);
    w.bucket_rows = reinterpret_cast<uint32_t*>(p + off); off += align_up(kMaxRadix * 4, 256);
    w.msd_flags = reinterpret_cast<unsigned long long*>(p + off); off += align_up((size_t)(2 + kMsdMaxBlocks * (kSortItems / kMsdItems)) * 8, 256);
    // in-place scratch of the bucket sort (only streams that take the MSD path can use it)
    const size_t big = nb_sort <= (size_t)kMsdMaxBlocks ? 2 * (size_t)cap + 8 : 8;
    w.big_keys = reinterpret_cast<uint64_t*>(p + off); off += align_up(big * 8, 256);
    w.big_idx = reinterpret_cast<uint32_t*>(p + off); off += align_up(big * 4, 256);
    w.os_ws = p + off;
    w.os_bytes = nb_sort > (size_t)kScanFreeMaxBlocks ? onesweep_workspace_bytes(cap) : 0;
    off += align_up(w.os_bytes, 256);
    w.rg_ws = p + off;
    // (streams beyond the MSD path's 4 M tuples; the runs themselves are at most that many, so the reduction the
    // run-grouped form runs on them never asks for a run workspace of its own)
    w.rg_bytes = (nb_sort > (size_t)kMsdMaxBlocks && runs_enabled(cap)) ? runs_workspace_bytes(cap) : 0;
    off += align_up(w.rg_bytes, 256);
    w.total = off;
    return w;
}

// one LSD pass over `bits`-bit digits selected by ds; zero_* != nullptr on the last pass
template <int BITS, int ITEMS = kSortItems>
void launch_pass(hipStream_t s, const RedWorkspace& w, uint32_t nb_sort, uint32_t cap, const uint32_t* n_tuples,
                 DigitSel ds, bool first, const uint64_t* kin, const uint32_t* iin, uint64_t* kout, uint32_t* iout,
                 uint32_t* zero_n, unsigned long long* zero_sum, unsigned long long* zero_sum_sq,
                 uint32_t* bucket_start = nullptr, int packed_bits = 0) {
    // nb_sort counts tiles of kSortThreads * ITEMS keys; the scanned table's rows are that long
    const int scanned = nb_sort > (uint32_t)kScanFreeMaxBlocks ? 1 : 0;
    const uint32_t stride = ITEMS == kSortItems ? w.stride : nb_sort;
    {
        ProfScope ps(s, kProfRadixHist);
        hipLaunchKernelGGL((radix_hist_kernel<BITS, ITEMS>), dim3(nb_sort), dim3(kSortThreads), 0, s, kin, n_tuples, cap, ds,
                           w.table, stride, scanned, zero_n, zero_sum, zero_sum_sq);
    }
    if (scanned) {
        ProfScope ps(s, kProfRadixScan);
        hipLaunchKernelGGL(radix_rowscan_kernel, dim3(1u << BITS), dim3(256), 0, s, n_tuples, cap, w.table, stride,
                           w.row_total, (uint32_t)(kSortThreads * ITEMS));
    }
    ProfScope ps(s, kProfRadixScatter);
    if (first)
        hipLaunchKernelGGL((radix_scatter_kernel<BITS, true, ITEMS>), dim3(nb_sort), dim3(kSortThreads), 0, s, kin, iin,
                           n_tuples, cap, ds, w.table, stride, w.row_total, scanned, kout, iout, bucket_start,
                           packed_bits);
    else
        hipLaunchKernelGGL((radix_scatter_kernel<BITS, false, ITEMS>), dim3(nb_sort), dim3(kSortThreads), 0, s, kin, iin,
                           n_tuples, cap, ds, w.table, stride, w.row_total, scanned, kout, iout, bucket_start,
                           packed_bits);
}

}  // namespace

size_t reduce_workspace_bytes(int64_t cap) {
    if (cap < 1) cap = 1;
    return carve(nullptr, cap).total;
}


// which streams launch_sort_reduce hands to the chained-scan passes (the test of its large-stream branch)
static bool takes_large_stream_path(int64_t cap, int key_bits) {
    const uint32_t nb_sort = (uint32_t)((cap + kSortTile - 1) / kSortTile);
    int cap_idx_bits = 1;
    while (((int64_t)1 << cap_idx_bits) < cap) ++cap_idx_bits;
    const bool packable = (key_bits > kMsdBits ? key_bits - kMsdBits : 0) + cap_idx_bits <= 64;
    return !(nb_sort <= (uint32_t)kScanFreeMaxBlocks || (packable && nb_sort <= (uint32_t)kMsdMaxBlocks));
}

bool sort_presort_spec(int64_t cap, int key_bits, uint64_t key_base, void* ws, size_t ws_bytes, PresortSpec* out) {
    if (!out || !ws || cap < 1 || cap >= ((int64_t)1 << 32) || key_bits < 1 || key_bits > 64) return false;
    const RedWorkspace w = carve(ws, cap);
    if (ws_bytes < w.total || !takes_large_stream_path(cap, key_bits)) return false;
    return onesweep_presort_spec(cap, key_bits, key_base, w.os_ws, out);
}

int launch_sort_reduce(hipStream_t s, int64_t cap, const uint32_t* n_tuples, int key_bits,
                       const uint64_t* keys, const uint64_t* payload, uint64_t* row_key,
                       uint32_t* row_mask, uint32_t* row_n, int64_t* row_sum, int64_t* row_sum_sq,
                       uint32_t* row_first, uint32_t* row_offset, int32_t* obs_lo, int32_t* obs_hi,
                       uint32_t* n_rows, void* ws, size_t ws_bytes, const uint32_t* first_map, uint64_t key_base,
                       bool hist_ready, const SegSource* seg, uint32_t flags) {
    // key_bits counts the significant bits of key - key_base: every path below sorts that difference (the order is
    // the same) and puts key_base back when it writes a row's key
    BESST_REQUIRE(cap >= 0 && cap < ((int64_t)1 << 32), "reduce: capacity out of range");
    BESST_REQUIRE(key_bits >= 1 && key_bits <= 64, "reduce: key_bits out of range");
    if (cap == 0) {
        BESST_HIP_TRY(hipMemsetAsync(n_rows, 0, sizeof(uint32_t), s));
        return BESST_OK;
    }
    const RedWorkspace w = carve(ws, cap);
    BESST_REQUIRE(ws != nullptr && ws_bytes >= w.total, "reduce: workspace too small");
    BESST_REQUIRE(!seg || takes_large_stream_path(cap, key_bits), "reduce: a segmented stream needs the large-stream sort");
    const uint32_t nb_sort = (uint32_t)((cap + kSortTile - 1) / kSortTile);
    const uint32_t nb_red = (uint32_t)((cap + kRedTile - 1) / kRedTile);
    auto* zsum = reinterpret_cast<unsigned long long*>(row_sum);
    auto* zsq = reinterpret_cast<unsigned long long*>(row_sum_sq);
    const uint64_t* kin = keys;
    const uint32_t* iin = nullptr;
    int packed_bits = 0;
    int cap_idx_bits = 1;
    while (((int64_t)1 << cap_idx_bits) < cap) ++cap_idx_bits;
    // the MSD digit is implied by the bucket, so only the key bits below it have to fit next to the index
    const bool packable = (key_bits > kMsdBits ? key_bits - kMsdBits : 0) + cap_idx_bits <= 64;
    if (nb_sort <= (uint32_t)kScanFreeMaxBlocks || (packable && nb_sort <= (uint32_t)kMsdMaxBlocks)) {
        // one MSD pass on the top 11 significant bits, then every bucket sorts and reduces itself
        const int shift = key_bits > kMsdBits ? key_bits - kMsdBits : 0;
        const DigitSel ds{0, shift, 0, 1u, kMsdBits, key_base};
        packed_bits = packable ? cap_idx_bits : 0;    // key and stream index in one word (always, in practice)
        const bool one_launch = packed_bits != 0 && msd_one_launch() != 0 && nb_sort <= (uint32_t)kMsdMaxBlocks;
        const unsigned long long nonce = one_launch ? msd_next_nonce() : 0ull;
        const unsigned long long* mflags = one_launch ? w.msd_flags : nullptr;
        if (one_launch) {
            ProfScope ps(s, kProfMsdPartition);
            const uint32_t nb_part = (uint32_t)((cap + kSortThreads * kMsdItems - 1) / (kSortThreads * kMsdItems));
            if (nb_part <= kMsdMutualTiles)
                hipLaunchKernelGGL((msd_partition_kernel<kMsdItems, false>), dim3(nb_part), dim3(kSortThreads), 0, s, keys,
                                   n_tuples, (uint32_t)cap, ds, w.row_total, w.msd_flags, nonce, msd_spin_limit(), w.keys[0],
                                   w.bucket_start, packed_bits, row_n, zsum, zsq, w.table);
            else
                hipLaunchKernelGGL((msd_partition_kernel<kMsdItems, true>), dim3(2u * nb_part), dim3(kSortThreads), 0, s, keys,
                                   n_tuples, (uint32_t)cap, ds, w.row_total, w.msd_flags, nonce, msd_spin_limit(), w.keys[0],
                                   w.bucket_start, packed_bits, row_n, zsum, zsq, w.table);
        } else if (nb_sort <= (uint32_t)kMsdSmallMaxBlocks) {
            // few sort tiles leave most of the chip idle: half-size tiles (and the row scan they then need)
            const uint32_t nb_small = (uint32_t)((cap + kSortThreads * kMsdSmallItems - 1) / (kSortThreads * kMsdSmallItems));
            launch_pass<kMsdBits, kMsdSmallItems>(s, w, nb_small, (uint32_t)cap, n_tuples, ds, true, keys, nullptr,
                                                  w.keys[0], w.idx[0], row_n, zsum, zsq, w.bucket_start, packed_bits);
        } else {
            launch_pass<kMsdBits>(s, w, nb_sort, (uint32_t)cap, n_tuples, ds, true, keys, nullptr, w.keys[0], w.idx[0],
                                  row_n, zsum, zsq, w.bucket_start, packed_bits);
        }
        if (shift > 0 || packed_bits) {   // the packed partition is unstable: buckets always need their sort
            ProfScope ps(s, kProfBucketSort);
            const dim3 grid(1u << kMsdBits), block(kBucketThreads);
            if (!packed_bits)
                hipLaunchKernelGGL((bucket_sort_kernel<false, kBucketLds>), grid, block, 0, s, w.keys[0], w.idx[0],
                                   w.bucket_start, n_tuples, w.big_keys, w.big_idx, w.bucket_rows, 0, shift, mflags, nonce);
            else
                hipLaunchKernelGGL((bucket_sort_kernel<true, 4096>), grid, block, 0, s, w.keys[0], w.idx[0],
                                   w.bucket_start, n_tuples, w.big_keys, w.big_idx, w.bucket_rows, packed_bits, shift, mflags, nonce);
        }
        if (packed_bits) {
            // rows are local to a bucket: one launch instead of head counts + tile-based reduction
            ProfScope ps(s, kProfBucketReduce);
            hipLaunchKernelGGL(bucket_reduce_kernel, dim3(1u << kMsdBits), dim3(256), 0, s, w.keys[0], payload, n_tuples,
                               w.bucket_start, w.bucket_rows, packed_bits, shift, n_rows, row_key, row_mask, row_n, zsum, zsq,
                               row_first, row_offset, obs_lo, obs_hi, first_map, key_base, mflags, nonce);
            BESST_HIP_TRY(hipGetLastError());
            return BESST_OK;
        }
        kin = w.keys[0];
        iin = w.idx[0];
    } else {
        // large streams whose keys cluster (any coordinate-sorted library): chunks of the stream -> runs of equal keys,
        // and only the runs are sorted (runs.hip); a stream that overflows the run buffers says so in *n_rows and the
        // caller repeats the call with BESST_REDUCE_NO_RUNS
        if (w.rg_bytes && !(flags & BESST_REDUCE_NO_RUNS))
            return launch_runs_reduce(s, cap, n_tuples, key_bits, keys, payload, seg, seg ? seg->payload_out : w.keys[0],
                                      onesweep_staged_rows(w.os_ws, cap), row_key, row_mask, row_n, row_sum, row_sum_sq,
                                      row_first, row_offset, obs_lo, obs_hi, n_rows, w.rg_ws, w.rg_bytes, first_map, key_base);
        // else: one histogram read, chained-scan partition passes, atomic-free row reduction (onesweep.hip)
        uint64_t* bk[2] = {w.keys[0], w.keys[1]};
        uint32_t* bi[2] = {w.idx[0], w.idx[1]};
        return launch_onesweep_sort_reduce(s, cap, n_tuples, key_bits, keys, payload, bk, bi, row_key, row_mask, row_n,
                                           row_sum, row_sum_sq, row_first, row_offset, obs_lo, obs_hi, n_rows, w.os_ws,
                                           w.os_bytes, first_map, key_base, hist_ready, seg);
    }
    const int rscanned = nb_red > (uint32_t)kRowScanFreeMaxBlocks ? 1 : 0;
    {
        ProfScope ps(s, kProfRowHeads);
        hipLaunchKernelGGL(row_heads_kernel, dim3(nb_red), dim3(kRedThreads), 0, s, kin, n_tuples, (uint32_t)cap,
                           w.blk_heads, packed_bits);
    }
    if (rscanned) {
        ProfScope ps(s, kProfRowScan);
        hipLaunchKernelGGL(row_scan_kernel, dim3(1), dim3(1024), 0, s, n_tuples, (uint32_t)cap, w.blk_heads, w.blk_base,
                           n_rows);
    }
    ProfScope ps(s, kProfRowReduce);
    hipLaunchKernelGGL(row_reduce_kernel, dim3(nb_red), dim3(kRedThreads), 0, s, kin, iin, payload, n_tuples,
                       (uint32_t)cap, w.blk_heads, w.blk_base, rscanned, n_rows, row_key, row_mask, row_n, zsum, zsq,
                       row_first, row_offset, obs_lo, obs_hi, first_map, packed_bits);
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

// ---------------------------------------------------------------------------------------------------
// multi-GPU exchange: stable partition of the tuple stream by owner rank, packed into fixed-capacity
// regions (one per destination) for a single equal-split all-to-all, and the matching unpack.
//   region d = [ header 64 B | keys pair_cap x 8 | payload pair_cap x 8 | idx pair_cap x 4 ]
//   header   = { count sent (<= pair_cap), tuples the source wanted to send, source's total n_out, 0.. }
// Stable partitioning keeps each source's tuples in BAM order; regions arrive ordered by source rank and
// ranks own contiguous slices of the stream, so the concatenation on the receiver is again in global BAM
// order.  idx carries the tuple's position in its source's stream; the receiver turns it into a global
// emit index (first-occurrence order across ranks).
// ---------------------------------------------------------------------------------------------------
namespace {

// Stable partition of the ordered tuple stream by owner rank, written straight into the all-to-all regions
// (keys, payload and stream index of every tuple; the header by tile 0).  Same ranking as radix_scatter_kernel
// (wave match-any on the owner byte, per-wave counters in LDS, per-tile counts from radix_hist_kernel in owner
// mode), but a tuple's destination is its rank INSIDE its owner's region, and the payload travels with the key:
// the thread that ranks tuple i copies payload[i] (coalesced), so no sorted copy and no gather are needed.
template <int ITEMS>
__global__ __launch_bounds__(kSortThreads) void partition_scatter_kernel(
    const uint64_t* __restrict__ keys_in, const uint64_t* __restrict__ payload, const uint32_t* __restrict__ n_ptr,
    uint32_t cap, DigitSel ds, const uint32_t* __restrict__ table, uint32_t stride,
    const uint32_t* __restrict__ row_total, int scanned, uint32_t world, uint32_t pair_cap,
    char* __restrict__ send, size_t region_bytes, uint32_t tile_blocks, const uint64_t* __restrict__ rider,
    uint32_t rider_words, size_t rider_offset, const int32_t* __restrict__ slice_info) {
    // slice_info (stitch_kernel's speculative mode): copied into words 4..11 of every region header; the thread that
    // places the slice head's provisional tuple tells every destination where it went (words 12, 13: owner, position
    // inside the owner's region), so that the owner can drop it without a search if it turns out to be a duplicate.
    constexpr int RADIX = kRadix;                 // one counter per thread (world <= 256)
    __shared__ uint32_t s_whist[4][RADIX];
    __shared__ uint32_t s_base[RADIX];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (blockIdx.x >= tile_blocks) {
        // the blocks behind the tiles copy the rider (coverage numerators + counters of this rank) behind the
        // tuples of EVERY region: the receivers sum the riders of all sources, which saves the all-reduce
        const uint32_t nb_r = gridDim.x - tile_blocks;
        for (uint32_t i = (blockIdx.x - tile_blocks) * kSortThreads + t; i < rider_words; i += nb_r * kSortThreads) {
            const uint64_t v = rider[i];
            for (uint32_t d = 0; d < world; ++d)
                reinterpret_cast<uint64_t*>(send + (size_t)d * region_bytes + rider_offset)[i] = v;
        }
        return;
    }
    const uint32_t b = blockIdx.x;
    const uint32_t wbase = b * (kSortThreads * ITEMS) + wave * (ITEMS * 64);
    uint64_t key[ITEMS], pl[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t i = wbase + r * 64 + lane;
        key[r] = i < cap ? keys_in[i] : ~0ull;
        pl[r] = i < cap ? payload[i] : 0ull;
    }
    uint32_t n = *n_ptr;
    n = n < cap ? n : cap;
    const uint32_t nb = nblocks_of(n, (kSortThreads * ITEMS));
    if (nb == 0 && b == 0 && (uint32_t)t < world) {          // empty stream: headers only
        uint32_t* hdr = reinterpret_cast<uint32_t*>(send + (size_t)t * region_bytes);
        hdr[0] = 0; hdr[1] = 0; hdr[2] = 0; hdr[3] = 0;
        if (slice_info)
            for (int k = 0; k < 8; ++k) hdr[4 + k] = (uint32_t)slice_info[k];
    }
    if (b >= nb) return;
#pragma unroll
    for (int w = 0; w < 4; ++w) s_whist[w][t] = 0;
    {
        uint32_t pre = 0, tot = 0;
        if (scanned) {
            tot = row_total[t];
            pre = table[(uint32_t)t * stride + b];
        } else {
            for (uint32_t bb = 0; bb < nb; ++bb) {           // <= 64 rows, all loads independent
                const uint32_t x = table[bb * RADIX + t];
                tot += x;
                if (bb < b) pre += x;
            }
        }
        s_base[t] = pre;
        if (b == 0 && (uint32_t)t < world) {
            uint32_t* hdr = reinterpret_cast<uint32_t*>(send + (size_t)t * region_bytes);
            hdr[0] = tot < pair_cap ? tot : pair_cap;        // tuples sent
            hdr[1] = tot;                                    // tuples the source wanted to send (overflow check)
            hdr[2] = n;                                      // the source's total: base of the next source's indexes
            hdr[3] = 0;
            if (slice_info)
                for (int k = 0; k < 8; ++k) hdr[4 + k] = (uint32_t)slice_info[k];
        }
    }
    __syncthreads();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t dig_rank[ITEMS];   // owner | rank << 8
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t i = wbase + r * 64 + lane;
        const bool valid = i < n;
        const uint32_t d = valid ? digit_of(key[r], ds) : (uint32_t)(RADIX - 1);
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < kRadixBits; ++bit) {
            const bool one = (d >> bit) & 1u;
            const unsigned long long bal = __ballot(one);
            peers &= one ? bal : ~bal;
        }
        uint32_t pre = 0;
        const int leader = __ffsll((long long)peers) - 1;
        if (valid && lane == leader) {
            volatile uint32_t* slot = &s_whist[wave][d];
            pre = *slot;
            *slot = pre + (uint32_t)__popcll(peers);
        }
        pre = __shfl(pre, leader < 0 ? 0 : leader, 64);
        dig_rank[r] = d | ((pre + (uint32_t)__popcll(peers & lt_mask)) << kRadixBits);
    }
    __syncthreads();
    {
        uint32_t run = s_base[t];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t c = s_whist[w][t];
            s_whist[w][t] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t i = wbase + r * 64 + lane;
        if (i < n) {
            const uint32_t d = dig_rank[r] & (RADIX - 1);
            const uint32_t p = s_whist[wave][d] + (dig_rank[r] >> kRadixBits);
            if (slice_info && (int32_t)i == slice_info[7])   // the slice head's provisional tuple
                for (uint32_t d2 = 0; d2 < world; ++d2) {
                    uint32_t* hdr = reinterpret_cast<uint32_t*>(send + (size_t)d2 * region_bytes);
                    hdr[12] = d; hdr[13] = p;
                }
            if (p < pair_cap) {                              // overflow is reported through hdr[1] > hdr[0]
                char* region = send + (size_t)d * region_bytes + 64;
                reinterpret_cast<uint64_t*>(region)[p] = key[r];
                reinterpret_cast<uint64_t*>(region + (size_t)pair_cap * 8)[p] = pl[r];
                reinterpret_cast<uint32_t*>(region + (size_t)pair_cap * 16)[p] = i;
            }
        }
    }
}

// Speculative heads (spec != 0, stitch_kernel's slice_info in the headers): no tail exchange ran before the emit
// stage, so every source left its first reaching record unresolved.  Each receiver replays the chain over the
// sources - the prev_obs entering source s is the tail of the nearest earlier source that reached CreateEdge, the
// reference's initial (-1, -1) for the first - applies CreateEdge to each head (CreateGraph.py:835-870), corrects
// the (already summed) counters by the heads' contributions and, where a head turns out to be a duplicate, drops its
// provisional tuple: the owner finds it at the position the source wrote into the header, every receiver shifts the
// global emit indexes behind it.  All ranks compute the same corrections from the same headers.
__global__ __launch_bounds__(256) void unpack_kernel(const char* __restrict__ recv, uint32_t world, uint32_t pair_cap,
                                                     size_t region_bytes, uint64_t* __restrict__ keys,
                                                     uint64_t* __restrict__ payload, uint32_t* __restrict__ gidx,
                                                     uint32_t* __restrict__ n_out, uint32_t* __restrict__ overflow,
                                                     uint32_t tuple_blocks, unsigned long long* __restrict__ rider_sum,
                                                     uint32_t rider_words, size_t rider_offset, int spec,
                                                     uint32_t my_rank, int detect, int32_t* __restrict__ all_info,
                                                     unsigned long long* __restrict__ counters) {
    __shared__ uint32_t s_off[257], s_base[257], s_cnt[256];
    __shared__ int32_t s_drop_local[256], s_drop_src[256];
    __shared__ long long s_delta[8];              // corrections of the 8 summed counter words
    if (threadIdx.x == 0) {
        uint32_t run = 0, gb = 0, over = 0;
        int32_t p1 = -1, p2 = -1;                 // Parameter.counters: prev_obs1 = prev_obs2 = -1
        long long dc = 0, dl = 0, dd = 0, dn = 0, dt = 0;
        for (uint32_t sidx = 0; sidx < world; ++sidx) {
            const uint32_t* hdr = reinterpret_cast<const uint32_t*>(recv + (size_t)sidx * region_bytes);
            uint32_t cnt = hdr[0], tot = hdr[2];
            int32_t drop_local = -1, drop_src = -1;
            if (spec) {
                const bool any = hdr[4] != 0, head = hdr[7] != 0;
                const uint32_t info = hdr[10];
                if (head) {
                    const CEDelta d = create_edge((int32_t)hdr[8], (int32_t)hdr[9], p1, p2, info & 1u, info & 2u,
                                                  info & 4u, detect != 0);
                    dc += d.count; dl += d.too_long; dd += d.dup; dn += d.nus;
                    if ((info & 8u) && !d.keep) {            // the provisional tuple has to go
                        drop_src = (int32_t)hdr[11];
                        if (hdr[12] == my_rank && hdr[13] < cnt) drop_local = (int32_t)hdr[13];
                        dt -= 1;
                    }
                }
                if (any) { p1 = (int32_t)hdr[5]; p2 = (int32_t)hdr[6]; }
                if (all_info && blockIdx.x == 0 && blockIdx.y == 0)
                    for (int k = 0; k < 8; ++k) all_info[sidx * 8 + k] = (int32_t)hdr[4 + k];
            }
            s_off[sidx] = run;
            s_base[sidx] = gb;
            s_cnt[sidx] = cnt;
            s_drop_local[sidx] = drop_local;
            s_drop_src[sidx] = drop_src;
            run += cnt - (drop_local >= 0 ? 1u : 0u);
            gb += tot - (drop_src >= 0 ? 1u : 0u);
            over |= hdr[1] > hdr[0] ? 1u : 0u;
        }
        s_off[world] = run;
        // besst_counters words: count 0, non_unique 1, non_unique_for_scaf 2, nr_of_duplicates 3, too_long 4, fishy 5,
        // n_tuples 6, n_reach 7
        s_delta[0] = dc; s_delta[1] = 0; s_delta[2] = dn; s_delta[3] = dd; s_delta[4] = dl; s_delta[5] = 0;
        s_delta[6] = dt; s_delta[7] = 0;
        if (blockIdx.x == 0 && blockIdx.y == 0) {
            *n_out = run;
            if (over) *overflow = 1u;
            if (spec && !rider_words)              // counters were all-reduced before: add the corrections here
                for (int k = 0; k < 8; ++k)
                    if (s_delta[k]) atomicAdd(&counters[k], (unsigned long long)s_delta[k]);
        }
    }
    __syncthreads();
    if (blockIdx.x >= tuple_blocks) {             // riders: sum over the sources, written over the local values
        if (blockIdx.y != 0) return;
        const uint32_t nb_r = gridDim.x - tuple_blocks;
        for (uint32_t i = (blockIdx.x - tuple_blocks) * blockDim.x + threadIdx.x; i < rider_words;
             i += nb_r * blockDim.x) {
            unsigned long long acc = 0;
            for (uint32_t sidx = 0; sidx < world; ++sidx)
                acc += reinterpret_cast<const unsigned long long*>(recv + (size_t)sidx * region_bytes + rider_offset)[i];
            if (spec && i + 8 >= rider_words) acc += (unsigned long long)s_delta[i + 8 - rider_words];   // the counter words
            rider_sum[i] = acc;
        }
        return;
    }
    const uint32_t sidx = blockIdx.y;
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= s_cnt[sidx]) return;
    const int32_t dl = s_drop_local[sidx], ds = s_drop_src[sidx];
    if ((int32_t)p == dl) return;
    const char* region = recv + (size_t)sidx * region_bytes + 64;
    const uint32_t dst = s_off[sidx] + p - (dl >= 0 && (int32_t)p > dl ? 1u : 0u);
    const uint32_t idx = reinterpret_cast<const uint32_t*>(region + (size_t)pair_cap * 16)[p];
    keys[dst] = reinterpret_cast<const uint64_t*>(region)[p];
    payload[dst] = reinterpret_cast<const uint64_t*>(region + (size_t)pair_cap * 8)[p];
    gidx[dst] = s_base[sidx] + idx - (ds >= 0 && (int32_t)idx > ds ? 1u : 0u);
}

}  // namespace

size_t exchange_region_bytes(int64_t pair_cap) { return 64 + (size_t)pair_cap * 20; }
// a region followed by a rider of `rider_bytes` (0: none); the rider starts 8-byte aligned behind the tuples
size_t exchange_stride_bytes(int64_t pair_cap, int64_t rider_bytes) {
    return align_up(exchange_region_bytes(pair_cap), 8) + align_up((size_t)rider_bytes, 8);
}

int launch_partition(hipStream_t s, int64_t cap, const uint32_t* n_tuples, int node_bits, int world,
                     const uint64_t* keys, const uint64_t* payload, int64_t pair_cap, void* send, void* ws,
                     size_t ws_bytes, const void* rider, int64_t rider_bytes, const int32_t* slice_info) {
    BESST_REQUIRE(rider_bytes >= 0 && (rider_bytes & 7) == 0 && rider_bytes < ((int64_t)1 << 31) &&
                      (rider_bytes == 0 || rider), "partition: bad rider");
    BESST_REQUIRE(world >= 1 && world <= 256, "partition: world size must be in [1, 256]");
    BESST_REQUIRE(pair_cap > 0 && (pair_cap & 1) == 0, "partition: pair capacity must be a positive even number");
    BESST_REQUIRE(cap >= 0 && cap < ((int64_t)1 << 32), "partition: capacity out of range");
    const RedWorkspace w = carve(ws, cap > 0 ? cap : 1);
    BESST_REQUIRE(ws != nullptr && ws_bytes >= w.total, "partition: workspace too small");
    const size_t region = exchange_stride_bytes(pair_cap, rider_bytes);
    const size_t rider_offset = align_up(exchange_region_bytes(pair_cap), 8);
    const uint32_t rider_words = (uint32_t)(rider_bytes / 8);
    const uint32_t rider_blocks = rider_words ? (rider_words + kSortThreads - 1) / kSortThreads : 0;
    // Tiles of 1024 tuples instead of the sort's 4096 for streams of up to 4 M tuples: a C2-sized slice emits ~140 k
    // tuples, i.e. 34 sort tiles on a 256-CU chip; with 135 small ones the two launches take 14 instead of 22 us
    // (one rank over RCCL, whole step 190 -> 173 us).  Their [owner][tile] table (256 x 4 x sort tiles) fits the
    // MSD table of the same workspace; larger streams keep the sort's tile.
    const uint32_t nb_sort = (uint32_t)((cap + kSortTile - 1) / kSortTile);
    const DigitSel ds{1, 0, node_bits, (uint32_t)world, kRadixBits, 0ull};
    auto run = [&](auto items_tag) {
        constexpr int kItems = decltype(items_tag)::value;
        constexpr uint32_t kTile = kSortThreads * kItems;
        const uint32_t nb = (uint32_t)((cap + kTile - 1) / kTile);
        const uint32_t nb_launch = nb ? nb : 1;
        // (scan-free up to 256 tiles instead of 64 - no row-scan launch for a C2-sized slice - was slower: 145 -> 159 us
        //  per step, every one of the ~170 scatter workgroups summing the rows before it)
        const int scanned = nb > (uint32_t)kScanFreeMaxBlocks ? 1 : 0;
        const uint32_t stride = nb_launch;                   // rows of the scanned layout: [owner][tile]
        hipLaunchKernelGGL((radix_hist_kernel<kRadixBits, kItems>), dim3(nb_launch), dim3(kSortThreads), 0, s, keys,
                           n_tuples, (uint32_t)cap, ds, w.table, stride, scanned, nullptr, nullptr, nullptr);
        if (scanned)
            hipLaunchKernelGGL(radix_rowscan_kernel, dim3(kRadix), dim3(256), 0, s, n_tuples, (uint32_t)cap, w.table,
                               stride, w.row_total, kTile);
        hipLaunchKernelGGL((partition_scatter_kernel<kItems>), dim3(nb_launch + rider_blocks), dim3(kSortThreads), 0, s,
                           keys, payload, n_tuples, (uint32_t)cap, ds, w.table, stride, w.row_total, scanned,
                           (uint32_t)world, (uint32_t)pair_cap, static_cast<char*>(send), region, nb_launch,
                           static_cast<const uint64_t*>(rider), rider_words, rider_offset, slice_info);
    };
    if (nb_sort <= (uint32_t)kMsdMaxBlocks) run(std::integral_constant<int, 4>{});
    else run(std::integral_constant<int, kSortItems>{});
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

int launch_unpack(hipStream_t s, int world, int64_t pair_cap, const void* recv, uint64_t* keys, uint64_t* payload,
                  uint32_t* gidx, uint32_t* n_out, uint32_t* overflow, void* rider_sum, int64_t rider_bytes,
                  int speculative, int rank, int detect_dup, int32_t* all_info, besst_counters* counters) {
    BESST_REQUIRE(world >= 1 && world <= 256, "unpack: world size must be in [1, 256]");
    BESST_REQUIRE(!speculative || (rank >= 0 && rank < world && counters), "unpack: bad speculative-head arguments");
    BESST_REQUIRE(!speculative || rider_bytes == 0 || rider_bytes >= 64, "unpack: the rider must end with the counters");
    BESST_REQUIRE(pair_cap > 0, "unpack: pair capacity must be positive");
    BESST_REQUIRE(rider_bytes >= 0 && (rider_bytes & 7) == 0 && rider_bytes < ((int64_t)1 << 31) &&
                      (rider_bytes == 0 || rider_sum), "unpack: bad rider");
    const size_t region = exchange_stride_bytes(pair_cap, rider_bytes);
    const uint32_t tuple_blocks = (uint32_t)((pair_cap + 255) / 256);
    const uint32_t rider_words = (uint32_t)(rider_bytes / 8);
    const uint32_t rider_blocks = rider_words ? (rider_words + 255) / 256 : 0;
    hipLaunchKernelGGL(unpack_kernel, dim3(tuple_blocks + rider_blocks, (uint32_t)world), dim3(256), 0, s,
                       static_cast<const char*>(recv), (uint32_t)world, (uint32_t)pair_cap, region, keys, payload,
                       gidx, n_out, overflow, tuple_blocks, static_cast<unsigned long long*>(rider_sum), rider_words,
                       align_up(exchange_region_bytes(pair_cap), 8), speculative ? 1 : 0, (uint32_t)rank, detect_dup,
                       all_info, reinterpret_cast<unsigned long long*>(counters));
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}
}  // namespace besst
