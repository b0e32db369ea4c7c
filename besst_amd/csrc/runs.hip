// Large tuple streams, run-grouped form: the edge table without a stream-wide sort of the tuples.
//
// A (tid,pos)-ordered alignment stream emits its link tuples contig by contig: 512 consecutive tuples of a mate-pair
// library concern one or two contigs and carry about six distinct keys (C3: mean 6.2; 9.4 in 1024 tuples, 99th
// percentile 18), each of them dozens to hundreds of times.  So the tuples are never sorted one by one:
//
//   1. rg_group_kernel    one wave per chunk of kRunChunk consecutive tuples, words in registers: the distinct keys of
//                         the chunk are peeled off in order of first occurrence (first lane not yet placed -> its key
//                         -> one compare + ballot + mbcnt per round of 64), the tuples of a key go - in stream order -
//                         right behind those placed so far, and the key's RUN is described by one 40-byte record
//                         (key, count, sum obs, sum obs^2, first stream index, start, graph mask).  Reads the record
//                         loop's block segments (or the compacted stream) once, writes the payload grouped by run.
//                         After 63 keys the rest of a chunk travels as runs of one tuple (chimeric pairs).
//   2. rg_compact_kernel  the chunks' run lists -> one dense, stream-ordered list of (key, count | start) pairs
//   3. the small-stream sort + reduction of sortreduce.hip on the RUNS (C3: 0.5 M runs for 42.7 M tuples): runs sorted
//                         stably by key = the tuples sorted stably by key, because a run holds the equal keys of a
//                         chunk in stream order and the chunks are in stream order
//   4. rg_tile_sums / rg_dst_kernel   exclusive scan of the sorted runs' counts: where every run's observations go
//   5. rg_rows_kernel     one thread per edge row: the sums of its runs, first index and mask of its first run
//   6. rg_copy_kernel     one wave per eight sorted runs: observations from their grouped places to their sorted places
//
// Semantics as everywhere in stage 2 (CreateGraph.py:842-862): an edge is the unordered node pair, its observations stay
// in BAM order, the row's first tuple is its first occurrence, nr_links / obs / obs_sq are exact integers.
//
// Measured on full C3 (42.7 M tuples, 263 k rows; round 2's two chained-scan passes + wave-per-bucket kernels: 0.87 ms
// and 3.3 GB of traffic): 0.49 ms - grouping 0.23 (0.68 GB read, 0.34 GB written), copy 0.14 (0.34 + 0.34 GB), the
// run sort and the scans 0.12 in nine small launches.  Where the grouping kernel's time goes (parts left out, timings
// only): segment reads alone 0.19 ms, the peel 0.08, the grouped stores 0.03 - it is bound by the life of a wave (block
// look-up -> window of block offsets -> tuples -> peel -> stores, nothing overlapped inside a wave), so occupancy was
// the lever: 16 words per lane 0.31 ms at three waves per SIMD, 8 words 0.23 at five.  The two wave sums per key are
// DPP reductions (as ds_bpermute shuffles they were a chain of 24 dependent LDS operations per key: 0.38 -> 0.33 ms).
// The segments' sparse layout (14 KB used of every 128 KB) costs nothing: tools/probe_stride.hip reads and writes that
// pattern at the rate of a dense array.
//
// A stream whose keys do not cluster (name-sorted input, mostly chimeric pairs) has about as many runs as tuples; when the
// runs do not fit their buffers (kRunCapMax) nothing is produced, *n_rows reads BESST_ROWS_RUN_OVERFLOW and the caller
// repeats the call with BESST_REDUCE_NO_RUNS (the chained-scan passes of onesweep.hip).
#include "common.h"

namespace besst {

namespace {

constexpr int kRgRounds = BESST_RG_ROUNDS;
constexpr int kRgMinWaves = kRgRounds >= 16 ? 3 : kRgRounds >= 8 ? 5 : 8;   // what the registers of a chunk allow per SIMD
constexpr int kRgChunk = 64 * kRgRounds;                 // tuples per wave
static_assert(kRgChunk == kRunChunk, "the record loop's chunk table is cut for this chunk size");
constexpr int kRgWaves = 4;
constexpr int kRgPeel = 63;                              // keys peeled per chunk; lane k of the wave keeps run k
constexpr uint32_t kRgNoSlot = 0xffffffffu;
constexpr int kRgCopyRuns = 8;                           // sorted runs per wave of the copy kernel
constexpr int kRsThreads = 256, kRsItems = 16, kRsTile = kRsThreads * kRsItems;

struct RunRec {                                          // 40 bytes: a run is written and read as a whole
    uint64_t key;
    unsigned long long sum, sq;
    uint32_t n, first, off, mask;
};
static_assert(sizeof(RunRec) == 40, "run record layout");

__device__ __forceinline__ uint64_t rg_readlane64(uint64_t v, int l) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
// Sum over the wave, in every lane's... no: in lane 63, read back as a wave-uniform value.  DPP moves (row_shr 1, 2, 4, 8
// inside the rows of 16, then row_bcast 15 and 31), no LDS round trips: with shuffles (ds_bpermute) the two sums per
// key were a chain of 24 dependent LDS operations and set the life of a chunk.
__device__ __forceinline__ unsigned long long rg_wave_sum64(unsigned long long v) {
#define BESST_RG_STEP(ctrl, rows)                                                                                  \
    {                                                                                                              \
        const uint32_t l2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, ctrl, rows, 0xf, false);    \
        const uint32_t h2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), ctrl, rows, 0xf, false); \
        v += ((unsigned long long)h2 << 32) | l2;                                                                  \
    }
    BESST_RG_STEP(0x111, 0xf)
    BESST_RG_STEP(0x112, 0xf)
    BESST_RG_STEP(0x114, 0xf)
    BESST_RG_STEP(0x118, 0xf)
    BESST_RG_STEP(0x142, 0xa)
    BESST_RG_STEP(0x143, 0xc)
#undef BESST_RG_STEP
    return rg_readlane64(v, 63);
}
__device__ __forceinline__ uint32_t rg_obs(uint64_t pl) { return (uint32_t)pl + ((uint32_t)(pl >> 32) & 0x3fffffffu); }

// ---------------------------------------------------------------------------------------------------
// 1. chunks of 512 tuples -> runs
// ---------------------------------------------------------------------------------------------------
template <bool kSeg>
__global__ __launch_bounds__(kRgWaves * 64, kRgMinWaves) void rg_group_kernel(
    const uint64_t* __restrict__ keys, const uint64_t* __restrict__ payload, SegSource seg,
    const uint32_t* __restrict__ n_ptr, uint32_t cap, const uint32_t* __restrict__ first_map,
    uint64_t* __restrict__ grouped, RunRec* __restrict__ staged, uint16_t* __restrict__ starts,
    uint32_t* __restrict__ chunk_runs) {
    constexpr int R = kRgRounds;
    const int lane = threadIdx.x & 63;
    const uint32_t chunk = blockIdx.x * kRgWaves + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t n_all = *n_ptr;
    const uint32_t n = n_all < cap ? n_all : cap;
    const uint32_t base = chunk * (uint32_t)kRgChunk;
    if (base >= n) {
        if (lane == 0) chunk_runs[chunk] = 0;
        return;
    }
    const uint32_t cnt = n - base < (uint32_t)kRgChunk ? n - base : (uint32_t)kRgChunk;
    const uint32_t end = base + cnt;
    uint64_t key[R], pl[R];
    if constexpr (kSeg) {
        // The stream lies in the record loop's block segments (SegSource): dense position i belongs to the last block
        // whose first-tuple offset is <= i (empty blocks share their successor's offset), slot i - offset, one further
        // when the stitch dropped that block's head tuple at or before it.  The chunk's first block by a 64-way search
        // (three round trips for 24 k blocks), the offsets of the 64 blocks from there on in the wave's lanes.
        uint32_t lo = 0, hi = seg.nblocks;
        if (seg.chunk_first) {                               // uniform: the record loop's side has left the answer
            lo = seg.chunk_first[chunk];
            hi = lo + 1u;
        }
        while (hi - lo > 1u) {                               // uniform: lo, hi come out of ballots
            const uint32_t step = (hi - lo + 63u) >> 6;
            const uint32_t b = lo + (uint32_t)(lane + 1) * step;
            const bool le = b < hi && seg.offsets[b] <= base;
            const uint32_t c = (uint32_t)__popcll(__ballot(le));     // the offsets are monotone: a prefix of the lanes
            const uint32_t nhi = lo + (c + 1u) * step;
            lo += c * step;
            hi = nhi < hi ? nhi : hi;
        }
        const uint32_t b0 = lo;
        const uint32_t wb = b0 + (uint32_t)lane;
        const uint32_t w_off = wb < seg.nblocks ? seg.offsets[wb] : 0xffffffffu;
        const uint32_t w_skip = wb < seg.nblocks ? seg.skip[wb] : kRgNoSlot;
        const uint32_t after = b0 + 64u < seg.nblocks ? seg.offsets[b0 + 64u] : 0xffffffffu;
        const bool in_window = after >= end;                 // uniform: every position of the chunk lies in a windowed block
        uint32_t wq[R];
#pragma unroll
        for (int r = 0; r < R; ++r) wq[r] = 0;
        if (in_window) {
            for (int q = 1; q < 64; ++q) {                   // uniform loop: the blocks that begin inside the chunk
                const uint32_t oq = (uint32_t)__builtin_amdgcn_readlane((int)w_off, q);
                if (oq >= end) break;
#pragma unroll
                for (int r = 0; r < R; ++r) wq[r] += base + (uint32_t)(r * 64 + lane) >= oq ? 1u : 0u;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t i = base + (uint32_t)(r * 64 + lane);
            key[r] = ~0ull;
            pl[r] = 0ull;
            uint32_t b, off, skip;
            if (in_window) {
                b = b0 + wq[r];
                off = (uint32_t)__shfl((int)w_off, (int)wq[r], 64);
                skip = (uint32_t)__shfl((int)w_skip, (int)wq[r], 64);
            } else if (i < end) {                            // a chunk that spans more than 64 blocks: search in memory
                uint32_t l2 = b0, h2 = seg.nblocks;
                while (h2 - l2 > 1u) {
                    const uint32_t mid = l2 + ((h2 - l2) >> 1);
                    if (seg.offsets[mid] <= i) l2 = mid; else h2 = mid;
                }
                b = l2; off = seg.offsets[l2]; skip = seg.skip[l2];
            } else {
                b = 0; off = 0; skip = kRgNoSlot;
            }
            if (i < end) {
                uint32_t j = i - off;
                j += j >= skip ? 1u : 0u;
                const size_t src = (size_t)b * seg.tile + j;
                key[r] = seg.seg_keys[src];
                pl[r] = seg.seg_payload[src];
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t i = base + (uint32_t)(r * 64 + lane);
            key[r] = i < end ? keys[i] : ~0ull;              // (a key is below 2^59: the padding matches nothing)
            pl[r] = i < end ? payload[i] : 0ull;
        }
    }
    // ---- peel the distinct keys off, first occurrence first
    uint32_t placed = 0;                                     // bit r: the lane's tuple of round r is placed (or padding)
    uint32_t pos[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (!(base + (uint32_t)(r * 64 + lane) < end)) placed |= 1u << r;
        pos[r] = 0;
    }
    uint32_t covered = 0;
    int K = 0;
    uint64_t my_key = 0;
    unsigned long long my_sum = 0, my_sq = 0;
    uint32_t my_start = 0, my_cnt = 0, my_first = 0, my_mask = 0;
#pragma unroll
    for (int r0 = 0; r0 < R; ++r0) {
        unsigned long long rest = __ballot(!((placed >> r0) & 1u));
        while (rest != 0ull && K < kRgPeel) {                // uniform
            const int src = __ffsll((long long)rest) - 1;
            const uint64_t f = rg_readlane64(key[r0], src);
            uint32_t run = covered;
            unsigned long long s = 0, q = 0;
#pragma unroll
            for (int r = r0; r < R; ++r) {
                const bool hit = key[r] == f;                // (a placed tuple has another key: keys are placed whole)
                const unsigned long long mm = __ballot(hit);
                const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, run));
                // (the observation is formed from the selected words: nothing here is invariant in the key loop, so
                // the compiler does not keep sixteen sums and squares alive across it)
                const uint32_t ol = hit ? (uint32_t)pl[r] : 0u;
                const uint32_t oh = (hit ? (uint32_t)(pl[r] >> 32) : 0u) & 0x3fffffffu;
                const unsigned long long o = (unsigned long long)(ol + oh);
                s += o;
                q += o * o;
                if (hit) {
                    pos[r] = rk;
                    placed |= 1u << r;
                }
                run += (uint32_t)__popcll(mm);
                if (r == r0) rest &= ~mm;
            }
            s = rg_wave_sum64(s);
            q = rg_wave_sum64(q);
            const uint32_t msk = (uint32_t)__builtin_amdgcn_readlane((int)(pl[r0] >> 32), src) >> 30;
            if (lane == K) {
                my_key = f; my_sum = s; my_sq = q; my_start = covered; my_cnt = run - covered;
                my_first = base + (uint32_t)(r0 * 64 + src); my_mask = msk;
            }
            covered = run;
            ++K;
        }
    }
    uint32_t n_runs = (uint32_t)K;
    if (covered < cnt) {
        // more than kRgPeel distinct keys: what is left travels as runs of one tuple (their order among themselves is
        // stream order, and a key that shows up here has none of its tuples among the peeled runs)
        uint32_t run = covered;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool un = !((placed >> r) & 1u);
            const unsigned long long mm = __ballot(un);
            const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, run));
            if (un) {
                pos[r] = rk;
                const uint32_t i = base + (uint32_t)(r * 64 + lane);
                const unsigned long long o = (unsigned long long)rg_obs(pl[r]);
                RunRec rec;
                rec.key = key[r]; rec.sum = o; rec.sq = o * o; rec.n = 1u;
                rec.first = first_map ? first_map[i] : i;
                rec.off = base + rk;
                rec.mask = (uint32_t)(pl[r] >> 62);
                staged[base + rk] = rec;
                starts[base + (uint32_t)kRgPeel + (rk - covered)] = (uint16_t)rk;
            }
            run += (uint32_t)__popcll(mm);
        }
        n_runs = (uint32_t)kRgPeel + (run - covered);
    }
    if (lane < K) {
        RunRec rec;
        rec.key = my_key; rec.sum = my_sum; rec.sq = my_sq; rec.n = my_cnt;
        rec.first = first_map ? first_map[my_first] : my_first;
        rec.off = base + my_start;
        rec.mask = my_mask;
        staged[base + my_start] = rec;                       // a run's start is its own: one record per run, found by `off`
        starts[base + (uint32_t)lane] = (uint16_t)my_start;
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (base + (uint32_t)(r * 64 + lane) < end) grouped[base + pos[r]] = pl[r];
    if (lane == 0) chunk_runs[chunk] = n_runs;
}

// ---------------------------------------------------------------------------------------------------
// 2. run lists of the chunks -> one dense list in stream order
// ---------------------------------------------------------------------------------------------------
// Workgroup g owns chunks [256 g, 256 g + 256): it sums the run counts of all chunks before them (coalesced, from L2),
// scans its own and moves its runs, one thread per run - 1024 threads for a couple of thousand runs: a run costs three
// dependent loads, so the depth per workgroup is what sets the kernel's time (1024 chunks per workgroup: 46 -> 24 us).  The
// last workgroup writes the run count - or, when the runs do not fit, the overflow word (and a count of zero: the
// stages behind find nothing to do).
constexpr int kRcThreads = 1024;
constexpr int kRcChunks = 256;

__global__ __launch_bounds__(kRcThreads) void rg_compact_kernel(
    const uint32_t* __restrict__ chunk_runs, const uint32_t* __restrict__ n_ptr, uint32_t cap,
    const RunRec* __restrict__ staged, const uint16_t* __restrict__ starts, uint32_t run_cap,
    uint64_t* __restrict__ run_keys, uint64_t* __restrict__ run_payload, uint32_t* __restrict__ n_runs,
    uint32_t* __restrict__ status) {
    __shared__ uint32_t s_excl[kRcChunks + 1];
    __shared__ uint32_t s_w[kRcChunks / 64], s_b[kRcThreads / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t n = *n_ptr;
    n = n < cap ? n : cap;
    const uint32_t nchunks = (n + (uint32_t)kRgChunk - 1u) / (uint32_t)kRgChunk;
    const uint32_t c0 = blockIdx.x * (uint32_t)kRcChunks;
    uint32_t before = 0;
    const uint32_t lim = c0 < nchunks ? c0 : nchunks;
    for (uint32_t i = t; i < lim; i += kRcThreads) before += chunk_runs[i];
    const uint32_t mine = (t < kRcChunks && c0 + t < nchunks) ? chunk_runs[c0 + t] : 0u;
    uint32_t x = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)x, d, 64);
        if (lane >= d) x += v;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) before += (uint32_t)__shfl_xor((int)before, d, 64);
    if (lane == 63 && wave < kRcChunks / 64) s_w[wave] = x;
    if (lane == 0) s_b[wave] = before;
    __syncthreads();
    uint32_t off = x - mine, basev = 0, total = 0;
#pragma unroll
    for (int q = 0; q < kRcChunks / 64; ++q) {
        if (q < wave) off += s_w[q];
        total += s_w[q];
    }
#pragma unroll
    for (int q = 0; q < kRcThreads / 64; ++q) basev += s_b[q];
    if (t < kRcChunks) s_excl[t] = off;
    if (t == 0) s_excl[kRcChunks] = total;
    if (blockIdx.x == gridDim.x - 1 && t == 0) {
        const uint32_t all = basev + total;                  // (the chunks of the last workgroup end the stream)
        if (all > run_cap) {
            status[1] = 1u;
            status[2] = all;
            *n_runs = 0u;
        } else {
            *n_runs = all;
        }
    }
    __syncthreads();
    for (uint32_t i = t; i < total; i += kRcThreads) {
        int lo = 0, hi = kRcChunks;                          // last chunk whose exclusive count is <= i
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_excl[mid] <= i) lo = mid; else hi = mid;
        }
        const uint32_t dst = basev + i;
        if (dst >= run_cap) continue;                        // (overflow: reported by the last workgroup)
        const uint32_t cb = (c0 + (uint32_t)lo) * (uint32_t)kRgChunk;
        const uint32_t at = cb + (uint32_t)starts[cb + (i - s_excl[lo])];
        run_keys[dst] = staged[at].key;
        run_payload[dst] = (uint64_t)staged[at].n | ((uint64_t)at << 32);   // "obs_lo" = the run's length, "obs_hi" = where it lies
    }
}

// ---------------------------------------------------------------------------------------------------
// 4. where the observations of every sorted run go: exclusive scan of the run lengths
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kRsThreads) void rg_tile_sums_kernel(const int32_t* __restrict__ run_len,
                                                                 const uint32_t* __restrict__ n_runs,
                                                                 uint32_t* __restrict__ tile_sum) {
    __shared__ uint32_t s_w[kRsThreads / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t n = *n_runs;
    const uint32_t j0 = blockIdx.x * (uint32_t)kRsTile;
    uint32_t v = 0;
    if (j0 < n) {
#pragma unroll
        for (int q = 0; q < kRsItems; ++q) {
            const uint32_t j = j0 + (uint32_t)(q * kRsThreads + t);
            v += j < n ? (uint32_t)run_len[j] : 0u;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
    if (lane == 0) s_w[wave] = v;
    __syncthreads();
    if (t == 0) {
        uint32_t tot = 0;
#pragma unroll
        for (int q = 0; q < kRsThreads / 64; ++q) tot += s_w[q];
        tile_sum[blockIdx.x] = tot;
    }
}

// (dst_of: also by the run's place in the stream-ordered list, for a consumer that walks the runs in that order)
__global__ __launch_bounds__(kRsThreads) void rg_dst_kernel(const int32_t* __restrict__ run_len,
                                                           const uint32_t* __restrict__ n_runs,
                                                           const uint32_t* __restrict__ tile_sum,
                                                           uint32_t* __restrict__ run_dst,
                                                           const int32_t* __restrict__ run_at,
                                                           uint32_t* __restrict__ dst_of) {
    __shared__ uint32_t s_w[kRsThreads / 64], s_b[kRsThreads / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t n = *n_runs;
    const uint32_t j0 = blockIdx.x * (uint32_t)kRsTile;
    if (j0 >= n) return;                                     // uniform
    uint32_t before = 0;
    for (uint32_t i = t; i < blockIdx.x; i += kRsThreads) before += tile_sum[i];
    uint32_t len[kRsItems];
    uint32_t tot = 0;
#pragma unroll
    for (int q = 0; q < kRsItems; ++q) {                     // thread t owns runs [16 t, 16 t + 16) of the tile
        const uint32_t j = j0 + (uint32_t)(t * kRsItems + q);
        len[q] = j < n ? (uint32_t)run_len[j] : 0u;
        tot += len[q];
    }
    uint32_t x = tot;
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
    uint32_t off = x - tot;
#pragma unroll
    for (int q = 0; q < kRsThreads / 64; ++q) {
        if (q < wave) off += s_w[q];
        off += s_b[q];
    }
#pragma unroll
    for (int q = 0; q < kRsItems; ++q) {
        const uint32_t j = j0 + (uint32_t)(t * kRsItems + q);
        if (j < n) {
            run_dst[j] = off;
            if (dst_of) dst_of[(uint32_t)run_at[j]] = off;
        }
        off += len[q];
    }
}

// ---------------------------------------------------------------------------------------------------
// 5. edge rows: the sums of a row's runs
// ---------------------------------------------------------------------------------------------------
// The sort of the runs has left per row its key (already in the table), the number of its runs and the sorted position
// of its first run.  status: a failed stage in front of this one turns into the value *n_rows carries out.
__global__ __launch_bounds__(256) void rg_rows_kernel(const uint32_t* __restrict__ status, uint32_t* __restrict__ n_rows,
                                                      const uint32_t* __restrict__ r_runs, const uint32_t* __restrict__ r_first_run,
                                                      const int32_t* __restrict__ run_at, const uint32_t* __restrict__ run_dst,
                                                      const RunRec* __restrict__ staged, uint32_t* __restrict__ row_mask,
                                                      uint32_t* __restrict__ row_n, unsigned long long* __restrict__ row_sum,
                                                      unsigned long long* __restrict__ row_sum_sq,
                                                      uint32_t* __restrict__ row_first, uint32_t* __restrict__ row_offset) {
    const uint32_t st_err = status[0], st_over = status[1];
    if (st_err | st_over) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *n_rows = st_err ? BESST_ROWS_SORT_FAILED : BESST_ROWS_RUN_OVERFLOW;
        return;
    }
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nr = *n_rows;
    if (nr >= BESST_ROWS_RUN_OVERFLOW || r >= nr) return;    // (the sort of the runs gave up: its word travels on)
    const uint32_t j0 = r_first_run[r], c = r_runs[r];
    uint32_t nn = 0, first = 0, mask = 0;
    unsigned long long s = 0, q = 0;
    for (uint32_t j = j0; j < j0 + c; ++j) {
        const RunRec rec = staged[(uint32_t)run_at[j]];
        nn += rec.n;
        s += rec.sum;
        q += rec.sq;
        if (j == j0) { first = rec.first; mask = rec.mask; }
    }
    row_mask[r] = mask;
    row_n[r] = nn;
    row_sum[r] = s;
    row_sum_sq[r] = q;
    row_first[r] = first;
    row_offset[r] = run_dst[j0];
}

// 5'. the same over the run columns of the record loop's runs (their sums come from rl_place_kernel)
__global__ __launch_bounds__(256) void rl_rows_kernel(const uint32_t* __restrict__ status, uint32_t* __restrict__ n_rows,
                                                      const uint32_t* __restrict__ r_runs, const uint32_t* __restrict__ r_first_run,
                                                      const int32_t* __restrict__ run_len, const int32_t* __restrict__ run_at,
                                                      const uint32_t* __restrict__ run_dst, const uint32_t* __restrict__ c_first,
                                                      const uint32_t* __restrict__ c_mask,
                                                      const unsigned long long* __restrict__ c_sum,
                                                      const unsigned long long* __restrict__ c_sq, uint32_t* __restrict__ row_mask,
                                                      uint32_t* __restrict__ row_n, unsigned long long* __restrict__ row_sum,
                                                      unsigned long long* __restrict__ row_sum_sq,
                                                      uint32_t* __restrict__ row_first, uint32_t* __restrict__ row_offset) {
    const uint32_t st_err = status[0], st_over = status[1];
    if (st_err | st_over) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *n_rows = st_err ? BESST_ROWS_SORT_FAILED : BESST_ROWS_RUN_OVERFLOW;
        return;
    }
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nr = *n_rows;
    if (nr >= BESST_ROWS_RUN_OVERFLOW || r >= nr) return;
    const uint32_t j0 = r_first_run[r], c = r_runs[r];
    uint32_t nn = 0;
    unsigned long long s = 0, q = 0;
    for (uint32_t j = j0; j < j0 + c; ++j) {
        const uint32_t at = (uint32_t)run_at[j];
        nn += (uint32_t)run_len[j];
        s += c_sum[at];
        q += c_sq[at];
    }
    const uint32_t at0 = (uint32_t)run_at[j0];
    row_mask[r] = c_mask[at0];
    row_n[r] = nn;
    row_sum[r] = s;
    row_sum_sq[r] = q;
    row_first[r] = c_first[at0];
    row_offset[r] = run_dst[j0];
}

// ---------------------------------------------------------------------------------------------------
// 6. observations to their sorted places, run by run
// ---------------------------------------------------------------------------------------------------
// One wave per kRgCopyRuns consecutive sorted runs, their tuples taken as one flat range: position p of the range lies
// in the run whose prefix is the last one <= p.  All loads of a wave's range (<= 1024 tuples at a time) are in flight
// before its first store.
__global__ __launch_bounds__(256) void rg_copy_kernel(const uint32_t* __restrict__ n_runs, const int32_t* __restrict__ run_len,
                                                      const int32_t* __restrict__ run_at, const uint32_t* __restrict__ run_dst,
                                                      const uint64_t* __restrict__ grouped, int32_t* __restrict__ obs_lo,
                                                      int32_t* __restrict__ obs_hi, const uint32_t* __restrict__ n_rows) {
    constexpr int G = kRgCopyRuns;
    const int lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t n = *n_runs;
    const uint32_t j = w * (uint32_t)G + (uint32_t)lane;
    if (w * (uint32_t)G >= n || *n_rows >= BESST_ROWS_RUN_OVERFLOW) return;   // uniform (no rows: the runs' places are not valid)
    const bool live = lane < G && j < n;
    const uint32_t len = live ? (uint32_t)run_len[j] : 0u;
    const uint32_t at = live ? (uint32_t)run_at[j] : 0u;
    const uint32_t dst = live ? run_dst[j] : 0u;
    uint32_t incl = len;                                     // prefix sums over the first G lanes
#pragma unroll
    for (int d = 1; d < G; d <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= d) incl += v;
    }
    const uint32_t excl = incl - len;
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, G - 1);
    uint32_t pre[G];                                         // uniform: prefix of run k
#pragma unroll
    for (int k = 0; k < G; ++k) pre[k] = (uint32_t)__builtin_amdgcn_readlane((int)excl, k);
    const uint32_t a_rel = at - excl, d_rel = dst - excl;    // lane k: source / target of range position p is *_rel + p
    for (uint32_t p0 = 0; p0 < total; p0 += 1024u) {
        uint64_t v[16];
        uint32_t to[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const uint32_t p = p0 + (uint32_t)(u * 64 + lane);
            int k = 0;
#pragma unroll
            for (int q = 1; q < G; ++q) k += p >= pre[q] ? 1 : 0;
            // (runs of length 0 do not exist: a prefix equal to p belongs to the run that starts there)
            const uint32_t src = (uint32_t)__shfl((int)a_rel, k, 64) + p;
            to[u] = (uint32_t)__shfl((int)d_rel, k, 64) + p;
            v[u] = 0ull;
            if (p0 + (uint32_t)(u * 64) < total) {           // uniform
                if (p < total) v[u] = grouped[src];
            }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const uint32_t p = p0 + (uint32_t)(u * 64 + lane);
            if (p < total) {
                obs_lo[to[u]] = (int32_t)(uint32_t)v[u];
                obs_hi[to[u]] = (int32_t)((uint32_t)(v[u] >> 32) & 0x3fffffffu);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The runs as the record loop leaves them (fused_wave_kernel<true>, RunLayout in common.h): steps 1 and 2 shrink to a
// listing pass over the blocks' run tables, and step 6 walks the blocks' chunks instead of the sorted runs.
// ---------------------------------------------------------------------------------------------------
struct RunCols {                                         // per run, by its place in the stream-ordered list
    uint32_t* first;                                     // stream index of its first tuple
    uint32_t* mask;                                      // graph mask of that tuple
    unsigned long long* sum;                             // sums of its observations (place kernel)
    unsigned long long* sq;
    uint32_t* dst;                                       // where its observations go (rg_dst_kernel)
};

__device__ __forceinline__ uint32_t rl_summ(const SegSource& seg, int plane, uint32_t b) {
    return seg.summ[(size_t)plane * seg.summ_stride + b];
}

// What a wave of the two kernels below knows about its block after ONE memory round trip: the summary words, the
// stitch's offsets, and the headers of all its chunks (two per lane, loaded before the chunk count is known - the region is
// there whatever it holds) with the exclusive scan of their run counts: chunk c's runs begin at run0 + ex(c) in the list.
struct RlBlock {
    uint32_t nch, off, skip, hslot, run0, head_idx;
    bool head_kept;
    uint64_t hdr[2];
    uint32_t ex[2], total;
    char* region;
    const uint64_t* pl_p;
    __device__ __forceinline__ uint64_t header(uint32_t c) const {
        const int l = (int)(c & 63u);
        const uint64_t h = c < 64u ? hdr[0] : hdr[1];
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(h >> 32), l) << 32) |
               (uint32_t)__builtin_amdgcn_readlane((int)h, l);
    }
    __device__ __forceinline__ uint32_t first_run(uint32_t c) const {
        return run0 + (uint32_t)__builtin_amdgcn_readlane((int)(c < 64u ? ex[0] : ex[1]), (int)(c & 63u));
    }
};
__device__ __forceinline__ RlBlock rl_block(const SegSource& seg, uint32_t b, int lane) {
    RlBlock k;
    k.region = reinterpret_cast<char*>(const_cast<uint64_t*>(seg.seg_keys) + (size_t)b * seg.tile);
    k.pl_p = seg.seg_payload + (size_t)b * seg.tile;
    const uint64_t* hp = reinterpret_cast<const uint64_t*>(k.region + kRlHdr);
    const uint64_t h0 = hp[lane], h1 = (uint32_t)(lane + 64) < (uint32_t)kRlMaxChunks ? hp[lane + 64] : 0ull;
    const uint32_t nch = rl_summ(seg, kSumChunks, b), hinfo = rl_summ(seg, kSumHeadInfo, b);
    k.hslot = rl_summ(seg, kSumHeadSlot, b);
    k.off = seg.offsets[b];
    k.skip = seg.skip[b];
    k.head_idx = seg.run_offsets[b];
    k.nch = nch < (uint32_t)kRlMaxChunks ? nch : (uint32_t)kRlMaxChunks;
    k.head_kept = (hinfo & 8u) && k.hslot != kRgNoSlot && k.skip == kRgNoSlot;
    k.run0 = k.head_idx + (k.head_kept ? 1u : 0u);
    k.hdr[0] = (uint32_t)lane < k.nch ? h0 : 0ull;
    k.hdr[1] = (uint32_t)(lane + 64) < k.nch ? h1 : 0ull;
    uint32_t before = 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const uint32_t n = (uint32_t)(k.hdr[q] >> 32) & 0xffffu;
        uint32_t x = n;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)x, d, 64);
            if (lane >= d) x += v;
        }
        k.ex[q] = before + x - n;
        before += (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
    }
    k.total = before;
    return k;
}
static_assert(kRlMaxChunks <= 64, "rl_list_kernel searches the chunk prefixes one wave holds");
constexpr int kRlWaves = 4;                              // waves per block: wave w takes chunks w, w + 4, ...
constexpr int kRlSingletonRuns = 32;                     // rl_place_kernel: chunks of more runs than this count their runs' tuples first

// 1'. the run list.  The record loop left per chunk a dense list of its runs - key, tuples | first slot (counted in LDS while
// it emitted), name - so listing is a copy: ONE WAVE per block walks the block's runs 64 at a time (run -> chunk by a
// search over the chunks' prefix counts, which the lanes hold), and every run becomes a (key, count | list index << 32)
// pair for the sort, a first stream index and a graph mask.  (Counting here instead - reading the run bytes back and
// peeling the names off, or per-slot LDS atomics - took 0.08 / 0.26 ms on full C3, a wave per chunk 0.075: bound by the
// number of short-lived waves, not by their work.)  The wave also lists the head's run of one (if the stitch kept the
// head); the last block's wave writes the run count - or the overflow word when the list does not fit or a block ran out
// of chunks.
__global__ __launch_bounds__(256) void rl_list_kernel(SegSource seg, uint32_t run_cap, uint64_t* __restrict__ run_keys,
                                                      uint64_t* __restrict__ run_payload, RunCols rc,
                                                      uint32_t* __restrict__ n_runs, uint32_t* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const uint32_t b = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (b >= seg.nblocks) return;
    // (chunk 0's first 64 runs travel with the block's words: one round trip for the loads of most blocks)
    const char* region0 = reinterpret_cast<const char*>(seg.seg_keys + (size_t)b * seg.tile);
    uint64_t key = reinterpret_cast<const uint64_t*>(region0 + kRlKeys)[lane];
    uint32_t meta = reinterpret_cast<const uint32_t*>(region0 + kRlMeta)[lane];
    const RlBlock k = rl_block(seg, b, lane);
    if (k.head_kept && lane == 0 && k.head_idx < run_cap) {
        run_keys[k.head_idx] = *reinterpret_cast<const uint64_t*>(k.region + kRlHeadKey);
        run_payload[k.head_idx] = 1ull | ((uint64_t)k.head_idx << 32);
        rc.first[k.head_idx] = k.off + k.hslot;
        rc.mask[k.head_idx] = (uint32_t)(k.pl_p[k.hslot] >> 62);
    }
    if (b == seg.nblocks - 1u && lane == 0) {
        const uint32_t all = k.run0 + k.total;
        if (all > run_cap || *seg.run_status != 0u) {
            status[1] = 1u;
            status[2] = all;
            *n_runs = 0u;
        } else {
            *n_runs = all;
        }
    }
    for (uint32_t r0 = 0; r0 < k.total; r0 += 64u) {         // uniform
        const uint32_t r = r0 + (uint32_t)lane;              // the lane's run of the block
        // its chunk: the last one whose exclusive prefix is <= r (chunks without runs share their successor's prefix and
        // lose the search to it, as they must); six steps over the 64 prefixes the lanes hold
        uint32_t c = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) {
            const uint32_t t = c + (uint32_t)step;
            const uint32_t ex_t = (uint32_t)__shfl((int)k.ex[0], (int)(t & 63u), 64);
            if (t < k.nch && ex_t <= r) c = t;
        }
        const uint32_t ex_c = (uint32_t)__shfl((int)k.ex[0], (int)c, 64);
        const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)k.hdr[0], (int)c, 64);
        const uint32_t e = c * (uint32_t)kRlSlots + (r - ex_c);
        if (r < k.total) {
            if (!(r0 == 0u && c == 0u)) {                    // (not what came with the first round trip)
                key = reinterpret_cast<const uint64_t*>(k.region + kRlKeys)[e];
                meta = reinterpret_cast<const uint32_t*>(k.region + kRlMeta)[e];
            }
            const uint32_t j = k.run0 + r, fs = (lo & 0xffffu) + (meta >> 16);
            if (j < run_cap) {
                run_keys[j] = key;
                run_payload[j] = (uint64_t)(meta & 0xffffu) | ((uint64_t)j << 32);
                rc.first[j] = k.off + fs - (fs > k.skip ? 1u : 0u);
                rc.mask[j] = (uint32_t)(k.pl_p[fs] >> 62);
            }
        }
    }
}

// 6'. the observations to their places.  A wave takes a chunk: the tuples' run bytes and payload (eight words per lane, as
// rg_group_kernel held them), per run of the chunk one compare + ballot + mbcnt per round of 64 - rank inside the run -,
// the observations stored at the run's sorted place + rank, the run's two sums left for the row kernel.  The payload is
// read ONCE, where the record loop wrote it, and written once, where the table wants it.
__global__ __launch_bounds__(kRlWaves * 64, 6) void rl_place_kernel(SegSource seg, const uint32_t* __restrict__ status,
                                                                   const uint32_t* __restrict__ n_rows, RunCols rc,
                                                                   int32_t* __restrict__ obs_lo, int32_t* __restrict__ obs_hi) {
    constexpr int R = kRlChunkTuples / 64, Q = kRlSlots / 64;
    __shared__ uint32_t s_cnt[kRlWaves][kRlSlots], s_dst[kRlWaves][kRlSlots];
    __shared__ uint8_t s_kk[kRlWaves][kRlSlots];
    const int lane = threadIdx.x & 63;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t b = blockIdx.x;
    if ((status[0] | status[1]) != 0u || *n_rows >= BESST_ROWS_RUN_OVERFLOW) return;   // uniform
    const RlBlock k = rl_block(seg, b, lane);
    const uint8_t* rid_p = reinterpret_cast<const uint8_t*>(k.region + kRlRid);
    for (uint32_t c = w; c < k.nch; c += (uint32_t)kRlWaves) {   // uniform
        const uint64_t hdr = k.header(c);
        const uint32_t start = (uint32_t)hdr & 0xffffu, cnt = (uint32_t)(hdr >> 16) & 0xffffu, K = (uint32_t)(hdr >> 32) & 0xffffu;
        const uint32_t idx = k.first_run(c);
        uint32_t my_dst[Q], my_slot[Q];                      // lane l: runs l and 64 + l of the chunk (list order)
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const uint32_t kk = (uint32_t)(q * 64 + lane);
            my_dst[q] = kk < K ? rc.dst[idx + kk] : 0u;
            my_slot[q] = kk < K ? (uint32_t)reinterpret_cast<const uint8_t*>(k.region + kRlOrd)[c * (uint32_t)kRlSlots + kk] : kRlNoRun;
        }
        uint32_t rid[R], to[R];
        uint64_t pl[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t i = (uint32_t)(r * 64 + lane);
            rid[r] = kRlNoRun + 1u;                          // padding: matches nothing
            pl[r] = 0ull;
            to[r] = 0u;
            if ((uint32_t)(r * 64) < cnt) {                  // uniform
                if (i < cnt) {
                    rid[r] = rid_p[start + i];
                    pl[r] = k.pl_p[start + i];
                }
            }
        }
        // the block's head lies in this chunk (uniform): its place comes with the chunk's other loads
        const bool has_head = k.head_kept && k.hslot >= start && k.hslot < start + cnt;
        const uint32_t head_dst = has_head ? rc.dst[k.head_idx] : 0u;
        // How many of the chunk's tuples each run holds (LDS counters, per wave): a run of ONE tuple - a chimeric pair's link,
        // an edge seen once - needs neither a rank nor a reduction: its tuple goes to the run's place and IS the run's
        // sums.  The loop over the runs (a ballot and two wave sums per run and round) is left to the runs of two tuples and
        // more: with 3 % chimeric pairs a chunk holds ~90 runs instead of ~15 and the loop was 0.39 of the 0.17 ms.
        // (only where a chunk holds many runs - uniform: the counters cost a chunk of ~15 runs more than they save)
        const bool sparse = K > (uint32_t)kRlSingletonRuns;
        uint32_t* cnt_s = s_cnt[w];
        uint32_t* dst_s = s_dst[w];
        uint8_t* kk_s = s_kk[w];
        unsigned long long my_s[Q], my_q[Q];
        if (sparse) {
#pragma unroll
            for (int q = 0; q < Q; ++q) cnt_s[q * 64 + lane] = 0u;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (rid[r] < kRlNoRun) atomicAdd(&cnt_s[rid[r]], 1u);
#pragma unroll
            for (int q = 0; q < Q; ++q)
                if (my_slot[q] < kRlNoRun) {
                    dst_s[my_slot[q]] = my_dst[q];
                    kk_s[my_slot[q]] = (uint8_t)(q * 64 + lane);
                }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (rid[r] < kRlNoRun && cnt_s[rid[r]] == 1u) {
                    const unsigned long long o = (unsigned long long)((uint32_t)pl[r] + ((uint32_t)(pl[r] >> 32) & 0x3fffffffu));
                    const uint32_t kk = kk_s[rid[r]];
                    to[r] = dst_s[rid[r]];
                    rc.sum[idx + kk] = o;
                    rc.sq[idx + kk] = o * o;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            my_s[q] = 0; my_q[q] = 0;
            const bool many = my_slot[q] < kRlNoRun && (!sparse || cnt_s[my_slot[q]] >= 2u);
            unsigned long long todo = __ballot(many);
            while (todo) {                                   // uniform: the runs of two tuples and more, in list order
                const uint32_t kk = (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1ull;
                uint32_t run = (uint32_t)__builtin_amdgcn_readlane((int)my_dst[q], (int)kk);
                const uint32_t sl = (uint32_t)__builtin_amdgcn_readlane((int)my_slot[q], (int)kk);
                unsigned long long s2 = 0, q2 = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const bool hit = rid[r] == sl;
                    const unsigned long long mm = __ballot(hit);
                    const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, run));
                    const uint32_t ol = hit ? (uint32_t)pl[r] : 0u;
                    const uint32_t oh = (hit ? (uint32_t)(pl[r] >> 32) : 0u) & 0x3fffffffu;
                    const unsigned long long o = (unsigned long long)(ol + oh);
                    s2 += o;
                    q2 += o * o;
                    if (hit) to[r] = rk;
                    run += (uint32_t)__popcll(mm);
                }
                s2 = rg_wave_sum64(s2);
                q2 = rg_wave_sum64(q2);
                if ((uint32_t)lane == kk) { my_s[q] = s2; my_q[q] = q2; }
            }
            if (many || (sparse && my_slot[q] < kRlNoRun && cnt_s[my_slot[q]] == 0u)) {     // (a listed run without a tuple here: zero sums, as the loop over all runs left them)
                rc.sum[idx + (uint32_t)(q * 64 + lane)] = my_s[q];
                rc.sq[idx + (uint32_t)(q * 64 + lane)] = my_q[q];
            }
        }
        __builtin_amdgcn_wave_barrier();                     // (the counters are the next chunk's again)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (rid[r] < kRlNoRun) {
                obs_lo[to[r]] = (int32_t)(uint32_t)pl[r];
                obs_hi[to[r]] = (int32_t)((uint32_t)(pl[r] >> 32) & 0x3fffffffu);
            } else if (rid[r] == kRlNoRun && has_head) {     // the block's head: a run of one
                const unsigned long long o = (unsigned long long)rg_obs(pl[r]);
                obs_lo[head_dst] = (int32_t)(uint32_t)pl[r];
                obs_hi[head_dst] = (int32_t)((uint32_t)(pl[r] >> 32) & 0x3fffffffu);
                rc.sum[k.head_idx] = o;
                rc.sq[k.head_idx] = o * o;
            }
        }
    }
}

struct RgWorkspace {
    uint32_t* status;           // [0] a look-back gave up, [1] run overflow, [2] runs wanted
    uint32_t* n_runs;
    uint32_t* chunk_runs;
    uint16_t* starts;
    uint64_t* run_keys;
    uint64_t* run_payload;
    int32_t* run_len;           // sorted runs: length ("obs_lo" of the nested reduction) ...
    int32_t* run_at;            // ... and where the run lies ("obs_hi")
    uint32_t* run_dst;
    uint32_t* tile_sum;
    uint32_t* r_mask;           // nested row outputs nobody reads, the number of runs per row, its first sorted run
    uint32_t* r_runs;
    int64_t* r_sum;
    int64_t* r_sq;
    uint32_t* r_first;
    uint32_t* r_first_run;
    char* nested;
    size_t nested_bytes;
    uint32_t run_cap;
    size_t total;
};

}  // namespace

// the small-stream form of launch_sort_reduce (MSD partition + per-bucket sort, kMsdMaxBlocks tiles) serves up to this many runs
constexpr int64_t kRunCapMax = (int64_t)4 << 20;

static int64_t rg_run_cap(int64_t cap) { return cap < kRunCapMax ? cap : kRunCapMax; }

static RgWorkspace rg_carve(void* ws, int64_t cap) {
    RgWorkspace w;
    char* p = static_cast<char*>(ws);
    size_t off = 0;
    const size_t rc = (size_t)rg_run_cap(cap);
    const size_t nchunks = align_up((size_t)((cap + kRgChunk - 1) / kRgChunk), kRgWaves);
    w.run_cap = (uint32_t)rc;
    w.status = reinterpret_cast<uint32_t*>(p + off); off += 256;
    w.n_runs = w.status + 8;
    w.chunk_runs = reinterpret_cast<uint32_t*>(p + off); off += align_up(nchunks * 4, 256);
    w.starts = reinterpret_cast<uint16_t*>(p + off); off += align_up((size_t)cap * 2, 256);
    w.run_keys = reinterpret_cast<uint64_t*>(p + off); off += align_up(rc * 8, 256);
    w.run_payload = reinterpret_cast<uint64_t*>(p + off); off += align_up(rc * 8, 256);
    w.run_len = reinterpret_cast<int32_t*>(p + off); off += align_up(rc * 4, 256);
    w.run_at = reinterpret_cast<int32_t*>(p + off); off += align_up(rc * 4, 256);
    w.run_dst = reinterpret_cast<uint32_t*>(p + off); off += align_up(rc * 4, 256);
    w.tile_sum = reinterpret_cast<uint32_t*>(p + off); off += align_up(((rc + kRsTile - 1) / kRsTile) * 4, 256);
    w.r_mask = reinterpret_cast<uint32_t*>(p + off); off += align_up(rc * 4, 256);
    w.r_runs = reinterpret_cast<uint32_t*>(p + off); off += align_up(rc * 4, 256);
    w.r_sum = reinterpret_cast<int64_t*>(p + off); off += align_up(rc * 8, 256);
    w.r_sq = reinterpret_cast<int64_t*>(p + off); off += align_up(rc * 8, 256);
    w.r_first = reinterpret_cast<uint32_t*>(p + off); off += align_up(rc * 4, 256);
    w.r_first_run = reinterpret_cast<uint32_t*>(p + off); off += align_up(rc * 4, 256);
    w.nested = p + off;
    w.nested_bytes = reduce_workspace_bytes((int64_t)rc);   // (rc <= 4 M: no run workspace inside)
    off += align_up(w.nested_bytes, 256);
    w.total = off;
    return w;
}

size_t runs_workspace_bytes(int64_t cap) {
    if (cap < 1) cap = 1;
    return rg_carve(nullptr, cap).total;
}

bool runs_enabled(int64_t cap) {
    return cap >= 1 && cap <= ((int64_t)1 << 30);
}

int launch_runs_reduce(hipStream_t s, int64_t cap, const uint32_t* n_tuples, int key_bits, const uint64_t* keys,
                       const uint64_t* payload, const SegSource* seg, uint64_t* grouped, void* staged_rows,
                       uint64_t* row_key, uint32_t* row_mask, uint32_t* row_n, int64_t* row_sum, int64_t* row_sum_sq,
                       uint32_t* row_first, uint32_t* row_offset, int32_t* obs_lo, int32_t* obs_hi, uint32_t* n_rows,
                       void* ws, size_t ws_bytes, const uint32_t* first_map, uint64_t key_base) {
    const RgWorkspace w = rg_carve(ws, cap);
    BESST_REQUIRE(ws != nullptr && ws_bytes >= w.total, "reduce: run workspace too small");
    BESST_REQUIRE(grouped != nullptr && staged_rows != nullptr, "reduce: run buffers missing");
    RunRec* staged = static_cast<RunRec*>(staged_rows);
    const uint32_t nchunks = (uint32_t)((cap + kRgChunk - 1) / kRgChunk);
    const uint32_t grid1 = (nchunks + kRgWaves - 1) / kRgWaves;
    BESST_HIP_TRY(hipMemsetAsync(w.status, 0, 64, s));
    if (seg && seg->run_offsets) {
        // the record loop grouped the runs (RunLayout): list them, sort the list, place the observations block by block
        BESST_REQUIRE(!first_map && seg->summ && seg->run_status, "reduce: incomplete description of the record loop's runs");
        RunCols rc;                                          // (in the row staging of the chained-scan form: 40 bytes per tuple slot)
        {
            char* p = static_cast<char*>(staged_rows);
            const size_t n = (size_t)w.run_cap;
            rc.sum = reinterpret_cast<unsigned long long*>(p); p += n * 8;
            rc.sq = reinterpret_cast<unsigned long long*>(p); p += n * 8;
            rc.first = reinterpret_cast<uint32_t*>(p); p += n * 4;
            rc.mask = reinterpret_cast<uint32_t*>(p); p += n * 4;
            rc.dst = reinterpret_cast<uint32_t*>(p);
        }
        const uint32_t bgrid = seg->nblocks;                // one workgroup per block, a wave per chunk
        {
            ProfScope ps(s, kProfRunList);
            hipLaunchKernelGGL(rl_list_kernel, dim3((bgrid + 3u) / 4u), dim3(256), 0, s, *seg, w.run_cap, w.run_keys, w.run_payload, rc,
                               w.n_runs, w.status);
        }
        const int rcode = launch_sort_reduce(s, (int64_t)w.run_cap, w.n_runs, key_bits, w.run_keys, w.run_payload, row_key, w.r_mask,
                                             w.r_runs, w.r_sum, w.r_sq, w.r_first, w.r_first_run, w.run_len, w.run_at, n_rows,
                                             w.nested, w.nested_bytes, nullptr, key_base, false, nullptr, BESST_REDUCE_NO_RUNS);
        if (rcode) return rcode;
        const uint32_t tiles = (w.run_cap + kRsTile - 1) / kRsTile;
        {
            ProfScope ps(s, kProfRunScan);
            hipLaunchKernelGGL(rg_tile_sums_kernel, dim3(tiles), dim3(kRsThreads), 0, s, w.run_len, w.n_runs, w.tile_sum);
            hipLaunchKernelGGL(rg_dst_kernel, dim3(tiles), dim3(kRsThreads), 0, s, w.run_len, w.n_runs, w.tile_sum, w.run_dst,
                               w.run_at, rc.dst);
        }
        {
            ProfScope ps(s, kProfRunPlace);
            hipLaunchKernelGGL(rl_place_kernel, dim3(bgrid), dim3(kRlWaves * 64), 0, s, *seg, w.status, n_rows, rc, obs_lo, obs_hi);
        }
        {
            ProfScope ps(s, kProfRunRows);
            hipLaunchKernelGGL(rl_rows_kernel, dim3((w.run_cap + 255) / 256), dim3(256), 0, s, w.status, n_rows, w.r_runs,
                               w.r_first_run, w.run_len, w.run_at, w.run_dst, rc.first, rc.mask, rc.sum, rc.sq, row_mask, row_n,
                               reinterpret_cast<unsigned long long*>(row_sum), reinterpret_cast<unsigned long long*>(row_sum_sq),
                               row_first, row_offset);
        }
        BESST_HIP_TRY(hipGetLastError());
        return BESST_OK;
    }
    {
        ProfScope ps(s, kProfRunGroup);
        if (seg)
            hipLaunchKernelGGL((rg_group_kernel<true>), dim3(grid1), dim3(kRgWaves * 64), 0, s, keys, payload, *seg, n_tuples,
                               (uint32_t)cap, first_map, grouped, staged, w.starts, w.chunk_runs);
        else
            hipLaunchKernelGGL((rg_group_kernel<false>), dim3(grid1), dim3(kRgWaves * 64), 0, s, keys, payload, SegSource{},
                               n_tuples, (uint32_t)cap, first_map, grouped, staged, w.starts, w.chunk_runs);
    }
    {
        ProfScope ps(s, kProfRunCompact);
        hipLaunchKernelGGL(rg_compact_kernel, dim3((nchunks + kRcChunks - 1) / kRcChunks), dim3(kRcThreads), 0, s,
                           w.chunk_runs, n_tuples, (uint32_t)cap, staged, w.starts, w.run_cap, w.run_keys, w.run_payload,
                           w.n_runs, w.status);
    }
    {
        // the runs, sorted stably by key and cut into rows: row_key is final, the other columns describe the runs
        const int rc = launch_sort_reduce(s, (int64_t)w.run_cap, w.n_runs, key_bits, w.run_keys, w.run_payload, row_key, w.r_mask,
                                          w.r_runs, w.r_sum, w.r_sq, w.r_first, w.r_first_run, w.run_len, w.run_at, n_rows,
                                          w.nested, w.nested_bytes, nullptr, key_base, false, nullptr, BESST_REDUCE_NO_RUNS);
        if (rc) return rc;
        ProfScope ps(s, kProfRunScan);
        const uint32_t tiles = (w.run_cap + kRsTile - 1) / kRsTile;
        hipLaunchKernelGGL(rg_tile_sums_kernel, dim3(tiles), dim3(kRsThreads), 0, s, w.run_len, w.n_runs, w.tile_sum);
        hipLaunchKernelGGL(rg_dst_kernel, dim3(tiles), dim3(kRsThreads), 0, s, w.run_len, w.n_runs, w.tile_sum, w.run_dst,
                           (const int32_t*)nullptr, (uint32_t*)nullptr);
        hipLaunchKernelGGL(rg_rows_kernel, dim3((w.run_cap + 255) / 256), dim3(256), 0, s, w.status, n_rows, w.r_runs,
                           w.r_first_run, w.run_at, w.run_dst, staged, row_mask, row_n,
                           reinterpret_cast<unsigned long long*>(row_sum), reinterpret_cast<unsigned long long*>(row_sum_sq),
                           row_first, row_offset);
    }
    {
        ProfScope ps(s, kProfRunCopy);
        const uint32_t waves = (w.run_cap + kRgCopyRuns - 1) / kRgCopyRuns;
        hipLaunchKernelGGL(rg_copy_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, w.n_runs, w.run_len, w.run_at, w.run_dst,
                           grouped, obs_lo, obs_hi, n_rows);
    }
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

}  // namespace besst
