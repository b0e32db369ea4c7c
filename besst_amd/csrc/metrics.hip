// Library-statistics sampling on gfx950: the three `for read in bam_file` scans of libmetrics
// (libmetrics.py:63-84 contamination, :293-303 insert-size sample) as one ordered pass.
//
// Per record (15 B: tid, mtid, tlen 4 B each, flag 2, mapq 1):
//   A  insert-size observation    fr: is_proper_aligned_unique_innie, rf: ..._outie, and tid among the
//                                 1000 longest references                     [bam_parser.py:22-29, :294-301]
//   B  tid among the 1000 longest references (sample_counter)                 [libmetrics.py:65-66]
//   C  B and not unmapped (counter_total)                                     [:67-68]
//   D  B and opposite-orientation pair with read_len < fragment size          [:71-81]
// The reference stops collecting A after exactly 1,000,000 observations and stops the contamination
// scan at the record where sample_counter reaches 1,000,000 - both are "first N in stream order".
// Ordered semantics come from exclusive prefixes of A, B and D in stream order (a single pass with a look-back
// over the tiles in front, metrics_onepass_kernel): they decide inclusion and the output slot, so isize_out /
// contam_out hold |tlen| in BAM order exactly like the Python lists.  The float finishing (means, trimming, GetDistr) replays the
// reference's operation order on the host from those lists.
#include "common.h"

namespace besst {

namespace {

constexpr int kMetThreads = 256;
constexpr int kMetVec = 4;
constexpr int kMetTile = kMetThreads * kMetVec;   // 1024 records per block
constexpr long long kSampleCap = 1000000;

// state layout (int64): 0 nA seen, 1 nB seen, 2 nD seen, 3 counter_total (C within cut-off),
//                        4 n_contam (D within cut-off), 5 records scanned
struct Flags4 {
    uint32_t a, b, c, d;   // bit k = record k of the thread
    int32_t val[kMetVec];  // |tlen|
};

// the predicates of one record (bam_parser.py:22-29, libmetrics.py:63-84, 293-303); branch free: `top` only gates the bits
__device__ __forceinline__ void eval_record(const MetricsArgs& m, int k, int32_t tid, int32_t mtid, int32_t tlen, uint32_t flag,
                                            uint32_t mapq, bool top, Flags4& f) {
    const int64_t t64 = tlen;
    const int64_t at = t64 < 0 ? -t64 : t64;
    f.val[k] = (int32_t)at;
    const uint32_t bit = top ? 1u << k : 0u;
    f.b |= bit;
    if (!(flag & kFlagUnmapped)) f.c |= bit;
    const bool rev = flag & kFlagReverse, mrev = flag & kFlagMateReverse;
    const bool base = (flag & kFlagRead2) && tid == mtid && !(flag & kFlagMateUnmapped) && (int32_t)mapq > m.min_mapq &&
                      !(flag & kFlagSecondary);
    const bool innie = base && ((rev && !mrev && tlen < 0) || (!rev && mrev && tlen > 0));
    const bool outie = base && ((rev && !mrev && tlen > 0) || (!rev && mrev && tlen < 0));
    if (m.rf ? outie : innie) f.a |= bit;
    const bool d = m.rf ? (innie && m.read_len < (double)at) : (outie && m.read_len < (double)at + 2.0 * m.read_len);
    if (d) f.d |= bit;
}

// four records of a thread, any alignment and any end (the count kernel, and the one-pass kernel's ragged tiles)
__device__ __forceinline__ Flags4 eval4(const MetricsArgs& m, int64_t i0, int64_t end) {
    Flags4 f;
    f.a = f.b = f.c = f.d = 0;
    int32_t tid[kMetVec], mtid[kMetVec], tlen[kMetVec];
    uint32_t flag[kMetVec], mapq[kMetVec];
    if (i0 + kMetVec <= end && (i0 & 3) == 0) {
        // (every record is read once: non-temporal, as in the record loop)
        typedef int v4i __attribute__((ext_vector_type(4)));
        typedef unsigned int v2u __attribute__((ext_vector_type(2)));
        const v4i v0 = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(m.tid + i0));
        const v4i v1 = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(m.mtid + i0));
        const v4i v2 = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(m.tlen + i0));
        const v2u fl = __builtin_nontemporal_load(reinterpret_cast<const v2u*>(m.flag + i0));
        const uint32_t mq = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(m.mapq + i0));
        tid[0] = v0.x; tid[1] = v0.y; tid[2] = v0.z; tid[3] = v0.w;
        mtid[0] = v1.x; mtid[1] = v1.y; mtid[2] = v1.z; mtid[3] = v1.w;
        tlen[0] = v2.x; tlen[1] = v2.y; tlen[2] = v2.z; tlen[3] = v2.w;
        flag[0] = fl.x & 0xffffu; flag[1] = fl.x >> 16; flag[2] = fl.y & 0xffffu; flag[3] = fl.y >> 16;
        mapq[0] = mq & 255u; mapq[1] = (mq >> 8) & 255u; mapq[2] = (mq >> 16) & 255u; mapq[3] = mq >> 24;
    } else {
#pragma unroll
        for (int k = 0; k < kMetVec; ++k) {
            const int64_t i = i0 + k;
            const bool in = i < end;
            tid[k] = in ? m.tid[i] : -1;
            mtid[k] = in ? m.mtid[i] : -2;
            tlen[k] = in ? m.tlen[i] : 0;
            flag[k] = in ? m.flag[i] : 0;
            mapq[k] = in ? m.mapq[i] : 0;
        }
    }
#pragma unroll
    for (int k = 0; k < kMetVec; ++k) {
        const bool top = (uint32_t)tid[k] < (uint32_t)m.n_contigs && m.top_mask[tid[k]] != 0;
        eval_record(m, k, tid[k], mtid[k], tlen[k], flag[k], mapq[k], top, f);
    }
    return f;
}

// A whole, aligned tile of the one-pass kernel: kSubs x four records per thread in TWO memory round trips - every column load
// of the tile is issued before the first is waited for, then every top-1000 look-up (a byte gather through the contig id,
// clamped instead of branched around), then the arithmetic.  Written sub-tile by sub-tile the compiler kept each sub-tile's
// loads, and each of its four look-ups, behind the branches of the one before: 20 round trips in a row, 15 of a tile's 27 us.
template <int kSubs>
__device__ __forceinline__ void eval_tile(const MetricsArgs& m, int64_t i0, int64_t stride, Flags4 (&f)[kSubs]) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    v4i tid[kSubs], mtid[kSubs], tlen[kSubs];
    v2u fl[kSubs];
    uint32_t mq[kSubs];
#pragma unroll
    for (int u = 0; u < kSubs; ++u) {
        const int64_t i = i0 + (int64_t)u * stride;
        tid[u] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(m.tid + i));
        mtid[u] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(m.mtid + i));
        tlen[u] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(m.tlen + i));
        fl[u] = __builtin_nontemporal_load(reinterpret_cast<const v2u*>(m.flag + i));
        mq[u] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(m.mapq + i));
    }
    uint8_t top[kSubs][kMetVec];
#pragma unroll
    for (int u = 0; u < kSubs; ++u)
#pragma unroll
        for (int k = 0; k < kMetVec; ++k) {
            const uint32_t c = (uint32_t)tid[u][k];
            top[u][k] = m.top_mask[c < (uint32_t)m.n_contigs ? c : 0u];
        }
#pragma unroll
    for (int u = 0; u < kSubs; ++u) {
        f[u].a = f[u].b = f[u].c = f[u].d = 0;
#pragma unroll
        for (int k = 0; k < kMetVec; ++k) {
            const uint32_t flag = (k & 1) ? fl[u][k >> 1] >> 16 : fl[u][k >> 1] & 0xffffu;
            const bool is_top = (uint32_t)tid[u][k] < (uint32_t)m.n_contigs && top[u][k] != 0;
            eval_record(m, k, tid[u][k], mtid[u][k], tlen[u][k], flag, (mq[u] >> (8 * k)) & 255u, is_top, f[u]);
        }
    }
}

__device__ __forceinline__ int wsum(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__global__ __launch_bounds__(kMetThreads) void metrics_count_kernel(MetricsArgs m, int64_t start, int64_t end,
                                                                    uint32_t* __restrict__ blk) {
    __shared__ int s[4][3];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t i0 = start + (int64_t)blockIdx.x * kMetTile + (int64_t)t * kMetVec;
    const Flags4 f = eval4(m, i0, end);
    const int a = wsum(__popc(f.a)), b = wsum(__popc(f.b)), d = wsum(__popc(f.d));
    if (lane == 0) { s[wave][0] = a; s[wave][1] = b; s[wave][2] = d; }
    __syncthreads();
    if (t < 3) blk[blockIdx.x * 3 + t] = (uint32_t)(s[0][t] + s[1][t] + s[2][t] + s[3][t]);
}

// exclusive scan of the three per-block counts; bases continue from the running state
__global__ __launch_bounds__(1024) void metrics_scan_kernel(uint32_t* __restrict__ blk, uint32_t nblocks,
                                                            const long long* __restrict__ state,
                                                            long long* __restrict__ base /* [nblocks*3] */,
                                                            long long* __restrict__ totals /* [3] */) {
    __shared__ long long s_w[16][3];
    __shared__ long long s_carry[3];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t < 3) s_carry[t] = state[t];
    __syncthreads();
    for (uint32_t c0 = 0; c0 < nblocks; c0 += 1024) {
        const uint32_t b = c0 + t;
        long long v[3], x[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { v[j] = b < nblocks ? (long long)blk[b * 3 + j] : 0; x[j] = v[j]; }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const long long o = __shfl_up(x[j], d, 64);
                if (lane >= d) x[j] += o;
            }
        }
        if (lane == 63) { s_w[wave][0] = x[0]; s_w[wave][1] = x[1]; s_w[wave][2] = x[2]; }
        __syncthreads();
        long long pre[3] = {s_carry[0], s_carry[1], s_carry[2]};
        for (int w = 0; w < wave; ++w) { pre[0] += s_w[w][0]; pre[1] += s_w[w][1]; pre[2] += s_w[w][2]; }
        if (b < nblocks) {
#pragma unroll
            for (int j = 0; j < 3; ++j) base[(size_t)b * 3 + j] = pre[j] + x[j] - v[j];
        }
        __syncthreads();
        if (t == 1023) { s_carry[0] = pre[0] + x[0]; s_carry[1] = pre[1] + x[1]; s_carry[2] = pre[2] + x[2]; }
        __syncthreads();
    }
    if (t < 3) totals[t] = s_carry[t];
}

// ---- count + scan + emit in ONE pass over the records (round 4: the three-kernel form read every record twice and ran
// at 0.11 of the pass's 22 B/pair roofline).  A workgroup takes a tile of kMetSubs x 1024 records - sixteen per thread, all
// their loads in flight together, the flags kept in registers -; tiles are numbered by an arrival ticket, so a tile's
// predecessors are always running or done.  A tile publishes its three counts in one {status, counts} word - status 1: the
// tile's own counts, 2: the inclusive counts of the stream up to and including the tile - and its first wave looks back over
// the words of the 64 tiles in front of it (one per lane), further while none of them is inclusive: the look-back is a
// serial chain of memory round trips (~2 us each across the chip), one per 64 tiles, which is why tiles are large (1024-record tiles: 58 k
// tiles, 1.5 ms for 60 M records - the chain, not the bytes); a wider window was slower, not faster: 512 words per step
// 0.17 of the roofline, 128 words 0.24, 64 words 0.26 - every waiting tile polls uncached words, and that traffic is what
// slows the tiles that could make progress.  The
// words are relaxed agent-scope atomics, each self-describing; the workspace is zeroed per call.
constexpr unsigned long long kMdShift = 62;
constexpr long long kMdSat = (1ll << 20) - 1;            // >= kSampleCap
#ifndef BESST_MET_SUBS
#define BESST_MET_SUBS 4
#endif
#ifndef BESST_MET_LOOK
#define BESST_MET_LOOK 1
#endif
constexpr int kMetSubs = BESST_MET_SUBS;
constexpr int kMetBig = kMetSubs * kMetTile;             // records per tile of the one-pass kernel
constexpr int kMetLook = BESST_MET_LOOK;                              // predecessors per lane and look-back step

__device__ __forceinline__ long long wsum64(long long v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__global__ __launch_bounds__(kMetThreads) void metrics_onepass_kernel(MetricsArgs m, int64_t start, int64_t end, uint32_t nb,
                                                                      unsigned long long* __restrict__ desc,
                                                                      uint32_t* __restrict__ ticket, int want_isize,
                                                                      int32_t* __restrict__ isize_out,
                                                                      int32_t* __restrict__ contam_out,
                                                                      unsigned long long* __restrict__ state) {
    __shared__ unsigned long long s_w[kMetSubs][4];          // per sub-tile and wave: the three counts, 16 bits apart
    __shared__ int s_tot[4][2];
    __shared__ long long s_pre[3];
    __shared__ uint32_t s_tile;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const int64_t tile0 = start + (int64_t)tile * kMetBig;
    Flags4 f[kMetSubs];
    if ((tile0 & 3) == 0 && tile0 + kMetBig <= end && m.n_contigs > 0) {           // uniform
        eval_tile<kMetSubs>(m, tile0 + (int64_t)t * kMetVec, kMetTile, f);
    } else {
#pragma unroll
        for (int u = 0; u < kMetSubs; ++u) f[u] = eval4(m, tile0 + (int64_t)u * kMetTile + (int64_t)t * kMetVec, end);
    }
    unsigned long long mine[kMetSubs], incl[kMetSubs];       // a | b << 16 | d << 32 of the thread / scanned over the wave
#pragma unroll
    for (int u = 0; u < kMetSubs; ++u) {
        mine[u] = (unsigned long long)__popc(f[u].a) | ((unsigned long long)__popc(f[u].b) << 16) | ((unsigned long long)__popc(f[u].d) << 32);
        unsigned long long x = mine[u];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long o = __shfl_up(x, d, 64);
            if (lane >= d) x += o;
        }
        incl[u] = x;
        if (lane == 63) s_w[u][wave] = x;
    }
    __syncthreads();
    if (wave == 0) {
        long long tot[3] = {0, 0, 0};
#pragma unroll
        for (int u = 0; u < kMetSubs; ++u)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const unsigned long long v = s_w[u][w];
                tot[0] += (long long)(v & 0xffffu); tot[1] += (long long)((v >> 16) & 0xffffu); tot[2] += (long long)(v >> 32);
            }
        // ONE word per tile: status << 62 | three 20-bit counts.  A tile's own counts are at most 4096; inclusive counts are
        // saturated at kMdSat = 2^20 - 1 >= the 1,000,000 cut-offs - a position at or beyond a cut-off is never written to and
        // every comparison with the cut-off comes out the same -, so that a look-back step polls 4 KB of words, not 12 KB
        // (three words per tile: hundreds of waiting tiles polling uncached words slowed the whole chip, 0.09 of the roofline).
        unsigned long long* own = desc + tile;
        auto pack = [](const long long (&v)[3], unsigned long long status) {
            unsigned long long w = status << kMdShift;
#pragma unroll
            for (int j = 0; j < 3; ++j) w |= (unsigned long long)(v[j] < kMdSat ? v[j] : kMdSat) << (20 * j);
            return w;
        };
        long long before[3] = {0, 0, 0};
        if (tile == 0u) {
#pragma unroll
            for (int j = 0; j < 3; ++j) before[j] = (long long)state[j];   // what earlier chunks of the stream counted
        } else {
            if (lane == 0) __hip_atomic_store(own, pack(tot, 1ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long long look = (long long)tile - 1;
            for (;;) {                                       // uniform: a step of 64 x kMetLook tiles back per turn
                // the lane's own tiles, nearest first: their counts up to and including the first inclusive one
                long long part[3] = {0, 0, 0};
                bool closed = false, again = false;
#pragma unroll
                for (int q = 0; q < kMetLook; ++q) {
                    const long long idx = look - (lane * kMetLook + q);
                    // (in front of tile 0: nothing, and tile 0 is always inclusive)
                    const unsigned long long w = idx >= 0 ? __hip_atomic_load(desc + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                          : (2ull << kMdShift);
                    if (!closed) {
                        again = again || (w >> kMdShift) == 0ull;
#pragma unroll
                        for (int j = 0; j < 3; ++j) part[j] += (long long)((w >> (20 * j)) & kMdSat);
                        closed = (w >> kMdShift) == 2ull;
                    }
                }
                const unsigned long long cm = __ballot(closed);
                const int first = cm ? __ffsll((long long)cm) - 1 : 64;
                if (__ballot(again && lane <= first) != 0ull) {   // (tiles beyond the first inclusive one do not matter)
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) before[j] += wsum64(lane <= first ? part[j] : 0ll);
                if (cm) break;
                look -= 64 * kMetLook;
            }
        }
        if (lane == 0) {
            long long after[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                before[j] = before[j] < kMdSat ? before[j] : kMdSat;
                after[j] = before[j] + tot[j];
                s_pre[j] = before[j];
                if (tile == nb - 1u) state[j] = (unsigned long long)(after[j] < kMdSat ? after[j] : kMdSat);   // (tile 0 has read it long ago)
            }
            __hip_atomic_store(own, pack(after, 2ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0 && tile == nb - 1u) state[5] += (unsigned long long)(end - start);
    }
    __syncthreads();
    long long run[3] = {s_pre[0], s_pre[1], s_pre[2]};       // the counts in front of the sub-tile at hand
    int n_c = 0, n_d = 0;
#pragma unroll
    for (int u = 0; u < kMetSubs; ++u) {
        unsigned long long sub = 0, mywaves = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const unsigned long long v = s_w[u][w];
            sub += v;                                        // (fields of at most 1024: no carries)
            if (w < wave) mywaves += v;
        }
        const unsigned long long ex = mywaves + incl[u] - mine[u];
        long long pa = run[0] + (long long)(ex & 0xffffu);
        long long pb = run[1] + (long long)((ex >> 16) & 0xffffu);
        long long pd = run[2] + (long long)(ex >> 32);
#pragma unroll
        for (int k = 0; k < kMetVec; ++k) {
            const uint32_t bit = 1u << k;
            if (f[u].a & bit) {
                if (want_isize && pa < kSampleCap) isize_out[pa] = f[u].val[k];
                pa++;
            }
            if (f[u].b & bit) {
                const bool in = pb < kSampleCap;   // this record is among the first 1,000,000 on the top contigs
                pb++;
                if (in) {
                    if (f[u].c & bit) n_c++;
                    if (f[u].d & bit) { contam_out[pd] = f[u].val[k]; n_d++; }
                }
            }
            if (f[u].d & bit) pd++;
        }
        run[0] += (long long)(sub & 0xffffu); run[1] += (long long)((sub >> 16) & 0xffffu); run[2] += (long long)(sub >> 32);
    }
    n_c = wsum(n_c);
    n_d = wsum(n_d);
    if (lane == 0) { s_tot[wave][0] = n_c; s_tot[wave][1] = n_d; }
    __syncthreads();
    if (t < 2) {
        const int v = s_tot[0][t] + s_tot[1][t] + s_tot[2][t] + s_tot[3][t];
        if (v) atomicAdd(&state[3 + t], (unsigned long long)v);
    }
}

__global__ void metrics_commit_kernel(long long* __restrict__ state, const long long* __restrict__ totals,
                                      long long scanned) {
    if (threadIdx.x < 3) state[threadIdx.x] = totals[threadIdx.x];
    if (threadIdx.x == 3) state[5] += scanned;
}

__global__ __launch_bounds__(256) void value_hist_kernel(const int32_t* __restrict__ values, int64_t n,
                                                         int64_t n_bins, unsigned long long* __restrict__ hist,
                                                         unsigned long long* __restrict__ overflow) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long over = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int32_t v = values[i];
        if (v >= 0 && v < n_bins) atomicAdd(&hist[v], 1ull);
        else over++;
    }
    if (over) atomicAdd(overflow, over);
}

}  // namespace

size_t metrics_workspace_bytes(int64_t n) {
    const size_t nb = (size_t)((n + kMetTile - 1) / kMetTile) + 1;
    return align_up(nb * 3 * 4, 256) + align_up(nb * 3 * 8, 256) + 256;
}

// Scan records [start, start+count).  state: 6 x int64 on the device (see layout above).
// count_only: only the three running counts (state[0..2]) advance - the first phase of a sharded scan, whose
// slices need the counts of the slices before them to place their samples.
int launch_metrics(hipStream_t s, const MetricsArgs& a, int64_t start, int64_t count, int32_t* isize_out,
                   int32_t* contam_out, int64_t* state, void* ws, size_t ws_bytes, bool count_only) {
    if (count <= 0) return BESST_OK;
    BESST_REQUIRE((start & 3) == 0, "metrics: chunk start must be a multiple of 4");
    BESST_REQUIRE(ws && ws_bytes >= metrics_workspace_bytes(count), "metrics: workspace too small");
    const uint32_t nb = (uint32_t)((count + kMetTile - 1) / kMetTile);
    char* p = static_cast<char*>(ws);
    uint32_t* blk = reinterpret_cast<uint32_t*>(p);
    long long* base = reinterpret_cast<long long*>(p + align_up((size_t)(nb + 1) * 3 * 4, 256));
    long long* totals = reinterpret_cast<long long*>(p + align_up((size_t)(nb + 1) * 3 * 4, 256) +
                                                     align_up((size_t)(nb + 1) * 3 * 8, 256));
    const int64_t end = start + count;
    ProfScope ps(s, kProfMetrics);
    if (!count_only) {
        // one pass: the descriptor words (in the place of the old form's bases) and the ticket start from zero
        const uint32_t nbig = (uint32_t)((count + kMetBig - 1) / kMetBig);
        BESST_HIP_TRY(hipMemsetAsync(base, 0, (size_t)nbig * 8 + 64, s));
        uint32_t* ticket = reinterpret_cast<uint32_t*>(base + (size_t)nbig);
        hipLaunchKernelGGL(metrics_onepass_kernel, dim3(nbig), dim3(kMetThreads), 0, s, a, start, end, nbig,
                           reinterpret_cast<unsigned long long*>(base), ticket, isize_out != nullptr ? 1 : 0, isize_out,
                           contam_out, reinterpret_cast<unsigned long long*>(state));
        BESST_HIP_TRY(hipGetLastError());
        return BESST_OK;
    }
    // count only (the first phase of a sharded scan): one read, the three totals
    hipLaunchKernelGGL(metrics_count_kernel, dim3(nb), dim3(kMetThreads), 0, s, a, start, end, blk);
    hipLaunchKernelGGL(metrics_scan_kernel, dim3(1), dim3(1024), 0, s, blk, nb,
                       reinterpret_cast<const long long*>(state), base, totals);
    hipLaunchKernelGGL(metrics_commit_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<long long*>(state), totals, 0ll);
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

int launch_value_histogram(hipStream_t s, const int32_t* values, int64_t n, int64_t n_bins,
                           unsigned long long* hist, unsigned long long* overflow) {
    if (n <= 0) return BESST_OK;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(value_hist_kernel, dim3(blocks), dim3(256), 0, s, values, n, n_bins, hist, overflow);
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

}  // namespace besst
