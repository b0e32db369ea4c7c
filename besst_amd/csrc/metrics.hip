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
// Ordered semantics come from a count / scan / emit triple per chunk: exclusive prefixes of A, B and D
// in stream order decide inclusion and the output slot, so isize_out / contam_out hold |tlen| in BAM
// order exactly like the Python lists.  The float finishing (means, trimming, GetDistr) replays the
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

__device__ __forceinline__ Flags4 eval4(const MetricsArgs& m, int64_t i0, int64_t end) {
    Flags4 f;
    f.a = f.b = f.c = f.d = 0;
    int32_t tid[kMetVec], mtid[kMetVec], tlen[kMetVec];
    uint32_t flag[kMetVec], mapq[kMetVec];
    if (i0 + kMetVec <= end && (i0 & 3) == 0) {
        const int4 v0 = *reinterpret_cast<const int4*>(m.tid + i0);
        const int4 v1 = *reinterpret_cast<const int4*>(m.mtid + i0);
        const int4 v2 = *reinterpret_cast<const int4*>(m.tlen + i0);
        const ushort4 fl = *reinterpret_cast<const ushort4*>(m.flag + i0);
        const uchar4 mq = *reinterpret_cast<const uchar4*>(m.mapq + i0);
        tid[0] = v0.x; tid[1] = v0.y; tid[2] = v0.z; tid[3] = v0.w;
        mtid[0] = v1.x; mtid[1] = v1.y; mtid[2] = v1.z; mtid[3] = v1.w;
        tlen[0] = v2.x; tlen[1] = v2.y; tlen[2] = v2.z; tlen[3] = v2.w;
        flag[0] = fl.x; flag[1] = fl.y; flag[2] = fl.z; flag[3] = fl.w;
        mapq[0] = mq.x; mapq[1] = mq.y; mapq[2] = mq.z; mapq[3] = mq.w;
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
        const int64_t t64 = tlen[k];
        const int64_t at = t64 < 0 ? -t64 : t64;
        f.val[k] = (int32_t)at;
        const bool top = (uint32_t)tid[k] < (uint32_t)m.n_contigs && m.top_mask[tid[k]] != 0;
        if (!top) continue;
        f.b |= 1u << k;
        if (!(flag[k] & kFlagUnmapped)) f.c |= 1u << k;
        const bool rev = flag[k] & kFlagReverse, mrev = flag[k] & kFlagMateReverse;
        const bool base = (flag[k] & kFlagRead2) && tid[k] == mtid[k] && !(flag[k] & kFlagMateUnmapped) &&
                          (int32_t)mapq[k] > m.min_mapq && !(flag[k] & kFlagSecondary);
        const bool innie = base && ((rev && !mrev && tlen[k] < 0) || (!rev && mrev && tlen[k] > 0));
        const bool outie = base && ((rev && !mrev && tlen[k] > 0) || (!rev && mrev && tlen[k] < 0));
        if (m.rf ? outie : innie) f.a |= 1u << k;
        if (m.rf) {
            if (innie && m.read_len < (double)at) f.d |= 1u << k;
        } else {
            if (outie && m.read_len < (double)at + 2.0 * m.read_len) f.d |= 1u << k;
        }
    }
    return f;
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

__global__ __launch_bounds__(kMetThreads) void metrics_emit_kernel(MetricsArgs m, int64_t start, int64_t end,
                                                                   const long long* __restrict__ base,
                                                                   int want_isize, int32_t* __restrict__ isize_out,
                                                                   int32_t* __restrict__ contam_out,
                                                                   unsigned long long* __restrict__ state) {
    __shared__ int s[4][3];
    __shared__ int s_tot[4][2];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t i0 = start + (int64_t)blockIdx.x * kMetTile + (int64_t)t * kMetVec;
    const Flags4 f = eval4(m, i0, end);
    int c[3] = {(int)__popc(f.a), (int)__popc(f.b), (int)__popc(f.d)};
    int x[3] = {c[0], c[1], c[2]};
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int o = __shfl_up(x[j], d, 64);
            if (lane >= d) x[j] += o;
        }
    }
    if (lane == 63) { s[wave][0] = x[0]; s[wave][1] = x[1]; s[wave][2] = x[2]; }
    __syncthreads();
    long long pa = base[(size_t)blockIdx.x * 3 + 0] + x[0] - c[0];
    long long pb = base[(size_t)blockIdx.x * 3 + 1] + x[1] - c[1];
    long long pd = base[(size_t)blockIdx.x * 3 + 2] + x[2] - c[2];
    for (int w = 0; w < wave; ++w) { pa += s[w][0]; pb += s[w][1]; pd += s[w][2]; }
    int n_c = 0, n_d = 0;
#pragma unroll
    for (int k = 0; k < kMetVec; ++k) {
        const uint32_t bit = 1u << k;
        if (f.a & bit) {
            if (want_isize && pa < kSampleCap) isize_out[pa] = f.val[k];
            pa++;
        }
        if (f.b & bit) {
            const bool in = pb < kSampleCap;   // this record is among the first 1,000,000 on the top contigs
            pb++;
            if (in) {
                if (f.c & bit) n_c++;
                if (f.d & bit) { contam_out[pd] = f.val[k]; n_d++; }
            }
        }
        if (f.d & bit) pd++;
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
    hipLaunchKernelGGL(metrics_count_kernel, dim3(nb), dim3(kMetThreads), 0, s, a, start, end, blk);
    hipLaunchKernelGGL(metrics_scan_kernel, dim3(1), dim3(1024), 0, s, blk, nb,
                       reinterpret_cast<const long long*>(state), base, totals);
    if (!count_only)
        hipLaunchKernelGGL(metrics_emit_kernel, dim3(nb), dim3(kMetThreads), 0, s, a, start, end, base,
                           isize_out != nullptr ? 1 : 0, isize_out, contam_out,
                           reinterpret_cast<unsigned long long*>(state));
    hipLaunchKernelGGL(metrics_commit_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<long long*>(state), totals,
                       (long long)(count_only ? 0 : count));
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
