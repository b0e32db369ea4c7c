// Per-edge scoring on gfx950: the data-parallel part of GiveScoreOnEdges, normal-distribution branch
// (CreateGraph.py:498-614).  One workgroup per scored edge:
//   * thread 0: naive gap, ML gap (bisection on d + sigma^2 g'(d)/g(d) = mu - mean_obs) and the
//     expected std-dev of the truncated/skewed spanning density.  These restate the un-vendored
//     mathstats 0.2.6.5 routines from the publication (besst_amd/mathstats_compat.py, same
//     expressions in the same order; parity unpinned, +-1 bp tolerance on the gap).
//   * all threads: the link-dispersity statistic.  l1 = sorted(obs of first endpoint) - mean,
//     l2 = sorted(max(obs2) - obs2) - mean (:582-594); the two lists are bitonic-sorted in LDS (global
//     scratch for edges with more than 8192 links) and h = max |#{l1 <= x} - #{l2 <= x}| over the pooled
//     points is found with fp64 upper-bound searches, so ties are decided exactly like Python's float
//     comparisons.  The KS statistic is h / n; the host forms the score.
// fp64 throughout, compiled with -ffp-contract=off.
#include <math.h>

#include "common.h"

namespace besst {

namespace {

constexpr int kScoreThreads = 256;
constexpr int kLdsCap = 8192;   // links per edge sorted in LDS (2 x 32 KB)

struct Moments {
    double m0, m1, m2, gprime;
};

// Weighted centred moments of the spanning-fragment density and g'(d); see mathstats_compat.py.
__device__ Moments gap_moments(double d, double mean, double sigma, double c_min, double c_max, double r) {
    const double s2 = sigma * sigma;
    const double inv = 1.0 / sqrt(2.0 * M_PI) / sigma;
    const double sq2s = sqrt(2.0) * sigma;
    const double lo1 = d + 2.0 * r - 1.0 - mean;
    const double hi1 = d + c_min + r - mean;
    const double lo3 = d + c_max + r - mean;
    const double hi3 = d + c_min + c_max + 1.0 - mean;
    const double pa_[3] = {lo1, hi1, lo3}, pb_[3] = {hi1, lo3, hi3};
    const double alpha_[3] = {1.0, 0.0, -1.0}, beta_[3] = {-lo1, c_min - r + 1.0, hi3};
    Moments M{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const double a = pa_[p], b = pb_[p], alpha = alpha_[p], beta = beta_[p];
        if (!(b > a)) continue;
        const double pa = inv * exp(-(a * a) / (2.0 * s2));
        const double pb = inv * exp(-(b * b) / (2.0 * s2));
        const double B0 = 0.5 * (1.0 + erf(b / sq2s)) - 0.5 * (1.0 + erf(a / sq2s));
        const double B1 = -s2 * (pb - pa);
        const double B2 = 1.0 * s2 * B0 - s2 * (b * pb - a * pa);
        const double B3 = 2.0 * s2 * B1 - s2 * ((b * b) * pb - (a * a) * pa);
        M.m0 += alpha * B1 + beta * B0;
        M.m1 += alpha * B2 + beta * B1;
        M.m2 += alpha * B3 + beta * B2;
    }
    if (hi3 > lo3) M.gprime += 0.5 * (1.0 + erf(hi3 / sq2s)) - 0.5 * (1.0 + erf(lo3 / sq2s));
    if (hi1 > lo1) M.gprime -= 0.5 * (1.0 + erf(hi1 / sq2s)) - 0.5 * (1.0 + erf(lo1 / sq2s));
    return M;
}

// ML gap: the reference restatement bisects d + sigma^2 g'(d)/g(d) = mean - mean_obs from
// [trunc(-4 sigma), trunc(mean + 4 sigma - 2r)] down to unit width (10-14 dependent steps, ~16 erf/exp each).  The
// workgroup replays EXACTLY that bisection, eight levels per round: every midpoint the bisection can reach within
// the next eight steps is a node of a 255-node dyadic tree over the current bracket; one thread evaluates one node
// (the interval ends are integers and the tree spacing a power-of-two fraction, so the fp64 midpoints are the very
// values (upper + lower) / 2.0 produces), then a single thread walks the eight comparisons.  Same result as the
// sequential loop whether or not the function is numerically monotone.
__device__ double gap_estimator_parallel(double mean, double sigma, double r, double mean_obs, double c1, double c2,
                                         unsigned char* s_cmp, double* s_bracket) {
    const double c_min = c1 < c2 ? c1 : c2, c_max = c1 < c2 ? c2 : c1;
    const double naive = mean - mean_obs;
    const int t = threadIdx.x;
    if (t == 0) {
        s_bracket[0] = trunc(-4 * sigma);                    // lower
        s_bracket[1] = trunc(mean + 4 * sigma - 2 * r);      // upper
    }
    __syncthreads();
    while (true) {
        const double lower = s_bracket[0], upper = s_bracket[1];
        if (!(upper - lower > 1)) break;                     // uniform: every thread reads the same bracket
        // node i (1..255) of the tree = lower + i * (upper - lower) / 256
        if (t >= 1 && t < 256) {
            const double mid = lower + (double)t * ((upper - lower) / 256.0);
            const Moments M = gap_moments(mid, mean, sigma, c_min, c_max, r);
            const double f = M.m0 > 0.0 ? mid + sigma * sigma * M.gprime / M.m0 : mid;
            s_cmp[t] = f > naive ? 1 : 0;
        }
        __syncthreads();
        if (t == 0) {
            double lo = lower, hi = upper;
            int ilo = 0, ihi = 256;
            for (int step = 0; step < 8 && hi - lo > 1; ++step) {
                const int imid = (ilo + ihi) >> 1;
                const double mid = (hi + lo) / 2.0;
                if (s_cmp[imid]) { hi = mid; ihi = imid; } else { lo = mid; ilo = imid; }
            }
            s_bracket[0] = lo;
            s_bracket[1] = hi;
        }
        __syncthreads();
    }
    return floor((s_bracket[1] + s_bracket[0]) / 2.0 + 0.5);
}

__device__ double tr_sk_std_dev(double mean, double sigma, double r, double c1, double c2, double d) {
    const double c_min = c1 < c2 ? c1 : c2, c_max = c1 < c2 ? c2 : c1;
    const Moments M = gap_moments(d, mean, sigma, c_min, c_max, r);
    if (!(M.m0 > 0.0)) return 4294967296.0;
    const double e1 = M.m1 / M.m0;
    const double var = M.m2 / M.m0 - e1 * e1;
    return var > 0.0 ? sqrt(var) : 0.0;
}

// in-place ascending bitonic sort of a[0..np) (np power of two) by the whole workgroup
__device__ void bitonic_sort(int32_t* a, int np) {
    for (int k = 2; k <= np; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < np; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const int32_t x = a[i], y = a[ixj];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { a[i] = y; a[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
}

// #{k < n : (double)a[k] - m <= x} for ascending a
__device__ __forceinline__ int upper_bound_centred(const int32_t* a, int n, double m, double x) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((double)a[mid] - m <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <int CAP, bool kSmall>
__global__ __launch_bounds__(kScoreThreads) void score_kernel(ScoreArgs a, double* __restrict__ gap_out,
                                                              double* __restrict__ sd0_out,
                                                              int32_t* __restrict__ ks_out,
                                                              uint8_t* __restrict__ flags_out,
                                                              int32_t* __restrict__ big_scratch,
                                                              const unsigned long long* __restrict__ big_off) {
    __shared__ int32_t s_buf[2 * CAP];
    __shared__ long long s_red[4];
    __shared__ int s_max[4];
    __shared__ unsigned char s_cmp[256];
    __shared__ double s_bracket[2];
    const int e = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t row = a.row[e];
    const int n = (int)a.row_n[row];
    {   // two launches share the edge list: small edges sort in a 2 x 1024-entry LDS buffer (many workgroups per
        // CU), the rest in 2 x 8192 entries or, beyond that, in global scratch
        int np_probe = 1;
        while (np_probe < n) np_probe <<= 1;
        if (kSmall != (np_probe <= 1024)) return;
    }
    const uint32_t off = a.row_offset[row];
    const bool swap = a.swap[e] != 0;
    const int32_t* src1 = (swap ? a.obs_hi : a.obs_lo) + off;
    const int32_t* src2 = (swap ? a.obs_lo : a.obs_hi) + off;

    {
        const double len1 = (double)a.len1[e], len2 = (double)a.len2[e];
        const double obs = (double)a.row_sum[row];
        const double nf = (double)n;
        const double mean_ = obs / nf;
        const double data_observation = (nf * a.mean - obs) / nf;
        const bool long_enough = 2 * a.sigma < len1 && 2 * a.sigma < len2;       // uniform per workgroup
        double gap = data_observation;
        if (long_enough) gap = gap_estimator_parallel(a.mean, a.sigma, a.read_len, mean_, len1, len2, s_cmp, s_bracket);
        if (t == 0) {
            uint8_t fl = long_enough ? 1 : 0;
            if (-gap > len1 || -gap > len2) fl |= 2;
            gap_out[e] = gap;
            sd0_out[e] = long_enough ? tr_sk_std_dev(a.mean, a.sigma, a.read_len, len1, len2, gap) : 4294967296.0;
            flags_out[e] = fl;
        }
    }

    int np = 1;
    while (np < n) np <<= 1;
    int32_t* l1;
    int32_t* l2;
    if (np <= CAP) {
        l1 = s_buf;
        l2 = s_buf + np;
    } else {
        l1 = big_scratch + big_off[e];
        l2 = l1 + np;
    }
    // pass 1: load, sums, max of the second list
    long long sum1 = 0;
    int mx = -2147483647 - 1;
    for (int i = t; i < n; i += kScoreThreads) {
        const int32_t x = src1[i], y = src2[i];
        l1[i] = x;
        l2[i] = y;
        sum1 += x;
        mx = y > mx ? y : mx;
    }
    for (int i = n + t; i < np; i += kScoreThreads) l1[i] = 2147483647;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        sum1 += __shfl_xor(sum1, d, 64);
        const int o = __shfl_xor(mx, d, 64);
        mx = o > mx ? o : mx;
    }
    if (lane == 0) { s_red[wave] = sum1; s_max[wave] = mx; }
    __syncthreads();
    sum1 = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    mx = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    __syncthreads();
    // l2 := max - obs2 (ascending after the sort), with its sum
    long long sum2 = 0;
    for (int i = t; i < n; i += kScoreThreads) {
        const int32_t dlt = mx - l2[i];
        l2[i] = dlt;
        sum2 += dlt;
    }
    for (int i = n + t; i < np; i += kScoreThreads) l2[i] = 2147483647;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum2 += __shfl_xor(sum2, d, 64);
    if (lane == 0) s_red[wave] = sum2;
    __syncthreads();
    sum2 = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    __syncthreads();
    bitonic_sort(l1, np);
    bitonic_sort(l2, np);
    const double m1 = (double)sum1 / (double)n;
    const double m2 = (double)sum2 / (double)n;
    int h = 0;
    for (int i = t; i < n; i += kScoreThreads) {
        {
            const double x = (double)l1[i] - m1;
            const int c1 = upper_bound_centred(l1, n, m1, x);
            const int c2 = upper_bound_centred(l2, n, m2, x);
            const int dff = c1 > c2 ? c1 - c2 : c2 - c1;
            h = dff > h ? dff : h;
        }
        {
            const double x = (double)l2[i] - m2;
            const int c1 = upper_bound_centred(l1, n, m1, x);
            const int c2 = upper_bound_centred(l2, n, m2, x);
            const int dff = c1 > c2 ? c1 - c2 : c2 - c1;
            h = dff > h ? dff : h;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const int o = __shfl_xor(h, d, 64);
        h = o > h ? o : h;
    }
    if (lane == 0) s_max[wave] = h;
    __syncthreads();
    if (t == 0) ks_out[e] = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
}

// d + sigma^2 g'(d) / g(d) for a run of consecutive gaps d = d_lower, d_lower + 1, ... between two contigs of the same
// length: the table mathstats' PreCalcMLvaluesOfdLongContigs is built from (MakeScaffolds.py:68,447 look gaps of long
// scaffold pairs up instead of bisecting per edge).  One thread per d.
__global__ __launch_bounds__(256) void gap_table_kernel(double mean, double sigma, double r, double c_len, int32_t d_lower,
                                                        int32_t n, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = (double)(d_lower + i);
    const Moments M = gap_moments(d, mean, sigma, c_len, c_len, r);
    out[i] = M.m0 > 0.0 ? d + sigma * sigma * M.gprime / M.m0 : d;
}

}  // namespace

int launch_gap_table(hipStream_t s, double mean, double sigma, double r, double c_len, int32_t d_lower, int32_t n, double* out) {
    if (n <= 0) return BESST_OK;
    hipLaunchKernelGGL(gap_table_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, mean, sigma, r, c_len, d_lower, n, out);
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

size_t score_workspace_bytes(int64_t n_edges, int64_t n_tuples) {
    // worst case every big edge needs 2 * next_pow2(n) <= 4 n ints of scratch
    return align_up((size_t)(n_edges > 0 ? n_edges : 1) * 8, 256) + align_up((size_t)(n_tuples > 0 ? n_tuples : 1) * 16, 256);
}

int launch_score(hipStream_t s, const ScoreArgs& a, double* gap, double* sd0, int32_t* ks_h, uint8_t* flags,
                 void* ws, size_t ws_bytes) {
    if (a.n_edges <= 0) return BESST_OK;
    BESST_REQUIRE(a.n_edges < ((int64_t)1 << 31), "score: too many edges");
    BESST_REQUIRE(ws != nullptr, "score: null workspace");
    char* p = static_cast<char*>(ws);
    auto* big_off = reinterpret_cast<unsigned long long*>(p);
    auto* big_scratch = reinterpret_cast<int32_t*>(p + align_up((size_t)a.n_edges * 8, 256));
    (void)ws_bytes;
    ProfScope ps(s, kProfScore);
    hipLaunchKernelGGL((score_kernel<1024, true>), dim3((uint32_t)a.n_edges), dim3(kScoreThreads), 0, s, a, gap, sd0,
                       ks_h, flags, big_scratch, big_off);
    hipLaunchKernelGGL((score_kernel<kLdsCap, false>), dim3((uint32_t)a.n_edges), dim3(kScoreThreads), 0, s, a, gap, sd0,
                       ks_h, flags, big_scratch, big_off);
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

}  // namespace besst
