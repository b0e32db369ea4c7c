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
// Log-normal branch (skewed libraries, param.lognormal; CreateGraph.py:485-494,522-531,549-553): the gap is the integer d
// maximising L(d) = sum_i log f(o_i + d) - n log g(d) over the raw observations of the edge (mathstats'
// log_normal_param_est.GapEstimator, restated in besst_amd/mathstats_compat.py: coarse scan with stride 64, then the 129
// gaps around the coarse optimum).  The workgroup evaluates the gaps of a scan side by side, a group of lanes per gap
// over the observations held in LDS; log g(d) comes from prefix tables F0 / F1 of the log-normal pmf built once per
// library (lognormal_tables).  The conditional sigma table of get_conditional_stddevs (:436-469) is one small kernel.
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

// The log-normal pmf on the integers and the scan of the likelihood -------------------------------------------------
struct LogNormal {
    double mu, sigma;          // of log x
    long long x_max;           // support 1 .. x_max
    const double* F0;          // F0[k] = sum_{x <= k} f(x), k = 0 .. x_max
    const double* F1;          // F1[k] = sum_{x <= k} x f(x)
    int32_t max_gap;           // len(conditional_stddevs) - 1 (CreateGraph.py:493,527-528)
};

__device__ __forceinline__ double lognormal_pmf(double x, double mu, double sigma) {
    const double lx = log(x);
    return exp(-((lx - mu) * (lx - mu)) / (2.0 * sigma * sigma)) / (x * sigma * sqrt(2.0 * M_PI));
}

// sums of f and x f over the integers of [a, b] clipped to the support
__device__ __forceinline__ void ln_segment(const LogNormal& ln, long long a, long long b, double& s0, double& s1) {
    a = a < 1 ? 1 : (a > ln.x_max + 1 ? ln.x_max + 1 : a);
    b = b < 0 ? 0 : (b > ln.x_max ? ln.x_max : b);
    if (b >= a) {
        s0 = ln.F0[b] - ln.F0[a - 1];
        s1 = ln.F1[b] - ln.F1[a - 1];
    } else {
        s0 = 0.0;
        s1 = 0.0;
    }
}

// log g(d): the three linear pieces of the placement weight against the prefix tables (mathstats_compat._lognormal_log_g)
__device__ double ln_log_g(const LogNormal& ln, long long d, long long c_min, long long c_max, long long r) {
    double s0, s1;
    ln_segment(ln, d + 2 * r, d + c_min + r - 1, s0, s1);                 // w = x - d - 2r + 1
    double g = s1 - (double)(d + 2 * r - 1) * s0;
    ln_segment(ln, d + c_min + r, d + c_max + r, s0, s1);                 // w = c_min - r + 1
    g = g + (double)(c_min - r + 1) * s0;
    ln_segment(ln, d + c_max + r + 1, d + c_min + c_max, s0, s1);         // w = c_min + c_max + d - x + 1
    g = g + (double)(c_min + c_max + d + 1) * s0 - s1;
    return g > 0.0 ? log(g) : -INFINITY;
}

// log x for x >= 1 by table + series (the scan evaluates it ~400 n times per edge; the library routine is ~3x the
// instructions): x = 2^e m, m in [1, 2); c_j the centre of the j-th 1/128 of [1, 2): m / c_j = 1 + r with |r| < 2^-8, and
// log x = e ln 2 - log(1 / c_j) + log1p(r), the series cut after r^6 / 6 (the next term is below 2e-18).  The table holds
// {1 / c_j rounded, -log of that rounded value}, so the rounding of the reciprocal cancels.  Absolute error ~2e-16 + 1 ulp.
struct LogTable {
    double inv_c, log_c;
};
__device__ __forceinline__ void build_log_table(LogTable* tab) {
    if (threadIdx.x < 128) {
        const double inv = 1.0 / (1.0 + ((double)threadIdx.x + 0.5) / 128.0);
        tab[threadIdx.x] = LogTable{inv, -log(inv)};
    }
}
__device__ __forceinline__ double table_log(double x, const LogTable* tab) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    const unsigned hi = (unsigned)(bits >> 32);
    const int e = (int)(hi >> 20) - 1023;
    const LogTable t = tab[(hi >> 13) & 127u];
    const double m = __longlong_as_double((long long)((bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull));
    const double r = __builtin_fma(m, t.inv_c, -1.0);
    double p = __builtin_fma(r, -1.0 / 6.0, 1.0 / 5.0);
    p = __builtin_fma(r, p, -1.0 / 4.0);
    p = __builtin_fma(r, p, 1.0 / 3.0);
    p = __builtin_fma(r, p, -1.0 / 2.0);
    p = __builtin_fma(r, p, 1.0);
    const double ef = (double)e;
    // ln 2 = hi + lo, hi with 11 trailing zero bits: e hi is exact for |e| < 2048
    return __builtin_fma(ef, 0x1.62e42fefa38p-1, t.log_c) + __builtin_fma(r, p, ef * 0x1.ef35793c7673p-45);
}

struct Best {
    double v;
    long long d;
};
__device__ __forceinline__ void best_take(Best& b, double v, long long d) {   // max, the earlier gap on a tie (argmax)
    if (v > b.v || (v == b.v && d < b.d)) { b.v = v; b.d = d; }
}

// argmax over d = d0, d0 + stride, ... (count gaps) of L(d).  lpg lanes share a gap (lpg a power of two <= 64): each sums
// its share of the observations, the group's sum is formed by xor-shuffles, the group's first lane adds the g term.
template <typename ObsAt>
__device__ long long ln_scan(const LogNormal& ln, long long d0, long long stride, long long count, int n, ObsAt obs_at,
                             int lpg, long long c_min, long long c_max, long long r, double* s_bv, long long* s_bd,
                             const LogTable* s_log, double* s_lg) {
    const int t = threadIdx.x;
    const int slots = kScoreThreads / lpg;
    const int slot = t / lpg, sub = t & (lpg - 1);
    const double k2 = 1.0 / (2.0 * ln.sigma * ln.sigma);
    Best best{-INFINITY, 0x7fffffffffffffffll};
    // 256 gaps at a time: first their log g, one gap per THREAD (inside the pass every lane of a group would evaluate its
    // gap's twelve table reads and one log again: an eighth of the scan for an edge of 400 links), then the passes over
    // them, 256 / lpg gaps per pass
    for (long long blk = 0; blk < count; blk += kScoreThreads) {
        __syncthreads();                                     // (the block before has been read)
        s_lg[t] = blk + t < count ? ln_log_g(ln, d0 + (blk + t) * stride, c_min, c_max, r) : -INFINITY;
        __syncthreads();
        for (int g0 = 0; g0 < kScoreThreads && blk + g0 < count; g0 += slots) {      // uniform
            const int g = g0 + slot;
            const bool valid = blk + g < count;
            const long long d = d0 + (valid ? blk + g : 0) * stride;
            // sum_i log f(o_i + d) + n mu = -(sum u + k2 sum u^2), u = log x - mu (the constant n mu is the same for every gap)
            double s1 = 0.0, s2 = 0.0;
            for (int i = sub; i < n; i += lpg) {
                const double u = table_log((double)(obs_at(i) + d), s_log) - ln.mu;
                s1 += u;
                s2 = __builtin_fma(u, u, s2);
            }
            for (int w = 1; w < lpg; w <<= 1) {
                s1 += __shfl_xor(s1, w, 64);
                s2 += __shfl_xor(s2, w, 64);
            }
            const double acc = -(s1 + k2 * s2) - (double)n * ln.mu;
            if (valid && sub == 0) {
                const double lg = s_lg[g];
                const double v = lg == -INFINITY ? -INFINITY : acc - (double)n * lg;
                best_take(best, v, d);
            }
        }
    }
#pragma unroll
    for (int w = 1; w < 64; w <<= 1) {
        const double ov = __shfl_xor(best.v, w, 64);
        const long long od = __shfl_xor(best.d, w, 64);
        best_take(best, ov, od);
    }
    __syncthreads();
    if ((t & 63) == 0) { s_bv[t >> 6] = best.v; s_bd[t >> 6] = best.d; }
    __syncthreads();
    Best all{s_bv[0], s_bd[0]};
#pragma unroll
    for (int w = 1; w < kScoreThreads / 64; ++w) best_take(all, s_bv[w], s_bd[w]);
    __syncthreads();
    return all.d;
}

// mathstats_compat.lognormal_GapEstimator for one edge, by the whole workgroup.  s_obs: LDS room for cap observations
// (an edge with more reads them from the columns every time).
__device__ double lognormal_gap(const LogNormal& ln, const int32_t* __restrict__ obs_lo, const int32_t* __restrict__ obs_hi,
                                int n, double read_len, long long len1, long long len2, int32_t* s_obs, int cap,
                                double* s_bv, long long* s_bd, int* s_mm, const LogTable* s_log, double* s_lg) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool in_lds = n <= cap;
    int mn = 2147483647, mx = -2147483647 - 1;
    long long sum = 0;
    for (int i = t; i < n; i += kScoreThreads) {
        const int o = obs_lo[i] + obs_hi[i];
        if (in_lds) s_obs[i] = o;
        mn = o < mn ? o : mn;
        mx = o > mx ? o : mx;
        sum += o;
    }
#pragma unroll
    for (int w = 32; w > 0; w >>= 1) {
        const int a = __shfl_xor(mn, w, 64), b = __shfl_xor(mx, w, 64);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
        sum += __shfl_xor(sum, w, 64);
    }
    if (lane == 0) { s_mm[wave] = mn; s_mm[4 + wave] = mx; s_bd[wave] = sum; }
    __syncthreads();
    mn = min(min(s_mm[0], s_mm[1]), min(s_mm[2], s_mm[3]));
    mx = max(max(s_mm[4], s_mm[5]), max(s_mm[6], s_mm[7]));
    sum = s_bd[0] + s_bd[1] + s_bd[2] + s_bd[3];
    __syncthreads();
    const long long r = (long long)rint(read_len);                               // Python's round(): half to even
    const long long c_min = len1 < len2 ? len1 : len2, c_max = len1 < len2 ? len2 : len1;
    const long long d_lo = 1 - (long long)mn, d_hi = ln.x_max - (long long)mx;
    if (d_hi < d_lo) return rint(exp(ln.mu) - (double)sum / (double)n);
    int lpg = 1;
    while (lpg < 16 && lpg * 2 <= n) lpg <<= 1;
    long long best;
    if (in_lds) {
        auto at = [s_obs](int i) { return (long long)s_obs[i]; };
        best = ln_scan(ln, d_lo, 64, (d_hi - d_lo) / 64 + 1, n, at, lpg, c_min, c_max, r, s_bv, s_bd, s_log, s_lg);
        const long long f_lo = best - 64 > d_lo ? best - 64 : d_lo, f_hi = best + 64 < d_hi ? best + 64 : d_hi;
        best = ln_scan(ln, f_lo, 1, f_hi - f_lo + 1, n, at, lpg, c_min, c_max, r, s_bv, s_bd, s_log, s_lg);
    } else {
        auto at = [obs_lo, obs_hi](int i) { return (long long)(obs_lo[i] + obs_hi[i]); };
        best = ln_scan(ln, d_lo, 64, (d_hi - d_lo) / 64 + 1, n, at, lpg, c_min, c_max, r, s_bv, s_bd, s_log, s_lg);
        const long long f_lo = best - 64 > d_lo ? best - 64 : d_lo, f_hi = best + 64 < d_hi ? best + 64 : d_hi;
        best = ln_scan(ln, f_lo, 1, f_hi - f_lo + 1, n, at, lpg, c_min, c_max, r, s_bv, s_bd, s_log, s_lg);
    }
    return (double)best;
}

template <int CAP, bool kSmall, bool kLogNormal>
__global__ __launch_bounds__(kScoreThreads) void score_kernel(ScoreArgs a, LogNormal ln, double* __restrict__ gap_out,
                                                              double* __restrict__ sd0_out,
                                                              int32_t* __restrict__ ks_out,
                                                              uint8_t* __restrict__ flags_out,
                                                              int32_t* __restrict__ big_scratch,
                                                              const unsigned long long* __restrict__ big_off) {
    __shared__ int32_t s_buf[2 * CAP];
    __shared__ long long s_red[4];
    __shared__ int s_max[8];
    __shared__ unsigned char s_cmp[256];
    __shared__ double s_bracket[4];
    __shared__ LogTable s_log[kLogNormal ? 128 : 1];
    __shared__ double s_lg[kLogNormal ? kScoreThreads : 1];
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
        if constexpr (kLogNormal) {
            if (long_enough) {
                build_log_table(s_log);                          // (the barriers of lognormal_gap stand before its use)
                gap = lognormal_gap(ln, a.obs_lo + off, a.obs_hi + off, n, a.read_len, (long long)a.len1[e],
                                    (long long)a.len2[e], s_buf, 2 * CAP, s_bracket, s_red, s_max, s_log, s_lg);
                if (gap > (double)ln.max_gap) gap = (double)ln.max_gap;          // :527-528
            }
        } else {
            if (long_enough) gap = gap_estimator_parallel(a.mean, a.sigma, a.read_len, mean_, len1, len2, s_cmp, s_bracket);
        }
        if (t == 0) {
            uint8_t fl = long_enough ? 1 : 0;
            if (-gap > len1 || -gap > len2) fl |= 2;
            gap_out[e] = gap;
            if constexpr (kLogNormal) {
                sd0_out[e] = 4294967296.0;       // the caller indexes the conditional sigma table with the gap (:549-553)
            } else {
                sd0_out[e] = long_enough ? tr_sk_std_dev(a.mean, a.sigma, a.read_len, len1, len2, gap) : 4294967296.0;
            }
            flags_out[e] = fl;
        }
        __syncthreads();
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

namespace {

template <bool kLogNormal>
int launch_score_impl(hipStream_t s, const ScoreArgs& a, const LogNormal& ln, double* gap, double* sd0, int32_t* ks_h,
                      uint8_t* flags, void* ws) {
    if (a.n_edges <= 0) return BESST_OK;
    BESST_REQUIRE(a.n_edges < ((int64_t)1 << 31), "score: too many edges");
    BESST_REQUIRE(ws != nullptr, "score: null workspace");
    char* p = static_cast<char*>(ws);
    auto* big_off = reinterpret_cast<unsigned long long*>(p);
    auto* big_scratch = reinterpret_cast<int32_t*>(p + align_up((size_t)a.n_edges * 8, 256));
    ProfScope ps(s, kProfScore);
    hipLaunchKernelGGL((score_kernel<1024, true, kLogNormal>), dim3((uint32_t)a.n_edges), dim3(kScoreThreads), 0, s, a, ln,
                       gap, sd0, ks_h, flags, big_scratch, big_off);
    hipLaunchKernelGGL((score_kernel<kLdsCap, false, kLogNormal>), dim3((uint32_t)a.n_edges), dim3(kScoreThreads), 0, s, a,
                       ln, gap, sd0, ks_h, flags, big_scratch, big_off);
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

// ---- prefix tables of the pmf: F0[k] = sum_{x <= k} f(x), F1[k] = sum_{x <= k} x f(x), k = 0 .. x_max ------------------
// (mathstats_compat._lognormal_tables).  Three launches: sums per 2048-value tile, one workgroup scans the tile sums, every
// tile writes its prefixes.  The tile and scan orders are fixed, so the tables are the same bits on every call.
constexpr int kLnThreads = 256, kLnPer = 8, kLnTile = kLnThreads * kLnPer;

__device__ __forceinline__ void block_scan2(double& a, double& b, double* s_a, double* s_b, double& tot_a, double& tot_b) {
    // EXCLUSIVE scan of (a, b) over the workgroup's threads (sums of the threads in front, added in thread order: nothing
    // is ever subtracted, a small prefix in front of a large term keeps its bits); totals in tot_*
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int w = 1; w < 64; w <<= 1) {
        const double oa = __shfl_up(a, w, 64), ob = __shfl_up(b, w, 64);
        if (lane >= w) { a = oa + a; b = ob + b; }
    }
    if (lane == 63) { s_a[wave] = a; s_b[wave] = b; }
    const double ea = __shfl_up(a, 1, 64), eb = __shfl_up(b, 1, 64);
    a = lane ? ea : 0.0;
    b = lane ? eb : 0.0;
    __syncthreads();
    double pa = 0.0, pb = 0.0, ta = 0.0, tb = 0.0;
    for (int w = 0; w < kLnThreads / 64; ++w) {
        if (w < wave) { pa += s_a[w]; pb += s_b[w]; }
        ta += s_a[w];
        tb += s_b[w];
    }
    a = pa + a;
    b = pb + b;
    tot_a = ta;
    tot_b = tb;
    __syncthreads();
}

template <bool kWrite>
__global__ __launch_bounds__(kLnThreads) void ln_tile_kernel(double mu, double sigma, long long x_max, double* __restrict__ tile0,
                                                             double* __restrict__ tile1, double* __restrict__ F0,
                                                             double* __restrict__ F1) {
    __shared__ double s_a[kLnThreads / 64], s_b[kLnThreads / 64];
    const long long first = (long long)blockIdx.x * kLnTile + (long long)threadIdx.x * kLnPer + 1;   // x of this thread's first value
    double f[kLnPer];
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int k = 0; k < kLnPer; ++k) {
        const long long x = first + k;
        f[k] = x <= x_max ? lognormal_pmf((double)x, mu, sigma) : 0.0;
        a += f[k];
        b += f[k] * (double)x;
    }
    double ta, tb;
    block_scan2(a, b, s_a, s_b, ta, tb);
    if (!kWrite) {
        if (threadIdx.x == 0) { tile0[blockIdx.x] = ta; tile1[blockIdx.x] = tb; }
        return;
    }
    // (a, b) = sums of the threads in front of this one; the tile offsets were scanned exclusively in place
    double ra = tile0[blockIdx.x] + a, rb = tile1[blockIdx.x] + b;
    if (blockIdx.x == 0 && threadIdx.x == 0) { F0[0] = 0.0; F1[0] = 0.0; }
#pragma unroll
    for (int k = 0; k < kLnPer; ++k) {
        const long long x = first + k;
        ra += f[k];
        rb += f[k] * (double)x;
        if (x <= x_max) { F0[x] = ra; F1[x] = rb; }
    }
}

__global__ __launch_bounds__(kLnThreads) void ln_tile_scan_kernel(double* __restrict__ tile0, double* __restrict__ tile1, long long n_tiles) {
    __shared__ double s_a[kLnThreads / 64], s_b[kLnThreads / 64];
    double carry_a = 0.0, carry_b = 0.0;
    for (long long base = 0; base < n_tiles; base += kLnThreads) {
        const long long i = base + threadIdx.x;
        const double va = i < n_tiles ? tile0[i] : 0.0, vb = i < n_tiles ? tile1[i] : 0.0;
        double a = va, b = vb, ta, tb;
        block_scan2(a, b, s_a, s_b, ta, tb);
        if (i < n_tiles) { tile0[i] = carry_a + a; tile1[i] = carry_b + b; }
        carry_a += ta;
        carry_b += tb;
    }
}

// get_conditional_stddevs (CreateGraph.py:436-469): for gap = steps[blockIdx.x], the sigma of the density
// f(x) * max(0, x - gap + 1) over x = 0 .. max_isize (f dense, 0 where the empirical distribution has no entry).
__global__ __launch_bounds__(kLnThreads) void cond_stddev_kernel(const double* __restrict__ f, long long max_isize,
                                                                 const int32_t* __restrict__ steps, double* __restrict__ out) {
    __shared__ double s_a[kLnThreads / 64], s_b[kLnThreads / 64];
    const long long gap = steps[blockIdx.x];
    double a = 0.0, b = 0.0, ta, tb;
    for (long long x = threadIdx.x; x <= max_isize; x += kLnThreads) {
        const long long w = x - gap + 1;
        const double v = w > 0 ? f[x] * (double)w : 0.0;
        a += v;
        b += (double)x * v;
    }
    block_scan2(a, b, s_a, s_b, ta, tb);
    const double tot = ta, mu = tb / tot;
    a = 0.0;
    b = 0.0;
    for (long long x = threadIdx.x; x <= max_isize; x += kLnThreads) {
        const long long w = x - gap + 1;
        const double v = w > 0 ? f[x] * (double)w : 0.0;
        const double dx = (double)x - mu;
        a += dx * dx * v;
    }
    block_scan2(a, b, s_a, s_b, ta, tb);
    if (threadIdx.x == 0) out[blockIdx.x] = sqrt(ta / tot);
}

}  // namespace

int launch_score(hipStream_t s, const ScoreArgs& a, double* gap, double* sd0, int32_t* ks_h, uint8_t* flags,
                 void* ws, size_t ws_bytes) {
    (void)ws_bytes;
    return launch_score_impl<false>(s, a, LogNormal{}, gap, sd0, ks_h, flags, ws);
}

int launch_score_lognormal(hipStream_t s, const ScoreArgs& a, const LogNormalArgs& l, double* gap, double* sd0, int32_t* ks_h,
                           uint8_t* flags, void* ws, size_t ws_bytes) {
    (void)ws_bytes;
    BESST_REQUIRE(l.sigma > 0.0 && l.x_max >= 1 && l.F0 && l.F1, "score: log-normal tables missing");
    LogNormal ln{l.mu, l.sigma, (long long)l.x_max, l.F0, l.F1, l.max_gap};
    return launch_score_impl<true>(s, a, ln, gap, sd0, ks_h, flags, ws);
}

size_t lognormal_tables_workspace_bytes(int64_t x_max) {
    const size_t tiles = (size_t)((x_max < 1 ? 1 : x_max) + kLnTile - 1) / kLnTile;
    return 2 * align_up(tiles * 8, 256);
}

int launch_lognormal_tables(hipStream_t s, double mu, double sigma, int64_t x_max, double* F0, double* F1, void* ws,
                            size_t ws_bytes) {
    BESST_REQUIRE(sigma > 0.0 && x_max >= 1 && x_max < ((int64_t)1 << 31), "lognormal_tables: parameters out of range");
    BESST_REQUIRE(F0 && F1 && ws, "lognormal_tables: null pointer");
    BESST_REQUIRE(ws_bytes >= lognormal_tables_workspace_bytes(x_max), "lognormal_tables: workspace too small");
    const long long tiles = (x_max + kLnTile - 1) / kLnTile;
    auto* tile0 = reinterpret_cast<double*>(ws);
    auto* tile1 = reinterpret_cast<double*>(static_cast<char*>(ws) + align_up((size_t)tiles * 8, 256));
    ProfScope ps(s, kProfScore);
    hipLaunchKernelGGL((ln_tile_kernel<false>), dim3((uint32_t)tiles), dim3(kLnThreads), 0, s, mu, sigma, (long long)x_max,
                       tile0, tile1, F0, F1);
    hipLaunchKernelGGL(ln_tile_scan_kernel, dim3(1), dim3(kLnThreads), 0, s, tile0, tile1, tiles);
    hipLaunchKernelGGL((ln_tile_kernel<true>), dim3((uint32_t)tiles), dim3(kLnThreads), 0, s, mu, sigma, (long long)x_max,
                       tile0, tile1, F0, F1);
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

int launch_conditional_stddevs(hipStream_t s, const double* f, int64_t max_isize, const int32_t* steps, int32_t n_steps,
                               double* out) {
    if (n_steps <= 0) return BESST_OK;
    BESST_REQUIRE(f && steps && out && max_isize >= 0, "conditional_stddevs: bad arguments");
    ProfScope ps(s, kProfScore);
    hipLaunchKernelGGL(cond_stddev_kernel, dim3((uint32_t)n_steps), dim3(kLnThreads), 0, s, f, (long long)max_isize, steps, out);
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

}  // namespace besst
