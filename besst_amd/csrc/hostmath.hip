// Host-side float finishing of libmetrics (no HIP code): the reference's statistics on the <= 1,000,000 sampled
// insert sizes, replayed in EXACTLY its operation order so that the doubles come out bit-identical to CPython:
//   mean / (n-1)-stddev by the expanded square      libmetrics.py:317-319 (same form :36,42,46,93,105)
//   iterative trimming  AdjustInsertsizeDist          :22-28, loops :322-332 and :99-108
//   skewness                                          :340-341
//   GetDistr bias-corrected distribution, median, 21-window mode, mu/sigma/skew   :141-223
// CPython evaluates `x ** 2`, `x ** 0.5`, `x ** 3` on floats with libm pow() and sums with naive left-to-right
// addition (3.10); the same libm calls are made here through volatile function pointers so the compiler cannot
// strength-reduce them.  Integer samples ('fr': abs(tlen)) use exact integer sums / squares like Python ints do;
// 'rf' samples are the floats abs(tlen) + 2*read_len.  Doing this natively takes ~40 ms instead of ~0.8 s of
// interpreted loops per library.
#include <math.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "common.h"

namespace {

double (*volatile libm_pow)(double, double) = pow;
double (*volatile libm_log)(double) = log;
double (*volatile libm_sqrt)(double) = sqrt;

// mathstats.normaldist.normal.MaxObsDistr (restated; see besst_amd/mathstats_compat.py)
double rational_approximation(double t) {
    const double numerator = (0.010328 * t + 0.802853) * t + 2.515517;
    const double denominator = ((0.001308 * t + 0.189269) * t + 1.432788) * t + 1.0;
    return t - numerator / denominator;
}
double max_obs_distr(int64_t n, double prob) {
    const double q = 1 - libm_pow(prob, 1 / (double)n);
    const double p = 1 - q;
    if (p < 0.5) return -rational_approximation(libm_sqrt(-2.0 * libm_log(p)));
    return rational_approximation(libm_sqrt(-2.0 * libm_log(1.0 - p)));
}

// The libm calls of a pass over the sample do not depend on each other - only the additions that follow them do: the terms
// are computed by the CPUs this process may use (its affinity mask and cgroup quota, 16 at most), the sum stays
// left-to-right.  (1,000,000 pow() calls per pass and five passes per library were 30 of get_metrics' 50 ms.)
int host_threads() {
    static const int n = [] {
        int k = 1;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) k = CPU_COUNT(&set);
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char quota[32];
            long long period = 0;
            if (fscanf(f, "%31s %lld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
                const long long c = (atoll(quota) + period - 1) / period;
                if (c >= 1 && c < k) k = (int)c;
            }
            fclose(f);
        }
        return k < 1 ? 1 : k > 16 ? 16 : k;
    }();
    return n;
}
template <class F>
void in_pieces(size_t n, F f) {                                // f(begin, end) over [0, n), on several threads when n is large
    const int t = n < 65536 ? 1 : host_threads();
    if (t <= 1) { f((size_t)0, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n + (size_t)t - 1) / (size_t)t;
    for (int k = 1; k < t; ++k) {
        const size_t b = per * (size_t)k, e = b + per < n ? b + per : n;
        if (b < e) th.emplace_back([=, &f] { f(b, e); });
    }
    f((size_t)0, per < n ? per : n);
    for (auto& x : th) x.join();
}

struct Sample {
    const int32_t* a;   // abs(tlen)
    bool is_float;      // value = a + offset (float) instead of the integer a
    double offset;
    double value(int64_t i) const { return (double)a[i] + offset; }
};

// mean and (n-1)-stddev exactly as  sum(x)/n  and  (sum(x**2 - 2*x*mean + mean**2)/(n-1)) ** 0.5
void mean_std(const Sample& s, const std::vector<int64_t>& keep, double* mean_out, double* std_out) {
    const double n = (double)keep.size();
    double mean;
    if (!s.is_float) {
        long long tot = 0;
        for (int64_t i : keep) tot += s.a[i];
        mean = (double)tot / n;
    } else {
        double tot = 0.0;                       // Python: 0 + x0 + x1 + ...
        bool first = true;
        for (int64_t i : keep) {
            if (first) { tot = 0 + s.value(i); first = false; }
            else tot = tot + s.value(i);
        }
        mean = tot / n;
    }
    const double m2 = libm_pow(mean, 2.0);
    double acc = 0.0;
    if (!s.is_float) {
        for (int64_t i : keep) {
            const long long x = s.a[i];
            const double term = ((double)(x * x) - (double)(2 * x) * mean) + m2;
            acc = acc + term;
        }
    } else {
        std::vector<double> term(keep.size());
        in_pieces(keep.size(), [&](size_t b, size_t e) {
            for (size_t j = b; j < e; ++j) {
                const double x = s.value(keep[j]);
                term[j] = (libm_pow(x, 2.0) - (2 * x) * mean) + m2;
            }
        });
        for (const double t : term) acc = acc + t;
    }
    *mean_out = mean;
    *std_out = libm_pow(acc / (n - 1), 0.5);
}

// one AdjustInsertsizeDist round; returns true if something was removed
bool trim_once(const Sample& s, std::vector<int64_t>& keep, double mean, double sd) {
    const double k = 1.5 * max_obs_distr((int64_t)keep.size(), 0.95);
    const double hi = mean + k * sd, lo = mean - k * sd;
    std::vector<int64_t> next;
    next.reserve(keep.size());
    for (int64_t i : keep) {
        const double x = s.is_float ? s.value(i) : (double)s.a[i];
        if (x < hi && x > lo) next.push_back(i);
    }
    const bool removed = next.size() < keep.size();
    keep.swap(next);
    return removed;
}

}  // namespace

extern "C" {

// Insert-size statistics (libmetrics.py:316-343).  values: abs(tlen) in BAM order; is_float/offset select the
// 'rf' form abs(tlen) + 2*read_len.  kept_out receives the indexes of the observations that survive trimming
// (capacity n).  stats_out: [0] mean before, [1] std before, [2] mean converged, [3] std converged, [4] skewness.
int besst_host_isize_stats(const int32_t* values, int64_t n, int32_t is_float, double offset, int64_t* kept_out,
                           int64_t* n_kept, double* stats_out) {
    BESST_REQUIRE(values && kept_out && n_kept && stats_out && n >= 2, "host_isize_stats: bad argument");
    const Sample s{values, is_float != 0, offset};
    std::vector<int64_t> keep((size_t)n);
    for (int64_t i = 0; i < n; ++i) keep[(size_t)i] = i;
    double mean, sd;
    mean_std(s, keep, &mean, &sd);
    stats_out[0] = mean;
    stats_out[1] = sd;
    bool again = true;
    while (again) {
        again = trim_once(s, keep, mean, sd);
        BESST_REQUIRE(keep.size() >= 2, "host_isize_stats: fewer than two observations left after trimming");
        mean_std(s, keep, &mean, &sd);
    }
    // (libmetrics.py:330-332 computes mean and deviation of the trimmed list once more: the same list, the same doubles)
    stats_out[2] = mean;
    stats_out[3] = sd;
    double m3 = 0.0;
    {
        std::vector<double> term(keep.size());
        in_pieces(keep.size(), [&](size_t b, size_t e) {
            for (size_t j = b; j < e; ++j) {
                const double x = s.is_float ? s.value(keep[j]) : (double)s.a[keep[j]];
                term[j] = libm_pow(x - mean, 3.0);
            }
        });
        bool first = true;
        for (const double t : term) {
            if (first) { m3 = 0 + t; first = false; } else m3 = m3 + t;
        }
    }
    m3 = m3 / (double)keep.size();
    stats_out[4] = m3 / libm_pow(sd, 3.0);
    for (size_t j = 0; j < keep.size(); ++j) kept_out[j] = keep[j];
    *n_kept = (int64_t)keep.size();
    return BESST_OK;
}

// Contamination trimming loop (libmetrics.py:88-110): stops as soon as <= 2 observations would remain.
// stats_out: [0] mean before, [1] std before, [2] mean converged, [3] std converged; *n_final = len(list).
int besst_host_contam_stats(const int32_t* values, int64_t n, int32_t is_float, double offset, int64_t* n_final,
                            double* stats_out) {
    BESST_REQUIRE(values && n_final && stats_out && n > 2, "host_contam_stats: bad argument");
    const Sample s{values, is_float != 0, offset};
    std::vector<int64_t> keep((size_t)n);
    for (int64_t i = 0; i < n; ++i) keep[(size_t)i] = i;
    double mean, sd;
    mean_std(s, keep, &mean, &sd);
    stats_out[0] = mean;
    stats_out[1] = sd;
    double n_contamine = (double)n;
    bool again = true;
    while (again) {
        std::vector<int64_t> cand = keep;
        again = trim_once(s, cand, mean, sd);
        n_contamine = (double)cand.size();
        if (cand.size() > 2) {
            mean_std(s, cand, &mean, &sd);
            keep.swap(cand);
        } else {
            break;
        }
    }
    stats_out[2] = mean;
    stats_out[3] = sd;
    *n_final = (int64_t)n_contamine;
    return BESST_OK;
}

// GetDistr (libmetrics.py:141-223).  kept: indexes into values (the trimmed sample, BAM order).
// adjusted_out must hold max_isize + 1 doubles where max_isize = int(max(sample)) (returned in *n_adjusted).
// out: [0] mu_adj, [1] sigma_adj, [2] skew_adj, [3] median_adj, [4] mode_adj, [5..25] the 21 per-window modes.
int besst_host_getdistr(const int32_t* values, const int64_t* kept, int64_t n_kept, int32_t is_float, double offset,
                        const int32_t* contig_lengths, int64_t n_contigs, double* adjusted_out, int64_t adjusted_cap,
                        int64_t* n_adjusted, double* out) {
    BESST_REQUIRE(values && kept && contig_lengths && adjusted_out && n_adjusted && out && n_kept > 0 && n_contigs > 0,
                  "host_getdistr: bad argument");
    const Sample s{values, is_float != 0, offset};
    // largest_contigs = ascending list of the (up to) 1000 largest lengths
    std::vector<long long> largest(contig_lengths, contig_lengths + n_contigs);
    std::sort(largest.begin(), largest.end());
    if (largest.size() > 1000) largest.erase(largest.begin(), largest.end() - 1000);
    double mx = -1e300;
    for (int64_t j = 0; j < n_kept; ++j) {
        const double x = s.is_float ? s.value(kept[j]) : (double)s.a[kept[j]];
        if (x > mx) mx = x;
    }
    const long long max_isize = (long long)mx;
    BESST_REQUIRE(max_isize + 1 <= adjusted_cap, "host_getdistr: adjusted_out too small");
    std::vector<double> adj((size_t)(max_isize + 1), 0.0);
    std::vector<char> touched((size_t)(max_isize + 1), 0);   // Python list starts as ints 0: 0 + 1/w
    long long cur_sum = 0;
    for (long long v : largest) cur_sum += v;
    long long cur_nr = (long long)largest.size();
    std::vector<long long> at_nr, at_sum;
    at_nr.push_back(cur_nr);
    at_sum.push_back(cur_sum);
    long long smallest = largest[0];
    size_t smallest_index = 0;
    const long long upper_isize = std::min(max_isize + 1, largest.back());
    for (long long isize = 0; isize < upper_isize; ++isize) {
        if (isize > smallest) {
            while (isize > largest[smallest_index]) {
                smallest_index++;
                cur_nr -= 1;
                cur_sum -= smallest;      // the not-yet-updated value, as in the reference (:164-170)
            }
            at_nr.push_back(cur_nr);
            at_sum.push_back(cur_sum);
            smallest = largest[smallest_index];
        } else {
            at_nr.push_back(cur_nr);
            at_sum.push_back(cur_sum);
        }
    }
    for (int64_t j = 0; j < n_kept; ++j) {
        const double o = s.is_float ? s.value(kept[j]) : (double)s.a[kept[j]];
        const long long obs = (long long)o;
        if (obs > upper_isize) continue;
        const long long w_int = std::max(at_sum[(size_t)obs] - (obs - 1) * at_nr[(size_t)obs], 10000LL);
        const double inc = 1 / (double)w_int;
        adj[(size_t)obs] = adj[(size_t)obs] + inc;
        touched[(size_t)obs] = 1;
    }
    double tot = 0.0;
    for (double v : adj) tot = tot + v;
    double cum = 0.0;
    long long curr = 0;
    const double half = tot / 2.0;
    while (cum <= half) {
        cum += adj[(size_t)curr];
        curr += 1;
    }
    const long long median_adj = curr;
    std::vector<long long> modes;
    int slot = 5;
    for (int chunk = 1; chunk < 102; chunk += 5) {
        long long best_i = 0;
        double best_v = 0.0;
        bool have = false;
        long long ci = 0;
        for (size_t start = 0; start < adj.size(); start += (size_t)chunk, ++ci) {
            double v = 0.0;
            const size_t end = std::min(adj.size(), start + (size_t)chunk);
            for (size_t q = start; q < end; ++q) v = v + adj[q];
            if (!have || v > best_v) { best_i = ci; best_v = v; have = true; }
        }
        const double mode = ((double)best_i + 0.5) * chunk;
        out[slot++] = mode;
        modes.push_back((long long)mode);
    }
    std::sort(modes.begin(), modes.end());
    const long long mode_adj = modes[modes.size() / 2];
    double s1 = 0.0;
    for (size_t i = 0; i < adj.size(); ++i) s1 = s1 + (double)i * adj[i];
    const double mu = s1 / tot;
    double s2 = 0.0, s3 = 0.0;
    for (size_t i = 0; i < adj.size(); ++i) {
        s2 = s2 + libm_pow((double)i - mu, 2.0) * adj[i];
        s3 = s3 + libm_pow((double)i - mu, 3.0) * adj[i];
    }
    const double sigma = libm_sqrt(s2 / tot);
    const double m3 = s3 / tot;
    out[0] = mu;
    out[1] = sigma;
    out[2] = m3 / libm_pow(sigma, 3.0);
    out[3] = (double)median_adj;
    out[4] = (double)mode_adj;
    memcpy(adjusted_out, adj.data(), adj.size() * sizeof(double));
    *n_adjusted = (int64_t)adj.size();
    (void)touched;
    return BESST_OK;
}

}  // extern "C"
