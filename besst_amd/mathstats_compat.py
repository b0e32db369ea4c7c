"""Restatement of the `mathstats==0.2.6.5` routines the hot path calls.

mathstats is a third-party dependency of the reference (requirements.txt:3,
setup.py:33) that is NOT vendored under /root/reference and is not installable
here, so nothing in this file can be checked against the real package:
**parity unpinned** (DESIGN.md).  Call sites replaced:

* ``mathstats.normaldist.normal.MaxObsDistr(n, 0.95)``
      libmetrics.py:14,23; CreateGraph.py:34,952,966
* ``mathstats.normaldist.truncatedskewed.param_est.GapEstimator(mean, sigma, read_len, mean_obs, c1_len, c2_len)``
      CreateGraph.py:36,537
* ``param_est.tr_sk_std_dev(mean, sigma, read_len, c1_len, c2_len, gap)``
      CreateGraph.py:555

The formulas are re-derived from the published model (Sahlin, Street, Lundeberg,
Arvestad, "Improved gap size estimation for scaffolding algorithms",
Bioinformatics 28(17):2215, 2012) as summarised in SURVEY.md Appendix C.1:

  fragment length x ~ N(mu, sigma^2); contigs c_min <= c_max; read length r; gap d.
  number of placements of a fragment that spans the gap with both reads inside
  their contigs

      w(x) = x - (d + 2r - 1)            on [d + 2r - 1 , d + c_min + r]
           = c_min - r + 1               on [d + c_min + r, d + c_max + r]
           = (d + c_min + c_max + 1) - x on [d + c_max + r, d + c_min + c_max + 1]

  g(d)  = Int w(x) phi(x) dx                         (normaliser)
  g'(d) = [Phi(hi3) - Phi(lo3)] - [Phi(hi1) - Phi(lo1)]
  ML condition:  d + sigma^2 g'(d) / g(d) = mu - mean(o),   o = obs1 + obs2 = x - d

``GapEstimator`` bisects that condition on [int(-4 sigma), int(mu + 4 sigma - 2r)]
to unit width and returns the rounded midpoint; ``tr_sk_std_dev`` is the standard
deviation of x under the density w(x) phi(x) / g(d).

The device kernels (besst_amd/csrc/score.hip) and the C oracle (oracle/besst_oracle.c)
implement the same expressions in the same evaluation order.
"""
import math

_SQRT2 = math.sqrt(2.0)
_INV_SQRT_2PI = 1.0 / math.sqrt(2.0 * math.pi)


def _rational_approximation(t):
    # Abramowitz & Stegun 26.2.23, |error| < 4.5e-4
    c0, c1, c2 = 2.515517, 0.802853, 0.010328
    d0, d1, d2 = 1.432788, 0.189269, 0.001308
    numerator = (c2 * t + c1) * t + c0
    denominator = ((d2 * t + d1) * t + d0) * t + 1.0
    return t - numerator / denominator


def normal_cdf_inverse(p):
    if not (0.0 < p < 1.0):
        raise ValueError('p must be in (0,1)')
    if p < 0.5:
        return -_rational_approximation(math.sqrt(-2.0 * math.log(p)))
    return _rational_approximation(math.sqrt(-2.0 * math.log(1.0 - p)))


def MaxObsDistr(nr_of_obs, prob):
    """k such that P(max of n standard normal draws < k) = prob.

    q = 1 - prob**(1/n) is the per-observation exceedance probability; the
    quantile is taken with the A&S 26.2.23 rational approximation.
    """
    q = 1 - prob ** (1 / float(nr_of_obs))
    return normal_cdf_inverse(1 - q)


def _phi0(y, sigma):
    """Density of N(0, sigma^2) at y."""
    return _INV_SQRT_2PI / sigma * math.exp(-(y * y) / (2.0 * sigma * sigma))


def _Phi0(y, sigma):
    return 0.5 * (1.0 + math.erf(y / (_SQRT2 * sigma)))


def _breakpoints(d, mean, c_min, c_max, r):
    """Piece boundaries of w(x), shifted by the library mean (y = x - mean)."""
    lo1 = d + 2.0 * r - 1.0 - mean
    hi1 = d + c_min + r - mean
    lo3 = d + c_max + r - mean
    hi3 = d + c_min + c_max + 1.0 - mean
    return lo1, hi1, lo3, hi3


def _weighted_moments(d, mean, sigma, c_min, c_max, r, kmax):
    """M_k = Int y^k w(y + mean) phi0(y) dy for k = 0..kmax (y = x - mean).

    Centred partial moments of N(0, sigma^2) on [a, b]:
      B0 = Phi(b) - Phi(a)
      B1 = -s2 [phi]            B2 = s2 B0 - s2 [y phi]
      B3 = 2 s2 B1 - s2 [y^2 phi]     B4 = 3 s2 B2 - s2 [y^3 phi]
    and for a piece with weight alpha*y + beta:  Int y^k w phi = alpha B_{k+1} + beta B_k.
    """
    s2 = sigma * sigma
    lo1, hi1, lo3, hi3 = _breakpoints(d, mean, c_min, c_max, r)
    pieces = ((lo1, hi1, 1.0, -lo1),
              (hi1, lo3, 0.0, c_min - r + 1.0),
              (lo3, hi3, -1.0, hi3))
    M = [0.0] * (kmax + 1)
    for a, b, alpha, beta in pieces:
        if not (b > a):
            continue
        pa = _phi0(a, sigma)
        pb = _phi0(b, sigma)
        B = [0.0] * (kmax + 2)
        B[0] = _Phi0(b, sigma) - _Phi0(a, sigma)
        B[1] = -s2 * (pb - pa)
        ya, yb = a, b
        for k in range(2, kmax + 2):
            B[k] = (k - 1) * s2 * B[k - 2] - s2 * (yb * pb - ya * pa)
            ya *= a
            yb *= b
        for k in range(kmax + 1):
            M[k] += alpha * B[k + 1] + beta * B[k]
    return M


def norm_const(d, mean, sigma, c_min, c_max, r):
    """g(d)."""
    return _weighted_moments(d, mean, sigma, c_min, c_max, r, 0)[0]


def norm_const_prime(d, mean, sigma, c_min, c_max, r):
    """g'(d)."""
    lo1, hi1, lo3, hi3 = _breakpoints(d, mean, c_min, c_max, r)
    t3 = (_Phi0(hi3, sigma) - _Phi0(lo3, sigma)) if hi3 > lo3 else 0.0
    t1 = (_Phi0(hi1, sigma) - _Phi0(lo1, sigma)) if hi1 > lo1 else 0.0
    return t3 - t1


def ml_condition(d, mean, sigma, c_min, c_max, r):
    """Left-hand side  d + sigma^2 g'(d)/g(d)  of the ML equation."""
    g = norm_const(d, mean, sigma, c_min, c_max, r)
    if not (g > 0.0):
        return d
    return d + sigma * sigma * norm_const_prime(d, mean, sigma, c_min, c_max, r) / g


def GapEstimator(mean, sigma, read_length, mean_obs, c1_len, c2_len=None):
    """ML gap between two contigs given the mean spanning observation."""
    if c2_len is None:
        c2_len = 10 * mean
    c_min = float(min(c1_len, c2_len))
    c_max = float(max(c1_len, c2_len))
    naive_gap = mean - mean_obs
    d_upper = float(int(mean + 4 * sigma - 2 * read_length))
    d_lower = float(int(-4 * sigma))
    while d_upper - d_lower > 1:
        d_mid = (d_upper + d_lower) / 2.0
        if ml_condition(d_mid, mean, sigma, c_min, c_max, read_length) > naive_gap:
            d_upper = d_mid
        else:
            d_lower = d_mid
    d_ml = (d_upper + d_lower) / 2.0
    return int(math.floor(d_ml + 0.5))


def tr_sk_std_dev(mean, sigma, read_length, c1_len, c2_len, d):
    """Std-dev of the spanning-fragment length under the truncated/skewed density."""
    c_min = float(min(c1_len, c2_len))
    c_max = float(max(c1_len, c2_len))
    M = _weighted_moments(float(d), mean, sigma, c_min, c_max, read_length, 2)
    if not (M[0] > 0.0):
        return float(2 ** 32)
    e1 = M[1] / M[0]
    var = M[2] / M[0] - e1 * e1
    if not (var > 0.0):
        return 0.0
    return math.sqrt(var)


def PreCalcMLvaluesOfdLongContigs(mean, sigma, read_length, ctx=None):
    """Table {round(naive_gap) -> ML gap} for two long contigs (MakeScaffolds.py:68,447).

    ctx: a besst_amd.device.GraphContext - the ML condition of every gap of the range is then evaluated by the
    device (besst_ctx_gap_condition_table, one thread per gap) and only the rounding and the inversion of the map
    stay on the host; without it everything runs here.  Both give the same table up to the last-bit differences of
    the device's erf / exp at a rounding boundary (tests/test_gpu_score_numeric.py).
    """
    big = 10.0 * (mean + 4 * sigma) + 10.0 * read_length
    d_upper = int(mean + 2 * sigma - 2 * read_length)
    d_lower = int(-2 * sigma)
    values = None
    if ctx is not None and d_upper >= d_lower:
        values = ctx.gap_condition_table(mean, sigma, read_length, big, d_lower, d_upper - d_lower + 1).tolist()
    table = {}
    prev = None
    for d in range(d_lower, d_upper + 1):
        v = values[d - d_lower] if values is not None else ml_condition(float(d), mean, sigma, big, big, read_length)
        f = int(math.floor(v + 0.5))
        if prev is None:
            prev = f
        for k in range(prev, f + 1):
            table.setdefault(k, d)
        prev = f
    return table


# ---------------------------------------------------------------------------------------------------------------
# mathstats.log_normal_param_est.GapEstimator (CreateGraph.py:37,526; MakeScaffolds.py:421) - also un-vendored.
# Restated from the same model as the normal case with the log-normal density: parity with 0.2.6.5 UNPINNED.
#   x ~ LogNormal(mu, sigma) on the integers, w(x; d) placements as in the normal case, g(d) = sum_x w(x; d) f(x),
#   log-likelihood of the observations o_i = x_i - d:   L(d) = sum_i log f(o_i + d) - n log g(d)
# (the placement weight of an observation does not depend on d).  Unlike the normal case L depends on every
# observation, not only on their mean.  The estimate is the integer d maximising L over all gaps that keep every
# x_i inside the support [1, exp(mu + 6 sigma)], found by a coarse scan (stride 64) and an exhaustive scan of the
# 129 gaps around the coarse optimum.
# ---------------------------------------------------------------------------------------------------------------
def lognormal_support(mu, sigma):
    """x_max: the pmf lives on the integers 1 .. x_max."""
    return int(min(math.exp(mu + 6.0 * sigma), 4.0e6))


def _lognormal_tables(mu, sigma):
    import numpy as np
    x_max = lognormal_support(mu, sigma)
    x = np.arange(1, x_max + 1, dtype=np.float64)
    lx = np.log(x)
    f = np.exp(-((lx - mu) ** 2) / (2.0 * sigma * sigma)) / (x * sigma * math.sqrt(2.0 * math.pi))
    F0 = np.concatenate(([0.0], np.cumsum(f)))              # F0[k] = sum_{x <= k} f(x)
    F1 = np.concatenate(([0.0], np.cumsum(f * x)))
    return x_max, F0, F1


def _lognormal_log_g(d, x_max, F0, F1, c_min, c_max, r):
    """log g(d) for an array of integer gaps d (three linear pieces of w, prefix sums of f and x f)."""
    import numpy as np
    d = np.asarray(d, dtype=np.int64)

    def seg(a, b):                                           # sums over integer x in [a, b] clipped to [1, x_max]
        a = np.clip(a, 1, x_max + 1)
        b = np.clip(b, 0, x_max)
        ok = b >= a
        a0 = np.where(ok, a, 1)
        b0 = np.where(ok, b, 0)
        return np.where(ok, F0[b0] - F0[a0 - 1], 0.0), np.where(ok, F1[b0] - F1[a0 - 1], 0.0)
    s0, s1 = seg(d + 2 * r, d + c_min + r - 1)               # w = x - d - 2r + 1
    g = s1 - (d + 2 * r - 1) * s0
    s0, s1 = seg(d + c_min + r, d + c_max + r)               # w = c_min - r + 1
    g = g + (c_min - r + 1) * s0
    s0, s1 = seg(d + c_max + r + 1, d + c_min + c_max)       # w = c_min + c_max + d - x + 1
    g = g + (c_min + c_max + d + 1) * s0 - s1
    with np.errstate(divide='ignore'):
        return np.where(g > 0.0, np.log(np.where(g > 0.0, g, 1.0)), -np.inf)


_LN_CACHE = {}


def lognormal_GapEstimator(mu, sigma, read_length, samples, c1_len, c2_len=None):
    import numpy as np
    obs = np.asarray(samples, dtype=np.int64)
    n = obs.shape[0]
    if n == 0:
        return 0
    key = (float(mu), float(sigma))
    if key not in _LN_CACHE:
        _LN_CACHE.clear()
        _LN_CACHE[key] = _lognormal_tables(mu, sigma)
    x_max, F0, F1 = _LN_CACHE[key]
    if c2_len is None:
        c2_len = 10 * x_max
    r = int(round(read_length))
    c_min, c_max = int(min(c1_len, c2_len)), int(max(c1_len, c2_len))
    d_lo, d_hi = 1 - int(obs.min()), x_max - int(obs.max())
    if d_hi < d_lo:
        return int(round(math.exp(mu) - float(obs.mean())))

    def loglik(ds):
        ds = np.asarray(ds, dtype=np.int64)
        out = np.empty(ds.shape[0], dtype=np.float64)
        step = max(16, (1 << 22) // max(1, n))               # (gaps x observations) blocks of at most 32 MB per temporary
        for a in range(0, ds.shape[0], step):
            blk = ds[a:a + step]
            lx = np.log((obs[None, :] + blk[:, None]).astype(np.float64))
            out[a:a + step] = (-lx - ((lx - mu) ** 2) / (2.0 * sigma * sigma)).sum(axis=1)
        lg = _lognormal_log_g(ds, x_max, F0, F1, c_min, c_max, r)
        with np.errstate(invalid='ignore'):
            return np.where(np.isfinite(lg), out - n * lg, -np.inf)      # no spanning fragment at this gap: never chosen
    coarse = np.arange(d_lo, d_hi + 1, 64, dtype=np.int64)
    best = int(coarse[int(np.argmax(loglik(coarse)))])
    fine = np.arange(max(d_lo, best - 64), min(d_hi, best + 64) + 1, dtype=np.int64)
    return int(fine[int(np.argmax(loglik(fine)))])
