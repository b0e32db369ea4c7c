"""Brute-force GapEst: an INDEPENDENT check of the restated mathstats routines (test infrastructure only).

besst_amd/mathstats_compat.py, csrc/score.hip and oracle/besst_oracle.c all evaluate ONE closed-form derivation of
the gap estimator of Sahlin et al. 2012 (Bioinformatics 28:2215) - the real mathstats 0.2.6.5 is not obtainable in
the build container - so their mutual agreement says nothing about a sign or +-1 slip in that derivation.  This file
shares no formula with them: it goes back to the model and counts.

Model (the paper's, as used at the reference's call sites CreateGraph.py:537,555): fragment
length x ~ N(mu, sigma^2); two contigs of lengths c1, c2 separated by a gap d (negative: overlap); read length r.  A
fragment is observed on this contig pair when its left read lies inside contig 1 and its right read inside contig 2.

  w(x; d)  = number of INTEGER start positions a of the fragment for which that holds, by direct counting:
             left read  [a, a + r)      inside [0, c1)
             right read [a + x - r, a + x) inside [c1 + d, c1 + d + c2)
  P(x | d) = w(x; d) phi(x) / g(d),   g(d) = sum over integer x of w(x; d) phi(x)
  the observation is o = obs1 + obs2 = x - d; in terms of o the weight w does not depend on d, so for n observations
  with mean obar the log-likelihood is, up to terms without d,
             F(d) = - (obar + d - mu)^2 / (2 sigma^2) - log g(d)
  ml_gap     = argmax of F over the integers the reference's search interval [int(-4 sigma), int(mu + 4 sigma - 2 r)]
  span_sd(d) = standard deviation of x under P(x | d)   (what tr_sk_std_dev must return)

Nothing here is used by the product; tests/test_gapest_numeric.py (CPU) and tests/test_gpu_score_numeric.py (GPU)
compare mathstats_compat and score_kernel with it.  Parity with mathstats 0.2.6.5 itself remains UNPINNED: what this
pins is agreement with the published model to the +-1 bp the north star asks for.
"""
import math

import numpy as np
from scipy.stats import norm


def placements(x, d, c1, c2, r):
    """w(x; d) by counting, vectorised over integer x."""
    x = np.asarray(x, dtype=np.int64)
    lo = np.maximum(0, c1 + d + r - x)                  # right read starts inside contig 2
    hi = np.minimum(c1 - r, c1 + d + c2 - x)            # left read ends inside contig 1, right read ends inside contig 2
    return np.maximum(0, hi - lo + 1)


def _support(mu, sigma):
    lo = int(math.floor(mu - 12 * sigma))
    hi = int(math.ceil(mu + 12 * sigma))
    return np.arange(lo, hi + 1, dtype=np.int64)


def log_g(d, mu, sigma, c1, c2, r):
    x = _support(mu, sigma)
    w = placements(x, d, c1, c2, r).astype(np.float64)
    phi = np.exp(-0.5 * ((x - mu) / sigma) ** 2)
    tot = float(np.dot(w, phi))
    return math.log(tot) if tot > 0.0 else -math.inf


def ml_gap(mu, sigma, r, mean_obs, c1, c2):
    """(argmax d, F values around it) over the reference's integer search interval."""
    d_lo, d_hi = int(-4 * sigma), int(mu + 4 * sigma - 2 * r)
    best, best_f = None, -math.inf
    fs = {}
    for d in range(d_lo, d_hi + 1):
        lg = log_g(d, mu, sigma, c1, c2, r)
        if lg == -math.inf:
            continue
        f = -((mean_obs + d - mu) ** 2) / (2.0 * sigma * sigma) - lg
        fs[d] = f
        if f > best_f:
            best, best_f = d, f
    return best, fs


def span_sd(d, mu, sigma, c1, c2, r):
    x = _support(mu, sigma)
    w = placements(x, d, c1, c2, r).astype(np.float64)
    p = w * np.exp(-0.5 * ((x - mu) / sigma) ** 2)
    tot = p.sum()
    if not tot > 0.0:
        return None
    m1 = float(np.dot(p, x) / tot)
    var = float(np.dot(p, (x - m1) ** 2) / tot)
    return math.sqrt(var)


def max_obs_quantile(n, prob):
    """k with P(max of n standard normal draws < k) = prob, from scipy's exact quantile (MaxObsDistr's definition)."""
    return float(norm.ppf(prob ** (1.0 / n)))
