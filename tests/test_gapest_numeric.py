"""The restated mathstats routines against a brute-force evaluation of the published GapEst model (oracle/gapest_numeric.py:
integer counting of fragment placements, numerical normalisation, argmax over the integer gaps - no closed form shared with
besst_amd/mathstats_compat.py).  Tolerances: gap +-1 bp (north star), expected sigma 0.5 %, MaxObsDistr within the stated
error of the Abramowitz-Stegun approximation.  mathstats 0.2.6.5 itself stays unpinned (not obtainable here)."""
import itertools

import pytest

from besst_amd import mathstats_compat as MC
from oracle import gapest_numeric as GN

GRID = [
    # (mu, sigma, r, c1, c2)   - incl. contigs shorter than 2 sigma .. mu, and long ones
    (500.0, 50.0, 100, 3000, 5000), (500.0, 50.0, 100, 400, 700), (500.0, 50.0, 100, 250, 260),
    (2500.0, 250.0, 100, 3000, 1800), (2500.0, 250.0, 100, 800, 20000), (5000.0, 500.0, 100, 8000, 8000),
    (5000.0, 500.0, 100.38, 1200, 30000), (480.5, 55.25, 75, 900, 1000), (350.0, 60.0, 100, 100000, 100000),
]


def cases():
    for mu, sigma, r, c1, c2 in GRID:
        for frac in (-0.6, -0.2, 0.0, 0.3, 0.7, 1.0):          # naive gap = frac * (mu - 2 r): from overlaps to the far end
            naive = frac * (mu - 2 * r)
            yield mu, sigma, r, c1, c2, mu - naive


@pytest.mark.parametrize('mu,sigma,r,c1,c2,mean_obs', list(cases()))
def test_gap_estimator_matches_brute_force_likelihood(mu, sigma, r, c1, c2, mean_obs):
    # the numeric model needs an integer read length; a fractional inferred one is rounded for the counting only
    ri = int(round(r))
    want, fs = GN.ml_gap(mu, sigma, ri, mean_obs, c1, c2)
    got = MC.GapEstimator(mu, sigma, r, mean_obs, c1, c2)
    lo, hi = int(-4 * sigma), int(mu + 4 * sigma - 2 * r)
    if want in (lo, hi, min(fs), max(fs)):
        # the likelihood is still rising at the end of the search interval: the bisection ends within a step of it
        assert abs(got - want) <= 2
    else:
        assert abs(got - want) <= 1, (got, want)
        # and the estimate is a maximum of the brute-force likelihood to within its flatness
        assert fs[got] >= fs[want] - 1e-3 * abs(fs[want]) - 1e-6 if got in fs else True


@pytest.mark.parametrize('mu,sigma,r,c1,c2', GRID)
def test_expected_sigma_matches_brute_force_density(mu, sigma, r, c1, c2):
    ri = int(round(r))
    for d in (int(-2 * sigma), -50, 0, 100, int(mu / 2), int(mu)):
        want = GN.span_sd(d, mu, sigma, c1, c2, ri)
        if want is None:
            continue
        got = MC.tr_sk_std_dev(mu, sigma, ri, c1, c2, d)
        assert abs(got - want) <= 0.005 * want + 0.05, (d, got, want)


@pytest.mark.parametrize('n', [10, 1000, 17345, 1000000])
def test_max_obs_distr_is_the_quantile_of_the_maximum(n):
    assert abs(MC.MaxObsDistr(n, 0.95) - GN.max_obs_quantile(n, 0.95)) < 4.5e-4 + 1e-9
