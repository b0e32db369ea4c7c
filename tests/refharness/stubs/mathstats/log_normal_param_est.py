"""mathstats.log_normal_param_est as the reference imports it (CreateGraph.py:37) -> the restatement under test."""
from besst_amd.mathstats_compat import lognormal_GapEstimator


def GapEstimator(mu, sigma, read_len, samples, c1_len, c2_len=None):
    return lognormal_GapEstimator(mu, sigma, read_len, samples, c1_len, c2_len)
