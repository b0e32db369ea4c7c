from besst_amd.mathstats_compat import MaxObsDistr, normal_cdf_inverse  # noqa: F401
from besst_amd.e_nr_links import normcdf  # noqa: F401


def normpdf(x, mu, sigma):
    import math
    u = (x - mu) / abs(sigma)
    return (1 / (math.sqrt(2 * math.pi) * abs(sigma))) * math.exp(-u * u / 2)
