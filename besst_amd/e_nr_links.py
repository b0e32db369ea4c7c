"""Expected number of read-pair links spanning a gap (Zerbino et al. 2009 closed form).

Host-side scalar restatement of the reference's ``e_nr_links.Param`` /
``ExpectedLinks`` / ``normcdf`` / ``normpdf`` (BESST/e_nr_links.py:18-93).  It is
evaluated once per library by ``infer_spurious_link_count_threshold``
(CreateGraph.py:323-353) to set the dense-region link threshold, so it stays on
the host; the vectorised form below is also used by the parity tests.
"""
import math
from decimal import Decimal, getcontext


def normcdf(x, mu, sigma):
    y = 0.5 * math.erfc(-(x - mu) / (sigma * math.sqrt(2.0)))
    return 1.0 if y > 1.0 else y


def normpdf(x, mu, sigma):
    # The reference evaluates the exponential in 100-digit decimal arithmetic and rounds the
    # product back to a double (e_nr_links.py:25-30); keep that exact chain of conversions.
    getcontext().prec = 100
    u = Decimal(str(x - mu)) / Decimal(str(abs(sigma)))
    scale = 1 / Decimal(str(math.sqrt(2 * math.pi) * abs(sigma)))
    return float(str(scale * Decimal(str(-u * u / 2)).exp()))


class Param(object):
    """Library-wide constants: mean, stddev, coverage, read length, allowed soft clipping."""

    def __init__(self, mean, stddev, cov, read_len, softclipped):
        self.mean = mean
        self.stddev = stddev
        self.read_len = read_len
        self.cov = cov
        self.softclipped = softclipped
        # expected distance between consecutive fragment starts
        self.readfrequency = 2 * self.read_len / self.cov


def ExpectedLinks(len1, len2, d, param):
    sd = float(param.stddev)
    gap = max(d, 0)          # negative gaps are not credited with extra links
    inside = param.read_len - param.softclipped     # bases that must lie inside a contig
    short, long_ = min(len1, len2), max(len1, len2)
    freq = param.readfrequency

    def part(a, b):
        cdf_a = normcdf(a, 0, 1)
        cdf_b = normcdf(b, 0, 1)
        e1 = (short - inside) / freq * cdf_a
        e2 = -(-param.softclipped) / freq * cdf_b
        e3 = (b * sd) / freq * (cdf_b - cdf_a)
        e4 = (sd / freq) * (normpdf(b, 0, 1) - normpdf(a, 0, 1))
        return e1 + e2 + e3 + e4

    b1 = (len1 + len2 + gap + 2 * param.softclipped - param.mean) / sd
    a1 = (long_ + gap + inside - param.mean) / sd
    b2 = (short + gap + inside - param.mean) / sd
    a2 = (gap + 2 * inside - param.mean) / sd
    return part(a1, b1) - part(a2, b2)
