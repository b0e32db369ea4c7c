"""The host finishing of get_metrics (csrc/hostmath.hip: besst_host_isize_stats) against the reference's own expressions
(BESST/libmetrics.py:316-341, :22-28) evaluated by CPython on the same sample - bit for bit, also where the native code
computes a pass's libm terms on several threads (samples of 65 536 and more) and where trimming takes several rounds."""
import numpy as np
import pytest

from besst_amd import libmetrics
from besst_amd.mathstats_compat import MaxObsDistr


def _reference(ins_size_reads):
    """libmetrics.py:316-341 on a Python list (ints for 'fr', floats for 'rf')."""
    kept = list(range(len(ins_size_reads)))
    n = float(len(ins_size_reads))
    mean_isize = sum(ins_size_reads) / n
    std_dev_isize = (sum(list(map((lambda x: x ** 2 - 2 * x * mean_isize + mean_isize ** 2), ins_size_reads))) / (n - 1)) ** 0.5
    before = (mean_isize, std_dev_isize)
    extreme_obs_occur = True
    while extreme_obs_occur:
        k = 1.5 * MaxObsDistr(len(ins_size_reads), 0.95)
        lo, hi = mean_isize - k * std_dev_isize, mean_isize + k * std_dev_isize
        keep = [j for j, x in enumerate(ins_size_reads) if (x < hi and x > lo)]
        extreme_obs_occur = len(keep) < len(ins_size_reads)
        filtered_list = [ins_size_reads[j] for j in keep]
        kept = [kept[j] for j in keep]
        n = float(len(filtered_list))
        mean_isize = sum(filtered_list) / n
        std_dev_isize = (sum(list(map((lambda x: x ** 2 - 2 * x * mean_isize + mean_isize ** 2), filtered_list))) / (n - 1)) ** 0.5
        ins_size_reads = filtered_list
    m_3 = sum([(x - mean_isize) ** 3 for x in ins_size_reads]) / n
    return kept, [before[0], before[1], mean_isize, std_dev_isize, m_3 / std_dev_isize ** 3]


@pytest.mark.parametrize('n', [1500, 65535, 65536, 200000])
@pytest.mark.parametrize('is_float', [False, True])
def test_isize_stats_equal_cpython(n, is_float):
    rng = np.random.default_rng(n + int(is_float))
    vals = np.abs(rng.normal(3000, 300, n)).astype(np.int32)
    wild = rng.random(n) < 0.002                                # chimeric pairs far outside: several trimming rounds
    vals[wild] = rng.integers(20000, 400000, int(wild.sum()))
    offset = 2 * 100.38
    sample = [int(v) + offset for v in vals] if is_float else [int(v) for v in vals]
    want_kept, want = _reference(sample)
    got_vals, got_kept, got = libmetrics._native_isize_stats(vals, is_float, offset)
    assert len(want_kept) < n                                   # something was trimmed
    assert got_kept.tolist() == want_kept
    assert got == want                                          # the same doubles, not close ones
