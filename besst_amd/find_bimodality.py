"""PE/MP bimodality splitter: cut an insert-size population into (at most) two modes.

Restates ``find_bimodality.split_distribution`` with helpers ``params`` and
``checkEqualIvo`` (BESST/find_bimodality.py:28-37,39-205).  The reference bins the
sorted sample into ``2*max`` equal bins, joins runs of non-empty bins into
clusters (:110-132) and then scans the split points left to right, keeping a split
only when BOTH the summed variance and the summed std-dev improve on the running
best (:173-195).  Because the bin width is below 0.5, integer samples make every
distinct value its own cluster, so the scan is a function of the count-per-value
histogram alone: prefix sums of (n, sum x, sum x^2) give each candidate split in
O(1).  That histogram is what the device produces (csrc/metrics.hip,
``besst_dev_value_histogram``); ``split_from_histogram`` consumes it directly and
``split_distribution`` is the list-in/list-out form with the reference's signature.

Reference quirk kept out: its single-cluster branch (:137-142) raises NameError;
here it returns the one cluster with its mean/stddev, which is what the code
evidently meant.
"""
import math

import numpy as np


def params(x):
    n = float(len(x))
    mu = sum(x) / float(len(x))
    x_sq = sum(v ** 2 for v in x)
    var = (x_sq - n * mu ** 2) / (n - 1)
    return mu, var


def checkEqualIvo(lst):
    return not lst or lst.count(lst[0]) == len(lst)


def _moments(n, s1, s2):
    """(mu, var) from exact integer count / sum / sum of squares, in the reference's float order."""
    nf = float(n)
    mu = s1 / nf
    var = (s2 - nf * mu ** 2) / (nf - 1)
    return mu, var


def _clusters_from_values(values):
    """Distinct-cluster table [(first_value_index, n, sum, sumsq)] following :110-132."""
    arr = np.sort(np.asarray(values))
    integral = arr.dtype.kind in 'iu' or bool(np.all(arr == np.floor(arr)))
    if integral:
        vals, counts = np.unique(arr.astype(np.int64), return_counts=True)
        return vals.tolist(), counts.tolist(), None
    # general (non-integer) samples: do the actual binning
    n_bins = int(arr[-1] * 2)
    _, edges = np.histogram(arr, bins=n_bins)
    idx = np.digitize(arr, edges)
    occupied = np.unique(idx)
    run_id = np.cumsum(np.concatenate(([0], (np.diff(occupied) > 1).astype(np.int64))))
    lut = dict(zip(occupied.tolist(), run_id.tolist()))
    member = [lut[i] for i in idx.tolist()]
    groups = {}
    for v, g in zip(arr.tolist(), member):
        groups.setdefault(g, []).append(v)
    return None, None, [groups[g] for g in sorted(groups)]


def split_from_histogram(values, counts):
    """Split from a count-per-distinct-value table (ascending integer ``values``).

    Returns ``(split_value_index, mean1, stddev1, mean2, stddev2)`` where clusters are
    ``values[:split]`` / ``values[split:]``; ``split == len(values)`` means "one cluster"
    and ``split == -1`` means the reference's early ``([],[],0,0,0,0)`` exit.
    """
    k = len(values)
    n_all = sum(counts)
    # base case: only observations > 100 take part (:153-158)
    fn = fs1 = fs2 = 0
    for v, c in zip(values, counts):
        if v > 100:
            fn += c
            fs1 += c * v
            fs2 += c * v * v
    if k < 2:
        s1 = sum(c * v for v, c in zip(values, counts))
        s2 = sum(c * v * v for v, c in zip(values, counts))
        if n_all < 2:
            return k, float(values[0]) if k else 0, 0, 0, 0
        mu, var = _moments(n_all, s1, s2)
        return k, mu, math.sqrt(var), 0, 0
    if fn < 2:
        return -1, 0, 0, 0, 0
    base_mu, base_var = _moments(fn, fs1, fs2)
    base_stddev = math.sqrt(base_var)
    lowest_var, lowest_stddev = base_var, base_stddev
    split_index = 0
    mean1 = stddev1 = mean2 = stddev2 = 0
    tot_s1 = sum(c * v for v, c in zip(values, counts))
    tot_s2 = sum(c * v * v for v, c in zip(values, counts))
    n1 = s1 = s2 = 0
    for i in range(1, k):
        v, c = values[i - 1], counts[i - 1]
        n1 += c
        s1 += c * v
        s2 += c * v * v
        n2 = n_all - n1
        # fewer than two observations, or all observations equal, on either side: skip (:178)
        if n1 < 2 or n2 < 2 or i == 1 or i == k - 1:
            continue
        mu1, var1 = _moments(n1, s1, s2)
        mu2, var2 = _moments(n2, tot_s1 - s1, tot_s2 - s2)
        sum_var = var1 + var2
        sum_stddev = math.sqrt(var1) + math.sqrt(var2)
        if sum_var < lowest_var and sum_stddev < lowest_stddev:
            lowest_var, lowest_stddev = sum_var, sum_stddev
            stddev1, stddev2 = math.sqrt(var1), math.sqrt(var2)
            mean1, mean2 = mu1, mu2
            split_index = i
    if lowest_var < base_var and lowest_stddev < base_stddev:
        return split_index, mean1, stddev1, mean2, stddev2
    return k, base_mu, base_stddev, 0, 0


def split_distribution(all_isizes):
    """List form with the reference signature: (cluster1, cluster2, mean1, sd1, mean2, sd2)."""
    values, counts, groups = _clusters_from_values(all_isizes)
    if values is None:
        return _split_general(groups)
    split, m1, s1, m2, s2 = split_from_histogram(values, counts)
    if split < 0:
        return [], [], 0, 0, 0, 0
    c1 = [v for v, c in zip(values[:split], counts[:split]) for _ in range(c)]
    c2 = [v for v, c in zip(values[split:], counts[split:]) for _ in range(c)]
    return c1, c2, m1, s1, m2, s2


def _split_general(joined_bins):
    """Literal list-based scan for non-integer samples (rare: float read_len on an rf library)."""
    observations = [v for grp in joined_bins for v in grp]
    if len(joined_bins) < 2:
        mu, var = params(observations)
        return observations, [], mu, math.sqrt(var), 0, 0
    kept = [v for v in observations if v > 100]
    if len(kept) < 2:
        return [], [], 0, 0, 0, 0
    base_mu, base_var = params(kept)
    base_stddev = math.sqrt(base_var)
    lowest_var, lowest_stddev, split_index = base_var, base_stddev, 0
    mean1 = stddev1 = mean2 = stddev2 = 0
    for i in range(1, len(joined_bins)):
        left = [v for grp in joined_bins[:i] for v in grp]
        right = [v for grp in joined_bins[i:] for v in grp]
        if len(left) < 2 or len(right) < 2 or checkEqualIvo(left) or checkEqualIvo(right):
            continue
        mu1, var1 = params(left)
        mu2, var2 = params(right)
        sum_var = var1 + var2
        sum_stddev = math.sqrt(var1) + math.sqrt(var2)
        if sum_var < lowest_var and sum_stddev < lowest_stddev:
            lowest_var, lowest_stddev = sum_var, sum_stddev
            stddev1, stddev2, mean1, mean2, split_index = math.sqrt(var1), math.sqrt(var2), mu1, mu2, i
    if lowest_var < base_var and lowest_stddev < base_stddev:
        return ([v for grp in joined_bins[:split_index] for v in grp],
                [v for grp in joined_bins[split_index:] for v in grp], mean1, stddev1, mean2, stddev2)
    return observations, [], base_mu, base_stddev, 0, 0
