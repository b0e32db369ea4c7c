"""Library statistics: drop-in for ``BESST/libmetrics.py`` with the record scans on the MI355X.

``get_metrics(bam_file, param, Information)`` keeps the reference signature and mutates ``param``
exactly like libmetrics.py:226-432 does (read_len, mean/std of the insert size, -T/-k thresholds,
skewness, GetDistr-adjusted distribution, log-normal switch, PE-contamination metrics).

Split of work:
  * device (csrc/metrics.hip, one ordered pass): the three ``for read in bam_file`` scans - predicate
    evaluation, membership in the 1000 longest references, the "first 1,000,000 in stream order"
    cut-offs, and order-preserving compaction of the |tlen| samples;
  * host: the float finishing on the <= 1,000,000 sampled values.  The reference's results depend on the
    exact order of its float operations (naive left-to-right ``sum``, expanded-square variance, repeated
    ``+= 1/w`` in GetDistr, libm ``pow`` for ``**``); ``get_metrics`` runs a native replay of that order
    (csrc/hostmath.hip, ``besst_host_*``: ~40 ms instead of ~0.8 s of interpreted loops, bit-identical on
    every golden vector).  ``AdjustInsertsizeDist`` and ``getdistr`` below are the same algorithms under
    the reference's names for callers that use them directly; they accumulate explicitly instead of
    ``sum()`` because CPython >= 3.12 sums floats with compensation.

There is no CPU path for the scans: without libbesst_amd.so and a GPU the call raises.
"""
from __future__ import print_function

import math
import sys

import ctypes as C

import numpy as np

from . import _lib, session
from .mathstats_compat import MaxObsDistr


def _native_isize_stats(abs_values, is_float, offset):
    """(kept indexes, [mean0, std0, mean, std, skewness]) via libbesst_amd's host finishing (csrc/hostmath.hip)."""
    lib = _lib.load()
    vals = np.ascontiguousarray(abs_values, dtype=np.int32)
    kept = np.empty(vals.shape[0], dtype=np.int64)
    n_kept = C.c_int64()
    stats = np.zeros(5, dtype=np.float64)
    _lib.check(lib.besst_host_isize_stats(_lib.ptr(vals), vals.shape[0], int(is_float), float(offset), _lib.ptr(kept),
                                          C.byref(n_kept), _lib.ptr(stats)), 'host_isize_stats')
    return vals, kept[:n_kept.value], stats.tolist()


def _native_getdistr(vals, kept, is_float, offset, cont_lengths_list):
    lib = _lib.load()
    lens = np.ascontiguousarray(cont_lengths_list, dtype=np.int32)
    cap = int(vals.max()) + int(offset) + 4
    adjusted = np.zeros(cap, dtype=np.float64)
    n_adj = C.c_int64()
    out = np.zeros(26, dtype=np.float64)
    _lib.check(lib.besst_host_getdistr(_lib.ptr(vals), _lib.ptr(kept), kept.shape[0], int(is_float), float(offset),
                                       _lib.ptr(lens), lens.shape[0], _lib.ptr(adjusted), cap, C.byref(n_adj),
                                       _lib.ptr(out)), 'host_getdistr')
    return adjusted[:n_adj.value].tolist(), out.tolist()


def _native_contam_stats(abs_values, is_float, offset):
    lib = _lib.load()
    vals = np.ascontiguousarray(abs_values, dtype=np.int32)
    n_final = C.c_int64()
    stats = np.zeros(4, dtype=np.float64)
    _lib.check(lib.besst_host_contam_stats(_lib.ptr(vals), vals.shape[0], int(is_float), float(offset),
                                           C.byref(n_final), _lib.ptr(stats)), 'host_contam_stats')
    return n_final.value, stats.tolist()


def _acc(values):
    total = 0
    for v in values:
        total = total + v
    return total


def _mean_and_std(xs):
    n = float(len(xs))
    mean = _acc(xs) / n
    sq = 0
    for x in xs:
        sq = sq + (x ** 2 - 2 * x * mean + mean ** 2)
    return mean, (sq / (n - 1)) ** 0.5


def AdjustInsertsizeDist(param, mean_insert, std_dev_insert, insert_list):
    """One trimming round: keep observations within 1.5 * MaxObsDistr(n, 0.95) sigmas (libmetrics.py:22-28)."""
    k = 1.5 * MaxObsDistr(len(insert_list), 0.95)
    lo, hi = mean_insert - k * std_dev_insert, mean_insert + k * std_dev_insert
    kept = [x for x in insert_list if (x < hi and x > lo)]
    return len(insert_list) > len(kept), kept


def largest_reference_indexes(lengths, k=1000):
    """Indexes of the k longest references; ties go to the smaller index (heapq.nlargest, libmetrics.py:233).  One
    partition instead of a sort of the whole header: everything longer than the k-th longest, then the first of those that
    are as long as it, in index order."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = lengths.shape[0]
    if n <= k:
        return np.lexsort((np.arange(n), -lengths))
    kth = np.partition(lengths, n - k)[n - k]                 # the k-th longest length
    above = np.flatnonzero(lengths > kth)
    equal = np.flatnonzero(lengths == kth)[:k - above.shape[0]]
    chosen = np.concatenate([above, equal])
    return chosen[np.lexsort((chosen, -lengths[chosen]))]


def getdistr(ins_size_reads, cont_lengths_list, param, Information):
    """GetDistr observation-bias correction (libmetrics.py:141-223), quirks included."""
    largest = sorted(int(x) for x in sorted(cont_lengths_list, reverse=True)[:1000])
    max_isize = int(max(ins_size_reads))
    adjusted = [0] * int(max_isize + 1)
    cur_sum = _acc(largest)
    cur_nr = len(largest)
    at_least = [(cur_nr, cur_sum)]           # index obs -> (#contigs, their total length) for isize obs-1
    smallest, smallest_index = largest[0], 0
    upper_isize = min(max_isize + 1, largest[-1])
    for isize in range(upper_isize):
        if isize > smallest:
            while isize > largest[smallest_index]:
                smallest_index += 1
                cur_nr -= 1
                cur_sum -= smallest          # the reference subtracts the not-yet-updated value (:164-170)
            at_least.append((cur_nr, cur_sum))
            smallest = largest[smallest_index]
        else:
            at_least.append((cur_nr, cur_sum))
    for o in ins_size_reads:
        obs = int(o)
        if obs > upper_isize:
            continue
        nr_ctgs, sum_ctgs = at_least[obs]
        adjusted[obs] += 1 / float(max(sum_ctgs - (obs - 1) * nr_ctgs, 10000))

    tot_density = float(_acc(adjusted))
    cum, curr = 0, 0
    while cum <= tot_density / 2.0:
        cum += adjusted[curr]
        curr += 1
    median_adj = curr

    modes = []
    for chunk_size in range(1, 102, 5):
        best_i, best_v = 0, None
        for ci, start in enumerate(range(0, len(adjusted), chunk_size)):
            v = _acc(adjusted[start:start + chunk_size])
            if best_v is None or v > best_v:
                best_i, best_v = ci, v
        mode = (best_i + 0.5) * chunk_size
        modes.append(int(mode))
        print('mode for chunk size ', chunk_size, ' : ', mode, file=Information)
    mode_adj = sorted(modes)[int(len(modes) / 2)]

    mu_adj = _acc([i * f for i, f in enumerate(adjusted)]) / tot_density
    sigma_adj = math.sqrt(_acc([(i - mu_adj) ** 2 * f for i, f in enumerate(adjusted)]) / tot_density)
    m_3 = _acc([(i - mu_adj) ** 3 * f for i, f in enumerate(adjusted)]) / tot_density
    skew_adj = m_3 / sigma_adj ** 3
    return adjusted, mu_adj, sigma_adj, skew_adj, median_adj, mode_adj


def _finish_contamination(abs_contam, is_float, offset, counter_total, param, Information):
    """Trimming and acceptance rule of get_contamination_metrics (libmetrics.py:86-128).

    ``abs_contam``: abs(tlen) of the opposite-orientation pairs; for an 'fr' library the fragment size is
    abs(tlen) + 2*read_len (a float), for 'rf' the integer abs(tlen) itself (:71-81)."""
    n_contamine = float(len(abs_contam))
    mean_isize, std_dev_isize = 0, 0
    if n_contamine > 2:
        n_final, st = _native_contam_stats(abs_contam, is_float, offset if is_float else 0.0)
        print('Contamine mean before filtering :', st[0], file=Information)
        print('Contamine stddev before filtering: ', st[1], file=Information)
        n_contamine = float(n_final)
        mean_isize, std_dev_isize = st[2], st[3]
        print('Contamine mean converged:', mean_isize, file=Information)
        print('Contamine std_est converged: ', std_dev_isize, file=Information)
    ratio = 2 * n_contamine / float(counter_total) if counter_total > 0 else 0
    if mean_isize >= param.mean_ins_size or std_dev_isize >= param.std_dev_ins_size or ratio <= 0.05:
        param.contamination_ratio = False
        param.contamination_mean = 0
        param.contamination_stddev = 0
    else:
        param.contamination_mean = mean_isize
        param.contamination_stddev = std_dev_isize
        param.contamination_ratio = ratio
    return n_contamine


def _set_thresholds(param):
    param.ins_size_threshold = param.mean_ins_size + 6 * param.std_dev_ins_size
    if param.extend_paths:
        param.contig_threshold = param.mean_ins_size + 4 * param.std_dev_ins_size
    else:
        param.contig_threshold = param.mean_ins_size + \
            (param.std_dev_ins_size / float(param.mean_ins_size)) * param.std_dev_ins_size


def get_metrics(bam_file, param, Information):
    sess = session.open_session(bam_file)
    batch = sess.batch
    cont_lengths_list = list(batch.lengths)
    top = largest_reference_indexes(cont_lengths_list)
    param.lognormal = False

    # libmetrics.py:237-241 refuses a BAM without an index - and only a coordinate-sorted file has one.  The kernels here
    # are exact on any order (and there is no index to ask for), so the drop-in does not refuse: it checks the order of the
    # resident stream and passes the reference's message on as a warning.
    unsorted_at = sess.stream_order()
    if unsorted_at is not None:
        text = ('WARNING: the alignments are not sorted by coordinate (record {0} lies in front of its predecessor). BESST '
                'itself stops here: "Need indexed bamfiles, index file should be located in the same directory as the BAM '
                'file" (only a coordinate-sorted BAM can be indexed).  Results stay exact, but the graph build runs '
                'several times slower on an unsorted stream - sort the file (samtools sort) for full speed.'
                .format(unsorted_at))
        sys.stderr.write(text + '\n')
        print(text, file=Information)
        param.stream_unsorted_at = unsorted_at

    if not param.read_len:                                        # libmetrics.py:246-273
        if len(batch) < 1000:
            sys.stderr.write('Did not get sufficient readmappings to calculate read_length from mappings. '
                             'Got {0} mappings. Please provide this parameter or more importantly check why '
                             'almost no reads are mapping to the contigs.\nterminating..\n'.format(len(batch)))
            sys.exit(0)
        rlen = (batch.rlen if batch.rlen is not None else batch.qlen)[:1000].astype(np.int64)
        alen = (batch.alen if batch.alen is not None else batch.qlen)[:1000].astype(np.int64)
        param.read_len = int(np.where(rlen != 0, rlen, alen).sum()) / float(1000)

    if param.mean_ins_size and param.std_dev_ins_size and not param.ins_size_threshold:   # :275-281
        _set_thresholds(param)
        print('-T', param.ins_size_threshold, '-t', param.contig_threshold, file=Information)

    want_isize = not param.mean_ins_size
    top_mask = np.zeros(len(cont_lengths_list), dtype=np.uint8)
    top_mask[top] = 1
    abs_isize, abs_contam, counts = sess.metrics_sample(top_mask, param.orientation, param.min_mapq,
                                                        param.read_len, want_isize)

    if want_isize:                                                # :283-390
        is_float = param.orientation != 'fr'
        offset = 2 * param.read_len if is_float else 0.0
        n_obs = int(abs_isize.shape[0])
        print('Estimating insert size from {0} mappings with quality over --min_mapq {1}.'.format(
            n_obs + 1, param.min_mapq), file=Information)
        if n_obs <= 1000:
            sys.stderr.write('To few valid read alignments exists to compute mean and variance of library (need at '
                             'least 1000 observations). Got only ' + str(n_obs) +
                             ' valid alignments. Please specify -m and -s to the program. \nPrinting out '
                             'scaffolds produced in earlier steps...\nterminating...\n')
            sys.exit(0)
        # mean / stddev, iterative trimming and skewness: native replay of the reference's float order
        vals, kept, st = _native_isize_stats(abs_isize, is_float, offset)
        print('Mean before filtering :', st[0], file=Information)
        print('Std_est  before filtering: ', st[1], file=Information)
        mean_isize, std_dev_isize = st[2], st[3]
        print('Mean converged:', mean_isize, file=Information)
        print('Std_est converged: ', std_dev_isize, file=Information)
        param.mean_ins_size = mean_isize
        param.std_dev_ins_size = std_dev_isize
        param.skewness = st[4]
        print('Skewness of distribution: ', param.skewness, file=Information)

        adj_distr, gd = _native_getdistr(vals, kept, is_float, offset, cont_lengths_list)
        mu_adj, sigma_adj, skew_adj, median_adj, mode_adj = gd[0], gd[1], gd[2], int(gd[3]), int(gd[4])
        for chunk_size, mode in zip(range(1, 102, 5), gd[5:26]):
            print('mode for chunk size ', chunk_size, ' : ', mode, file=Information)
        param.skew_adj = skew_adj
        param.empirical_distribution = dict(zip(range(len(adj_distr)), adj_distr))
        print('Mean of getdistr adjusted distribution: ', mu_adj, file=Information)
        print('Sigma of getdistr adjusted distribution: ', sigma_adj, file=Information)
        print('Skewness of getdistr adjusted distribution: ', skew_adj, file=Information)
        print('Median of getdistr adjusted distribution: ', median_adj, file=Information)
        print('Mode of getdistr adjusted distribution: ', mode_adj, file=Information)
        print('Using mean and stddev of getdistr adjusted distribution from here: ', mu_adj, sigma_adj,
              file=Information)
        param.mean_ins_size = mu_adj
        param.std_dev_ins_size = sigma_adj
        if param.skew_adj > 0.5 and math.log(median_adj) > math.log(mode_adj):
            print('Mode on getdistr adjusted: ', mode_adj, file=Information)
            print('Median on getdistr adjusted:', median_adj, file=Information)
            param.lognormal_mean = math.log(median_adj)
            param.lognormal_sigma = math.sqrt(param.lognormal_mean - math.log(mode_adj))
            print('Lognormal mean getdistr adjusted: ', param.lognormal_mean, file=Information)
            print('Lognormal stddev getdistr adjusted', param.lognormal_sigma, file=Information)
            param.lognormal = True

    if not param.ins_size_threshold:                              # :404-409
        _set_thresholds(param)

    # contamination: opposite-orientation pairs on the 1000 longest references (:49-131,412)
    n_contamine = _finish_contamination(abs_contam, param.orientation == 'fr', 2 * param.read_len,
                                        counts.counter_total, param, Information)

    print('', file=Information)
    print('LIBRARY STATISTICS', file=Information)
    print('Mean of library set to:', param.mean_ins_size, file=Information)
    print('Standard deviation of library set to: ', param.std_dev_ins_size, file=Information)
    print('MP library PE contamination:', file=Information)
    print('Contamine rate (rev comp oriented) estimated to: ', param.contamination_ratio, file=Information)
    print('lib contamine mean (avg fragmentation size): ', param.contamination_mean, file=Information)
    print('lib contamine stddev: ', param.contamination_stddev, file=Information)
    print('Number of contamined reads used for this calculation: ', n_contamine, file=Information)
    print('-T (library insert size threshold) set to: ', param.ins_size_threshold, file=Information)
    print('-k set to (Scaffolding with contigs larger than): ', param.contig_threshold, file=Information)
    print('Number of links required to create an edge: ', param.edgesupport, file=Information)
    print('Maximum identical contig-end overlap-length to merge of contigs that are adjacent in a scaffold: ',
          param.max_contig_overlap, file=Information)
    print('Read length set to: ', param.read_len, file=Information)
    print('', file=Information)
    return ()
