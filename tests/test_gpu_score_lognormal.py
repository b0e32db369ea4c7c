"""GPU: the log-normal branch of GiveScoreOnEdges on the device (csrc/score.hip: lognormal_gap, ln_tile_kernel,
cond_stddev_kernel) against the host restatement (besst_amd/mathstats_compat.lognormal_GapEstimator, prefix tables) and
against the oracle's direct-sum form (oracle/py_oracle.lognormal_gap_estimator, no tables), which share no code.
Reference call sites: CreateGraph.py:485-494 (conditional sigma table), :522-531 (gap), :549-553 (sigma look-up).
mathstats 0.2.6.5 itself is not vendored: PARITY UNPINNED, the estimator is the restatement's.
Tolerances: gap exact (it is an argmax over integers; a tie in the last bits may move it by 1 bp, counted and bounded),
tables 1e-11, sigmas 1e-12 relative."""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MU, SIGMA, R = math.log(3000.0), 0.35, 100


def spanning_observations(rng, n, d, c1, c2, mu=MU, sigma=SIGMA, r=R):
    """n link observations (obs on contig 1, obs on contig 2) of fragments x ~ LogNormal(mu, sigma) that span a gap d with
    both reads inside their contigs."""
    lo, hi = [], []
    while len(lo) < n:
        m = max(64, 4 * (n - len(lo)))
        x = np.rint(np.exp(rng.normal(mu, sigma, m))).astype(np.int64)
        a = rng.integers(r, c1 + 1, m)                       # bases of the fragment on contig 1
        b = x - d - a                                        # ... on contig 2
        ok = (b >= r) & (b <= c2) & (x >= 2 * r)
        lo.extend(a[ok].tolist())
        hi.extend(b[ok].tolist())
    return lo[:n], hi[:n]


def build_rows(edges, seed):
    """edges: [(n_links, true_gap, c1, c2)] -> a DeviceGraphBuilder holding one row per edge + the host copy of its lists."""
    import torch
    from besst_amd import pipeline
    rng = np.random.default_rng(seed)
    keys, lo, hi, samples = [], [], [], []
    for e, (n, d, c1, c2) in enumerate(edges):
        a, b = spanning_observations(rng, n, d, c1, c2)
        keys.extend([(((2 * e + 2) << 20) | (2 * e + 3)) << 1] * n)           # node_bits 20: a node pair per edge
        lo.extend(a)
        hi.extend(b)
        samples.append([x + y for x, y in zip(a, b)])
    keys = np.array(keys, np.uint64)
    payload = np.array(lo, np.uint64) | ((np.array(hi, np.uint64) | (np.uint64(3) << np.uint64(30))) << np.uint64(32))
    n = len(keys)
    dev = torch.device('cuda', 0)
    lib = dict(read_len=float(R), ins_size_threshold=1.0e6, min_mapq=11, orientation='fr', detect_duplicate=True,
               extend_paths=True, no_score=False)
    gb = pipeline.DeviceGraphBuilder(dev, 4, 20, lib, n, n)
    dk = torch.from_numpy(keys.view(np.int64)).to(dev)
    dp = torch.from_numpy(payload.view(np.int64)).to(dev)
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    gb.reduce(keys=dk, payload=dp, n_tuples_ptr=C.c_void_p(cnt.data_ptr()), capacity=n)
    torch.cuda.synchronize()
    assert gb.read_sizes()[1] == len(edges)
    return gb, samples


def test_prefix_tables_equal_the_host_cumsum():
    import torch
    from besst_amd import mathstats_compat as MC, pipeline
    for mu, sigma in ((MU, SIGMA), (math.log(400.0), 0.2), (math.log(8000.0), 0.8)):
        x_max, F0, F1 = MC._lognormal_tables(mu, sigma)
        assert x_max == MC.lognormal_support(mu, sigma)
        gb = pipeline.DeviceGraphBuilder(torch.device('cuda', 0), 4, 8, dict(
            read_len=100.0, ins_size_threshold=1e6, min_mapq=11, orientation='fr', detect_duplicate=True, extend_paths=True,
            no_score=False), 64, 64)
        d0, d1 = gb.lognormal_tables(mu, sigma, x_max)
        d0, d1 = d0.cpu().numpy(), d1.cpu().numpy()
        assert d0.shape[0] == x_max + 1 and d0[0] == 0.0 and d1[0] == 0.0
        assert np.allclose(d0, F0, rtol=1e-11, atol=0.0) and np.allclose(d1, F1, rtol=1e-11, atol=0.0)
        assert np.all(np.diff(d0) >= 0)
        again0, _ = gb.lognormal_tables(mu, sigma, x_max)    # cached: the same tensor
        assert again0.data_ptr() == gb._ln_F.data_ptr()


def test_conditional_sigmas_equal_the_reference_loop():
    """cond_stddev_kernel == besst_amd.CreateGraph.get_conditional_stddevs (the reference's loop, :436-469) on the
    fr_lognormal golden's empirical distribution and on a ragged one."""
    from besst_amd import CreateGraph as CG, device
    from tests import golden_util as GU
    doc, _ = GU.load('fr_lognormal')
    ed = doc['metrics']['empirical_distribution']
    dists = [{i: v for i, v in enumerate(ed) if v}]
    rng = np.random.default_rng(5)
    keys = np.unique(rng.integers(150, 9000, 700))
    dists.append({int(k): float(v) for k, v in zip(keys, rng.random(keys.shape[0]))})
    with device.GraphContext(0) as ctx:
        for emp in dists:
            max_isize = sorted(emp)[-1]
            steps = list(range(0, int(max_isize * 0.8), max_isize // 50))
            want = CG.get_conditional_stddevs(steps, emp, max_isize)
            got = CG.expand_conditional_stddevs(steps, ctx.conditional_stddevs(CG.dense_distribution(emp, max_isize), steps))
            assert len(got) == len(want)
            assert np.allclose(got, want, rtol=1e-12, atol=0.0)


def _edges(rng, count, n_lo, n_hi):
    out = []
    for _ in range(count):
        n = int(np.exp(rng.uniform(np.log(n_lo), np.log(n_hi))))
        c1, c2 = (int(v) for v in rng.integers(1200, 60000, 2))
        d = int(rng.integers(-300, 2500))
        out.append((n, d, c1, c2))
    return out


def test_lognormal_gaps_equal_the_host_restatement_and_the_oracle():
    """All three forms of an edge inside the kernel: observations in the small LDS buffer (next_pow2(n) <= 1024), in the
    large one (<= 16384) and read from the columns (more); lists of one and two links; a short scaffold (naive gap kept)."""
    from besst_amd import mathstats_compat as MC
    from oracle import py_oracle as O
    rng = np.random.default_rng(11)
    edges = _edges(rng, 260, 1, 900) + _edges(rng, 24, 1025, 9000) + _edges(rng, 2, 16500, 20000)
    edges += [(1, 400, 5000, 7000), (2, 0, 3000, 3000), (7, 100, 500, 9000), (40, 900, 40000, 650)]
    gb, samples = build_rows(edges, 12)
    x_max = MC.lognormal_support(MU, SIGMA)
    mean, sd = 3200.0, 1100.0                                # the library's normal parameters: only 2 sd < len is read
    max_gap = 100000
    rows = np.arange(len(edges), dtype=np.uint32)
    len1 = np.array([e[2] for e in edges], np.int32)
    len2 = np.array([e[3] for e in edges], np.int32)
    for swap in (0, 1):
        l1, l2 = (len1, len2) if not swap else (len2, len1)
        gap, sd0, ks, flags = gb.score_edges(rows, np.full(len(edges), swap, np.uint8), l1, l2, mean, sd, R,
                                             lognormal=(MU, SIGMA, x_max, max_gap))
        off_by_one = 0
        for i, (n, d, c1, c2) in enumerate(edges):
            long_enough = 2 * sd < c1 and 2 * sd < c2
            assert bool(flags[i] & 1) == long_enough
            if not long_enough:
                assert gap[i] == (n * mean - sum(samples[i])) / float(n)
                continue
            want = MC.lognormal_GapEstimator(MU, SIGMA, R, samples[i], int(l1[i]), c2_len=int(l2[i]))
            assert abs(int(gap[i]) - want) <= 1, (edges[i], gap[i], want)
            off_by_one += int(gap[i]) != want
            assert bool(flags[i] & 2) == (-want > c1 or -want > c2)
            if n <= 60 and i % 4 == 0:
                assert O.lognormal_gap_estimator(MU, SIGMA, R, samples[i], int(l1[i]), int(l2[i])) == want
        assert off_by_one == 0
        assert np.all(sd0 == 2.0 ** 32)
    # the clamp of CreateGraph.py:527-528
    gap, _, _, _ = gb.score_edges(rows, np.zeros(len(edges), np.uint8), len1, len2, mean, sd, R, lognormal=(MU, SIGMA, x_max, 150))
    full, _, _, _ = gb.score_edges(rows, np.zeros(len(edges), np.uint8), len1, len2, mean, sd, R,
                                   lognormal=(MU, SIGMA, x_max, max_gap))
    long_enough = (2 * sd < len1) & (2 * sd < len2)
    assert np.array_equal(gap[long_enough], np.minimum(full[long_enough], 150.0))
    assert np.array_equal(gap[~long_enough], full[~long_enough])


def test_observations_outside_the_support_fall_back_to_the_median_rule():
    """d_hi < d_lo (an observation beyond exp(mu + 6 sigma)): int(round(exp(mu) - mean(obs))), as the restatement."""
    from besst_amd import mathstats_compat as MC
    mu, sigma = math.log(400.0), 0.1                         # support 1 .. 728
    x_max = MC.lognormal_support(mu, sigma)
    edges = [(30, 0, 5000, 5000)]
    gb, samples = build_rows(edges, 3)                       # observations of the 3000-bp library: far outside
    assert max(samples[0]) - min(samples[0]) > x_max - 1
    gap, _, _, flags = gb.score_edges(np.zeros(1, np.uint32), np.zeros(1, np.uint8), np.array([5000], np.int32),
                                      np.array([5000], np.int32), 400.0, 40.0, R, lognormal=(mu, sigma, x_max, 10 ** 6))
    want = MC.lognormal_GapEstimator(mu, sigma, R, samples[0], 5000, c2_len=5000)
    assert int(gap[0]) == want and flags[0] & 1


def test_ks_numerators_are_those_of_the_normal_branch():
    """The link-dispersity part of the kernel is shared: same h whichever gap estimator ran."""
    rng = np.random.default_rng(21)
    edges = _edges(rng, 40, 5, 3000)
    gb, _ = build_rows(edges, 22)
    rows = np.arange(len(edges), dtype=np.uint32)
    len1 = np.array([e[2] for e in edges], np.int32)
    len2 = np.array([e[3] for e in edges], np.int32)
    from besst_amd import mathstats_compat as MC
    a = gb.score_edges(rows, np.zeros(len(edges), np.uint8), len1, len2, 3200.0, 1100.0, R)
    b = gb.score_edges(rows, np.zeros(len(edges), np.uint8), len1, len2, 3200.0, 1100.0, R,
                       lognormal=(MU, SIGMA, MC.lognormal_support(MU, SIGMA), 10 ** 6))
    assert np.array_equal(a[2], b[2])
