"""GPU: score_kernel's ML gap and expected sigma against the brute-force GapEst model (oracle/gapest_numeric.py), which
shares no closed form with the kernel (csrc/score.hip), its host mirror (mathstats_compat.py) or the C oracle.
Reference call sites: CreateGraph.py:537 (gap), :555 (sigma).  Tolerances: gap +-1 bp, sigma 0.5 %."""
import ctypes as C

import numpy as np
import pytest

from oracle import gapest_numeric as GN
from tests.test_gapest_numeric import GRID

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('mu,sigma,r', sorted({(g[0], g[1], g[2]) for g in GRID}))
def test_score_kernel_gap_and_sigma_vs_brute_force(mu, sigma, r):
    import torch
    from besst_amd import pipeline
    pairs = [(g[3], g[4]) for g in GRID if (g[0], g[1], g[2]) == (mu, sigma, r)]
    fracs = (-0.6, -0.2, 0.0, 0.3, 0.7, 1.0)
    edges = [(c1, c2, f) for c1, c2 in pairs for f in fracs]
    n_links = 6
    keys, lo, hi, mean_obs = [], [], [], []
    for e, (c1, c2, f) in enumerate(edges):
        target = mu - f * (mu - 2 * r)                       # mean observation of this edge
        tot = int(round(target * n_links))
        obs = [tot // n_links] * n_links
        for k in range(tot - sum(obs)):
            obs[k] += 1
        mean_obs.append(sum(obs) / float(n_links))
        for k, o in enumerate(obs):
            keys.append((((2 * e + 2) << 12) | (2 * e + 3)) << 1)      # node_bits 12: a distinct node pair per edge
            a = max(26, o // 3 + 7 * k)
            lo.append(a)
            hi.append(o - a)
    keys = np.array(keys, np.uint64)
    lo_a, hi_a = np.array(lo, np.uint64), np.array(hi, np.uint64) | (np.uint64(3) << np.uint64(30))
    payload = lo_a | (hi_a << np.uint64(32))
    n = len(keys)
    dev = torch.device('cuda', 0)
    lib = dict(read_len=float(r), ins_size_threshold=mu + 6 * sigma, min_mapq=11, orientation='fr', detect_duplicate=True,
               extend_paths=True, no_score=False)
    gb = pipeline.DeviceGraphBuilder(dev, 4, 12, lib, n, n)
    dk = torch.from_numpy(keys.view(np.int64)).to(dev)
    dp = torch.from_numpy(payload.view(np.int64)).to(dev)
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    gb.reduce(keys=dk, payload=dp, n_tuples_ptr=C.c_void_p(cnt.data_ptr()), capacity=n)
    torch.cuda.synchronize()
    rows = np.arange(len(edges), dtype=np.uint32)
    len1 = np.array([e[0] for e in edges], np.int32)
    len2 = np.array([e[1] for e in edges], np.int32)
    gap, sd0, ks, flags = gb.score_edges(rows, np.zeros(len(edges), np.uint8), len1, len2, mu, sigma, r)
    ri = int(round(r))
    for i, (c1, c2, f) in enumerate(edges):
        if not (c1 > 2 * sigma and c2 > 2 * sigma):          # the reference keeps the naive gap there (CreateGraph.py:536)
            continue
        want, fs = GN.ml_gap(mu, sigma, ri, mean_obs[i], c1, c2)
        tol = 2 if want in (min(fs), max(fs)) else 1
        assert abs(int(gap[i]) - want) <= tol, (edges[i], gap[i], want)
        want_sd = GN.span_sd(int(gap[i]), mu, sigma, c1, c2, ri)
        if want_sd is not None and sd0[i] < 2 ** 31:
            assert abs(sd0[i] - want_sd) <= 0.005 * want_sd + 0.05, (edges[i], sd0[i], want_sd)


@pytest.mark.parametrize('mu,sigma,r', [(500.0, 50.0, 100), (5199.56, 499.55, 100), (2500.0, 250.0, 100.38)])
def test_gap_table_of_long_contigs(mu, sigma, r):
    """PreCalcMLvaluesOfdLongContigs (MakeScaffolds.py:68,447): the table built from the device's evaluation of the ML
    condition equals the host restatement's (an entry may move by one where the device's erf/exp land on the other side of
    a rounding boundary), and reading a gap off it agrees with the brute-force likelihood to +-1 bp."""
    from besst_amd import device, mathstats_compat as MC
    host = MC.PreCalcMLvaluesOfdLongContigs(mu, sigma, r)
    with device.GraphContext(0) as ctx:
        dev = MC.PreCalcMLvaluesOfdLongContigs(mu, sigma, r, ctx=ctx)
    assert set(dev) == set(host) or len(set(dev) ^ set(host)) <= 2
    common = sorted(set(dev) & set(host))
    assert len(common) > 100
    diff = [k for k in common if dev[k] != host[k]]
    assert all(abs(dev[k] - host[k]) <= 1 for k in diff) and len(diff) <= len(common) // 100 + 1
    big = int(10.0 * (mu + 4 * sigma) + 10.0 * r)
    ri = int(round(r))
    for k in common[5:-5:max(1, len(common) // 12)]:
        want, fs = GN.ml_gap(mu, sigma, ri, mu - k, big, big)      # naive gap k  <=>  mean observation mu - k
        if want in (min(fs), max(fs)):
            continue
        assert abs(dev[k] - want) <= 1, (k, dev[k], want)
