"""GPU parity of ScorePaths (besst_score_paths, besst_amd.ExtendLargeScaffolds) against the fixture captured from
the reference, and against the CPU restatement on a large batch."""
import numpy as np
import pytest

from tests import scorepaths_util as PU

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', PU.case_names())
def test_mirror_matches_reference_fixture(name):
    from besst_amd import ExtendLargeScaffolds as ELS
    case = PU.by_name(name)
    G = PU.build_graph(case)
    paths = [[tuple(n) for n in p] for p in case['paths']]
    all_paths = []
    ELS.ScorePaths(G, paths, all_paths, PU.Param(case))
    want = [[s, b, paths[i], n] for s, b, i, n in case['all_paths']]
    assert all_paths == want
    assert all(type(a[0]) is type(w[0]) for a, w in zip(all_paths, want))     # int stays int when nothing is bad


def test_large_batch_matches_cpu_restatement():
    """200 k paths on a 50 k-scaffold link graph, incl. paths longer than the LDS window (global-memory scan)."""
    from besst_amd import _lib
    from oracle import scorepaths_oracle as PO
    rng = np.random.default_rng(11)
    n_scaf = 50000
    n_nodes = 2 * n_scaf
    m = 150000
    a = rng.integers(0, n_nodes, m)
    b = rng.integers(0, n_nodes, m)
    keep = (a >> 1) != (b >> 1)
    a, b = a[keep], b[keep]
    w = rng.integers(1, 50, a.shape[0])
    src = np.concatenate([a, b])
    dst = np.concatenate([b, a]).astype(np.int32)
    ww = np.concatenate([w, w]).astype(np.int32)
    order = np.argsort(src, kind='stable')
    col, weight = np.ascontiguousarray(dst[order]), np.ascontiguousarray(ww[order])
    row_ptr = np.zeros(n_nodes + 1, np.int64)
    np.cumsum(np.bincount(src, minlength=n_nodes), out=row_ptr[1:])
    # paths: random walks over the link graph (so that partners really sit in the path), plus two very long ones
    n_paths = 200000
    lens = rng.choice([1, 2, 3, 4, 6, 8, 12, 20, 40], n_paths)
    lens[:2] = (700, 1300)
    path_ptr = np.zeros(n_paths + 1, np.int64)
    np.cumsum(lens, out=path_ptr[1:])
    nodes = np.empty(int(path_ptr[-1]), np.int32)
    cur = rng.integers(0, n_nodes, n_paths)
    for step in range(int(lens.max())):
        live = np.nonzero(lens > step)[0]
        nodes[path_ptr[live] + step] = cur[live]
        if step % 2 == 0:                                   # follow a link edge if there is one, else jump
            deg = row_ptr[cur[live] + 1] - row_ptr[cur[live]]
            pick = row_ptr[cur[live]] + (rng.integers(0, 1 << 30, live.shape[0]) % np.maximum(deg, 1))
            nxt = np.where(deg > 0, col[np.minimum(pick, col.shape[0] - 1)], rng.integers(0, n_nodes, live.shape[0]))
        else:                                               # cross the scaffold
            nxt = cur[live] ^ 1
        cur[live] = nxt
    lib = _lib.load()
    for contamination in (0, 1):
        good = np.zeros(n_paths, np.int64)
        bad = np.zeros(n_paths, np.int64)
        _lib.check(lib.besst_score_paths(0, n_nodes, _lib.ptr(row_ptr), _lib.ptr(col), _lib.ptr(weight), n_paths,
                                         _lib.ptr(path_ptr), _lib.ptr(nodes), contamination, _lib.ptr(good),
                                         _lib.ptr(bad)), 'besst_score_paths')
        rp, cl, wl, nl = row_ptr.tolist(), col.tolist(), weight.tolist(), nodes.tolist()
        check = list(range(0, 2)) + rng.integers(0, n_paths, 3000).tolist()
        for p in check:
            want = PO.link_weights(rp, cl, wl, nl[path_ptr[p]:path_ptr[p + 1]], bool(contamination))
            assert (int(good[p]), int(bad[p])) == want, (p, contamination)
        assert good.sum() > 0 and bad.sum() > 0


def test_empty_and_bad_arguments():
    from besst_amd import ExtendLargeScaffolds as ELS
    from besst_amd import _lib
    case = PU.by_name(PU.case_names()[0])
    G = PU.build_graph(case)
    out = []
    assert ELS.ScorePaths(G, [], out, PU.Param(case)) == () and out == []
    lib = _lib.load()
    row_ptr = np.zeros(3, np.int64)
    path_ptr = np.array([0, 1], np.int64)
    nodes = np.array([7], np.int32)
    g = np.zeros(1, np.int64)
    assert lib.besst_score_paths(0, 2, _lib.ptr(row_ptr), None, None, 1, _lib.ptr(path_ptr), _lib.ptr(nodes), 0,
                                 _lib.ptr(g), _lib.ptr(g)) == 1
    assert 'out of range' in _lib.last_error()
