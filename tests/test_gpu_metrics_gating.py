"""The library-metrics pass (besst_ctx_metrics_sample, csrc/metrics.hip) reads a wave's reference ids first and the other four
columns only if one of the wave's records lies on a top-1000 contig.  Libraries in which such records are rare, clustered,
absent or everywhere - with unplaced reads (id -1), ids beyond the contig count, sizes around the tile and chunk edges - against
the C oracle's statement of libmetrics.py:63-84, 293-303."""
import numpy as np
import pytest

from besst_amd import device
from besst_amd.records import RecordBatch
from oracle import c_oracle as CO

pytestmark = pytest.mark.gpu


def _library(rng, n, nc, unplaced=0.02, beyond=0.0):
    """A coordinate-sorted stream of n records on nc contigs; flags / mapq / tlen at random so that every predicate fires."""
    tid = np.sort(rng.integers(0, nc, n)).astype(np.int32)
    k = int(n * unplaced)
    if k:
        tid[n - k:] = -1                                       # unplaced reads sort last
    if beyond:
        tid[rng.random(n) < beyond] = nc + 7                   # (never written by an aligner: the clamp, not the branch)
    same = rng.random(n) < 0.85
    mtid = np.where(same, tid, rng.integers(0, nc, n)).astype(np.int32)
    tlen = rng.integers(-3000, 3001, n).astype(np.int32)
    tlen[rng.random(n) < 0.01] = 0
    tlen[rng.random(n) < 0.001] = np.iinfo(np.int32).min
    flag = (rng.integers(0, 2, n) * 0x80 + (rng.random(n) < 0.5) * 0x10 + (rng.random(n) < 0.5) * 0x20 +
            (rng.random(n) < 0.03) * 0x4 + (rng.random(n) < 0.03) * 0x8 + (rng.random(n) < 0.02) * 0x100).astype(np.uint16)
    mapq = rng.integers(0, 61, n).astype(np.uint8)
    z = np.zeros(n, np.int32)
    return RecordBatch(['c%d' % i for i in range(nc)], [1000] * nc, tid=tid, mtid=mtid, pos=z, mpos=z, tlen=tlen, flag=flag,
                       mapq=mapq, qlen=np.full(n, 100, np.uint16))


def _check(batch, nc, top, orientation, read_len, want_isize=True, min_mapq=10):
    with device.GraphContext(0) as ctx:
        one = np.ones(nc, np.int32)
        ctx.set_contigs(scaf_id=np.arange(1, nc + 1, dtype=np.int32), scaf_len=one * 1000, ctg_pos=one * 0, ctg_len=one * 1000,
                        direction=one.astype(np.uint8), cls=one.astype(np.uint8))
        ctx.push_records(batch)
        isize, contam, counts = ctx.metrics_sample(top, orientation, min_mapq, read_len, want_isize)
        isize, contam = isize.copy(), contam.copy()
    w_isize, w_contam, w = CO.metrics_sample(batch, top, orientation, min_mapq, read_len, want_isize)
    assert [counts.n_isize, counts.n_contam, counts.counter_total, counts.sample_counter] == w.tolist()
    assert np.array_equal(isize, w_isize)
    assert np.array_equal(contam, w_contam)
    return counts


@pytest.mark.parametrize('n', [1, 3, 255, 1024, 4095, 4096, 4097, 12289, 70001, 1 << 20])
@pytest.mark.parametrize('orientation', ['fr', 'rf'])
def test_sparse_top_contigs_every_size(n, orientation):
    rng = np.random.default_rng(n * 2 + (orientation == 'rf'))
    nc = 3000
    batch = _library(rng, n, nc)
    top = np.zeros(nc, np.uint8)
    top[rng.choice(nc, 60, replace=False)] = 1                 # 2 % of the contigs: most waves hold none of their records
    _check(batch, nc, top, orientation, 100.38)


@pytest.mark.parametrize('share', [0.0, 0.001, 0.3, 1.0])
def test_from_no_top_contig_to_all_of_them(share):
    rng = np.random.default_rng(int(share * 1000) + 5)
    n, nc = 600_000, 2000
    batch = _library(rng, n, nc, beyond=0.01)
    top = (rng.random(nc) < share).astype(np.uint8) if 0.0 < share < 1.0 else np.full(nc, int(share), np.uint8)
    for orientation in ('fr', 'rf'):
        c = _check(batch, nc, top, orientation, 99.999)
        if share == 0.0:
            assert (c.n_isize, c.n_contam, c.counter_total, c.sample_counter) == (0, 0, 0, 0)
    _check(batch, nc, top, 'rf', 0.5, want_isize=False)


def test_one_top_contig_in_the_middle_of_a_tile():
    """A single short top contig whose records straddle two waves of one tile, everything around it ungated."""
    rng = np.random.default_rng(77)
    n, nc = 40_000, 400
    batch = _library(rng, n, nc, unplaced=0.0)
    batch.tid[batch.tid == 200] = 201
    batch.tid[4096 + 250:4096 + 262] = 200                      # lanes 62, 63 of wave 0 and 0, 1 of wave 1 (sub-tile 0 of tile 1)
    batch.mtid[:] = batch.tid
    top = np.zeros(nc, np.uint8)
    top[200] = 1
    c = _check(batch, nc, top, 'fr', 100.0)
    assert c.sample_counter == 12


def test_cut_off_inside_a_gated_stream():
    """More than 1,000,000 qualifying records, the top contigs in clusters far apart: the cut-offs of both lists fall inside
    tiles that are evaluated, behind long stretches that are not."""
    rng = np.random.default_rng(123)
    n, nc = 6_000_000, 5000
    batch = _library(rng, n, nc, unplaced=0.0)
    top = np.zeros(nc, np.uint8)
    for lo in (100, 1500, 2900, 4300):
        top[lo:lo + 350] = 1
    c = _check(batch, nc, top, 'fr', 100.38)
    assert c.sample_counter == 1_000_000 and c.n_isize < 1_000_000
    c = _check(batch, nc, top, 'rf', 100.38)
    assert c.sample_counter == 1_000_000
