"""Seeded synthetic read-pair streams shaped like the BASELINE.json configs.

Implements the generator described in SURVEY.md section 8(d) / BASELINE.md section 3:
a genome of ``nc`` contigs (log-normal lengths, clipped to [500, 200000]) laid end
to end with U[0,500] gaps; fragments start uniformly on the genome; a pair is kept
only if both 100-bp reads lie fully inside contigs; PE libraries are ``fr`` with
N(mean, sd) inserts, MP libraries are ``rf`` with an optional opposite-orientation
(PE contamination) sub-population; mapq is 60 / 0 / U[1,59] with probability
0.8 / 0.1 / 0.1 (same for both mates); 1 % of pairs are exact duplicates
(adjacent after sorting); 0.1 % of contig-spanning pairs get a read1 record that is
flagged unmapped while keeping its own placement (the BWA quirk the reference
guards against, CreateGraph.py:141-163); every pair emits two records and the
stream is sorted by (tid, pos) like a coordinate-sorted BAM.

Everything is numpy on the host: this is test/bench scaffolding, not the product.
"""
import numpy as np

from .records import (FLAG_MATE_REVERSE, FLAG_MATE_UNMAPPED, FLAG_PAIRED, FLAG_PROPER, FLAG_READ1,
                      FLAG_READ2, FLAG_REVERSE, FLAG_UNMAPPED, RecordBatch)

BASE_SEED = 20240929
_DTYPES = dict(tid=np.int32, mtid=np.int32, pos=np.int32, mpos=np.int32, tlen=np.int32, flag=np.uint16,
               mapq=np.uint8, qlen=np.uint16)


class Assembly(object):
    def __init__(self, names, lengths, gaps):
        self.names = list(names)
        self.lengths = np.asarray(lengths, dtype=np.int64)
        self.gaps = np.asarray(gaps, dtype=np.int64)
        self.starts = np.concatenate(([0], np.cumsum(self.lengths + self.gaps)[:-1]))
        self.total = int(self.starts[-1] + self.lengths[-1])

    @property
    def nc(self):
        return len(self.names)


def make_assembly(nc, median_len, seed, sigma_log=1.0, min_len=500, max_len=200000):
    rng = np.random.default_rng(seed)
    lengths = np.exp(rng.normal(np.log(median_len), sigma_log, nc))
    lengths = np.clip(lengths, min_len, max_len).astype(np.int64)
    gaps = rng.integers(0, 501, nc)
    names = ['ctg%07d' % i for i in range(nc)]
    return Assembly(names, lengths, gaps)


class LibrarySpec(object):
    def __init__(self, orientation='fr', mean=500.0, sd=50.0, contam_frac=0.0, contam_mean=350.0,
                 contam_sd=60.0, read_len=100, dup_frac=0.01, fishy_frac=0.001, softclip_frac=0.05, lognormal_sigma=None,
                 chimeric_frac=0.0):
        self.orientation = orientation
        self.mean = mean
        self.sd = sd
        self.contam_frac = contam_frac
        self.contam_mean = contam_mean
        self.contam_sd = contam_sd
        self.read_len = read_len
        self.dup_frac = dup_frac
        self.fishy_frac = fishy_frac
        self.softclip_frac = softclip_frac
        self.lognormal_sigma = lognormal_sigma      # insert sizes exp(N(ln mean, .)) instead of N(mean, sd): a skewed library
        # share of the contig-spanning pairs whose second read is moved, at the same offset, to a contig drawn at random
        # (chimeric fragments: links on edges of their own; simulate_library_device only)
        self.chimeric_frac = chimeric_frac


def simulate_library(asm, spec, n_pairs, seed, chunk=4_000_000):
    """Generate exactly ``n_pairs`` placed read pairs (2 records each) -> RecordBatch.

    Candidate fragments whose reads are not both fully inside contigs are dropped and
    replaced by further draws, so the stream holds ``n_pairs`` pairs, duplicates included.
    """
    rng = np.random.default_rng(seed)
    r = spec.read_len
    cols = {k: [] for k in ('tid', 'mtid', 'pos', 'mpos', 'tlen', 'flag', 'mapq', 'qlen')}
    done = 0
    while done < n_pairs:
        m = min(chunk, max(1024, int((n_pairs - done) * 1.3)))
        start = rng.integers(0, asm.total, m)
        contam = rng.random(m) < spec.contam_frac
        main = rng.normal(spec.mean, spec.sd, m)
        if getattr(spec, 'lognormal_sigma', None):
            main = np.exp(np.log(spec.mean) + spec.lognormal_sigma * (main - spec.mean) / spec.sd)
        x = np.where(contam, rng.normal(spec.contam_mean, spec.contam_sd, m), main)
        x = np.maximum(np.rint(x).astype(np.int64), 2 * r)
        lpos = start                       # left read  [lpos, lpos + r)
        rpos = start + x - r               # right read [rpos, rpos + r)
        lt = np.searchsorted(asm.starts, lpos, side='right') - 1
        rt = np.searchsorted(asm.starts, rpos, side='right') - 1
        ok = (lpos + r <= asm.starts[lt] + asm.lengths[lt]) & (rpos + r <= asm.starts[rt] + asm.lengths[rt])
        lt, rt, lpos, rpos, x, contam = lt[ok], rt[ok], lpos[ok], rpos[ok], x[ok], contam[ok]
        k = lt.shape[0]
        lp = (lpos - asm.starts[lt]).astype(np.int64)
        rp = (rpos - asm.starts[rt]).astype(np.int64)
        # innie: left forward / right reverse; outie: left reverse / right forward
        innie = (spec.orientation == 'fr') ^ contam
        l_rev = ~innie
        r_rev = innie
        first_is_left = rng.random(k) < 0.5
        u = rng.random(k)
        mapq = np.where(u < 0.8, 60, np.where(u < 0.9, 0, rng.integers(1, 60, k))).astype(np.int64)
        same = lt == rt
        tl = np.where(same, x, 0)
        ql = np.full(k, r, dtype=np.int64)
        qr = np.full(k, r, dtype=np.int64)
        sc = rng.random(k) < spec.softclip_frac
        ql[sc] = rng.integers(r // 2, r, int(sc.sum()))
        sc = rng.random(k) < spec.softclip_frac
        qr[sc] = rng.integers(r // 2, r, int(sc.sum()))
        base = FLAG_PAIRED + np.where(same, FLAG_PROPER, 0)
        fl = base + np.where(l_rev, FLAG_REVERSE, 0) + np.where(r_rev, FLAG_MATE_REVERSE, 0) \
            + np.where(first_is_left, FLAG_READ1, FLAG_READ2)
        fr_ = base + np.where(r_rev, FLAG_REVERSE, 0) + np.where(l_rev, FLAG_MATE_REVERSE, 0) \
            + np.where(first_is_left, FLAG_READ2, FLAG_READ1)
        # BWA quirk: read1 flagged unmapped but keeping its placement, on contig-spanning pairs
        fishy = (~same) & (rng.random(k) < spec.fishy_frac * 20)
        l_is_r1 = first_is_left
        fl = np.where(fishy & l_is_r1, fl | FLAG_UNMAPPED, fl)
        fr_ = np.where(fishy & ~l_is_r1, fr_ | FLAG_UNMAPPED, fr_)
        fl = np.where(fishy & ~l_is_r1, fl | FLAG_MATE_UNMAPPED, fl)
        fr_ = np.where(fishy & l_is_r1, fr_ | FLAG_MATE_UNMAPPED, fr_)
        # exact duplicates: repeat a subset of pairs verbatim
        base_n = min(k, int(np.ceil((n_pairs - done) / (1.0 + spec.dup_frac))))
        dup = np.nonzero(rng.random(base_n) < spec.dup_frac)[0]
        sel = np.concatenate((np.arange(base_n), dup))[:n_pairs - done]
        done += sel.shape[0]
        for name, left, right in (('tid', lt, rt), ('mtid', rt, lt), ('pos', lp, rp), ('mpos', rp, lp),
                                  ('tlen', tl, -tl), ('flag', fl, fr_), ('mapq', mapq, mapq),
                                  ('qlen', ql, qr)):
            # final column dtypes right away: the 200 M-pair configs would not fit as int64 intermediates
            cols[name].append(np.concatenate((left[sel], right[sel])).astype(_DTYPES[name]))
    cat = {name: np.concatenate(parts) for name, parts in cols.items()}
    cols.clear()
    order = np.argsort((cat['tid'].astype(np.int64) << 32) | cat['pos'].astype(np.int64), kind='stable')
    for name in list(cat):
        cat[name] = cat[name][order]
    n = cat['tid'].shape[0]
    return RecordBatch(asm.names, asm.lengths.tolist(), rlen=np.full(n, r, dtype=np.int32),
                       alen=cat['qlen'].astype(np.int32), **cat)


def chain_scaffolds(asm, seed, max_run=5, first_scaffold_id=1):
    """Contig table as left by a previous pass (MakeScaffolds.py:362-414): random runs of 1..max_run
    adjacent contigs chained into scaffolds with random per-contig direction, cumulative position
    (+ true gap) and scaffold length.  Returns dict of int arrays indexed by tid."""
    rng = np.random.default_rng(seed)
    nc = asm.nc
    scaf_id = np.zeros(nc, dtype=np.int64)
    position = np.zeros(nc, dtype=np.int64)
    direction = rng.random(nc) < 0.5
    scaf_len = np.zeros(nc, dtype=np.int64)
    i = 0
    sid = first_scaffold_id
    while i < nc:
        run = int(rng.integers(1, max_run + 1))
        j = min(nc, i + run)
        cur = 0
        for c in range(i, j):
            position[c] = cur
            cur += int(asm.lengths[c])
            if c + 1 < j:
                cur += int(asm.gaps[c])
        scaf_id[i:j] = sid
        scaf_len[i:j] = cur
        sid += 1
        i = j
    return dict(scaf_id=scaf_id, scaf_len=scaf_len, position=position, direction=direction,
                next_scaffold_id=sid)


# (contigs, median contig length, candidate pairs, [library specs]) per BASELINE.json config
CONFIGS = {
    'C1': dict(nc=1836, median=1000, pairs=1_000_000, libs=[LibrarySpec('fr', 4000.0, 500.0)]),
    'C2': dict(nc=10_000, median=4000, pairs=10_000_000, libs=[LibrarySpec('fr', 500.0, 50.0)]),
    'C3': dict(nc=100_000, median=8000, pairs=200_000_000,
               libs=[LibrarySpec('rf', 5000.0, 500.0, contam_frac=0.22)]),
    'C4': dict(nc=500_000, median=8000, pairs=1_000_000_000,
               libs=[LibrarySpec('fr', 500.0, 50.0), LibrarySpec('rf', 5000.0, 500.0, contam_frac=0.22)]),
    'C5': dict(nc=2_000_000, median=8000, pairs=4_000_000_000,
               libs=[LibrarySpec('fr', 500.0, 50.0), LibrarySpec('rf', 5000.0, 500.0, contam_frac=0.22),
                     LibrarySpec('rf', 10000.0, 1000.0, contam_frac=0.2)]),
}


def config_seed(name):
    return BASE_SEED + int(name[1:])


def simulate_library_device(asm, spec, n_pairs, seed, device, chunk=32_000_000, order='coordinate', window=None):
    """The generator of ``simulate_library`` written with torch ops, so that the full-size configs (C3: 400 M
    records) are drawn, sorted and left resident on the GPU in seconds instead of minutes of numpy on the host.
    Same model, same record semantics, its own random stream (torch.Generator seeded with ``seed``).
    Returns a dict of device tensors {tid mtid pos mpos tlen: int32, flag qlen: int16 bit patterns, mapq: uint8}
    in (tid, pos) order - or, order='name', the two records of a pair next to each other and the pairs in the order
    they were drawn (a name-sorted BAM).  window=(k, W): the k-th of W equal cuts of the genome - what rank k of a sharded
    run holds of ONE coordinate-sorted file: pairs are drawn around the cut and only the RECORDS that lie in it are kept (a
    pair across a cut leaves one record on each side, each side drawing its own), 2 n_pairs records in all.
    Bench / test scaffolding, not the product."""
    import torch
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    r = int(spec.read_len)
    starts = torch.from_numpy(np.ascontiguousarray(asm.starts, dtype=np.int64)).to(dev)
    lengths = torch.from_numpy(np.ascontiguousarray(asm.lengths, dtype=np.int64)).to(dev)
    names = ('tid', 'mtid', 'pos', 'mpos', 'tlen', 'flag', 'mapq', 'qlen')
    tdt = dict(tid=torch.int32, mtid=torch.int32, pos=torch.int32, mpos=torch.int32, tlen=torch.int32,
               flag=torch.int16, mapq=torch.uint8, qlen=torch.int16)
    parts = {k: [] for k in names}
    done = 0

    def rand(m):
        return torch.rand(m, generator=g, device=dev)

    w_lo, w_hi = 0, int(asm.total)
    if window is not None:
        if order == 'name':
            raise ValueError('a window of the genome is a cut of the coordinate-sorted stream')
        k_w, n_w = int(window[0]), int(window[1])
        w_lo, w_hi = int(asm.total) * k_w // n_w, int(asm.total) * (k_w + 1) // n_w
    pad = int(max(spec.mean + 8 * spec.sd, spec.contam_mean + 8 * spec.contam_sd)) if window is not None else 0
    while done < n_pairs:
        m = int(min(chunk, max(1024, int((n_pairs - done) * 1.3))))
        start = torch.randint(max(0, w_lo - pad), w_hi, (m,), generator=g, device=dev)
        contam = rand(m) < spec.contam_frac
        z = torch.randn(m, generator=g, device=dev, dtype=torch.float64)
        x = torch.where(contam, z * spec.contam_sd + spec.contam_mean, z * spec.sd + spec.mean)
        x = torch.clamp(torch.round(x).to(torch.int64), min=2 * r)
        del z
        lpos = start
        rpos = start + x - r
        lt = torch.searchsorted(starts, lpos, right=True) - 1
        rt = torch.searchsorted(starts, rpos, right=True) - 1
        ok = (lpos + r <= starts[lt] + lengths[lt]) & (rpos + r <= starts[rt] + lengths[rt])
        lt, rt, lpos, rpos, x, contam = lt[ok], rt[ok], lpos[ok], rpos[ok], x[ok], contam[ok]
        k = int(lt.shape[0])
        in_l = (lpos >= w_lo) & (lpos < w_hi)                # which of the pair's two records lie in the window
        in_r = (rpos >= w_lo) & (rpos < w_hi)
        lp = lpos - starts[lt]
        rp = rpos - starts[rt]
        del lpos, rpos, start, ok
        if getattr(spec, 'chimeric_frac', 0.0) > 0:
            other = torch.randint(0, asm.nc, (k,), generator=g, device=dev)
            move = (lt != rt) & (rand(k) < spec.chimeric_frac) & (other != lt) & (rp + r <= lengths[other])
            rt = torch.where(move, other, rt)
            del other, move
        innie = contam.logical_not() if spec.orientation == 'fr' else contam
        l_rev = innie.logical_not()
        r_rev = innie
        first_is_left = rand(k) < 0.5
        u = rand(k)
        mapq = torch.where(u < 0.8, torch.full((k,), 60, device=dev, dtype=torch.int64),
                           torch.where(u < 0.9, torch.zeros(k, device=dev, dtype=torch.int64),
                                       torch.randint(1, 60, (k,), generator=g, device=dev)))
        del u
        same = lt == rt
        tl = torch.where(same, x, torch.zeros_like(x))
        ql = torch.full((k,), r, device=dev, dtype=torch.int64)
        qr = torch.full((k,), r, device=dev, dtype=torch.int64)
        sc = rand(k) < spec.softclip_frac
        ql = torch.where(sc, torch.randint(r // 2, r, (k,), generator=g, device=dev), ql)
        sc = rand(k) < spec.softclip_frac
        qr = torch.where(sc, torch.randint(r // 2, r, (k,), generator=g, device=dev), qr)
        base = FLAG_PAIRED + same.to(torch.int64) * FLAG_PROPER
        fl = base + l_rev.to(torch.int64) * FLAG_REVERSE + r_rev.to(torch.int64) * FLAG_MATE_REVERSE \
            + torch.where(first_is_left, FLAG_READ1, FLAG_READ2)
        fr_ = base + r_rev.to(torch.int64) * FLAG_REVERSE + l_rev.to(torch.int64) * FLAG_MATE_REVERSE \
            + torch.where(first_is_left, FLAG_READ2, FLAG_READ1)
        fishy = same.logical_not() & (rand(k) < spec.fishy_frac * 20)
        l_is_r1 = first_is_left
        fl = torch.where(fishy & l_is_r1, fl | FLAG_UNMAPPED, fl)
        fr_ = torch.where(fishy & ~l_is_r1, fr_ | FLAG_UNMAPPED, fr_)
        fl = torch.where(fishy & ~l_is_r1, fl | FLAG_MATE_UNMAPPED, fl)
        fr_ = torch.where(fishy & l_is_r1, fr_ | FLAG_MATE_UNMAPPED, fr_)
        base_n = min(k, int(np.ceil((n_pairs - done) / (1.0 + spec.dup_frac))))
        dup = torch.nonzero(rand(base_n) < spec.dup_frac).flatten()
        sel = torch.cat((torch.arange(base_n, device=dev), dup))[:n_pairs - done]
        keep = None
        if window is not None:
            keep = torch.cat((in_l[sel], in_r[sel]))
            budget = 2 * (n_pairs - done)                    # records still wanted
            over = torch.cumsum(keep.to(torch.int64), 0) > budget
            keep = keep & ~over
            done += (int(keep.sum().item()) + 1) // 2
        else:
            done += int(sel.shape[0])
        for name, left, right in (('tid', lt, rt), ('mtid', rt, lt), ('pos', lp, rp), ('mpos', rp, lp),
                                  ('tlen', tl, -tl), ('flag', fl, fr_), ('mapq', mapq, mapq), ('qlen', ql, qr)):
            if order == 'name':
                parts[name].append(torch.stack((left[sel], right[sel]), dim=1).reshape(-1).to(tdt[name]))
            elif keep is not None:
                parts[name].append(torch.cat((left[sel], right[sel]))[keep].to(tdt[name]))
            else:
                parts[name].append(torch.cat((left[sel], right[sel])).to(tdt[name]))
        del in_l, in_r
        del lt, rt, lp, rp, tl, fl, fr_, mapq, ql, qr, x, contam, same, sel
    cat = {name: torch.cat(parts[name]) for name in names}
    parts.clear()
    if order == 'name':
        return cat
    key = (cat['tid'].to(torch.int64) << 32) | cat['pos'].to(torch.int64)
    order = torch.sort(key, stable=True)[1]
    del key
    for name in names:
        cat[name] = cat[name][order].contiguous()
    return cat


def device_columns_to_batch(asm, cols, read_len=100):
    """Host RecordBatch of a simulate_library_device result (the oracle's view of the same stream)."""
    h = {k: v.cpu().numpy() for k, v in cols.items()}
    h['flag'] = h['flag'].view(np.uint16)
    h['qlen'] = h['qlen'].view(np.uint16)
    n = h['tid'].shape[0]
    return RecordBatch(asm.names, asm.lengths.tolist(), rlen=np.full(n, read_len, dtype=np.int32),
                       alen=h['qlen'].astype(np.int32), **h)
