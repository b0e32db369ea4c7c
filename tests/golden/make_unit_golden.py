#!/usr/bin/env python
"""Unit-level golden vectors from the REAL reference: tests/golden/unit_golden.json.

Run in the build container only (needs /root/reference):    python tests/golden/make_unit_golden.py

  expected_links   e_nr_links.ExpectedLinks over a parameter grid                 (e_nr_links.py:62-93)
  split            find_bimodality.split_distribution on the docstring-style lists
                   (x -> 50x + 200) and on seeded bimodal integer samples           (find_bimodality.py:39-205)
  predicates       is_proper_aligned_unique_innie / outie, is_unique_read_link over every combination of the
                   seven flag bits x tlen sign x mapq around the threshold x same/other contig (bam_parser.py:22-36)
  posdir           PosDirCalculatorPE / MP over all direction combinations, integer and fractional read_len,
                   negative results included (int() truncates toward zero)       (CreateGraph.py:1024-1076)
Only inputs and the reference's outputs are stored; the reference modules are imported in place.
"""
import itertools
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from tests.refharness import loader  # noqa: E402

FLAG_BITS = (0x4, 0x8, 0x10, 0x20, 0x40, 0x80, 0x100)


class Rec(object):
    """The attribute surface of a pysam record that the three predicates read."""

    def __init__(self, flag, tlen, mapq, rname, mrnm):
        self.flag, self.tlen, self.mapq, self.rname, self.mrnm = flag, tlen, mapq, rname, mrnm
        self.is_unmapped = bool(flag & 0x4)
        self.mate_is_unmapped = bool(flag & 0x8)
        self.is_reverse = bool(flag & 0x10)
        self.mate_is_reverse = bool(flag & 0x20)
        self.is_read1 = bool(flag & 0x40)
        self.is_read2 = bool(flag & 0x80)
        self.is_secondary = bool(flag & 0x100)


def main():
    mods = loader.load()
    enl, fb, bp, cg = mods['e_nr_links'], mods['find_bimodality'], mods['bam_parser'], mods['CreateGraph']
    out = {}

    grid = []
    for mean, sd, cov, rl, soft in ((500, 50, 30, 100, 0), (3000, 400, 12.5, 100.38, 5), (350, 60, 80, 75, 0),
                                    (10000, 1000, 4, 150, 10)):
        p = enl.Param(mean, sd, cov, rl, soft)
        for len1, len2, d in itertools.product((600, 3000, 100000), (450, 5000, 100000),
                                               (-250, -100, 0, 100, 350, int(mean), int(mean + 3 * sd))):
            grid.append(dict(mean=mean, sd=sd, cov=cov, read_len=rl, softclipped=soft, len1=len1, len2=len2, d=d,
                             value=enl.ExpectedLinks(len1, len2, d, p)))
    out['expected_links'] = grid

    base_lists = [
        [2, 1, 2, 1, 2, 3, 3, 3, 2, 1, 2, 3, 4, 5, 6, 6, 7, 8, 9, 10, 11, 11, 12, 12, 11, 13, 11, 12, 13, 11, 12, 13,
         1, 2, 3, 2, 1, 3],
        [1, 1, 2, 2, 2, 3, 10, 11, 11, 12, 12, 12, 13],
        [5, 5, 5, 5, 6, 6, 7, 20, 21, 21, 22, 40, 41, 41, 41],
        [1, 2, 3, 4, 5, 6, 7, 8, 9, 10],
        [3, 3, 3, 3, 9, 9, 9, 9, 9, 30],
    ]
    rng = random.Random(7)
    splits = []
    cases = [[50 * x + 200 for x in lst] for lst in base_lists]
    for _ in range(6):
        n1, n2 = rng.randint(30, 400), rng.randint(30, 400)
        cases.append([max(101, int(rng.gauss(350, 40))) for _ in range(n1)] +
                     [max(101, int(rng.gauss(2500, 300))) for _ in range(n2)])
    for lst in cases:
        c1, c2, m1, s1, m2, s2 = fb.split_distribution(list(lst))
        splits.append(dict(values=list(lst), cluster1=[int(x) for x in c1], cluster2=[int(x) for x in c2],
                           mean1=float(m1), stddev1=float(s1), mean2=float(m2), stddev2=float(s2)))
    out['split'] = splits

    thr = 10
    preds = []
    for bits in range(1 << len(FLAG_BITS)):
        flag = sum(b for k, b in enumerate(FLAG_BITS) if bits >> k & 1)
        for tlen, mapq, same in itertools.product((-420, 0, 420), (thr - 1, thr, thr + 1, 0), (True, False)):
            r = Rec(flag, tlen, mapq, 3, 3 if same else 5)
            preds.append([flag, tlen, mapq, int(same),
                          int(bool(bp.is_proper_aligned_unique_innie(r, thr))),
                          int(bool(bp.is_proper_aligned_unique_outie(r, thr))),
                          int(bool(bp.is_unique_read_link(r, thr)))])
    out['predicates'] = dict(mapq_threshold=thr, columns=['flag', 'tlen', 'mapq', 'same_contig', 'innie', 'outie',
                                                            'unique_link'], rows=preds)

    pos = []
    geo = [dict(c1pos=0, rpos=120, s1=5000, c1len=5000, c2pos=0, mpos=4700, s2=6000, c2len=6000),
           dict(c1pos=2500, rpos=30, s1=9000, c1len=3100, c2pos=7000, mpos=15, s2=12000, c2len=800),
           dict(c1pos=8800, rpos=3090, s1=9000, c1len=3100, c2pos=100, mpos=790, s2=900, c2len=800),
           dict(c1pos=10, rpos=3, s1=40, c1len=30, c2pos=5, mpos=700, s2=60, c2len=20)]      # negative observations
    for calc, name in ((cg.PosDirCalculatorPE, 'fr'), (cg.PosDirCalculatorMP, 'rf')):
        for g in geo:
            for cd1, rd, cd2, md in itertools.product((True, False), repeat=4):
                for rl in (100, 100.38, 99.999, 0.5):
                    o1, o2, s1, s2 = calc(cd1, rd, g['c1pos'], g['rpos'], g['s1'], g['c1len'], cd2, md, g['c2pos'],
                                          g['mpos'], g['s2'], g['c2len'], rl)
                    pos.append(dict(orientation=name, cont_dir1=cd1, read_dir=rd, cont_dir2=cd2, mate_dir=md,
                                    read_len=rl, obs1=int(o1), obs2=int(o2), side1=s1, side2=s2, **g))
    out['posdir'] = pos

    path = os.path.join(HERE, 'unit_golden.json')
    with open(path, 'w') as fh:
        json.dump(out, fh, separators=(',', ':'))
    print('wrote', path, {k: (len(v['rows']) if isinstance(v, dict) else len(v)) for k, v in out.items()},
          os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
