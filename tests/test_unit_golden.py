"""Unit-level golden vectors captured from the real reference (tests/golden/make_unit_golden.py):
the host mirrors and the oracles on CPU, the device predicates and PosDir arithmetic on the GPU."""
import json
import os

import numpy as np
import pytest

from besst_amd import bam_parser, e_nr_links, find_bimodality
from besst_amd.records import RecordBatch
from oracle import py_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, 'golden', 'unit_golden.json')) as _fh:
    GOLD = json.load(_fh)


def test_expected_links_grid():
    for g in GOLD['expected_links']:
        p = e_nr_links.Param(g['mean'], g['sd'], g['cov'], g['read_len'], g['softclipped'])
        assert e_nr_links.ExpectedLinks(g['len1'], g['len2'], g['d'], p) == g['value'], g
        assert O.expected_links(g['len1'], g['len2'], g['d'], g['mean'], g['sd'], g['cov'], g['read_len'],
                                g['softclipped']) == g['value'], g


def test_split_distribution_cases():
    for g in GOLD['split']:
        c1, c2, m1, s1, m2, s2 = find_bimodality.split_distribution(list(g['values']))
        assert [int(x) for x in c1] == g['cluster1'] and [int(x) for x in c2] == g['cluster2']
        assert (float(m1), float(s1), float(m2), float(s2)) == (g['mean1'], g['stddev1'], g['mean2'], g['stddev2'])


class _Rec(object):
    def __init__(self, flag, tlen, mapq, rname, mrnm):
        self.tlen, self.mapq, self.rname, self.mrnm = tlen, mapq, rname, mrnm
        self.is_unmapped, self.mate_is_unmapped = bool(flag & 0x4), bool(flag & 0x8)
        self.is_reverse, self.mate_is_reverse = bool(flag & 0x10), bool(flag & 0x20)
        self.is_read1, self.is_read2, self.is_secondary = bool(flag & 0x40), bool(flag & 0x80), bool(flag & 0x100)


def _predicate_columns():
    rows = np.asarray(GOLD['predicates']['rows'], dtype=np.int64)
    flag, tlen, mapq, same = rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3]
    tid = np.full(len(rows), 3, np.int32)
    mtid = np.where(same == 1, 3, 5).astype(np.int32)
    return rows, tid, mtid, tlen.astype(np.int32), flag.astype(np.uint16), mapq.astype(np.uint8)


def test_predicate_truth_table_host():
    thr = GOLD['predicates']['mapq_threshold']
    rows, tid, mtid, tlen, flag, mapq = _predicate_columns()
    for r in rows:                                       # the oracle's scalar predicates
        assert (int(O.is_innie(int(r[0]), int(r[1]), 3, 3 if r[3] else 5, int(r[2]), thr)),
                int(O.is_outie(int(r[0]), int(r[1]), 3, 3 if r[3] else 5, int(r[2]), thr))) == (r[4], r[5])
    for r in rows[::7]:                                  # the object forms (every 7th row keeps the test quick)
        rec = _Rec(int(r[0]), int(r[1]), int(r[2]), 3, 3 if r[3] else 5)
        assert (int(bool(bam_parser.is_proper_aligned_unique_innie(rec, thr))),
                int(bool(bam_parser.is_proper_aligned_unique_outie(rec, thr))),
                int(bool(bam_parser.is_unique_read_link(rec, thr)))) == (r[4], r[5], r[6])
    assert np.array_equal(bam_parser.innie_mask(tid, mtid, tlen, flag, mapq, thr).astype(np.int64), rows[:, 4])
    assert np.array_equal(bam_parser.outie_mask(tid, mtid, tlen, flag, mapq, thr).astype(np.int64), rows[:, 5])
    assert np.array_equal(bam_parser.unique_read_link_mask(tid, mtid, flag, mapq, thr).astype(np.int64), rows[:, 6])


def test_posdir_enumeration_oracle():
    for g in GOLD['posdir']:
        o1, s1 = O.posdir(g['orientation'], g['cont_dir1'], g['read_dir'], g['c1pos'], g['rpos'], g['s1'], g['c1len'],
                          g['read_len'])
        o2, s2 = O.posdir(g['orientation'], g['cont_dir2'], g['mate_dir'], g['c2pos'], g['mpos'], g['s2'], g['c2len'],
                          g['read_len'])
        assert (o1, o2, 'R' if s1 else 'L', 'R' if s2 else 'L') == (g['obs1'], g['obs2'], g['side1'], g['side2']), g


@pytest.mark.gpu
@pytest.mark.parametrize('orientation', ['fr', 'rf'])
def test_predicate_truth_table_device(orientation):
    """The metrics kernels' predicates on the reference's truth table: the ordered insert-size sample must be |tlen|
    of exactly the rows the reference calls innie (fr) / outie (rf), the contamination sample the opposite class."""
    from besst_amd import device
    thr = GOLD['predicates']['mapq_threshold']
    rows, tid, mtid, tlen, flag, mapq = _predicate_columns()
    n = len(rows)
    z = np.zeros(n, np.int32)
    batch = RecordBatch(['c%d' % i for i in range(8)], [5000] * 8, tid=tid, mtid=mtid, pos=z, mpos=z, tlen=tlen,
                        flag=flag, mapq=mapq, qlen=np.full(n, 100, np.uint16), rlen=None, alen=None)
    read_len = 100.0
    with device.GraphContext(0) as ctx:
        ctx.set_contigs(scaf_id=np.arange(1, 9, dtype=np.int32), scaf_len=np.full(8, 5000, np.int32),
                        ctg_pos=np.zeros(8, np.int32), ctg_len=np.full(8, 5000, np.int32),
                        direction=np.ones(8, np.uint8), cls=np.ones(8, np.uint8))
        ctx.push_records(batch)
        isize, contam, counts = ctx.metrics_sample(np.ones(8, np.uint8), orientation, thr, read_len, True)
    innie, outie = rows[:, 4] == 1, rows[:, 5] == 1
    at = np.abs(rows[:, 1])
    want_isize = at[outie if orientation == 'rf' else innie]
    if orientation == 'rf':
        want_contam = at[innie & (read_len < at)]
    else:
        want_contam = at[outie & (read_len < at + 2 * read_len)]
    assert np.array_equal(np.asarray(isize, np.int64), want_isize)
    assert np.array_equal(np.asarray(contam, np.int64), want_contam)
    assert counts.counter_total == int(((rows[:, 0] & 0x4) == 0).sum())


@pytest.mark.gpu
@pytest.mark.parametrize('orientation,read_len', [('fr', 100), ('fr', 100.38), ('fr', 99.999), ('rf', 100.38),
                                                  ('rf', 0.5), ('rf', 100)])
def test_posdir_enumeration_device(orientation, read_len):
    """Every PosDir branch on the device: one read-2 record per fixture case, each between its own pair of
    single-contig scaffolds, so that the tuple's key identifies the case.  Cases with an observation
    <= 25 are not accepted by CreateEdge (:840) and must show up in reads_with_too_long_insert instead."""
    from tests import gpu_util as DU
    cases = [g for g in GOLD['posdir'] if g['orientation'] == orientation and g['read_len'] == read_len]
    assert len(cases) == 64
    cls, scaf, slen, cpos, clen, cdir = [], [], [], [], [], []
    tid, mtid, pos, mpos, flag = [], [], [], [], []
    for i, g in enumerate(cases):
        for which in (1, 2):
            cls.append(1)
            scaf.append(2 * i + which)
            slen.append(g['s%d' % which])
            cpos.append(g['c%dpos' % which])
            clen.append(g['c%dlen' % which])
            cdir.append(bool(g['cont_dir%d' % which]))
        tid.append(2 * i)
        mtid.append(2 * i + 1)
        pos.append(g['rpos'])
        mpos.append(g['mpos'])
        flag.append(0x80 | (0 if g['read_dir'] else 0x10) | (0 if g['mate_dir'] else 0x20))
    n = len(cases)
    batch = RecordBatch(['c%d' % i for i in range(2 * n)], clen, tid=np.asarray(tid, np.int32),
                        mtid=np.asarray(mtid, np.int32), pos=np.asarray(pos, np.int32), mpos=np.asarray(mpos, np.int32),
                        tlen=np.zeros(n, np.int32), flag=np.asarray(flag, np.uint16), mapq=np.full(n, 60, np.uint8),
                        qlen=np.full(n, 100, np.uint16), rlen=None, alen=None)
    tab = dict(cls=cls, scaf=scaf, slen=slen, cpos=cpos, clen=clen, cdir=cdir)
    p = O.LibParams(orientation=orientation, read_len=read_len, ins_size_threshold=1.0e9 / 2, extend_paths=False)
    table, aligned, ctr = DU.device_build(batch, tab, p)
    accepted = {}
    for i, g in enumerate(cases):
        if g['obs1'] > 25 and g['obs2'] > 25:
            u = (2 * i + 1) * 2 + (1 if g['side1'] == 'R' else 0)
            v = (2 * i + 2) * 2 + (1 if g['side2'] == 'R' else 0)
            accepted[(u, v)] = (g['obs1'], g['obs2'])
    assert ctr.count == len(accepted) and ctr.reads_with_too_long_insert == n - len(accepted)
    got = {}
    for r in range(len(table)):
        assert int(table.n[r]) == 1
        off = int(table.offset[r])
        got[(int(table.u[r]), int(table.v[r]))] = (int(table.obs_lo[off]), int(table.obs_hi[off]))
    assert got == accepted
