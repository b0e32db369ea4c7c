"""Load the golden scenarios captured from the reference (tests/golden/*.json + stream_*.npz)."""
import glob
import json
import os

import numpy as np

from besst_amd.records import RecordBatch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
COLS = ('tid', 'mtid', 'pos', 'mpos', 'tlen', 'flag', 'mapq', 'qlen', 'rlen', 'alen')


def scenario_names():
    """The end-to-end scenarios (unit_golden.json holds the unit-level vectors of tests/test_unit_golden.py)."""
    names = (os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, '*.json')))
    return sorted(n for n in names if not n.startswith('unit_'))


_streams = {}


def load(name):
    with open(os.path.join(GOLDEN_DIR, name + '.json')) as fh:
        doc = json.load(fh)
    key = doc['stream']
    if key not in _streams:
        z = np.load(os.path.join(GOLDEN_DIR, key + '.npz'))
        _streams[key] = {c: z[c] for c in COLS}
    cols = _streams[key]
    batch = RecordBatch(doc['references'], doc['lengths'], **cols)
    return doc, batch


def tolerances(doc):
    """How a scenario's `gap` / `score` fields are compared.  Documents made through the shim (no "mathstats" tag) hold the
    restatement's own numbers: the device must give the gap within +-1 bp and, where the gap agrees, the score within
    1e-9 (`exact` - host paths that run the restatement itself reproduce them bit for bit).  Documents made with the real
    package (BESST_MATHSTATS_PATH, tag = its version) pin the third-party arithmetic: gap +-1 bp (BASELINE.json's
    tolerance), score within 2 % - the expected sigma of the restatement is model-pinned to 0.5 % (oracle/gapest_numeric.py)
    and enters the score as a ratio; an edge whose two sub-scores sit at the 0.5 threshold may flip to 0 and is counted."""
    if doc.get('mathstats') is None:
        return dict(source='restatement', exact=True, gap=1, score=1e-9)
    return dict(source='mathstats ' + str(doc['mathstats']), exact=False, gap=1, score=2e-2)


def assert_scored_rows(got, want, doc, what=''):
    """Edge rows incl. `gap` and `score` against a scenario's stored rows, by the scenario's tolerances: equality against
    the restatement's own fixtures, structure exactly + gap / score within tolerance against package-made ones."""
    tol = tolerances(doc)
    if tol['exact']:
        assert got == want, what
        return
    assert len(got) == len(want), what
    flipped = 0
    for g, w in zip(got, want):
        assert {k: v for k, v in g.items() if k not in ('gap', 'score')} == \
            {k: v for k, v in w.items() if k not in ('gap', 'score')}, what
        if 'gap' in w:
            assert abs(g['gap'] - w['gap']) <= tol['gap'], (what, g, w)
            if (g['score'] == 0) != (w['score'] == 0):
                flipped += 1
            else:
                assert abs(g['score'] - w['score']) <= tol['score'] * max(1.0, abs(w['score'])), (what, g, w)
    assert flipped <= max(1, len(want) // 50), (what, flipped)


def rec_lists(batch):
    rec = {c: getattr(batch, c).tolist() for c in COLS}
    return rec
