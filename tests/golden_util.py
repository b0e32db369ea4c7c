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


def rec_lists(batch):
    rec = {c: getattr(batch, c).tolist() for c in COLS}
    return rec
