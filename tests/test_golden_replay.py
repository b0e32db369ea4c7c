"""The committed golden fixtures are what the REAL reference produces on the committed inputs.

Build container only (needs /root/reference; skipped elsewhere): every end-to-end scenario's stored inputs - record
stream, header, parameter overrides, FASTA names, object layout - go through the reference's own
libmetrics.get_metrics (BESST/libmetrics.py:226) and CreateGraph.PE (BESST/CreateGraph.py:45) again, and the result
must equal the stored document field by field: library metrics incl. empirical_distribution, the raw edge tables
right after the record loop, counters, coverage, fishy table, the final graphs in iteration order, object dicts and
param fields.  This is what pins oracle/py_oracle.py (tests/test_oracle_golden.py compares it with the same files), and
it is also the statement that `python tests/golden/make_golden.py` leaves `git diff tests/golden` empty.
"""
import importlib.util
import os

import pytest

from tests import golden_util as GU
from tests.refharness import loader

pytestmark = pytest.mark.skipif(not loader.available(), reason='reference checkout not present')

_HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def maker():
    spec = importlib.util.spec_from_file_location('make_golden', os.path.join(_HERE, 'golden', 'make_golden.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, loader.load()


@pytest.mark.parametrize('name', GU.scenario_names())
def test_reference_reproduces_committed_fixture(maker, name):
    mod, mods = maker
    stored, fresh = mod.replay(mods, name)
    assert set(stored) == set(fresh)
    for field in ('references', 'lengths', 'fasta_names', 'overrides', 'layout', 'layout_threshold', 'stream'):
        assert stored[field] == fresh[field], field
    assert stored['metrics'] == fresh['metrics']
    for part in ('counter', 'fishy', 'fishy_reads', 'aligned', 'G', 'G_prime'):
        assert stored['after_loop'][part] == fresh['after_loop'][part], part
    for part in stored['final']:
        assert stored['final'][part] == fresh['final'][part], part
    assert stored == fresh


def test_scenarios_cover_the_committed_streams():
    """Every committed stream is the input of at least one scenario (none is dead weight, none is missing)."""
    import glob
    streams = {os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GU.GOLDEN_DIR, 'stream_*.npz'))}
    used = {GU.load(n)[0]['stream'] for n in GU.scenario_names()}
    assert streams == used
