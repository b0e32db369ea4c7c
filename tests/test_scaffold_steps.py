"""Scaffold-graph linearisation steps 1-4: the CPU restatement against the fixture captured
from the reference's own MakeScaffolds functions, and - where /root/reference is present - against those functions
run live on fresh random graphs."""
import random

import pytest

from oracle import scaffold_oracle as SO
from tests import scaffold_util as SU


@pytest.mark.parametrize('name', SU.case_names())
def test_oracle_matches_reference_fixture(name):
    case = SU.by_name(name)
    n_scaf, _, a, b, score = SU.to_arrays(case)
    res = SO.linearize(n_scaf, a.tolist(), b.tolist(), score.tolist())
    SU.check_result(case, res)


def test_fixture_covers_the_interesting_events():
    cs = SU.cases()
    assert sum(c['cycles_removed'] for c in cs) > 20
    assert sum(len(c['ambivalent']) for c in cs) > 500
    assert any(c['isolated_removed'][0] > 0 for c in cs) and any(c['isolated_removed'][1] > 0 for c in cs)
    assert any(not c['extend_paths'] for c in cs)


def test_oracle_matches_live_reference_on_fresh_graphs():
    from tests.refharness import loader
    if not loader.available():
        pytest.skip('reference checkout not present')
    import importlib
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    gen = importlib.import_module('make_scaffold_golden')
    mods = loader.load()
    ms = importlib.import_module('BESST.MakeScaffolds')
    from besst_amd import nxcompat
    for k in range(6):
        rng = random.Random(777 + k)
        nodes, links = gen.random_case(rng, 150 + 40 * k, 1.0 + 0.4 * k, ('mixed', 'random', 'chains')[k % 3])
        case = gen.run_reference(ms, mods, nxcompat.Graph, nodes, links, list(links), extend_paths=True)
        n_scaf, _, a, b, score = SU.to_arrays(case)
        SU.check_result(case, SO.linearize(n_scaf, a.tolist(), b.tolist(), score.tolist()))
