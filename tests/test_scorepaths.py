"""ScorePaths: the CPU restatement against the fixture captured from the reference."""
import pytest

from oracle import scorepaths_oracle as PO
from tests import scorepaths_util as PU


@pytest.mark.parametrize('name', PU.case_names())
def test_oracle_matches_reference_fixture(name):
    case = PU.by_name(name)
    _, row_ptr, col, weight, path_ptr, path_nodes = PU.to_arrays(case)
    got = PO.score_paths(row_ptr.tolist(), col.tolist(), weight.tolist(), path_ptr.tolist(), path_nodes.tolist(),
                         bool(case['contamination_ratio']), case['no_score'], case['score_cutoff'])
    assert got == case['all_paths']


def test_fixture_has_long_and_degenerate_paths():
    cs = PU.cases()
    assert max(len(p) for c in cs for p in c['paths']) >= 200
    assert any(len(p) == 1 for c in cs for p in c['paths'])
    assert any(c['contamination_ratio'] for c in cs) and any(c['no_score'] for c in cs)
    assert sum(c['n_dfs_paths'] for c in cs) > 1000
