"""BESST_MATHSTATS_PATH: the one switch that re-pins the third-party arithmetic (tests/refharness/loader.py).

mathstats 0.2.6.5 (requirements.txt:3; call sites BESST/CreateGraph.py:34-37,526,537,555, BESST/libmetrics.py:14,23) is not
in this image, so the switch is exercised with a stand-in: a directory holding a package called `mathstats` that answers
like the restatement but moves every ML gap by one base pair and carries a version.  What must hold then: the reference
imports THAT package (not the shim), the documents make_golden writes are tagged with its version and differ in `gap`,
and the comparison helpers of the tests switch from equality to the tolerances of tests/golden_util.tolerances.
Build container only (needs /root/reference)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

from tests import golden_util as GU
from tests.refharness import loader

pytestmark = pytest.mark.skipif(not loader.available(), reason='reference checkout not present')

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_stand_in(root):
    pkg = os.path.join(root, 'mathstats')
    for sub in ('', 'normaldist', os.path.join('normaldist', 'truncatedskewed')):
        os.makedirs(os.path.join(pkg, sub), exist_ok=True)
    open(os.path.join(pkg, '__init__.py'), 'w').write("__version__ = '0.2.6.5-standin'\n")
    open(os.path.join(pkg, 'normaldist', '__init__.py'), 'w').write('')
    open(os.path.join(pkg, 'normaldist', 'truncatedskewed', '__init__.py'), 'w').write('')
    open(os.path.join(pkg, 'normaldist', 'normal.py'), 'w').write('from besst_amd.mathstats_compat import MaxObsDistr\n')
    open(os.path.join(pkg, 'normaldist', 'truncatedskewed', 'param_est.py'), 'w').write(textwrap.dedent('''
        from besst_amd import mathstats_compat as MC
        tr_sk_std_dev = MC.tr_sk_std_dev
        def GapEstimator(mean, sigma, read_length, mean_obs, c1_len, c2_len=None):
            return MC.GapEstimator(mean, sigma, read_length, mean_obs, c1_len, c2_len) + 1
    '''))
    open(os.path.join(pkg, 'log_normal_param_est.py'), 'w').write(textwrap.dedent('''
        from besst_amd.mathstats_compat import lognormal_GapEstimator
        def GapEstimator(mu, sigma, read_len, samples, c1_len, c2_len=None):
            return lognormal_GapEstimator(mu, sigma, read_len, samples, c1_len, c2_len) + 1
    '''))


SCRIPT = '''
import importlib.util, json, os, sys
sys.path.insert(0, %(repo)r)
from tests.refharness import loader
mods = loader.load()
import mathstats
spec = importlib.util.spec_from_file_location('make_golden', os.path.join(%(repo)r, 'tests', 'golden', 'make_golden.py'))
mk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mk)
stored, fresh = mk.replay(mods, 'fr_given')
print(json.dumps(dict(file=mathstats.__file__, tag=loader.mathstats_tag(), stored_tag=stored.get('mathstats'),
                      fresh_tag=fresh.get('mathstats'), stored_G=stored['final']['G'], fresh_G=fresh['final']['G'],
                      same_structure=stored['after_loop'] == fresh['after_loop'] and stored['metrics'] == fresh['metrics'])))
'''


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, '-c', SCRIPT % dict(repo=REPO)], cwd=REPO, env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_without_the_switch_the_shim_answers_and_documents_carry_no_tag():
    env = {k: v for k, v in os.environ.items() if k != 'BESST_MATHSTATS_PATH'}
    out = subprocess.run([sys.executable, '-c', SCRIPT % dict(repo=REPO)], cwd=REPO, env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert os.path.join('refharness', 'stubs', 'mathstats') in d['file']
    assert d['tag'] is None and d['fresh_tag'] is None and d['stored_tag'] is None
    assert d['fresh_G'] == d['stored_G']


def test_the_switch_imports_the_given_package_and_tags_the_documents(tmp_path):
    make_stand_in(str(tmp_path))
    d = _run({'BESST_MATHSTATS_PATH': str(tmp_path)})
    assert d['file'].startswith(str(tmp_path))
    assert d['tag'] == d['fresh_tag'] == '0.2.6.5-standin' and d['stored_tag'] is None
    assert d['same_structure']                               # everything the package does not touch is unchanged
    moved = [(a['gap'], b['gap']) for a, b in zip(d['stored_G'], d['fresh_G']) if 'gap' in a and a['gap'] != b['gap']]
    assert moved and all(b == a + 1 for a, b in moved)       # the stand-in's gaps, not the shim's
    # the comparison the tests make: equality against the restatement's fixtures, tolerances against tagged ones
    doc, _ = GU.load('fr_given')
    assert GU.tolerances(doc)['exact'] is True
    tagged = dict(doc, mathstats=d['fresh_tag'])
    assert GU.tolerances(tagged) == dict(source='mathstats 0.2.6.5-standin', exact=False, gap=1, score=2e-2)
    with pytest.raises(AssertionError):
        GU.assert_scored_rows(d['stored_G'], d['fresh_G'], doc)          # restatement-made: a moved gap is a failure
    GU.assert_scored_rows(d['stored_G'], d['fresh_G'], tagged)           # package-made: +-1 bp is the contract


def test_a_directory_without_the_package_is_refused(tmp_path):
    out = subprocess.run([sys.executable, '-c', SCRIPT % dict(repo=REPO)], cwd=REPO,
                         env=dict(os.environ, BESST_MATHSTATS_PATH=str(tmp_path)), capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and 'holds no mathstats/__init__.py' in out.stderr
