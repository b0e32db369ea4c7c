"""Shared by the CPU and GPU tests of the chain-extraction step: the fixture captured from the reference's
NewContigsScaffolds (tests/golden/scaffold_chains.json.gz, tests/golden/make_chain_golden.py)."""
import gzip
import json
import os

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'scaffold_chains.json.gz')
_cases = None


def cases():
    global _cases
    if _cases is None:
        with gzip.open(FIXTURE, 'rt') as fh:
            _cases = json.load(fh)['cases']
    return _cases


def case_names():
    return [c['name'] for c in cases()]


def by_name(name):
    return next(c for c in cases() if c['name'] == name)


class Param(object):
    def __init__(self, case):
        self.extend_paths = case['extend_paths']
        self.plots = False
        self.mean_ins_size, self.std_dev_ins_size, self.read_len = case['mean'], case['sd'], case['read_len']
        self.lognormal = False
        self.scaffold_indexer = (max(int(s) for s in case['scaffolds']) + 5) if case['scaffolds'] else 5
        self.gap_estimations = []
