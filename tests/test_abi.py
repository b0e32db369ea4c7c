"""CPU checks of the C-ABI boundary: the library loads and exports every symbol the header declares."""
import os
import re

from besst_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(REPO, 'include', 'besst_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(besst_(?:abi|last|release|device|prof|ctx|dev|owner|bam|bgzf|host)_\w+|besst_linearize|besst_score_paths|besst_chain_scaffolds)\s*\(', text)))


def test_header_symbols_exported_and_bound():
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), name
    assert names == _lib.exported_symbols()
    assert lib.besst_abi_version() == 3


def test_argument_errors_do_not_need_a_gpu():
    lib = _lib.load()
    assert lib.besst_ctx_set_library(None, None) == 1
    assert 'null' in _lib.last_error()
    assert lib.besst_dev_classify_workspace_bytes(4096) > 2 * 4096 * 8
