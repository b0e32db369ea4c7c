"""Import the reference hot-path modules from /root/reference (this container only).

Nothing under /root/reference is copied: the modules are imported in place, with
  * an empty `pysam` stub and a `mathstats` shim (tests/refharness/stubs),
  * `networkx.Graph` swapped for the 1.x-compatible facade before CreateGraph is imported
    (the reference calls graph.node[...] / graph.edge[...], CreateGraph.py:715,844),
  * bytecode writing disabled (the reference tree is read-only).
The GPU box has no /root/reference; everything here is used only to GENERATE the
fixtures committed under tests/golden/ and by CPU tests that skip when it is absent.
"""
import importlib
import os
import sys

REFERENCE_ROOT = os.environ.get('BESST_REFERENCE_ROOT', '/root/reference')
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'stubs')
_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'BESST', 'CreateGraph.py'))


def load():
    """Return a dict of the reference modules on the hot path."""
    if not available():
        raise RuntimeError('reference checkout not present at %s' % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    for p in (_REPO, REFERENCE_ROOT, _STUBS):
        if p not in sys.path:
            sys.path.insert(0, p)
    import networkx
    from besst_amd import nxcompat
    networkx.Graph = nxcompat.Graph
    mods = {}
    for name in ('bam_parser', 'e_nr_links', 'find_bimodality', 'Parameter', 'Contig', 'Scaffold',
                 'libmetrics', 'CreateGraph'):
        mods[name] = importlib.import_module('BESST.' + name)
    return mods
