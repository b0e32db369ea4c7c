"""Import the reference hot-path modules from /root/reference (this container only).

Nothing under /root/reference is copied: the modules are imported in place, with
  * an empty `pysam` stub and a `mathstats` shim (tests/refharness/stubs),
  * `networkx.Graph` swapped for the 1.x-compatible facade before CreateGraph is imported
    (the reference calls graph.node[...] / graph.edge[...], CreateGraph.py:715,844),
  * bytecode writing disabled (the reference tree is read-only).
The GPU box has no /root/reference; everything here is used only to GENERATE the
fixtures committed under tests/golden/ and by CPU tests that skip when it is absent.

ONE SWITCH for the third-party arithmetic: ``BESST_MATHSTATS_PATH=<dir>`` (a directory that holds the real
``mathstats`` package, e.g. an unpacked mathstats-0.2.6.5 sdist or a site-packages) makes the reference import THAT
package instead of the shim; ``python tests/golden/make_golden.py`` then writes fixtures whose ``gap`` / ``score`` fields
come from the package and tags every document ``"mathstats": "<version>"`` (tests/golden_util.tolerances reads the tag:
gap +-1 bp and a relative score tolerance against package-made fixtures, exact against the restatement's).  Without the
variable nothing changes: documents carry no tag and mean "restatement".
"""
import importlib
import os
import sys

REFERENCE_ROOT = os.environ.get('BESST_REFERENCE_ROOT', '/root/reference')
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'stubs')
_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'BESST', 'CreateGraph.py'))


def mathstats_path():
    """The directory of the real package, or None: the shim (tests/refharness/stubs/mathstats) answers."""
    p = os.environ.get('BESST_MATHSTATS_PATH')
    if not p:
        return None
    p = os.path.abspath(p)
    if not os.path.isfile(os.path.join(p, 'mathstats', '__init__.py')):
        raise RuntimeError('BESST_MATHSTATS_PATH=%s holds no mathstats/__init__.py' % p)
    return p


def mathstats_tag():
    """What the fixtures' gap / score fields were made with: None (the restatement, besst_amd.mathstats_compat, through
    the shim) or the version of the real package imported from BESST_MATHSTATS_PATH."""
    if mathstats_path() is None:
        return None
    import mathstats
    here = os.path.dirname(os.path.abspath(mathstats.__file__))
    if os.path.dirname(here) != mathstats_path():
        raise RuntimeError('mathstats was imported from %s, not from BESST_MATHSTATS_PATH' % here)
    version = getattr(mathstats, '__version__', None)
    if version is None:
        try:
            from importlib import metadata
            version = metadata.version('mathstats')
        except Exception:                                    # noqa: BLE001 - an unpacked sdist has no metadata
            version = os.environ.get('BESST_MATHSTATS_VERSION', 'unknown')
    return str(version)


def load():
    """Return a dict of the reference modules on the hot path."""
    if not available():
        raise RuntimeError('reference checkout not present at %s' % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    # (inserted last = searched first: the real mathstats, when given, shadows the shim; pysam stays the stub)
    for p in (_REPO, REFERENCE_ROOT, _STUBS) + ((mathstats_path(),) if mathstats_path() else ()):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    import networkx
    from besst_amd import nxcompat
    networkx.Graph = nxcompat.Graph
    mods = {}
    for name in ('bam_parser', 'e_nr_links', 'find_bimodality', 'Parameter', 'Contig', 'Scaffold',
                 'libmetrics', 'CreateGraph'):
        mods[name] = importlib.import_module('BESST.' + name)
    return mods
