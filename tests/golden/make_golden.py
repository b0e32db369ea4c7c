#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py                # replay: the committed inputs through the reference again
    python tests/golden/make_golden.py --resimulate   # draw NEW input streams from besst_amd.synth (changes the fixtures)

The default mode reads every committed scenario's inputs - the record stream (stream_*.npz), header, parameter overrides,
FASTA names, object layout, all stored in the scenario's JSON - runs the reference on them and writes the JSON again:
with an unchanged reference and harness `git diff tests/golden` stays empty, whatever has happened to the synthetic
generator since the streams were drawn (tests/test_golden_replay.py asserts exactly that).  --resimulate is how the
streams were made in the first place; it re-pins every oracle to new data and is only for adding scenarios.

The reference modules are imported in place through tests/refharness (no reference source is
copied).  Each scenario stores its INPUTS (record columns, header, parameters, initial object
state) and the reference's OUTPUTS (library metrics, the raw edge tables right after the record
loop, counters, coverage, and the final filtered/scored graphs).  `gap`/`score` values involve the
mathstats shim (tests/refharness/stubs/mathstats -> besst_amd.mathstats_compat) and therefore pin
plumbing only; every other number is produced by reference code alone.

    BESST_MATHSTATS_PATH=/path/to/dir-holding-mathstats python tests/golden/make_golden.py

re-pins those two fields against the real package the day it is at hand (tests/refharness/loader.py): the documents
are then tagged "mathstats": "<version>" and the tests compare with the tolerances of tests/golden_util.tolerances.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from besst_amd import synth  # noqa: E402
from besst_amd.records import RecordBatch  # noqa: E402
from tests.refharness import driver, loader  # noqa: E402

COLS = ('tid', 'mtid', 'pos', 'mpos', 'tlen', 'flag', 'mapq', 'qlen', 'rlen', 'alen')


def save_batch(name, batch):
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **{c: getattr(batch, c) for c in COLS})


def jsonable(o):
    if isinstance(o, dict):
        return {str(k): jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [jsonable(v) for v in o]
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.floating,)):
        return float(o)
    if isinstance(o, np.bool_):
        return bool(o)
    return o


def run_reference(mods, name, stream, batch, overrides, fasta_names=None, layout=None, layout_threshold=None):
    """One scenario through the reference's get_metrics + PE -> the document that is stored (as jsonable objects)."""
    import builtins
    cg = mods['CreateGraph']
    patched = name == 'fr_lognormal'          # see lognormal_scenario: Python 2's integer `range` arguments (:490)
    if patched:
        cg.range = lambda *a: builtins.range(*[int(v) for v in a])
    try:
        param = driver.make_param(mods, **overrides)
        metrics = driver.run_get_metrics(mods, batch, param)
        state = None
        if layout is not None:
            thr = layout_threshold if layout_threshold is not None else param.contig_threshold
            state = driver.build_state(mods, layout, batch.references, batch.lengths, thr)
            param.scaffold_indexer = int(layout['next_scaffold_id'])
            param.tot_assembly_length = int(sum(batch.lengths))
        fasta = list(batch.references) if fasta_names is None else list(fasta_names)
        snap, final, _ = driver.run_pe(mods, batch, param, fasta, state)
    finally:
        if patched:
            del cg.range
    doc = dict(name=name, stream=stream, references=list(batch.references), lengths=list(batch.lengths),
               fasta_names=fasta, overrides=overrides, metrics=metrics, after_loop=snap, final=final,
               layout=None if layout is None else {k: np.asarray(v).tolist() for k, v in layout.items()},
               layout_threshold=layout_threshold)
    if loader.mathstats_tag() is not None:                   # BESST_MATHSTATS_PATH: gap / score come from the real package
        doc['mathstats'] = loader.mathstats_tag()
    return jsonable(doc)


def replay(mods, name):
    """The committed scenario `name` - its stored inputs - through the reference again -> (stored doc, fresh doc)."""
    from tests import golden_util as GU
    doc, batch = GU.load(name)
    fresh = run_reference(mods, doc['name'], doc['stream'], batch, doc['overrides'], doc['fasta_names'], doc['layout'],
                          doc['layout_threshold'])
    return doc, json.loads(json.dumps(fresh))            # (what a reader of the file would get: tuples as lists, ...)


def write_doc(doc):
    with open(os.path.join(HERE, doc['name'] + '.json'), 'w') as fh:
        json.dump(doc, fh)
    snap, final = doc['after_loop'], doc['final']
    print('%-26s G=%d Gp=%d counters=%s' % (doc['name'], len(final['G']), len(final['G_prime']), snap and snap['counter']))


def scenario(mods, name, stream, batch, overrides, fasta_names=None, layout=None, layout_threshold=None):
    write_doc(run_reference(mods, name, stream, batch, overrides, fasta_names, layout, layout_threshold))


def edgecase_stream(asm, spec, seed):
    """Stream with unknown tids, a zero-length header entry, a repeat and an unsampled contig."""
    b = synth.simulate_library(asm, spec, 14000, seed)
    rng = np.random.default_rng(seed + 1)
    cols = {c: getattr(b, c).copy() for c in COLS}
    n = len(b)
    # unplaced reads (tid -1) and an out-of-range mate id
    idx = rng.choice(n, 60, replace=False)
    cols['tid'][idx[:30]] = -1
    cols['mtid'][idx[30:]] = len(b.references) + 3
    # mapq exactly at / just below the default --min_mapq 11
    idx = rng.choice(n, 400, replace=False)
    cols['mapq'][idx[:200]] = 10
    cols['mapq'][idx[200:]] = 11
    # secondary alignments and mate-unmapped flags sprinkled in
    idx = rng.choice(n, 200, replace=False)
    cols['flag'][idx[:100]] |= 0x100
    cols['flag'][idx[100:]] |= 0x8
    # a repeat: pile extra multi-mapping (mapq 0) reads on one long contig
    big = int(np.argsort(asm.lengths)[-5])
    extra = 4000
    add = dict(tid=np.full(extra, big), mtid=np.full(extra, big),
               pos=rng.integers(0, asm.lengths[big] - 100, extra), mpos=rng.integers(0, asm.lengths[big] - 100, extra),
               tlen=np.zeros(extra, int), flag=np.full(extra, 0x1 | 0x40), mapq=np.zeros(extra, int),
               qlen=np.full(extra, 100), rlen=np.full(extra, 100), alen=np.full(extra, 100))
    # drop every record touching one mid-sized contig so its coverage is 0 (< -z_min)
    quiet = int(np.argsort(asm.lengths)[len(asm.lengths) // 2])
    keep = (cols['tid'] != quiet) & (cols['mtid'] != quiet)
    cat = {c: np.concatenate((cols[c][keep], add[c])) for c in COLS}
    order = np.argsort((cat['tid'].astype(np.int64) << 32) | (cat['pos'].astype(np.int64) & 0xffffffff), kind='stable')
    cat = {c: v[order] for c, v in cat.items()}
    refs = list(b.references) + ['zero_len_ctg']
    lens = list(b.lengths) + [0]
    return RecordBatch(refs, lens, **cat)


def lognormal_scenario(mods):
    """A skewed library (insert sizes exp(N(ln 1500, 0.35))): libmetrics sets param.lognormal and GiveScoreOnEdges takes
    its log-normal branch (CreateGraph.py:485-531).  That branch is Python 2 code - `range(0, ..., max_isize/50)` raises
    TypeError on Python 3 (:490) - so the reference module gets ONE patch for this scenario: a `range` that truncates
    its arguments to int, i.e. Python 2's integer division.  lnpe.GapEstimator is the shim's restatement
    (tests/refharness/stubs/mathstats/log_normal_param_est.py): like the normal gap it pins plumbing only."""
    asm = synth.make_assembly(300, 6000, 501)
    ln = synth.simulate_library(asm, synth.LibrarySpec('fr', 1500.0, 150.0, lognormal_sigma=0.35), 30000, 502)
    save_batch('stream_ln', ln)
    scenario(mods, 'fr_lognormal', 'stream_ln', ln, {})          # (run_reference applies the `range` patch by name)


def dense_stream(asm, spec, seed, hubs=12):
    """A PE stream plus a tangle: chimeric pairs, one to four per edge, between the ends of the `hubs` longest contigs, so
    that most hub ends have a dozen thin link edges - the case remove_edges_below_threshold (CreateGraph.py:355-404)
    exists for.  Which of them survive depends on the order G_prime.edges() lists them in: an edge goes only while both
    of its ends still have more than four neighbours."""
    from besst_amd.records import FLAG_MATE_REVERSE, FLAG_PAIRED, FLAG_READ1, FLAG_READ2, FLAG_REVERSE
    b = synth.simulate_library(asm, spec, 12000, seed)
    rng = np.random.default_rng(seed + 1)
    hub = np.argsort(asm.lengths)[-hubs:]
    ends = [(int(c), side) for c in hub for side in (0, 1)]              # side 1 = 'R': forward read near the right end
    add = {c: [] for c in COLS}

    def place(c, side):
        jitter = int(rng.integers(30, 200))
        return (int(asm.lengths[c]) - jitter - 100, 0) if side else (jitter, FLAG_REVERSE)

    for i in range(len(ends)):
        for j in range(i + 1, len(ends)):
            (ca, sa), (cb, sb) = ends[i], ends[j]
            if ca == cb or rng.random() < 0.45:
                continue
            for _ in range(int(rng.integers(1, 5))):
                (pa, ra), (pb, rb) = place(ca, sa), place(cb, sb)
                first_a = rng.random() < 0.5
                fa = FLAG_PAIRED + ra + (FLAG_MATE_REVERSE if rb else 0) + (FLAG_READ1 if first_a else FLAG_READ2)
                fb = FLAG_PAIRED + rb + (FLAG_MATE_REVERSE if ra else 0) + (FLAG_READ2 if first_a else FLAG_READ1)
                for tid, mtid, pos, mpos, flag in ((ca, cb, pa, pb, fa), (cb, ca, pb, pa, fb)):
                    for c, v in (('tid', tid), ('mtid', mtid), ('pos', pos), ('mpos', mpos), ('tlen', 0), ('flag', flag),
                                 ('mapq', 60), ('qlen', 100), ('rlen', 100), ('alen', 100)):
                        add[c].append(v)
    cat = {c: np.concatenate((getattr(b, c), np.asarray(add[c], dtype=getattr(b, c).dtype))) for c in COLS}
    order = np.argsort((cat['tid'].astype(np.int64) << 32) | cat['pos'].astype(np.int64), kind='stable')
    return RecordBatch(list(b.references), list(b.lengths), **{c: v[order] for c, v in cat.items()})


def dense_scenario(mods):
    """Hub contigs whose ends are tangled by chimeric pairs: the dense-region pruning and the link-count filters of
    CreateGraph.py:323-404 all remove edges here (in the other scenarios the pruning finds nothing to do)."""
    asm = synth.make_assembly(220, 2500, 601)
    dn = dense_stream(asm, synth.LibrarySpec('fr', 500.0, 50.0), 602)
    save_batch('stream_dense', dn)
    scenario(mods, 'fr_dense', 'stream_dense', dn, {})
    scenario(mods, 'fr_dense_e2', 'stream_dense', dn, dict(edgesupport=2))


def main():
    mods = loader.load()
    if '--resimulate' not in sys.argv[1:]:
        from tests import golden_util as GU
        for name in GU.scenario_names():
            write_doc(replay(mods, name)[1])
        return None
    if 'lognormal' in sys.argv[1:]:                          # only the scenario added in round 2
        return lognormal_scenario(mods)
    if 'dense' in sys.argv[1:]:                              # only the scenarios added in round 3
        return dense_scenario(mods)

    # ---- PE library, short contigs: many contig-spanning pairs --------------------------------
    asm = synth.make_assembly(300, 1500, 101)
    fr = synth.simulate_library(asm, synth.LibrarySpec('fr', 500.0, 50.0), 15000, 102)
    save_batch('stream_fr', fr)
    scenario(mods, 'fr_infer', 'stream_fr', fr, {})
    scenario(mods, 'fr_noscore', 'stream_fr', fr, dict(no_score=True))
    scenario(mods, 'fr_noextend', 'stream_fr', fr, dict(extend_paths=False))
    scenario(mods, 'fr_nodup', 'stream_fr', fr, dict(detect_duplicate=False))
    # user-given -m -s -T -k -r -e, mapq threshold 0, tight -T so that long inserts are rejected
    scenario(mods, 'fr_given', 'stream_fr', fr,
             dict(mean_ins_size=480.0, std_dev_ins_size=55.0, ins_size_threshold=430, contig_threshold=900,
                  read_len=100, edgesupport=3, min_mapq=0))

    # ---- MP library with PE contamination, fractional inferred read length ---------------------
    asm2 = synth.make_assembly(250, 5000, 201)
    rf = synth.simulate_library(asm2, synth.LibrarySpec('rf', 2500.0, 250.0, contam_frac=0.25), 16000, 202)
    rf.rlen[:1000:7] = 99          # first-1000-records mean read length becomes fractional
    rf.rlen[5:1000:11] = 0         # rlen == 0 falls back to alen (libmetrics.py:258-263)
    save_batch('stream_rf', rf)
    scenario(mods, 'rf_contam', 'stream_rf', rf, dict(orientation='rf'))

    # ---- later-library state: chained scaffolds with mixed directions --------------------------
    layout = synth.chain_scaffolds(asm2, 301, max_run=4)
    scenario(mods, 'rf_second_lib', 'stream_rf', rf, dict(orientation='rf', first_lib=False),
             layout=layout, layout_threshold=3000)
    scenario(mods, 'rf_second_lib_nodup_noext', 'stream_rf', rf,
             dict(orientation='rf', first_lib=False, detect_duplicate=False, extend_paths=False),
             layout=layout, layout_threshold=3000)

    # ---- edge cases: unknown tids, contig missing from FASTA, zero-length contig, repeat, low coverage
    asm3 = synth.make_assembly(200, 1500, 401)
    ec = edgecase_stream(asm3, synth.LibrarySpec('fr', 500.0, 50.0, fishy_frac=0.01), 402)
    save_batch('stream_edge', ec)
    fasta = [n for i, n in enumerate(ec.references) if i % 37 != 5]
    scenario(mods, 'fr_edgecases', 'stream_edge', ec, {}, fasta_names=fasta)

    lognormal_scenario(mods)
    dense_scenario(mods)


if __name__ == '__main__':
    main()
