"""GPU parity: device record loop + edge table vs the oracle, through the C ABI (ctypes)."""
import numpy as np
import pytest

from oracle import py_oracle as O
from besst_amd import synth
from tests import golden_util as GU
from tests import gpu_util as DU
from tests.test_oracle_golden import oracle_state_from_layout

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=['two_pass', 'fused'])
def record_path(request, monkeypatch):
    """Every scenario runs through both forms of the record loop (besst_lib_params.record_path): stream_kernel +
    ordered_kernel, and fused_kernel."""
    monkeypatch.setenv('BESST_RECORD_PATH', '0' if request.param == 'two_pass' else '1')


def _oracle_inputs(doc, batch):
    p = O.LibParams(**doc['overrides'])
    rec = GU.rec_lists(batch)
    O.get_metrics(rec, batch.lengths, p)
    if doc['layout'] is not None:
        st = oracle_state_from_layout(doc, batch, doc['layout_threshold'])
        p.tot_assembly_length = sum(batch.lengths)
        O.clean_objects(st, p)
    else:
        st = O.State()
        fasta = {n: l for n, l in zip(batch.references, batch.lengths) if n in set(doc['fasta_names'])}
        O.initialize_objects(batch.references, batch.lengths, fasta, st, p)
    return p, rec, O.contig_table(batch.references, st)


@pytest.mark.parametrize('name', GU.scenario_names())
def test_golden_streams(name):
    doc, batch = GU.load(name)
    p, rec, tab = _oracle_inputs(doc, batch)
    loop = O.record_loop(rec, tab, p)
    table, aligned, ctr = DU.device_build(batch, tab, p)
    DU.assert_matches_oracle(table, aligned, ctr, loop, len(batch.references))
    # and against the reference's own numbers captured in the fixture
    c = doc['after_loop']['counter']
    if doc['layout'] is None:
        assert (ctr.count, ctr.nr_of_duplicates, ctr.reads_with_too_long_insert) == \
            (c['count'], c['nr_of_duplicates'], c['reads_with_too_long_insert'])


@pytest.mark.parametrize('orientation,read_len,chunks', [('fr', 100, 1), ('rf', 100.38, 3), ('fr', 99.999, 2)])
def test_seeded_stream_vs_oracle(orientation, read_len, chunks):
    asm = synth.make_assembly(3000, 2500, 11)
    spec = synth.LibrarySpec(orientation, 1200.0, 150.0, contam_frac=0.2 if orientation == 'rf' else 0.0)
    batch = synth.simulate_library(asm, spec, 400000, 12)
    lay = synth.chain_scaffolds(asm, 13, max_run=3)
    thr = 2600
    big = lay['scaf_len'] >= thr
    tab = dict(cls=np.where(big, 1, 2).tolist(), scaf=lay['scaf_id'].tolist(), slen=lay['scaf_len'].tolist(),
               cpos=lay['position'].tolist(), clen=asm.lengths.tolist(), cdir=lay['direction'].tolist())
    # a few contigs dropped from the table (repeats removed by an earlier pass)
    for t in range(7, asm.nc, 211):
        tab['cls'][t] = 0
    p = O.LibParams(orientation=orientation, read_len=read_len, ins_size_threshold=2100.5, min_mapq=11)
    rec = GU.rec_lists(batch)
    loop = O.record_loop(rec, tab, p)
    assert loop.count > 1000 and loop.nr_of_duplicates > 0 and loop.too_long > 0 and loop.fishy_reads > 0
    table, aligned, ctr = DU.device_build(batch, tab, p, chunks=chunks)
    DU.assert_matches_oracle(table, aligned, ctr, loop, asm.nc)


def test_empty_and_tiny_inputs():
    asm = synth.make_assembly(5, 3000, 1)
    tab = dict(cls=[1] * 5, scaf=[1, 2, 3, 4, 5], slen=asm.lengths.tolist(), cpos=[0] * 5,
               clen=asm.lengths.tolist(), cdir=[True] * 5)
    p = O.LibParams(read_len=100, ins_size_threshold=800.0)
    batch = synth.simulate_library(asm, synth.LibrarySpec('fr', 500.0, 50.0), 50, 2)
    empty = batch.slice(0, 0)
    table, aligned, ctr = DU.device_build(empty, tab, p)
    assert len(table) == 0 and ctr.count == 0 and aligned.tolist() == [0] * 5
    assert (ctr.prev_obs1, ctr.prev_obs2) == (-1, -1)
    for n in (1, 3, 5, 63, 100):
        sub = batch.slice(0, n)
        loop = O.record_loop(GU.rec_lists(sub), tab, p)
        table, aligned, ctr = DU.device_build(sub, tab, p)
        DU.assert_matches_oracle(table, aligned, ctr, loop, 5)


def test_unsorted_stream_is_still_exact():
    """A name-sorted / shuffled BAM breaks every sortedness assumption the kernels exploit for speed
    (wave-uniform tids, clustered candidates); results must not change."""
    asm = synth.make_assembly(900, 1500, 21)
    batch = synth.simulate_library(asm, synth.LibrarySpec('fr', 600.0, 80.0), 120000, 22)
    rng = np.random.default_rng(23)
    perm = rng.permutation(len(batch))
    cols = {c: getattr(batch, c)[perm] for c in ('tid', 'mtid', 'pos', 'mpos', 'tlen', 'flag', 'mapq', 'qlen')}
    from besst_amd.records import RecordBatch
    shuffled = RecordBatch(batch.references, batch.lengths, rlen=batch.rlen[perm], alen=batch.alen[perm], **cols)
    lens = asm.lengths.tolist()
    tab = dict(cls=[1 if l >= 900 else 2 for l in lens], scaf=list(range(1, asm.nc + 1)), slen=lens,
               cpos=[0] * asm.nc, clen=lens, cdir=[True] * asm.nc)
    p = O.LibParams(read_len=100, ins_size_threshold=1080.0)
    loop = O.record_loop(GU.rec_lists(shuffled), tab, p)
    assert loop.count > 1000
    table, aligned, ctr = DU.device_build(shuffled, tab, p)
    DU.assert_matches_oracle(table, aligned, ctr, loop, asm.nc)


def test_fragmented_assembly_many_contigs_per_wave():
    """20 records per contig: every wave of the streaming pass spans a dozen contigs and takes the run-segmented
    coverage path (lanes straddling a boundary add their records directly); coverage must stay exact, also with
    unmapped (-1) and absent contigs in the mix."""
    asm = synth.make_assembly(6000, 600, 41)
    batch = synth.simulate_library(asm, synth.LibrarySpec('fr', 450.0, 40.0), 60000, 42)
    lens = asm.lengths.tolist()
    tab = dict(cls=[1 if l >= 500 else 2 for l in lens], scaf=list(range(1, asm.nc + 1)), slen=lens,
               cpos=[0] * asm.nc, clen=lens, cdir=[True] * asm.nc)
    for t in range(3, asm.nc, 17):
        tab['cls'][t] = 0
    p = O.LibParams(read_len=100, ins_size_threshold=690.0)
    loop = O.record_loop(GU.rec_lists(batch), tab, p)
    assert loop.count > 500 and sum(1 for x in loop.aligned if x) > 4000
    table, aligned, ctr = DU.device_build(batch, tab, p)
    DU.assert_matches_oracle(table, aligned, ctr, loop, asm.nc)


def test_small_genome_high_coverage():
    """A few dozen contigs at very high coverage: thousands of links per edge (MSD buckets beyond the LDS sort:
    bitonic network in global scratch), and thousands of waves per contig (coverage through the group records)."""
    asm = synth.make_assembly(40, 3000, 51)
    batch = synth.simulate_library(asm, synth.LibrarySpec('fr', 450.0, 40.0), 2500000, 52)
    lens = asm.lengths.tolist()
    tab = dict(cls=[1] * asm.nc, scaf=list(range(1, asm.nc + 1)), slen=lens, cpos=[0] * asm.nc, clen=lens,
               cdir=[True] * asm.nc)
    p = O.LibParams(read_len=100, ins_size_threshold=690.0)
    loop = O.record_loop(GU.rec_lists(batch), tab, p)
    assert max(r.n for r in loop.edges.values()) > 2000
    table, aligned, ctr = DU.device_build(batch, tab, p)
    DU.assert_matches_oracle(table, aligned, ctr, loop, asm.nc)
