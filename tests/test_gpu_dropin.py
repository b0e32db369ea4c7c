"""GPU parity of the drop-in entry points (get_metrics + PE) against the golden vectors of the real reference."""
import io
import os
import tempfile

import pytest

from besst_amd import Contig, CreateGraph, Parameter, Scaffold, libmetrics, session
from tests import golden_util as GU

pytestmark = pytest.mark.gpu


def make_param(overrides):
    p = Parameter.parameter()
    p.scaffold_indexer = 1
    p.min_mapq = 11
    p.lower_cov_cutoff = 0.001
    p.cov_cutoff = None
    p.first_lib = True
    p.orientation = 'fr'
    p.detect_duplicate = True
    p.extend_paths = True
    p.no_score = False
    p.detect_haplotype = False
    p.print_scores = False
    p.max_contig_overlap = 200
    p.pass_number = 1
    p.information_file = io.StringIO()
    p.output_directory = tempfile.mkdtemp(prefix='besst_amd_')
    for k, v in overrides.items():
        setattr(p, k, v)
    return p


def state_from_layout(doc, batch, threshold):
    lay = doc['layout']
    Contigs, Scaffolds, small_contigs, small_scaffolds = {}, {}, {}, {}
    by_scaf = {}
    for tid in range(len(batch.references)):
        by_scaf.setdefault(lay['scaf_id'][tid], []).append(tid)
    for sid, tids in by_scaf.items():
        objs = []
        for t in tids:
            c = Contig.contig(batch.references[t], contig_scaffold=sid, contig_direction=bool(lay['direction'][t]),
                              contig_position=lay['position'][t], contig_length=batch.lengths[t], contig_sequence='')
            objs.append(c)
        s = Scaffold.scaffold(sid, objs, lay['scaf_len'][tids[0]])
        big = s.s_length >= threshold
        (Scaffolds if big else small_scaffolds)[sid] = s
        for c in objs:
            (Contigs if big else small_contigs)[c.name] = c
    return Contigs, Scaffolds, small_contigs, small_scaffolds


def edge_rows(G, with_score):
    rows = []
    for u, v in G.edges():
        d = G[u][v]
        if d['nr_links'] is None:
            continue
        row = dict(u=list(u), v=list(v), nr_links=d['nr_links'], obs=d['obs'], obs_sq=d['obs_sq'])
        if with_score:
            for k in ('gap', 'score'):
                if k in d:
                    row[k] = d[k]
        rows.append(row)
    return rows


@pytest.mark.parametrize('name', GU.scenario_names())
def test_dropin_matches_reference_golden(name):
    doc, batch = GU.load(name)
    param = make_param(doc['overrides'])
    info = param.information_file
    libmetrics.get_metrics(batch, param, info)
    for k, want in doc['metrics'].items():
        if k == 'empirical_distribution':
            ed = getattr(param, 'empirical_distribution', None)
            got = None if ed is None else [ed[i] for i in range(len(ed))]
        else:
            got = getattr(param, k, None)
        assert got == want, (name, k, got, want)     # library metrics: bit exact
    if doc['layout'] is not None:
        Contigs, Scaffolds, small_contigs, small_scaffolds = state_from_layout(doc, batch, doc['layout_threshold'])
        param.scaffold_indexer = doc['layout']['next_scaffold_id']
        param.tot_assembly_length = sum(batch.lengths)
    else:
        Contigs, Scaffolds, small_contigs, small_scaffolds = {}, {}, {}, {}
    lens = dict(zip(batch.references, batch.lengths))
    C_dict = {n: 'A' * int(lens.get(n, 10)) for n in doc['fasta_names']}
    G, G_prime = CreateGraph.PE(Contigs, Scaffolds, info, C_dict, param, small_contigs, small_scaffolds, batch)
    session.close_session(batch)
    fin = doc['final']
    # structure, counts and integer sums: bit exact, same iteration order as the reference's graphs
    assert edge_rows(G, False) == [{k: e[k] for k in ('u', 'v', 'nr_links', 'obs', 'obs_sq')} for e in fin['G']]
    assert edge_rows(G_prime, False) == [{k: e[k] for k in ('u', 'v', 'nr_links', 'obs', 'obs_sq')} for e in fin['G_prime']]
    assert [list(n) for n in G.nodes()] == fin['G_nodes']
    assert [list(n) for n in G_prime.nodes()] == fin['G_prime_nodes']
    assert [[c.name, c.scaffold, c.coverage] for c in Contigs.values()] == fin['contigs']
    assert [[c.name, c.scaffold, c.coverage] for c in small_contigs.values()] == fin['small_contigs']
    assert list(Scaffolds) == fin['scaffolds'] and list(small_scaffolds) == fin['small_scaffolds']
    for k in ('mean_coverage', 'std_dev_coverage', 'edgesupport', 'expected_links_over_mean_plus_stddev',
              'scaffold_indexer', 'tot_assembly_length', 'current_N50', 'current_L50'):
        assert getattr(param, k) == fin['param'][k], (name, k)
    # gap within +-1 bp, score within 1e-9 (device erf/exp are not bit-identical to libm)
    got = {(tuple(e['u']), tuple(e['v'])): e for e in edge_rows(G, True)}
    n_exact = 0
    tol = GU.tolerances(doc)
    for e in fin['G']:
        g = got[(tuple(e['u']), tuple(e['v']))]
        assert abs(g['gap'] - e['gap']) <= tol['gap'], (name, e, g)
        if g['gap'] == e['gap']:
            n_exact += 1
            assert abs(g['score'] - e['score']) <= tol['score'] * (1.0 if tol['exact'] else max(1.0, abs(e['score']))), (name, e, g)
    assert n_exact >= (0.9 if tol['exact'] else 0.5) * len(fin['G'])
    # observation lists keep BAM order
    snap = doc['after_loop']
    want_obs = {frozenset((tuple(e['u']), tuple(e['v']))): e['observations'] for e in snap['G_prime']}
    for u, v in G_prime.edges():
        d = G_prime[u][v]
        if d['nr_links'] is not None:
            assert d['observations'] == want_obs[frozenset((u, v))]


@pytest.mark.parametrize('name', ['fr_infer', 'rf_contam'])
def test_dropin_from_bam_file(name, tmp_path):
    """BAM bytes -> library's own BGZF/BAM reader -> get_metrics + PE: same metrics and graphs as from columns."""
    from besst_amd import bamio
    from tests import bam_writer
    doc, batch = GU.load(name)
    path = str(tmp_path / 'mapped.bam')
    bam_writer.write_bam(path, batch)
    records = bamio.read_bam(path, threads=4)
    param = make_param(doc['overrides'])
    info = param.information_file
    libmetrics.get_metrics(records, param, info)
    for k, want in doc['metrics'].items():
        if k != 'empirical_distribution':
            assert getattr(param, k, None) == want, (name, k)
    lens = dict(zip(batch.references, batch.lengths))
    C_dict = {n: 'A' * int(lens.get(n, 10)) for n in doc['fasta_names']}
    Contigs, Scaffolds, small_contigs, small_scaffolds = {}, {}, {}, {}
    G, G_prime = CreateGraph.PE(Contigs, Scaffolds, info, C_dict, param, small_contigs, small_scaffolds, records)
    session.close_session(records)
    fin = doc['final']
    assert edge_rows(G, False) == [{k: e[k] for k in ('u', 'v', 'nr_links', 'obs', 'obs_sq')} for e in fin['G']]
    assert edge_rows(G_prime, False) == [{k: e[k] for k in ('u', 'v', 'nr_links', 'obs', 'obs_sq')} for e in fin['G_prime']]


@pytest.mark.parametrize('mode', ['device', 'host'])
@pytest.mark.parametrize('name,chunk', [('fr_infer', 4096), ('rf_contam', 1024), ('fr_edgecases', 1 << 22)])
def test_dropin_from_streamed_bam(name, chunk, mode, tmp_path):
    """BAM file -> resident records - besst_ctx_push_bam_device (the compressed file uploaded, inflate + record decode on the
    GPU) or besst_ctx_push_bam (decode on host threads, pinned staging, copies under the next chunk's decode; small chunks:
    many of them, the columns grow on the way) -> get_metrics + PE on the resident records: metrics, graphs and object
    dicts as the reference's goldens, no host record columns anywhere."""
    from besst_amd import bamio
    doc, batch = GU.load(name)
    path = str(tmp_path / 'mapped.bam')
    bamio.write_bam(path, batch, threads=3)
    bam = bamio.ResidentBam(path, threads=4, chunk_records=chunk, mode=mode, chunk_blocks=64)
    assert len(bam) == len(batch)
    if mode == 'host':
        assert bam.ingest.on_device == 0 and bam.ingest.chunks == -(-len(batch) // max(1024, chunk))
        assert bam.ingest.bytes_h2d == 25 * len(batch)
    else:
        assert bam.ingest.on_device == 1 and bam.ingest.bytes_h2d <= os.path.getsize(path)
    param = make_param(doc['overrides'])
    info = param.information_file
    libmetrics.get_metrics(bam, param, info)
    for k, want in doc['metrics'].items():
        if k == 'empirical_distribution':
            ed = getattr(param, 'empirical_distribution', None)
            got = None if ed is None else [ed[i] for i in range(len(ed))]
        else:
            got = getattr(param, k, None)
        assert got == want, (name, k)
    lens = dict(zip(batch.references, batch.lengths))
    C_dict = {n: 'A' * int(lens.get(n, 10)) for n in doc['fasta_names']}
    Contigs, Scaffolds, small_contigs, small_scaffolds = {}, {}, {}, {}
    G, G_prime = CreateGraph.PE(Contigs, Scaffolds, info, C_dict, param, small_contigs, small_scaffolds, bam)
    session.close_session(bam)
    fin = doc['final']
    assert edge_rows(G, False) == [{k: e[k] for k in ('u', 'v', 'nr_links', 'obs', 'obs_sq')} for e in fin['G']]
    assert edge_rows(G_prime, False) == [{k: e[k] for k in ('u', 'v', 'nr_links', 'obs', 'obs_sq')} for e in fin['G_prime']]
    # (rf_contam's stream carries a few records whose rlen disagrees with their qlen - inputs of the read-length step -, and
    # a BAM record has ONE sequence length: their coverage does not survive the file)
    cov = (lambda c: None) if name == 'rf_contam' else (lambda c: c.coverage)
    assert [[c.name, c.scaffold, cov(c)] for c in Contigs.values()] == [[n, sc, cov(Contigs[n])] for n, sc, _ in fin['contigs']]
    if name != 'rf_contam':
        assert [[c.name, c.scaffold, c.coverage] for c in Contigs.values()] == fin['contigs']
    assert list(Scaffolds) == fin['scaffolds'] and list(small_scaffolds) == fin['small_scaffolds']


def test_cli_end_to_end(tmp_path):
    """FASTA + BAM on disk -> besst_amd.cli -> scored edge table, equal to the reference golden."""
    from besst_amd import cli
    from tests import bam_writer
    doc, batch = GU.load('fr_infer')
    bam = str(tmp_path / 'lib.bam')
    bam_writer.write_bam(bam, batch)
    fasta = str(tmp_path / 'contigs.fa')
    lens = dict(zip(batch.references, batch.lengths))
    with open(fasta, 'w') as fh:
        for n in doc['fasta_names']:
            fh.write('>%s\n%s\n' % (n, 'A' * lens[n]))
    assert cli.main(['-c', fasta, '-f', bam, '-orientation', 'fr', '-o', str(tmp_path), '--linearize']) == 0
    rows = [l.rstrip('\n').split('\t') for l in open(str(tmp_path / 'BESST_output' / 'pass1' / 'edges_G.tsv'))][1:]
    got = [(int(r[0]), r[1], int(r[2]), r[3], int(r[4]), int(r[5]), int(r[6])) for r in rows]
    want = [(e['u'][0], e['u'][1], e['v'][0], e['v'][1], e['nr_links'], e['obs'], e['obs_sq']) for e in doc['final']['G']]
    assert got == want
    assert (tmp_path / 'BESST_output' / 'Statistics.txt').exists()
    # --linearize: a subset of G's link edges, at most one per scaffold end
    lin = [l.rstrip('\n').split('\t') for l in open(str(tmp_path / 'BESST_output' / 'pass1' / 'edges_G_linear.tsv'))][1:]
    assert 0 < len(lin) < len(rows)
    assert set(tuple(r[:4]) for r in lin) <= set(tuple(r[:4]) for r in rows)
    ends = [(r[0], r[1]) for r in lin] + [(r[2], r[3]) for r in lin]
    assert len(ends) == len(set(ends))
    assert 'cycles removed from graph' in open(str(tmp_path / 'BESST_output' / 'Statistics.txt')).read()


@pytest.mark.parametrize('name', ['fr_infer', 'rf_contam', 'rf_second_lib'])
def test_linearize_the_graph_pe_returns(name):
    """CreateGraph.PE -> MakeScaffolds.LinearizeGraph (steps 1-4) on the real scored graph: the surviving nodes and
    link edges equal the sequential CPU restatement run on the same graph's edge list."""
    from besst_amd import MakeScaffolds as MS
    from oracle import scaffold_oracle as SO
    doc, batch = GU.load(name)
    param = make_param(doc['overrides'])
    info = param.information_file
    libmetrics.get_metrics(batch, param, info)
    if doc['layout'] is not None:
        Contigs, Scaffolds, small_contigs, small_scaffolds = state_from_layout(doc, batch, doc['layout_threshold'])
        param.scaffold_indexer = doc['layout']['next_scaffold_id']
        param.tot_assembly_length = sum(batch.lengths)
    else:
        Contigs, Scaffolds, small_contigs, small_scaffolds = {}, {}, {}, {}
    lens = dict(zip(batch.references, batch.lengths))
    C_dict = {n: 'A' * int(lens.get(n, 10)) for n in doc['fasta_names']}
    G, G_prime = CreateGraph.PE(Contigs, Scaffolds, info, C_dict, param, small_contigs, small_scaffolds, batch)
    session.close_session(batch)
    arr = MS._GraphArrays(G, scored_only=True)
    # (a graph straight from PE: its edge arrays come from the columns behind it, no container has been made)
    assert getattr(arr, 'from_columns', False) and G.link_columns() is not None
    assert len(arr.edges) > 20
    want = SO.linearize(arr.n_scaffolds, arr.a, arr.b, arr.score)
    keep_nodes = [n for n in G.nodes() if want['present'][arr.index[n[0]]]]
    keep_edges = [e for e, k in zip(arr.edges, want['alive2'])
                  if k and want['present'][arr.index[e[0][0]]] and want['present'][arr.index[e[1][0]]]]
    nodes_before = len(G.nodes())
    G, _, _ = MS.LinearizeGraph(G, G_prime, Contigs, Scaffolds, info, param)
    assert list(G.nodes()) == keep_nodes
    assert [(u, v) for u, v in G.edges() if G[u][v]['nr_links'] is not None] == keep_edges
    assert len(keep_nodes) < nodes_before or len(keep_edges) < len(arr.edges)
    # every remaining node has at most one link edge: the graph is a set of paths
    for n in G.nodes():
        assert sum(1 for m in G.neighbors(n) if G[n][m]['nr_links'] is not None) <= 1


def test_skewed_library_end_to_end_against_the_oracle():
    """A skewed library fifteen times the `fr_lognormal` golden (insert sizes exp(N(ln 1500, 0.35)), 250 k pairs on 800
    contigs: ~450 scored edges; with 450 k pairs on 1500 contigs - 900 edges - and with 1.2 M pairs on 4000 contigs - 2400 edges - the same assertions held, the oracle took
    three minutes) through the two entry points: get_metrics must flag it
    (param.lognormal, libmetrics.py:380-390) and PE must score it by the log-normal branch ON THE DEVICE
    (CreateGraph.py:485-494, 522-531, 549-553) - against the Python oracle end to end (its estimator sums directly, no
    prefix tables): library parameters and graph structure exactly, every gap exactly, every score within 1e-9."""
    from besst_amd import device, synth
    from oracle import py_oracle as O
    from tests.test_oracle_golden import edge_rows as oracle_rows, run_oracle
    asm = synth.make_assembly(800, 8000, 77)
    batch = synth.simulate_library(asm, synth.LibrarySpec('fr', 1500.0, 150.0, lognormal_sigma=0.35), 250_000, 78)
    doc = dict(overrides={}, layout=None, layout_threshold=None, fasta_names=list(batch.references))
    p, st, out = run_oracle(doc, batch)
    assert p.lognormal is True
    param = make_param({})
    info = param.information_file
    device.CALL_SECONDS = {}
    try:
        libmetrics.get_metrics(batch, param, info)
        assert param.lognormal is True
        for k in ('mean_ins_size', 'std_dev_ins_size', 'read_len', 'lognormal_mean', 'lognormal_sigma', 'ins_size_threshold',
                  'contig_threshold'):
            assert getattr(param, k) == getattr(p, k), k
        C_dict = {n: 'A' * int(l) for n, l in zip(batch.references, batch.lengths)}
        Contigs, Scaffolds, small_contigs, small_scaffolds = {}, {}, {}, {}
        G, G_prime = CreateGraph.PE(Contigs, Scaffolds, info, C_dict, param, small_contigs, small_scaffolds, batch)
        assert 'conditional_stddevs' in device.CALL_SECONDS          # the sigma table came from the device
    finally:
        device.CALL_SECONDS = None
        session.close_session(batch)
    want, got = oracle_rows(out['G']), edge_rows(G, True)
    strip = lambda rows: [{k: e[k] for k in ('u', 'v', 'nr_links', 'obs', 'obs_sq')} for e in rows]
    assert strip(got) == strip(want) and len(want) > 300
    assert strip(edge_rows(G_prime, False)) == strip(oracle_rows(out['Gp'], keys=()))
    assert [g['gap'] for g in got] == [w['gap'] for w in want]
    worst = max(abs(g['score'] - w['score']) for g, w in zip(got, want))
    assert worst <= 1e-9, worst
    assert len({w['gap'] for w in want}) > 40 and sum(1 for w in want if w['score'] > 0) > 40
