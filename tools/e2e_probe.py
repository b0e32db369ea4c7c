"""Development probe: wall time of the drop-in entry points on the full C2 workload."""
import io, os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from besst_amd import CreateGraph, Parameter, libmetrics, session, workload
wl = workload.make('C2', 0)
batch = wl['batch']
p = Parameter.parameter()
p.scaffold_indexer = 1; p.min_mapq = 11; p.lower_cov_cutoff = 0.001; p.cov_cutoff = None; p.first_lib = True
p.orientation = 'fr'; p.detect_duplicate = True; p.extend_paths = True; p.no_score = False; p.detect_haplotype = False
p.print_scores = False; p.max_contig_overlap = 200; p.pass_number = 1
p.information_file = io.StringIO(); p.output_directory = tempfile.mkdtemp()
t0 = time.perf_counter(); sess = session.open_session(batch); t1 = time.perf_counter()
libmetrics.get_metrics(batch, p, p.information_file); t2 = time.perf_counter()
C_dict = {n: '' for n in batch.references}
# sequences are irrelevant for timing; InitializeObjects only needs len() for N50
C_dict = {n: 'A' * l for n, l in zip(batch.references, batch.lengths)}
t3 = time.perf_counter()
G, Gp = CreateGraph.PE({}, {}, p.information_file, C_dict, p, {}, {}, batch); t4 = time.perf_counter()
print('records %d  upload %.3f s  get_metrics %.3f s  PE %.3f s' % (len(batch), t1 - t0, t2 - t1, t4 - t3))
print('G edges', len(G.edges()), 'Gp edges', len(Gp.edges()), 'mean/sd', p.mean_ins_size, p.std_dev_ins_size)
import cProfile, pstats
if os.environ.get('PROFILE_PE'):
    p2 = Parameter.parameter()
    for k, v in vars(p).items():
        if k not in ('information_file',):
            try:
                setattr(p2, k, v)
            except Exception:
                pass
    p2.scaffold_indexer = 1; p2.first_lib = True; p2.information_file = io.StringIO()
    C2 = {n: 'A' * l for n, l in zip(batch.references, batch.lengths)}
    pr = cProfile.Profile(); pr.enable()
    CreateGraph.PE({}, {}, p2.information_file, C2, p2, {}, {}, batch)
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
