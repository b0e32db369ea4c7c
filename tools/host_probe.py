"""Development probe (no GPU): the HOST side of the drop-in - besst_amd.libmetrics.get_metrics + besst_amd.CreateGraph.PE -
on a C3-shaped assembly, with the device stages answered by the C oracle (tests/fake_device.py) and the scoring kernel by
zeros: what is timed is the Python around the C ABI.  usage: host_probe.py [contigs] [pairs] [profile]"""
import cProfile, io, os, pickle, pstats, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from besst_amd import CreateGraph, Parameter, libmetrics, session, workload
from tests import fake_device

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
prof = len(sys.argv) > 3
cache = '/tmp/host_probe_%d_%d.pkl' % (nc, pairs)
t0 = time.perf_counter()
if os.path.isfile(cache):
    wl = pickle.load(open(cache, 'rb'))
else:
    wl = workload.make('C3', 0, pairs=pairs, nc=nc)
    pickle.dump(wl, open(cache, 'wb'), protocol=4)
batch = wl['batch']
print('workload %.1f s, %d records' % (time.perf_counter() - t0, len(batch)), flush=True)


class Fast(fake_device.FakeGraphContext):
    def score_edges(self, rows, swap, len1, len2, mean, sigma, read_len):
        m = len(rows)
        return np.zeros(m), np.full(m, 100.0), np.ones(m, np.int32), np.ones(m, np.uint8)


session.device.GraphContext = Fast
p = Parameter.parameter()
p.scaffold_indexer = 1; p.min_mapq = 11; p.lower_cov_cutoff = 0.001; p.cov_cutoff = None; p.first_lib = True
p.orientation = wl['lib']['orientation']; p.detect_duplicate = True; p.extend_paths = True; p.no_score = False
p.detect_haplotype = False; p.print_scores = False; p.max_contig_overlap = 200; p.pass_number = 1
p.information_file = io.StringIO(); p.output_directory = tempfile.mkdtemp(prefix='besst_amd_')
p.contig_index = dict(enumerate(batch.references))
lengths = dict(zip(batch.references, batch.lengths))


class L(object):
    __slots__ = ('n',)
    def __init__(self, n): self.n = n
    def __len__(self): return self.n
    def __getitem__(self, k): return 'N' * len(range(*k.indices(self.n))) if isinstance(k, slice) else 'N'


C_dict = {name: L(int(lengths[name])) for name in batch.references}
t0 = time.perf_counter()
libmetrics.get_metrics(batch, p, p.information_file)
t1 = time.perf_counter()
ctx = session.open_session(batch).ctx
orig = ctx.build_graph
spent = {}
def timed_build():
    a = time.perf_counter(); r = orig(); spent['build'] = time.perf_counter() - a; return r
ctx.build_graph = timed_build
Contigs, Scaffolds, sc, ss = {}, {}, {}, {}
stage = {}
def timed(owner, name):
    fn = getattr(owner, name)
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            stage[name] = stage.get(name, 0.0) + time.perf_counter() - t
    setattr(owner, name, w)
for name in ('InitializeObjects', 'contig_table', 'LinkTable', 'GiveScoreOnEdges', 'remove_edges_below_threshold',
             'RemoveBugEdges', 'RepeatDetector', 'CalculateMeanCoverage', 'filter_low_coverage_contigs',
             'infer_spurious_link_count_threshold'):
    timed(CreateGraph, name)
timed(CreateGraph.GraphPlan, 'build'); timed(CreateGraph.GraphPlan, 'take')
pr = cProfile.Profile()
if prof: pr.enable()
t2 = time.perf_counter()
G, Gp = CreateGraph.PE(Contigs, Scaffolds, p.information_file, C_dict, p, sc, ss, batch)
t3 = time.perf_counter()
if prof: pr.disable()
print('get_metrics %.2f s   PE %.2f s of which oracle build_graph %.2f s -> host %.2f s   (G %d edges, G_prime %d edges)' % (
    t1 - t0, t3 - t2, spent['build'], t3 - t2 - spent['build'], G.number_of_edges(), Gp.number_of_edges()))
print('  '.join('%s %.2f' % kv for kv in sorted(stage.items(), key=lambda kv: -kv[1])))
if prof:
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:5000])
