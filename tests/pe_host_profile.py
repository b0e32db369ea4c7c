"""TEST INFRASTRUCTURE (asks the oracle; not part of the product path).  Where does CreateGraph.PE's HOST time go?  No GPU needed: the device stages are answered once by the C oracle (record
loop, edge rows) and by zeros (scores), then PE runs under cProfile / perf_counter on the product's host code with the
answers at hand.  The contig count is what the host time scales with (objects, graph assembly, filters); the pair count
only sizes the observation columns.  usage: python tests/pe_host_profile.py [config] [pairs] [contigs] [--cprofile]"""
import cProfile, io, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from besst_amd import CreateGraph, Parameter, device, libmetrics, session, workload
from besst_amd._lib import Counters
from oracle import c_oracle as CO

config = sys.argv[1] if len(sys.argv) > 1 else 'C3'
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
contigs = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
t0 = time.perf_counter()
wl = workload.make(config, 0, pairs=pairs, nc=contigs)
batch, lib = wl['batch'], wl['lib']
print('workload: %d records, %d contigs (%.1f s)' % (len(batch), contigs, time.perf_counter() - t0), flush=True)


class Answers(object):
    """GraphContext's interface with the oracle's answers (computed at the first build_graph, outside the timed region)."""
    def __init__(self, device_index=0):
        self.n_contigs = 0
        self.seconds = {}
        self.ready = None

    def close(self): pass
    def push_records(self, b): pass
    def set_library(self, *a): pass

    def set_contigs(self, **cols):
        self.table = {k: np.asarray(v) for k, v in cols.items()}
        self.n_contigs = len(self.table['cls'])

    def stream_order(self): return None, (0, 0), (0, 0)

    def prepare(self):
        nb = workload.node_bits_for(self.table)
        keys, payload, aligned, c = CO.record_loop(batch, self.table, lib, nb, threads=os.cpu_count())
        rows = CO.edge_rows(keys, payload)
        table = device.EdgeTable(rows['key'].astype(np.uint64), rows['mask'].astype(np.uint32), rows['n'].astype(np.uint32),
                                 rows['sum_obs'].astype(np.int64), rows['sum_obs_sq'].astype(np.int64),
                                 rows['first_idx'].astype(np.uint32), rows['offset'].astype(np.uint32), nb,
                                 rows['obs_lo'].astype(np.int32), rows['obs_hi'].astype(np.int32))
        self.ready = (table, aligned, Counters(*[int(x) for x in c[:8]], int(c[8]), int(c[9])))

    def build_graph(self, lazy_observations=False):
        return self.ready

    def score_edges(self, rows, swap, len1, len2, mean, sigma, read_len, lognormal=None):
        m = len(rows)
        return np.zeros(m), np.full(m, 100.0), np.zeros(m, np.int32), np.ones(m, np.uint8)


def run(profile):
    p = Parameter.parameter()
    p.scaffold_indexer = 1; p.min_mapq = 11; p.lower_cov_cutoff = 0.001; p.cov_cutoff = None; p.first_lib = True
    p.orientation = lib['orientation']; p.detect_duplicate = True; p.extend_paths = True; p.no_score = False
    p.detect_haplotype = False; p.print_scores = False; p.max_contig_overlap = 200; p.pass_number = 1
    p.information_file = io.StringIO(); p.output_directory = tempfile.mkdtemp(prefix='besst_amd_')
    p.read_len = lib['read_len']; p.mean_ins_size = lib['mean']; p.std_dev_ins_size = lib['sd']
    p.ins_size_threshold = lib['ins_size_threshold']; p.contig_threshold = lib['mean'] + 4 * lib['sd']
    p.contamination_ratio = 0.2 if lib['orientation'] == 'rf' else False
    p.contamination_mean, p.contamination_stddev, p.edgesupport, p.lognormal = 350.0, 60.0, None, False
    C_dict = {name: 'A' for name in batch.references}

    class Len(str):
        pass
    sess = session.Session.__new__(session.Session)
    sess.batch = batch
    sess.ctx = Answers()
    session._sessions[batch] = sess
    # the answers for this table, outside the timed call: a dry InitializeObjects to get the table PE will set
    objs = ({}, {}, {}, {})
    dry = dict(C_dict)
    CreateGraph.InitializeObjects(batch, objs[0], objs[1], p, io.StringIO(), None, objs[2], objs[3], dry)
    cols, _ = CreateGraph.contig_table(batch.references, objs[0], objs[2], objs[1], objs[3])
    sess.ctx.set_contigs(**cols)
    sess.ctx.prepare()
    p.scaffold_indexer = 1
    Contigs, Scaffolds, small_contigs, small_scaffolds = {}, {}, {}, {}
    CreateGraph.STAGE_SECONDS = {}
    pr = cProfile.Profile() if profile else None
    t0 = time.perf_counter()
    if pr:
        pr.enable()
    G, Gp = CreateGraph.PE(Contigs, Scaffolds, p.information_file, C_dict, p, small_contigs, small_scaffolds, batch)
    if pr:
        pr.disable()
    dt = time.perf_counter() - t0
    session._sessions.pop(batch, None)
    print('PE host side: %.3f s  (G %d edges, G_prime %d edges, %d tuples)' % (dt, G.number_of_edges(), Gp.number_of_edges(),
                                                                               len(sess.ctx.ready[0].obs_lo)))
    if pr:
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
        print(s.getvalue())
    stages = getattr(CreateGraph, 'STAGE_SECONDS', None)
    if stages:
        print('stages: ' + '  '.join('%s %.3f' % kv for kv in stages.items()))


# C_dict values: InitializeObjects takes len() of the sequence; a real FASTA string of the contig's length is not needed for
# the timing except for tot_assembly_length
run(False)
run(False)
if '--cprofile' in sys.argv:
    run(True)
