"""Development probe: wall time of the drop-in entry points (libmetrics.get_metrics + CreateGraph.PE) from a host
RecordBatch, with a cProfile summary.  usage: dropin_probe.py C2|C3 [pairs] [contigs]"""
import cProfile, io, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from besst_amd import CreateGraph, Parameter, libmetrics, session, workload

config = sys.argv[1] if len(sys.argv) > 1 else 'C2'
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else None
nc = int(sys.argv[3]) if len(sys.argv) > 3 else None
dev = torch.device('cuda', 0)
t0 = time.perf_counter()
wl = workload.make_device(dev, config, 0, pairs=pairs, nc=nc)
batch = wl['batch']
print('generate + host copy %.2f s, %d records' % (time.perf_counter() - t0, len(batch)), flush=True)
p = Parameter.parameter()
p.scaffold_indexer = 1; p.min_mapq = 11; p.lower_cov_cutoff = 0.001; p.cov_cutoff = None; p.first_lib = True
p.orientation = wl['lib']['orientation']; p.detect_duplicate = True; p.extend_paths = True; p.no_score = False
p.detect_haplotype = False; p.print_scores = False; p.max_contig_overlap = 200; p.pass_number = 1
p.information_file = io.StringIO(); p.output_directory = tempfile.mkdtemp(prefix='besst_amd_')
p.contig_index = dict(enumerate(batch.references))
C_dict = {name: '' for name in batch.references}
C_dict = {name: 'A' * 1 for name in batch.references}      # sequences are only stored, never read, on this path
lengths = dict(zip(batch.references, batch.lengths))
class Seq(str):
    pass
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
libmetrics.get_metrics(batch, p, p.information_file)
t1 = time.perf_counter()
Contigs, Scaffolds, sc, ss = {}, {}, {}, {}
C_dict = {name: 'A' * int(lengths[name]) for name in batch.references} if len(batch.references) <= 20000 else None
if C_dict is None:
    # avoid allocating the assembly as python strings: PE only takes len() of the sequences
    class L(object):
        __slots__ = ('n',)
        def __init__(self, n): self.n = n
        def __len__(self): return self.n
    C_dict = {name: L(int(lengths[name])) for name in batch.references}
t2 = time.perf_counter()
G, Gp = CreateGraph.PE(Contigs, Scaffolds, p.information_file, C_dict, p, sc, ss, batch)
t3 = time.perf_counter()
pr.disable()
print('get_metrics %.3f s   PE %.3f s   (G %d edges, G_prime %d edges)' % (t1 - t0, t3 - t2, G.number_of_edges(), Gp.number_of_edges()))
print('mean %.2f sd %.2f T %.1f' % (p.mean_ins_size, p.std_dev_ins_size, p.ins_size_threshold), 'lognormal', p.lognormal)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28); print(s.getvalue()[:6000])
