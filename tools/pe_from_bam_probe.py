"""Development probe (GPU box): cProfile of CreateGraph.PE fed by a bamio.ResidentBam (device ingest), C3 at full size or
[pairs].  usage: python tools/pe_from_bam_probe.py [pairs]"""
import cProfile, io, os, pstats, sys, tempfile, time, shutil
sys.path.insert(0, '.')
import torch, numpy as np
from besst_amd import CreateGraph, Parameter, libmetrics, session, workload, bamio
import bench
dev = torch.device('cuda', 0)
wl = workload.make_device(dev, 'C3', 0, pairs=int(sys.argv[1]) if len(sys.argv) > 1 else None)
batch = wl['batch']; del wl['cols']
tmp = tempfile.mkdtemp(prefix='pe_', dir='/dev/shm'); path = os.path.join(tmp, 'lib.bam')
bamio.write_bam(path, batch)
p = Parameter.parameter()
p.scaffold_indexer = 1; p.min_mapq = 11; p.lower_cov_cutoff = 0.001; p.cov_cutoff = None; p.first_lib = True
p.orientation = wl['lib']['orientation']; p.detect_duplicate = True; p.extend_paths = True; p.no_score = False
p.detect_haplotype = False; p.print_scores = False; p.max_contig_overlap = 200; p.pass_number = 1
p.information_file = io.StringIO(); p.output_directory = tmp
p.contig_index = dict(enumerate(batch.references))
C_dict = {name: bench._SeqLen(int(n)) for name, n in zip(batch.references, batch.lengths)}
del batch, wl
bam = bamio.ResidentBam(path)
libmetrics.get_metrics(bam, p, p.information_file)
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
G, Gp = CreateGraph.PE({}, {}, p.information_file, C_dict, p, {}, {}, bam)
pr.disable(); print('PE', time.perf_counter() - t0)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(22); print(s.getvalue()[:4500])
shutil.rmtree(tmp)
