"""Development probe: per-phase time of ordered_kernel (library built with -DBESST_PHASE_TIMER; counters are ticks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from besst_amd import pipeline, workload
dev = torch.device('cuda', 0)
wl = workload.make(os.environ.get('CFG', 'C2'), 0)
rec = pipeline.DeviceRecords(wl['batch'], dev)
rec2 = pipeline.DeviceRecords(wl['batch'], dev)
gb = pipeline.DeviceGraphBuilder(dev, wl['asm'].nc, wl['node_bits'], wl['lib'], rec.n, rec.n)
gb.set_contigs(**wl['table'])
for r in (rec, rec2, rec):
    gb.reset(); gb.classify(r)
torch.cuda.synchronize()
c = gb.read_counters()
nb = c.n_reach
names = ['bits+scan', 'hot/record loads', 'cold+rows loads', 'eval+coverage', 'chain', 'summary']
vals = [c.count, c.non_unique, c.non_unique_for_scaf, c.nr_of_duplicates, c.reads_with_too_long_insert, c.fishy_reads]
print('blocks', nb)
for n, v in zip(names, vals):
    print('%-18s %7.2f us / block' % (n, v / max(1, nb) * 0.01))
