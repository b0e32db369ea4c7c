"""Development probe (GPU box): size and distinct-key distribution of the top-16-bit buckets of a config's tuple stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from besst_amd import workload
dev = torch.device('cuda', 0)
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
wl = workload.make_device(dev, cfg, 0)
r = bench.SingleGpu(dev, wl, 1)
r.gb.reset(); r.gb.classify(r.rec)
n, _ = r.gb.read_sizes()
k = r.gb.keys[:n] - r.gb.key_base
kb = r.gb.key_bits
print('tuples', n, 'key_bits', kb, 'key_base', r.gb.key_base, 'max', int(k.max()))
L = kb - 16
b = k >> L
sz = torch.bincount(b, minlength=65536)
q = torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], device=dev)
print('bucket size quantiles', torch.quantile(sz.double(), q.double()).tolist(), 'empty', int((sz == 0).sum()),
      '>1024:', int((sz > 1024).sum()), 'tuples in them', int(sz[sz > 1024].sum()), '>4096:', int((sz > 4096).sum()), int(sz[sz > 4096].sum()))
u = torch.unique(k)
ub = torch.bincount(u >> L, minlength=65536)
print('distinct keys per bucket quantiles', torch.quantile(ub.double(), q.double()).tolist(), '>32:', int((ub > 32).sum()),
      'tuples in those', int(sz[ub > 32].sum()), ' small (<=1024) with >32:', int(((ub > 32) & (sz <= 1024)).sum()))
# coverage by the eight most frequent... (first eight in stream order is what the kernel uses; frequency is a proxy)
