"""Development probe (GPU box): candidates (tid != mtid) per 1024-record sub-tile of a config's record stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from besst_amd import workload
dev = torch.device('cuda', 0)
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
wl = workload.make_device(dev, cfg, 0)
c = wl['cols']
cand = (c['tid'] != c['mtid'])
n = cand.numel() // 1024 * 1024
for sub in (1024, 768, 512):
    m = cand.numel() // sub * sub
    per = cand[:m].view(-1, sub).sum(1).double()
    q = torch.tensor([0.01, 0.1, 0.5, 0.9, 0.99, 1.0], device=dev, dtype=torch.double)
    rounds = torch.ceil(per / 256).clamp(min=0)
    print(sub, 'share', float(cand.double().mean()), 'per sub-tile quantiles', torch.quantile(per[:4000000], q).tolist(), 'mean rounds', float(rounds.mean()),
          'lane use', float(per.sum() / (rounds.sum() * 256)))
for g in (8, 16, 64, 256, 1024):
    m = cand.numel() // g * g
    print('groups of', g, 'records without a candidate:', float((~cand[:m].view(-1, g).any(1)).double().mean()))
