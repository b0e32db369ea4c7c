"""Development probe: which kernels DeviceGraphBuilder.reset() launches (run under tools/kstats_cmd.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from besst_amd import pipeline
dev = torch.device('cuda', 0)
lib = dict(read_len=100.0, ins_size_threshold=800.0, min_mapq=11, orientation='fr', detect_duplicate=True, extend_paths=True, no_score=False)
gb = pipeline.DeviceGraphBuilder(dev, 100000, 18, lib, 1 << 20, 1 << 20)
torch.cuda.synchronize()
for _ in range(50):
    gb.reset()
torch.cuda.synchronize()
