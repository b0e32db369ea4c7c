"""Development probe: the sharded step (one rank over RCCL, default modes) in a loop, for rocprofv3 --kernel-trace +
tools/timeline.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29534')
import torch, torch.distributed as dist
from besst_amd import distributed, workload
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
wl = workload.make('C2', 0)
job = distributed.ShardedGraphBuild(dev, wl, 0, 1)
for _ in range(5): job.step()
torch.cuda.synchronize()
K = 50
t0 = time.perf_counter()
for _ in range(K): job.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host enqueue us/step %.1f   wall us/step %.1f' % ((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
dist.destroy_process_group()
