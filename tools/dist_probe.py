"""Development probe: is the sharded step (forced world = 1 over RCCL) host-bound?  Host enqueue time vs wall time,
and the time of each stage when synchronised individually."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
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
b = job.backend
def timed(name, fn, reps=30):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    h = time.perf_counter(); torch.cuda.synchronize(); e = time.perf_counter()
    print('%-22s host %.1f us  wall %.1f us' % (name, (h - t) / reps * 1e6, (e - t) / reps * 1e6))
tails = [torch.empty_like(b.tail)]
recv = torch.empty_like(b.send)
timed('reset', b.reset)
timed('classify_scan', b.classify_scan)
timed('classify_tail', b.classify_tail)
flat = torch.zeros(4, dtype=torch.int32, device=dev)
timed('all_gather_into_tensor', lambda: dist.all_gather_into_tensor(flat, b.tail))
timed('all_gather(16B)', lambda: dist.all_gather(tails, b.tail))
timed('classify_emit', lambda: b.classify_emit(flat))
timed('all_reduce(80KB)', lambda: dist.all_reduce(b.pack_for_allreduce()))
timed('partition', b.partition)
timed('all_to_all', lambda: dist.all_to_all_single(recv, b.send))
timed('unpack', lambda: b.unpack(recv))
timed('reduce', b.reduce)
dist.destroy_process_group()
