"""Development probe: eager vs hipGraph replay of one graph-build step (not part of the product)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from besst_amd import pipeline, workload
wl = workload.make('C2', 0)
dev = torch.device('cuda', 0)
recs = [pipeline.DeviceRecords(wl['batch'], dev) for _ in range(3)]
gb = pipeline.DeviceGraphBuilder(dev, wl['asm'].nc, wl['node_bits'], wl['lib'], recs[0].n, 200000)
gb.set_contigs(**wl['table'])
for r in recs: gb.step(r)
torch.cuda.synchronize()
def timeit(fn, n=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print('eager  us/step', timeit(lambda i: gb.step(recs[i % 3])))
graphs = []
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for r in recs:
        gb.step(r)
    torch.cuda.synchronize()
    for r in recs:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            gb.step(r)
        graphs.append(g)
print('graph  us/step', timeit(lambda i: graphs[i % 3].replay()))
print(gb.read_sizes())
