"""Do independent library passes overlap when issued on separate HIP streams?  (S builders, round-robin steps;
eager launches vs one captured hipGraph per (builder, record copy); stream_kernel duration under overlap.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from besst_amd import _lib, pipeline, workload

dev = torch.device('cuda', 0)
wl = workload.make('C2', 0)
n = len(wl['batch'])
lib = _lib.load()
recs = [pipeline.DeviceRecords(wl['batch'], dev) for _ in range(4)]
for S in (1, 2, 3, 4):
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    gbs = []
    for s in streams:
        with torch.cuda.stream(s):
            gb = pipeline.DeviceGraphBuilder(dev, wl['asm'].nc, wl['node_bits'], wl['lib'], n, 200_000)
            gb.set_contigs(**wl['table'])
            gbs.append(gb)
    torch.cuda.synchronize()
    def run(k):
        for i in range(k):
            with torch.cuda.stream(streams[i % S]):
                gbs[i % S].step(recs[i % S])
    def timeit(fn, K=48):
        fn(6); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(K); torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / K * 1e6, 1)
    eager = timeit(run)
    lib.besst_prof_enable(1)
    ev = timeit(run)
    prof = pipeline.prof_collect()
    lib.besst_prof_enable(0)
    sk = prof['stream_kernel']
    graphs = []
    for j, s in enumerate(streams):
        with torch.cuda.stream(s):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                gbs[j].step(recs[j])
            graphs.append(g)
    def rung(k):
        for i in range(k):
            with torch.cuda.stream(streams[i % S]):
                graphs[i % S].replay()
    gr = timeit(rung)
    print('streams', S, 'eager us/step', eager, 'with events', ev, 'stream_kernel us', round(sk[0] / sk[1] * 1e3, 1),
          'graph us/step', gr, 'sizes', [g.read_sizes() for g in gbs])
    del gbs, graphs
    torch.cuda.empty_cache()
