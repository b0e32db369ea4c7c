"""Development probe: run tools/probe_stream.hip variants on the C2 columns (not part of the product)."""
import ctypes as C, os, subprocess, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, 'probe_stream.so')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
                       os.path.join(here, 'probe_stream.hip'), '-o', so])
lib = C.CDLL(so)
n = 20_000_000 // 8192 * 8192
rng = np.random.default_rng(0)
tid = np.sort(rng.integers(0, 10000, n)).astype(np.int32)
mtid = tid.copy()
sw = rng.random(n) < 0.015
mtid[sw] = np.minimum(tid[sw] + 1, 9999)
dev = torch.device('cuda', 0)
T = torch.from_numpy(tid).to(dev); M = torch.from_numpy(mtid).to(dev)
Q = torch.from_numpy(rng.choice([60, 0, 30], n, p=[.8, .1, .1]).astype(np.uint8)).to(dev)
L = torch.full((n,), 100, dtype=torch.int16, device=dev)
al = torch.zeros(10000, dtype=torch.int64, device=dev)
bm = torch.zeros(n // 64 + 64, dtype=torch.int64, device=dev)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
p = lambda t: C.c_void_p(t.data_ptr())
lib.probe_all(p(T), p(M), p(Q), p(L), C.c_int64(n), p(al), p(bm), p(sink))
print('bytes level0 %.1f MB, level>=1 %.1f MB' % (n * 4 / 1e6, n * 11 / 1e6))
