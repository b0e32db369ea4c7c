"""Development probe: time of the sort/reduce stage alone (besst_dev_reduce) on synthetic tuple streams of growing size.
usage: sort_probe.py [node_bits] [sizes ...]; BESST_PROBE_SORTED=1: keys nearly sorted by min node, like a BAM-ordered stream."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from besst_amd import pipeline
dev = torch.device('cuda', 0)
lib = dict(read_len=100.0, ins_size_threshold=800.0, min_mapq=11, orientation='fr', detect_duplicate=True,
           extend_paths=True, no_score=False)
node_bits = int(sys.argv[1]) if len(sys.argv) > 1 else 21        # key_bits = 43 (1 M scaffolds)
sizes = [int(x) for x in sys.argv[2:]] or [130_000, 260_000, 300_000, 600_000, 1_000_000, 2_000_000, 4_000_000, 8_000_000]
per_row = int(os.environ.get('BESST_PROBE_PER_ROW', '15'))
for n in sizes:
    rng = np.random.default_rng(n)
    rows = max(1000, n // per_row)                               # ~15 links per edge, like a PE library
    if os.environ.get('BESST_PROBE_SORTED') == '1':
        a = np.sort(rng.integers(1, (1 << node_bits) - 8, n, dtype=np.int64))
        b = a + rng.integers(1, 8, n)
        flip = rng.random(n) < 0.5                               # own node is the larger one for half the tuples
        a2 = np.where(flip, a - rng.integers(0, 6, n), a); b2 = np.where(flip, a + 1, b)
        pair = (np.maximum(a2, 0) << node_bits) | b2
    else:
        pair = rng.integers(0, 1 << (2 * node_bits), rows, dtype=np.int64)[rng.integers(0, rows, n)]
        noise = float(os.environ.get('BESST_PROBE_NOISE', '0'))          # share of tuples on edges of their own (chimeras)
        if noise > 0:
            m = rng.random(n) < noise
            pair[m] = rng.integers(0, 1 << (2 * node_bits), int(m.sum()), dtype=np.int64)
    keys = (pair << 1).astype(np.uint64)
    lo = rng.integers(26, 5000, n).astype(np.uint64); hi = rng.integers(26, 5000, n).astype(np.uint64) | (np.uint64(3) << np.uint64(30))
    payload = lo | (hi << np.uint64(32))
    gb = pipeline.DeviceGraphBuilder(dev, 4, node_bits, lib, n, n)
    dk = torch.from_numpy(keys.view(np.int64)).to(dev); dp = torch.from_numpy(payload.view(np.int64)).to(dev)
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    args = dict(keys=dk, payload=dp, n_tuples_ptr=C.c_void_p(cnt.data_ptr()), capacity=n)
    for _ in range(3): gb.reduce(**args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): gb.reduce(**args)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    gb.lib.besst_prof_enable(0xffffffff)
    for _ in range(3): gb.reduce(**args)
    torch.cuda.synchronize()
    prof = {k: round(v[0] / 3 * 1e3, 1) for k, v in pipeline.prof_collect().items()}
    gb.lib.besst_prof_enable(0)
    print('%9d tuples  %8.1f us  %6.1f M tuples/s  %s' % (n, dt * 1e6, n / dt / 1e6, prof), flush=True)
    del gb, dk, dp
