"""Development probe: the library-metrics pass on a C3-shaped library (prefix scan and forced full scan), kernel time by HIP events."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from besst_amd import workload
wl = workload.make("C3", 0, pairs=int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000, nc=20000)
st = bench.stage_timings(wl)
print(json.dumps({k: st[k] for k in st if k.startswith("metrics_roofline")}))
