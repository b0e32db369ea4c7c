import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from besst_amd import workload
for nc in (600, 5000):
    wl = workload.make("C2", 0, pairs=20_000_000, nc=nc)
    st = bench.stage_timings(wl)
    print(nc, json.dumps({k: (st[k]['kernel_ms'], st[k]['frac'], st[k]['records_scanned']) for k in st if k.startswith("metrics_roofline")}))
