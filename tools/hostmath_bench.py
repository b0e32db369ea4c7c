import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from besst_amd import libmetrics, _lib
rng=np.random.default_rng(1)
vals=np.abs(rng.normal(3000,300,1_000_000)).astype(np.int32)
for is_float in (1,0):
    for k in range(4):
        t=time.perf_counter(); r=libmetrics._native_isize_stats(vals,is_float,200.76); dt=time.perf_counter()-t
        print(is_float, round(dt*1e3,1),'ms', len(r[1]), r[2][2:4])
print('effective cpus', _lib.effective_cpus(), 'of', os.cpu_count())
