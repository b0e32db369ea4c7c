import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from besst_amd import _lib, bamio, device
path = sys.argv[1]
lib = _lib.load()
os.environ['BESST_INGEST_PROFILE'] = '1'
for k in range(3):
    t = [time.perf_counter()]
    handle, refs, lens = bamio._open(lib, path, bamio.reader_threads()); t.append(time.perf_counter())
    ctx = device.GraphContext(0); t.append(time.perf_counter())
    z = np.zeros(len(refs), dtype=np.int32)
    ctx.set_contigs(scaf_id=z, scaf_len=z, ctg_pos=z, ctg_len=z, direction=z, cls=z); t.append(time.perf_counter())
    st = ctx.push_bam(handle, 4 << 20, mode='device'); t.append(time.perf_counter())
    lib.besst_bam_close(handle); t.append(time.perf_counter())
    print('open %.1f ms  ctx %.1f  set_contigs %.1f  push_bam %.1f  close %.1f  | total %.1f ms, %d records' % tuple(
        [(b - a) * 1e3 for a, b in zip(t[:-1], t[1:])] + [(t[-1] - t[0]) * 1e3, st[0].records]), flush=True)
    ctx.close()
