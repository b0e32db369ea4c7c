"""Development probe (GPU box): the inflate kernel alone over a sequencer-like BAM (level 17), through the test hook
besst_bgzf_inflate_device (4096 blocks per launch, synchronous: the kernel's durations under rocprofv3 are its own).
usage: python tools/inflate_time.py [pairs] [check]"""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from besst_amd import bamio, workload

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 3000000
check = len(sys.argv) > 2 and sys.argv[2] == 'check'
dev = torch.device('cuda', 0)
wl = workload.make_device(dev, 'C3', 0, pairs=pairs)
path = '/dev/shm/inflate_time.bam'
bamio.write_bam(path, wl['batch'], level=17)
data = open(path, 'rb').read()
os.remove(path)
cap = 8 * len(data)
for _ in range(3):
    t0 = time.perf_counter()
    try:
        got = bamio.inflate_bgzf_device(data, out_cap=cap)
    except Exception as e:                                   # (a variant that skips a phase: the first launch only)
        print(str(e)[-80:])
        got = b''
    dt = time.perf_counter() - t0
    print('%d -> %d bytes, %.3f s in the call' % (len(data), len(got), dt), flush=True)
if check:
    import gzip, io
    want = gzip.GzipFile(fileobj=io.BytesIO(data)).read()
    print('equal to zlib:', want == got)
