"""Development probe (GPU box): where the time of bamio.ResidentBam(path) goes outside besst_ctx_push_bam_device.
usage: python tools/resident_steps.py [pairs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from besst_amd import _lib, bamio, device, workload

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 20000000
wl = workload.make_device(torch.device('cuda', 0), 'C3', 0, pairs=pairs)
path = '/dev/shm/resident_steps.bam'
bamio.write_bam(path, wl['batch'], level=17)
del wl
lib = _lib.load()
for rep in range(3):
    t = [time.perf_counter()]
    handle, refs, lens = bamio._open(lib, path, bamio.reader_threads()); t.append(time.perf_counter())
    ctx = device.GraphContext(0); t.append(time.perf_counter())
    zeros = np.zeros(len(refs), dtype=np.int32)
    ctx.set_contigs(scaf_id=zeros, scaf_len=zeros, ctg_pos=zeros, ctg_len=zeros, direction=zeros, cls=zeros); t.append(time.perf_counter())
    stats, rlen, alen, qlen = ctx.push_bam(handle, 4 << 20); t.append(time.perf_counter())
    lib.besst_bam_close(handle); t.append(time.perf_counter())
    ctx.close(); t.append(time.perf_counter())
    names = ('open', 'context', 'set_contigs', 'push_bam', 'bam_close', 'ctx.close')
    print('  '.join('%s %.1f ms' % (n, 1e3 * (b - a)) for n, a, b in zip(names, t, t[1:])), '| in the call %.1f ms' % (1e3 * stats.seconds), flush=True)
os.remove(path)
