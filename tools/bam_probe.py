"""Development probe: decode rate of the native BAM front-end (besst_bam_*) on this host's cores."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from besst_amd import bamio, workload
from tests import bam_writer
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
wl = workload.make('C2', 0, pairs=n_pairs, nc=2000)
path = os.path.join(tempfile.mkdtemp(), 'probe.bam')
t0 = time.perf_counter(); bam_writer.write_bam(path, wl['batch'], align_records=True); t1 = time.perf_counter()   # htslib layout
size = os.path.getsize(path)
print('wrote %d records, %.1f MB BAM in %.1f s (python writer)' % (len(wl['batch']), size / 1e6, t1 - t0))
for threads in (1, 8, 32, os.cpu_count()):
    t0 = time.perf_counter(); b = bamio.read_bam(path, threads=threads); dt = time.perf_counter() - t0
    print('threads %3d: %.3f s  %.1f M records/s  %.0f MB/s compressed' % (threads, dt, len(b) / dt / 1e6, size / dt / 1e6))
import numpy as np
from besst_amd import _lib
lib = _lib.load()
spec = (np.int32, np.int32, np.int32, np.int32, np.int32, np.uint16, np.uint8, np.uint16, np.int32, np.int32)
cap = len(wl['batch']) + 16
bufs = [np.zeros(cap, dtype=dt) for dt in spec]          # pre-touched: the native call alone
for threads in (1, 8, 32, 64):
    h = lib.besst_bam_open(os.fsencode(path), threads)
    t0 = time.perf_counter(); got = lib.besst_bam_read_records(h, cap, *[_lib.ptr(x) for x in bufs]); dt = time.perf_counter() - t0
    lib.besst_bam_close(h)
    print('native, threads %3d: %.3f s  %.1f M records/s' % (threads, dt, got / dt / 1e6))
assert (b.tid == wl['batch'].tid).all() and (b.pos == wl['batch'].pos).all()
