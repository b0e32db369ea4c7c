"""Development probe (GPU box): BAM file -> resident records through besst_ctx_push_bam, by reader threads and chunk size.
usage: python tools/ingest_probe.py [config] [pairs] [dir]      (the BAM is written with the native writer first)"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from besst_amd import bamio, workload

config = sys.argv[1] if len(sys.argv) > 1 else 'C2'
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else None
where = sys.argv[3] if len(sys.argv) > 3 else tempfile.mkdtemp()
dev = torch.device('cuda', 0)
wl = workload.make_device(dev, config, 0, pairs=pairs)
batch = wl['batch']
del wl['cols']
torch.cuda.empty_cache()
path = os.path.join(where, 'probe_%s.bam' % config)
t0 = time.perf_counter(); bamio.write_bam(path, batch, threads=min(128, os.cpu_count() or 1)); dt = time.perf_counter() - t0
size = os.path.getsize(path)
n = len(batch)
print('%d records -> %.2f GB BAM in %.1f s (native writer), %.1f B/record compressed, host cores %d' % (
    n, size / 1e9, dt, size / n, os.cpu_count()), flush=True)
for threads, chunk in ((16, 4 << 20), (32, 4 << 20), (64, 4 << 20), (128, 4 << 20), (64, 1 << 20), (64, 16 << 20), (128, 16 << 20)):
    if threads > (os.cpu_count() or 1):
        continue
    t0 = time.perf_counter()
    bam = bamio.ResidentBam(path, threads=threads, chunk_records=chunk)
    dt = time.perf_counter() - t0
    s = bam.ingest
    print('threads %3d chunk %8d: %.3f s = %.1f M records/s (%.2f GB/s compressed) | in the call %.3f s: decode %.3f, waiting for '
          'copies %.3f, %d chunks, %.1f GB to HBM' % (threads, chunk, dt, n / dt / 1e6, size / dt / 1e9, s.seconds, s.decode_seconds,
                                                       s.copy_wait_seconds, s.chunks, s.bytes_h2d / 1e9), flush=True)
    bam.close()
t0 = time.perf_counter(); b2 = bamio.read_bam(path, threads=64); dt = time.perf_counter() - t0
print('read_bam to host columns (64 threads): %.3f s = %.1f M records/s' % (dt, n / dt / 1e6))
os.remove(path)
