"""Development probe (GPU box): BAM file -> resident records, device form (besst_ctx_push_bam_device: inflate + decode on
the GPU) against host form (besst_ctx_push_bam: reader threads + pinned staging), by chunk size / threads.
usage: python tools/ingest_probe.py [config] [pairs] [dir] [level]      (the BAM is written with the native writer first)"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from besst_amd import _lib, bamio, workload

config = sys.argv[1] if len(sys.argv) > 1 else 'C2'
pairs = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else None
where = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != '-' else ('/dev/shm' if os.path.isdir('/dev/shm') else tempfile.mkdtemp())
level = int(sys.argv[4]) if len(sys.argv) > 4 else 1          # + 16: sequencer-like bases and qualities
dev = torch.device('cuda', 0)
wl = workload.make_device(dev, config, 0, pairs=pairs)
batch = wl['batch']
del wl['cols']
torch.cuda.empty_cache()
path = os.path.join(where, 'probe_%s.bam' % config)
t0 = time.perf_counter(); bamio.write_bam(path, batch, level=level); dt = time.perf_counter() - t0
size = os.path.getsize(path)
n = len(batch)
print('%d records -> %.2f GB BAM in %.1f s (native writer, level %d), %.1f B/record compressed, usable cpus %d of %d' % (
    n, size / 1e9, dt, level, size / n, _lib.effective_cpus(), os.cpu_count()), flush=True)
ref = None
modes = os.environ.get('PROBE_MODES', 'device:0,device:0,device:4096,device:65536,host:0,host:0')     # mode:blocks per chunk
for mode, blocks, threads in [(m.split(':')[0], int(m.split(':')[1]), 0) for m in modes.split(',')]:
    t0 = time.perf_counter()
    bam = bamio.ResidentBam(path, threads=threads or None, mode=mode, chunk_blocks=blocks)
    dt = time.perf_counter() - t0
    s = bam.ingest
    print('%-6s blocks/chunk %5d: %.3f s = %6.1f M records/s (%.2f GB/s compressed) | in the call %.3f s: staging / decode %.3f, waiting '
          '%.3f, %d chunks, %.2f GB to HBM, %.2f GB inflated, %d blocks, %d starts repaired' % (
              mode, blocks, dt, n / dt / 1e6, size / dt / 1e9, s.seconds, s.decode_seconds, s.copy_wait_seconds, s.chunks,
              s.bytes_h2d / 1e9, s.inflated_bytes / 1e9, s.blocks, s.starts_repaired), flush=True)
    if ref is None:
        ref = bam.ctx.fetch_records()
        for k in ('tid', 'mtid', 'pos', 'mpos', 'tlen', 'flag', 'mapq', 'qlen'):
            assert np.array_equal(ref[k], getattr(batch, k)), k
        print('   device form: all %d records equal to the batch the file was written from' % n, flush=True)
    bam.close()
os.remove(path)
