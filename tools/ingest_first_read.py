"""Development probe (GPU box): is the slow FIRST ingest of a file about the file (page cache state right after it was
written) or about the process (fresh pinned slots, thread pool, runtime warm-up)?  usage:
python tools/ingest_first_read.py write <path> [pairs]   |   python tools/ingest_first_read.py read <path> [times]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
what, path = sys.argv[1], sys.argv[2]
if what == 'write':
    import torch
    from besst_amd import bamio, workload
    wl = workload.make_device(torch.device('cuda', 0), 'C3', 0, pairs=int(sys.argv[3]) if len(sys.argv) > 3 else 20_000_000)
    batch = wl['batch']
    t0 = time.perf_counter(); bamio.write_bam(path, batch, level=17); print('written %d records, %.2f GB in %.1f s' % (len(batch), os.path.getsize(path) / 1e9, time.perf_counter() - t0))
else:
    from besst_amd import bamio
    os.environ['BESST_INGEST_PROFILE'] = '1'
    for k in range(int(sys.argv[3]) if len(sys.argv) > 3 else 2):
        t0 = time.perf_counter()
        bam = bamio.ResidentBam(path, mode='device')
        dt = time.perf_counter() - t0
        s = bam.ingest
        print('read %d: %.3f s = %.1f M records/s; staging %.3f wait %.3f chunks %d' % (k, dt, len(bam) / dt / 1e6, s.decode_seconds, s.copy_wait_seconds, s.chunks), flush=True)
        bam.close()
