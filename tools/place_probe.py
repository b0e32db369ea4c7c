"""Does the record loop's time depend on where the columns lie?  One process, the same records re-allocated several times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (the repository root)
import torch
import bench
from besst_amd import pipeline, workload, _lib
dev = torch.device('cuda', 0)
wl = workload.make_device(dev, 'C3', 0)
runner = bench.SingleGpu(dev, wl, 1)
lib = _lib.load()
def timed(rec, label):
    for _ in range(3): runner.gb.step(rec)
    torch.cuda.synchronize()
    lib.besst_prof_enable(0xffffffff); pipeline.prof_collect()
    t0 = time.perf_counter()
    for _ in range(10): runner.gb.step(rec)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    p = pipeline.prof_collect(); lib.besst_prof_enable(0)
    print('%-28s step %.4f ms  record loop %.4f ms  ptr tid %x' % (label, dt * 1e3, p['fused_wave_kernel'][0] / p['fused_wave_kernel'][1], rec.tid.data_ptr()), flush=True)
rec = runner.rec
timed(rec, 'as generated')
keep = []
# one arena, columns at staggered offsets
n = rec.n
sizes = {'tid': 4, 'mtid': 4, 'pos': 4, 'mpos': 4, 'tlen': 4, 'flag': 2, 'mapq': 1, 'qlen': 2}
for stagger in [int(x) for x in os.environ.get('STAGGERS', '0,1024,8192,16384,32768,65536,66560,131072,262144,524288,1048576,66560,0').split(',')]:
    del keep[:]
    torch.cuda.empty_cache()
    total = sum(((n * s + (2 << 20) - 1) // (2 << 20)) * (2 << 20) + (4 << 20) for s in sizes.values()) + 16 * stagger + (1 << 20)
    arena = torch.empty(total, dtype=torch.uint8, device=dev)
    off, cols, j = 0, {}, 0
    for c, s in sizes.items():
        o = off + j * stagger
        o = (o + 255) // 256 * 256
        view = arena[o:o + n * s].view({4: torch.int32, 2: torch.int16 if c != 'flag' and c != 'qlen' else torch.int16, 1: torch.uint8}[s])
        src = getattr(rec, c)
        view.view(torch.uint8).copy_(src.view(torch.uint8))
        cols[c] = view.view(src.dtype)
        off += ((n * s + (2 << 20) - 1) // (2 << 20)) * (2 << 20) + (4 << 20)
        j += 1
    new = pipeline.DeviceRecords.from_columns(cols)
    timed(new, 'arena stagger %d' % stagger)
    del new, cols, arena
