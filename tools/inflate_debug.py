import sys, zlib, struct
sys.path.insert(0, '.')
from tests import test_gpu_ingest as T
from besst_amd import bamio
pay = T._payloads()
cases = [(1, 0), (6, 0), (9, 0), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)]
names = sorted(pay)
if len(sys.argv) > 1:
    cases = [(int(sys.argv[1]), int(sys.argv[2]))]
    names = sys.argv[3:]
for level, strat in cases:
    for name in names:
        raw = pay[name]
        data = T._bgzf(raw, level, strat)
        try:
            got = bamio.inflate_bgzf_device(data, out_cap=len(raw) + 16)
            if got == raw:
                res = 'ok'
            else:
                n = min(len(got), len(raw))
                first = next((i for i in range(n) if got[i] != raw[i]), n)
                nd = sum(1 for i in range(n) if got[i] != raw[i])
                res = 'DIFF len %d/%d first %d ndiff %d' % (len(got), len(raw), first, nd)
        except Exception as e:
            res = 'ERR ' + str(e)[-60:]
        print(level, strat, name, len(data), res, flush=True)
