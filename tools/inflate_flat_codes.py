"""Development probe (GPU box): the inflate kernel on blocks whose codes all have ONE length (four letters, Huffman only: a
decoder that starts inside a symbol never falls into step) - the worst case of the second form's hand-overs.
usage: python tools/inflate_flat_codes.py [blocks]"""
import os, random, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from besst_amd import bamio
from tests.test_gpu_ingest import _bgzf

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rnd = random.Random(1)
kinds = {'acgt_huffman_only': lambda: _bgzf(bytes(rnd.choice(b'ACGT') for _ in range(60000)), 6, zlib.Z_HUFFMAN_ONLY),
         'sixteen_values_huffman_only': lambda: _bgzf(bytes(rnd.randrange(16) for _ in range(60000)), 6, zlib.Z_HUFFMAN_ONLY)}
for name, make in kinds.items():
    some = [make() for _ in range(16)]
    data = b''.join(some[i % 16] for i in range(n))
    for _ in range(2):
        t0 = time.perf_counter()
        out = bamio.inflate_bgzf_device(data, out_cap=60000 * n + 16)
        dt = time.perf_counter() - t0
    print('%s: %d blocks, %.3f s in the call (form %s)' % (name, n, dt, os.environ.get('BESST_INFLATE', '2')), flush=True)
