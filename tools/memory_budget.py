"""Per-GPU HBM budget of the sharded graph build for BASELINE.json configs[3] (C4) and configs[4] (C5) on eight GPUs
(besst_amd.distributed.memory_budget; no GPU needed).  usage: python tools/memory_budget.py [world]

Pair capacity as ShardedGraphBuild._probe_pair_capacity sizes it: 1.5 x the slice's tuples / world + 4096, with the
tuples per record measured on the libraries' single-GPU streams (PE 500 bp: 0.007 per record, MP with PE
contamination: 0.107 per record; twice that is budgeted)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from besst_amd import distributed, synth

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
TUPLES_PER_RECORD = {'fr': 2 * 0.007, 'rf': 2 * 0.107}
for config in ('C4', 'C5'):
    cfg = synth.CONFIGS[config]
    per_lib = cfg['pairs'] // len(cfg['libs']) // world
    total = 0
    print('%s: %d contigs, %d libraries, %d read pairs per library and GPU (world %d)' % (config, cfg['nc'], len(cfg['libs']), per_lib, world))
    for li, spec in enumerate(cfg['libs']):
        n_rec = 2 * per_lib
        tuples = int(n_rec * TUPLES_PER_RECORD[spec.orientation])
        pair_cap = int(tuples * 1.5 / world) + 4096
        b = distributed.memory_budget(n_rec, cfg['nc'], world, pair_cap, int(tuples * 1.25) + 4096)
        total += b['total']
        print('  library %d (%s %g): %.2f GB' % (li + 1, spec.orientation, spec.mean, b['total'] / 1e9))
        for k, v in b['items'].items():
            print('      %-62s %8.3f GB' % (k, v / 1e9))
    print('  all libraries resident at once: %.1f GB of 288 GB per GPU\n' % (total / 1e9))
