#!/bin/bash
cd "$(dirname "$0")/.."
for flags in "-DBESST_DBG_PHASE=0" "-DBESST_DBG_PHASE=1" "-DBESST_DBG_PHASE=2" "-DBESST_DBG_PHASE=9"; do
  BESST_EXTRA_FLAGS="$flags" besst_amd/csrc/build.sh > /dev/null 2>&1
  python bench.py --steps 11 --warmup 3 --no-stages --no-cpu-baseline --no-verify --breakdown-steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('$flags', round(d['ms_per_step']*1000,1), k['candidate_kernel'])"
done
