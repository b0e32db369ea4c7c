#!/bin/bash
cd "$(dirname "$0")/.."
for flags in "-DBESST_BUCKET_THREADS=64" "-DBESST_BUCKET_THREADS=128" "-DBESST_BUCKET_THREADS=256"; do
  BESST_EXTRA_FLAGS="$flags" besst_amd/csrc/build.sh > /dev/null 2>&1
  python bench.py --steps 21 --warmup 3 --no-stages --no-cpu-baseline --breakdown-steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('$flags', round(d['ms_per_step']*1000,1), k['bucket_sort_kernel'], d['verified_vs_c_oracle'])"
done
