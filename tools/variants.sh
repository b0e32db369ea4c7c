#!/bin/bash
# Development helper (runs on the GPU box): rebuild the library with each flag set and print the step and kernel times.
#   tools/variants.sh "-DBESST_CAND_GROUPS=32" "-DBESST_CAND_GROUPS=16"       (BENCH_ARGS: extra bench.py arguments)
cd "$(dirname "$0")/.."
for flags in "$@"; do
  BESST_EXTRA_FLAGS="$flags" besst_amd/csrc/build.sh > /dev/null 2>&1
  python bench.py ${BENCH_ARGS:---steps 10 --warmup 2 --also=} --no-stages --cpu-sample-records 0 --breakdown-steps 3 --in-flight 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('$flags', round(d['ms_per_step']*1000,1), {a: round(b*1000) for a,b in k.items()}, d['verified_vs_c_oracle'])"
done
besst_amd/csrc/build.sh > /dev/null 2>&1
