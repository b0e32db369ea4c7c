#!/bin/bash
# Development helper (GPU box): rocprofv3 kernel stats of a short bench run, top kernels printed.   tools/kstats.sh [bench args]
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/kstats; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
ARGS="${@:---config C3 --also= --steps 5 --warmup 1 --no-cpu-baseline --breakdown-steps 0 --no-verify --no-stages --in-flight 0}"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py $ARGS > $O/stats.log 2>&1)
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv; rm -rf $O/stats
python - <<PY
import csv
for r in list(csv.DictReader(open('$O/kernel_stats.csv')))[:22]:
    print('%-90s %6s %10.1f us' % (r['Name'].replace('besst::(anonymous namespace)::','')[:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
