#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats of a short C3 bench run -> gpurun_out/$1/{bench.json,kernel_stats.csv}
# usage: tools/prof_c3.sh TAG [extra bench args...]   (environment knobs are inherited)
cd "$(dirname "$0")/.."
R=$PWD; TAG=${1:-prof}; shift
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --config C3 --also= --steps 10 --warmup 2 --no-cpu-baseline --breakdown-steps 0 --no-verify --no-stages --in-flight 0 "$@" > $O/bench.json 2> $O/bench.err)
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv 2>/dev/null
rm -rf $O/stats
python - <<PY
import json
try:
    d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
    print('$TAG ms_per_step', d['ms_per_step'])
except Exception as e:
    print('$TAG bench failed', e); print(open('$O/bench.err').read()[-1500:])
PY
python tools/kstats.py $O/kernel_stats.csv | head -24
