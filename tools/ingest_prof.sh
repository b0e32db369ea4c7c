#!/bin/bash
# Development helper (GPU box): rocprofv3 kernel stats of a device ingest.   tools/ingest_prof.sh [config] [pairs] [level]
# (level 17, the default: sequencer-like bases and qualities at zlib level 1 - what a real BAM compresses like)
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/ingest_prof; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && PROBE_MODES=device:0,device:0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/ingest_probe.py ${1:-C3} ${2:-50000000} - ${3:-17} > $O/stats.log 2>&1)
tail -4 $O/stats.log
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv; rm -rf $O/stats
python - <<PY
import csv
for r in list(csv.DictReader(open('$O/kernel_stats.csv')))[:12]:
    print('%-70s %6s calls %10.1f us avg %10.1f ms total' % (r['Name'].replace('besst::(anonymous namespace)::','')[:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
