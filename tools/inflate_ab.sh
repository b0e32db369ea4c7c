#!/bin/bash
# Development (GPU box): kernel time of the inflate kernel alone.  tools/inflate_ab.sh [pairs]
cd "$(dirname "$0")/.."
R=$PWD; O=/tmp/infab; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/inflate_time.py ${1:-3000000} $2 > $O/stats.log 2>&1)
grep -v "^[EW]2026" $O/stats.log | tail -5
python - <<PY
import csv, glob
f = glob.glob('$O/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:3]:
    print('%-50s %6s calls %10.1f us avg %10.1f ms total' % (r['Name'].replace('besst::(anonymous namespace)::','')[:50], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
