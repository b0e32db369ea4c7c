#!/bin/bash
# usage: met_ab.sh label
cd /root/repo
O=/tmp/metab_$1; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python /root/repo/tools/metrics_probe.py 30000000 > $O/stats.log 2>&1)
tail -1 $O/stats.log | cut -c1-300
python - <<PY
import csv, glob
f = glob.glob('$O/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'metrics_' in r['Name']:
        print('  $1 %-28s %4s calls %9.1f us avg %9.1f min' % (r['Name'].split('::')[-1][:28], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
