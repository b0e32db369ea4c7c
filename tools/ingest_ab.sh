#!/bin/bash
# Development (GPU box): inflate kernel time of a device ingest by rocprofv3, for the library in BESST_AMD_LIB.
#   tools/ingest_ab.sh label [level] [pairs]
cd "$(dirname "$0")/.."
R=$PWD; O=/tmp/ingab_$1; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && PROBE_MODES=device:0,device:0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/ingest_probe.py C3 ${3:-20000000} - ${2:-17} > $O/stats.log 2>&1)
grep "records/s" $O/stats.log | tail -1 | cut -c1-60
python - <<PY
import csv, glob
f = glob.glob('$O/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    n = r['Name'].replace('besst::(anonymous namespace)::','')[:40]
    if 'bgzf' in n or 'bam_' in n:
        print('  $1 %-40s %6s calls %10.1f us avg %10.1f ms total' % (n, r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
