#!/bin/bash
# Development helper (GPU box): one rocprofv3 --pmc pass per counter group over an arbitrary python command.
#   tools/pmc_cmd.sh "tools/sort_probe.py 18 5000000" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES ..."
cd "$(dirname "$0")/.."
R=$PWD; export TMPDIR=/tmp
CMD="$1"; shift
i=0
for grp in "$@"; do
  O=/tmp/pmcc_$i; rm -rf $O
  (cd /tmp && timeout ${PMC_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O -- python $R/$CMD > /tmp/pmcc_$i.log 2>&1)
  python - "$O" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    with open(path, newline='') as fh:
        for row in csv.DictReader(fh):
            name = re.sub(r'\(.*$', '', row['Kernel_Name'].replace('besst::(anonymous namespace)::', '').replace('void ', '')).strip()
            if 'kernel' not in name or 'at::' in name or 'elementwise' in name or 'rocprim' in name:
                continue
            acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
for k in sorted(acc):
    print('%-40s' % k[:40], '  '.join('%s=%.4g' % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())))
PY
  i=$((i+1))
done
