#!/bin/bash
# Runs on the GPU box: one rocprofv3 --pmc pass per counter group over a short bench run; prints per-kernel means.
#   tools/pmc_probe.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" ...
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp
ARGS=${BENCH_ARGS:---steps 3 --warmup 1 --also= --no-stages --no-cpu-baseline --no-verify --breakdown-steps 0 --in-flight 0}
i=0
for grp in "$@"; do
  O=/tmp/pmc_$i; rm -rf $O
  (cd /tmp && rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O -- python $R/bench.py $ARGS > /tmp/pmc_$i.log 2>&1)
  python - "$O" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    with open(path, newline='') as fh:
        for row in csv.DictReader(fh):
            name = re.sub(r'\(.*$', '', row['Kernel_Name'].replace('besst::(anonymous namespace)::', '').replace('void ', '')).strip()
            if 'kernel' not in name or 'at::' in name or 'elementwise' in name:
                continue
            acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
for k in sorted(acc):
    print('%-46s' % k[:46], '  '.join('%s=%.4g' % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())), ' n=%d' % len(next(iter(acc[k].values()))))
PY
  i=$((i+1))
done
