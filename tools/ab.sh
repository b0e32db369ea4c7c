#!/bin/bash
# Development: the headline step with several builds / knobs on ONE box.  tools/ab.sh "label:ENV=.. ENV=.." ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for spec in "$@"; do
    label="${spec%%:*}"; envs="${spec#*:}"
    env ${envs} python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-stages --no-robustness --no-verify --also "" \
        > "gpurun_out/ab_${label}.json" 2> "gpurun_out/ab_${label}.err"
    python - "$label" <<'PY'
import json, sys
label = sys.argv[1]
try:
    d = json.loads(open('gpurun_out/ab_%s.json' % label).read().strip().splitlines()[-1])
    k = d['kernel_ms']
    print('%-10s step %.4f  3fl %.4f | %s' % (label, d['ms_per_step'], d.get('overlapped', {}).get('ms_per_step', 0),
          '  '.join('%s %.4f' % (n.split('_kernel')[0][:14], v) for n, v in k.items())))
except Exception as e:
    print(label, 'FAILED', e, open('gpurun_out/ab_%s.err' % label).read()[-800:])
PY
done
