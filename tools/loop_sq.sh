#!/bin/bash
# Runs on the GPU box: SQ counters of the record-loop kernels (fused_wave_kernel on C3, stream_kernel / ordered_kernel on C2),
# one rocprofv3 --pmc pass per counter group (--kernel-trace only), summarised per kernel as means per dispatch.
#   tools/loop_sq.sh [out.json]   -> gpurun_out/r06/loop_sq.json
cd "$(dirname "$0")/.."
R=$PWD; export TMPDIR=/tmp
OUT=${1:-$R/gpurun_out/r06/loop_sq.json}; mkdir -p "$(dirname "$OUT")"
GROUPS_=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM"
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH"
         "GRBM_GUI_ACTIVE GRBM_COUNT SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES"
         "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum")
rm -rf /tmp/lsq; mkdir -p /tmp/lsq
for CFG in C3 C2; do
  ARGS="--config $CFG --also= --steps 5 --warmup 1 --no-cpu-baseline --breakdown-steps 0 --no-verify --no-stages --in-flight 0 --no-robustness"
  i=0
  for grp in "${GROUPS_[@]}"; do
    O=/tmp/lsq/${CFG}_$i
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O -- python $R/bench.py $ARGS > $O.log 2>&1) || echo "group $i of $CFG failed: $(tail -2 $O.log)"
    i=$((i+1))
  done
done
python - "$OUT" <<'PY'
import csv, glob, json, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(list)))
for d in sorted(glob.glob('/tmp/lsq/C*_*')):
    if not os.path.isdir(d):
        continue
    cfg = os.path.basename(d).split('_')[0]
    for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(path, newline='') as fh:
            for row in csv.DictReader(fh):
                name = re.sub(r'\(.*$', '', row['Kernel_Name'].replace('besst::(anonymous namespace)::', '').replace('void ', '')).strip()
                if not any(k in name for k in ('fused_wave_kernel', 'stream_kernel', 'ordered_kernel', 'rl_place_kernel')):
                    continue
                acc[cfg][name][row['Counter_Name']].append(float(row['Counter_Value']))
doc = {'_about': 'rocprofv3 --pmc, one pass per counter group, --kernel-trace only (tools/loop_sq.sh); means per dispatch over the '
                 'dispatches of `python bench.py --config <cfg> --steps 5 --warmup 1 ...`; SQ cycle counters are quad-cycles '
                 '(MI355X_MICROARCH.md), summed over the chip'}
for cfg in acc:
    doc[cfg] = {k: dict({c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}, dispatches=max(len(v) for v in cs.values()))
                for k, cs in acc[cfg].items()}
    for k, c in doc[cfg].items():
        if 'SQ_WAVE_CYCLES' in c and c['SQ_WAVE_CYCLES']:
            w = c['SQ_WAVE_CYCLES']
            c['derived'] = {'wait_any_of_wave_cycles': round(c.get('SQ_WAIT_ANY', 0) / w, 3),
                            'wait_inst_of_wave_cycles': round(c.get('SQ_WAIT_INST_ANY', 0) / w, 3),
                            'active_inst_of_wave_cycles': round(c.get('SQ_ACTIVE_INST_ANY', 0) / w, 3),
                            'mean_waves_resident': round(w / c['SQ_BUSY_CYCLES'], 2) if c.get('SQ_BUSY_CYCLES') else None}
json.dump(doc, open(sys.argv[1], 'w'), indent=1)
print(json.dumps(doc, indent=1)[:6000])
PY
