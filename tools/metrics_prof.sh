#!/bin/bash
# Development helper (GPU box): the library-metrics pass (tools/metrics_probe.py: prefix scan + forced full scan of a C3-shaped
# library) under rocprofv3 - kernel stats, then FETCH_SIZE / WRITE_SIZE of metrics_stage_kernel, the pass's one read of the records (separate --pmc passes).
#   tools/metrics_prof.sh [pairs]   -> gpurun_out/metrics_prof/{kernel_stats.csv,pmc.json,probe.json}
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/metrics_prof; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
P=${1:-30000000}
python tools/metrics_probe.py $P > $O/probe.json 2> $O/probe.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/metrics_probe.py $P > $O/stats.log 2>&1)
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv; rm -rf $O/stats
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -- python $R/tools/metrics_probe.py $P > $O/$c.log 2>&1)
done
python - "$O" "$P" <<'PY'
import csv, glob, json, os, sys
O, pairs = sys.argv[1], int(sys.argv[2])
out = {'_about': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/metrics_probe.py: KB per dispatch of '
                 'metrics_stage_kernel, largest dispatch = one 64 M-record part of the forced full scan (or the whole of a shorter one); traffic = (2 * FETCH_SIZE + WRITE_SIZE) KB '
                 '(gfx950 correction of MI355X_MICROARCH.md); algorithmic = 22 B per pair', 'pairs': pairs}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    vals = []
    for path in glob.glob(os.path.join(O, c, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(path, newline='')):
            if 'metrics_stage_kernel' in row['Kernel_Name'] and row['Counter_Name'] == c:
                vals.append(float(row['Counter_Value']))
    out[c + '_KB_per_dispatch'] = {'dispatches': len(vals), 'max': max(vals) if vals else None, 'sum': sum(vals)}
f, w = out['FETCH_SIZE_KB_per_dispatch'], out['WRITE_SIZE_KB_per_dispatch']
if f['max'] and w['max'] is not None:
    out['largest_dispatch_traffic_bytes'] = int((2 * f['max'] + w['max']) * 1024)
    # the largest dispatch is the forced full scan's one part: min(2 * pairs, 64 Mi) records
    out['moved_bytes_per_record'] = round(out['largest_dispatch_traffic_bytes'] / float(min(2 * pairs, 64 << 20)), 3)
out['probe'] = json.loads(open(os.path.join(O, 'probe.json')).read().strip().splitlines()[-1])
json.dump(out, open(os.path.join(O, 'pmc.json'), 'w'), indent=1)
print(json.dumps(out))
PY
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
