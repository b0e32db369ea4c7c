#!/bin/bash
# Development helper (GPU box): rocprofv3 kernel stats of an arbitrary python command, own kernels printed.  tools/kstats_cmd.sh tools/sort_probe.py 18 5000000
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/kstats; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/"$@" > $O/stats.log 2>&1)
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv; rm -rf $O/stats
python tools/ownk.py $O/kernel_stats.csv
