#!/bin/bash
# Runs on the GPU box: kernel stats + the two PMC passes + the full bench line for C2; results under gpurun_out/prof_c2/.
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/prof_c2
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --breakdown-steps 0 --no-verify --no-stages --in-flight 0"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py $ARGS > $O/stats.log 2>&1)
PARGS="--steps 5 --warmup 1 --no-stages --no-cpu-baseline --no-verify --breakdown-steps 0 --in-flight 0"
(cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $R/bench.py $PARGS > $O/fetch.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python $R/bench.py $PARGS > $O/write.log 2>&1)
python tools/pmc_summary.py $O/fetch $O/write 20000000 $O/pmc_traffic.json
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
rm -rf $O/stats $O/fetch $O/write          # traces are large; the summaries are what is kept
cp $O/pmc_traffic.json profiles/r01_c2_pmc_traffic.json   # bench.py reads the traffic figure from here
python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
head -12 $O/kernel_stats.csv
