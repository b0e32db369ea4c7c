#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats + the two PMC passes for one config, then the full default bench line.
#   tools/profile_round.sh C3 400000000     -> gpurun_out/prof_c3/{kernel_stats.csv,pmc_traffic.json,bench.json}
cd "$(dirname "$0")/.."
R=$PWD
CFG=${1:-C3}; NREC=${2:-400000000}; TAG=$(echo $CFG | tr A-Z a-z)
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
STEPS=5
ARGS="--config $CFG --also= --steps $STEPS --warmup 1 --no-cpu-baseline --breakdown-steps 0 --no-verify --no-stages --in-flight 0"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py $ARGS > $O/stats.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $R/bench.py $ARGS > $O/fetch.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python $R/bench.py $ARGS > $O/write.log 2>&1)
# steps + warmup + the capacity probe = dispatches of the record loop per run
python tools/pmc_summary.py $O/fetch $O/write $CFG $NREC $((STEPS + 2)) $O/pmc_traffic.json
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
rm -rf $O/stats $O/fetch $O/write          # traces are large; the summaries are what is kept
mkdir -p profiles; cp $O/pmc_traffic.json profiles/${BESST_ROUND_TAG:-r06}_${TAG}_pmc_traffic.json   # bench.py reads the traffic figure from here
head -14 $O/kernel_stats.csv
