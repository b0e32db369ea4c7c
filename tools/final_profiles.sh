#!/bin/bash
# Runs on the GPU box: the round's profile set from ONE box - kernel stats + PMC traffic of C3 and C2, the default bench line,
# the ingest kernels - under gpurun_out/final/.
cd "$(dirname "$0")/.."
O=gpurun_out/final; rm -rf $O; mkdir -p $O
tools/profile_round.sh C3 400000000 > /dev/null 2>&1
tools/profile_round.sh C2 20000000 > /dev/null 2>&1
cp gpurun_out/prof_c3/kernel_stats.csv $O/c3_kernel_stats.csv; cp gpurun_out/prof_c3/pmc_traffic.json $O/c3_pmc_traffic.json
cp gpurun_out/prof_c2/kernel_stats.csv $O/c2_kernel_stats.csv; cp gpurun_out/prof_c2/pmc_traffic.json $O/c2_pmc_traffic.json
T=${BESST_ROUND_TAG:-r06}
cp $O/c3_pmc_traffic.json profiles/${T}_c3_pmc_traffic.json; cp $O/c2_pmc_traffic.json profiles/${T}_c2_pmc_traffic.json
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench.err
tools/ingest_prof.sh C3 50000000 17 > $O/ingest_prof.txt 2>&1
cp gpurun_out/ingest_prof/kernel_stats.csv $O/ingest_kernel_stats.csv
tools/ingest_pmc.sh final 17 > $O/ingest_pmc.txt 2>&1
tools/metrics_prof.sh 30000000 > $O/metrics_prof.txt 2>&1
cp gpurun_out/metrics_prof/kernel_stats.csv $O/metrics_kernel_stats.csv; cp gpurun_out/metrics_prof/pmc.json $O/metrics_pmc.json
cp $O/metrics_pmc.json profiles/${T}_metrics_pmc.json
tools/loop_sq.sh $O/record_loop_sq.json > $O/loop_sq.txt 2>&1
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["verified_vs_c_oracle"], d["c2"]["ms_per_step"], d["roofline"]["record_loop_kernel"]["avg_launch_ms"], d["overlapped"]["ms_per_step"], d["roofline"]["measured_d2d_copy_GBps"])
PY
# the sharded orchestration: ONE rank over RCCL (the C4 shape), and two ranks on the one GPU over gloo FROM A FILE per library
timeout 900 python bench.py --gpus 1 --config C4 --steps 10 --warmup 2 > $O/c4_rank1.json 2> $O/c4_rank1.err
BESST_DIST_BACKEND=gloo timeout 1200 python bench.py --gpus 2 --from-bam --steps 10 --warmup 2 > $O/from_bam_2ranks_one_gpu.json 2> $O/from_bam.err
tail -c 600 $O/from_bam_2ranks_one_gpu.json
# full-size C3 from BAM bytes (a one-off of bench.bam_to_graph_timing with all 200 M pairs: a 28 GB sequencer-like file)
timeout 1500 python -c "import bench, torch, json; print(json.dumps(bench.bam_to_graph_timing(torch.device('cuda', 0), 'C3', pairs=None, realistic=True)))" > $O/bam_to_graph_c3_full.json 2> $O/bam_full.err
tail -c 900 $O/bam_to_graph_c3_full.json
