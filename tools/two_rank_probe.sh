#!/bin/bash
# Two ranks on the one GPU of the development box: the driver's N>1 bench command line with the gloo transport
# (RCCL refuses two ranks on one device - the first command shows that).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== RCCL, two ranks on cuda:0 (expected to be refused)"
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
    tools/rccl_dup_probe.py 2>&1 | tail -4
echo "== bench.py --gpus 2 over gloo, both ranks on cuda:0"
BESST_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 29612 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/two_rank_bench.json 2> gpurun_out/two_rank_bench.err
echo "exit $?"; tail -3 gpurun_out/two_rank_bench.err; cat gpurun_out/two_rank_bench.json
