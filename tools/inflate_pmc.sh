#!/bin/bash
# Development (GPU box): SQ counters of the inflate kernel alone (library in BESST_AMD_LIB), means per launch of 4096 blocks.
cd "$(dirname "$0")/.."
echo "== $1"
tools/pmc_cmd.sh "tools/inflate_time.py 3000000" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" 2>&1 | grep "bgzf_inflate"
