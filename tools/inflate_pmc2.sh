#!/bin/bash
# Development (GPU box): memory-side counters of the inflate kernel alone (library in BESST_AMD_LIB).
cd "$(dirname "$0")/.."
echo "== $1"
tools/pmc_cmd.sh "tools/inflate_time.py 3000000" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_FLAT" "SQ_IFETCH SQ_IFETCH_LEVEL" "TA_TA_BUSY_sum" 2>&1 | grep "bgzf_inflate\|rror"
