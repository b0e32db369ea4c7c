#!/bin/bash
# Development (GPU box): PMC counters of the inflate kernel over a device ingest (library in BESST_AMD_LIB).  tools/ingest_pmc.sh label [level]
cd "$(dirname "$0")/.."
export PROBE_MODES=device:0
echo "== $1"
tools/pmc_cmd.sh "tools/ingest_probe.py C3 10000000 - ${2:-17}" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" 2>&1 | grep "bgzf_inflate"
