#!/bin/bash
# Build libbesst_amd.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
#   -ffp-contract=off : the fp64 paths (PosDir truncation, KS centring, gap estimator) must round
#                       exactly like the reference's Python floats - no fused multiply-adds.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="${here}/../libbesst_amd.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"${HIPCC}" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
    -Wall -Wno-unused-result ${BESST_EXTRA_FLAGS:-} \
    "${here}/api.hip" "${here}/classify.hip" "${here}/sortreduce.hip" "${here}/onesweep.hip" "${here}/metrics.hip" "${here}/score.hip" "${here}/bam_reader.hip" "${here}/hostmath.hip" "${here}/linearize.hip" "${here}/chain.hip" "${here}/scorepaths.hip" \
    -lz -ldl -lpthread -o "${out}"
echo "built ${out}"
