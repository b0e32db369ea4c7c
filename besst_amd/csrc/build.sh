#!/bin/bash
# Build libbesst_amd.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
#   -ffp-contract=off : the fp64 paths (PosDir truncation, KS centring, gap estimator) must round
#                       exactly like the reference's Python floats - no fused multiply-adds.
# One object per source file, compiled in parallel and only when the source (or a header) is newer; objects live in
# csrc/_build (not tracked).  BESST_EXTRA_FLAGS changes force a full rebuild (the flags are part of the stamp).
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="${here}/../libbesst_amd.so"
obj="${here}/_build"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result ${BESST_EXTRA_FLAGS:-}"
SRCS="api classify sortreduce onesweep runs metrics score bam_reader bgzf_gpu hostmath linearize chain scorepaths"
mkdir -p "${obj}"
stamp="${obj}/flags.txt"
if [ ! -f "${stamp}" ] || [ "$(cat "${stamp}")" != "${FLAGS}" ]; then
    rm -f "${obj}"/*.o
    echo "${FLAGS}" > "${stamp}"
fi
pids=()
for f in ${SRCS}; do
    src="${here}/${f}.hip"
    o="${obj}/${f}.o"
    if [ ! -f "${o}" ] || [ "${src}" -nt "${o}" ] || [ "${here}/common.h" -nt "${o}" ] || [ "${here}/../../include/besst_amd.h" -nt "${o}" ]; then
        "${HIPCC}" ${FLAGS} -c "${src}" -o "${o}" &
        pids+=($!)
    fi
done
for p in "${pids[@]:-}"; do
    [ -n "${p}" ] && wait "${p}"
done
objs=""
for f in ${SRCS}; do objs="${objs} ${obj}/${f}.o"; done
"${HIPCC}" --offload-arch=gfx950 -shared -fPIC ${objs} -lz -ldl -lpthread -o "${out}"
echo "built ${out}"
