#!/bin/bash
# Development: build libbesst_amd.so variants that differ in ONE source file's -D flags (A/B runs of kernel experiments).
#   tools/variant.sh NAME FILE "-DFLAG ..."   ->  _variants/libbesst_amd_NAME.so   (run with BESST_AMD_LIB=that path)
set -euo pipefail
repo="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; file="$2"; flags="$3"
src="${repo}/besst_amd/csrc"; obj="${src}/_build"; out="${repo}/_variants"
mkdir -p "${out}"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result ${flags} \
    -c "${src}/${file}.hip" -o "${out}/${file}_${name}.o"
objs=""
for f in api classify sortreduce onesweep runs metrics score bam_reader bgzf_gpu hostmath linearize chain scorepaths; do
    if [ "$f" = "${file}" ]; then objs="${objs} ${out}/${file}_${name}.o"; else objs="${objs} ${obj}/${f}.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ${objs} -lz -ldl -lpthread -o "${out}/libbesst_amd_${name}.so"
echo "${out}/libbesst_amd_${name}.so"
