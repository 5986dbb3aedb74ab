#!/bin/bash
# Builds the variants of the stand-alone GEMM driver (KS / NBUF / ablation) into tools/microbench/bin/.
set -e
cd "$(dirname "$0")"
mkdir -p bin
for v in "16 4 0" "16 3 0" "16 5 0" "8 4 0" "16 4 1" "16 4 2" ${EXTRA_VARIANTS}; do
    set -- $v
    hipcc --offload-arch=gfx950 -O3 -I ../../camliflow_amd/csrc/hip -DMB_KS=$1 -DMB_NBUF=$2 -DMB_ABL=$3 ${MB_DEFS} gemm_w128_mb.hip -o bin/gemm_w128_ks$1_nb$2_abl$3 &
done
wait
ls -la bin
