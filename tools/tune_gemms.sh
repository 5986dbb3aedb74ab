#!/bin/bash
# Regenerate camliflow_amd/gemm_tuning_gfx950.csv on an MI355X box: every bench configuration once under PyTorch's TunableOp
# (each GEMM shape is timed against every hipBLASLt / rocBLAS solution the first time it occurs; ~7 minutes in all).
#   bash tools/tune_gemms.sh            -> gpurun_out/gemm_tuning_raw.csv  (copy it over the shipped table)
set -u
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=/tmp/camli_tun.csv
cp camliflow_amd/gemm_tuning_gfx950.csv /tmp/camli_tun0.csv      # start from the shipped table: only new shapes are timed
for cfg in "" "--config eval" "--config camlipwc" "--config kitti"; do
  timeout 900 python bench.py $cfg --steps 3 --warmup 2 --no-cpu-baseline --no-isolated --no-side-configs --time-budget 800 > /dev/null 2>&1
  wc -l /tmp/camli_tun0.csv
done
mkdir -p gpurun_out && cp /tmp/camli_tun0.csv gpurun_out/gemm_tuning_raw.csv
