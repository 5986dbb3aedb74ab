#!/bin/bash
# A/B of the round-5 device work on the headline step, alternating runs on ONE box: everything on (default) against
# CAMLI_GRU_CL=0 CAMLI_CONVCL=0 CAMLI_GEMM_W128=0 (= the round-4 kernels: library GRU convolutions behind _CatConvCL, the
# 128x128-tile all-pairs GEMM).   tools/ab_round5.sh [rounds] [steps]
rounds=${1:-3}; steps=${2:-12}
run() { env $1 timeout 500 python bench.py --steps $steps --warmup 4 --no-cpu-baseline --no-isolated --no-side-configs 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['host_enqueue_ms_per_step'])"; }
OLD="CAMLI_GRU_CL=0 CAMLI_CONVCL=0 CAMLI_GEMM_W128=0"
NEW="CAMLI_GRU_CL=1"
run "$NEW" > /dev/null
for i in $(seq $rounds); do echo "new: $(run "$NEW")"; echo "old: $(run "$OLD")"; done
