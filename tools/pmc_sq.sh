#!/bin/bash
# SQ stall breakdown + LDS conflicts of one kernel of tools/kernel_bench.py:  bash tools/pmc_sq.sh <kernel regex> <--only substr> <tag>
set -u
ROOT=$(pwd); RX=$1; ONLY=$2; TAG=${3:-sq}
export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM \
  --kernel-trace --kernel-include-regex "$RX" --output-format csv -d /tmp/pmc_$TAG -o p -- python $ROOT/tools/kernel_bench.py --reps 2 --only "$ONLY" > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py /tmp/pmc_$TAG/p_counter_collection.csv
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE \
  --kernel-trace --kernel-include-regex "$RX" --output-format csv -d /tmp/pmc2_$TAG -o p -- python $ROOT/tools/kernel_bench.py --reps 2 --only "$ONLY" > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py /tmp/pmc2_$TAG/p_counter_collection.csv
