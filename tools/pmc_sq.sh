#!/bin/bash
# SQ stall breakdown + LDS conflicts of the kernels matching a regex:
#   bash tools/pmc_sq.sh <kernel regex> <tag> -- <command ...>          (default command: tools/kernel_bench.py --reps 2)
set -u
ROOT=$(pwd); RX=$1; TAG=${2:-sq}; shift 2
if [ "${1:-}" = "--" ]; then shift; CMD=("$@"); else CMD=(python $ROOT/tools/kernel_bench.py --reps 2); fi
export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM \
  --kernel-trace --kernel-include-regex "$RX" --output-format csv -d /tmp/pmc_$TAG -o p -- "${CMD[@]}" > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py /tmp/pmc_$TAG/p_counter_collection.csv
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES \
  --kernel-trace --kernel-include-regex "$RX" --output-format csv -d /tmp/pmc2_$TAG -o p -- "${CMD[@]}" > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py /tmp/pmc2_$TAG/p_counter_collection.csv
