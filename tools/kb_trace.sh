#!/bin/bash
# rocprofv3 kernel statistics of one group of tools/kernel_bench.py rows (gpurun):  bash tools/kb_trace.sh <only> <tag> [reps]
# -> gpurun_out/<tag>/kb_<only>_kernel_stats.csv (+ the bench rows as json); prints the rows and the top kernels
set -u
ONLY=$1; TAG=$2; REPS=${3:-10}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kbt_$ONLY -o t -- \
    python $ROOT/tools/kernel_bench.py --reps $REPS --only $ONLY --json $OUT/kb_$ONLY.json > $OUT/kb_$ONLY.log 2>&1
grep -E "^(case|$ONLY)" $OUT/kb_$ONLY.log
cp /tmp/kbt_$ONLY/t_kernel_stats.csv $OUT/kb_${ONLY}_kernel_stats.csv 2>/dev/null
python - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/kb_${ONLY}_kernel_stats.csv')))
for r in rows[:14]:
    print('%-90s calls %5s avg_us %9.1f total_ms %8.2f' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
rm -rf /tmp/kbt_$ONLY
