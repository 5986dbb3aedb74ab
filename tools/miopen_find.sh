#!/bin/bash
# Solver selection for the library convolutions, MEASURED on the box (gpurun): this image ships no gfx950 find-db /
# perf-db for MIOpen (only the kernel-tuning nets), so torch's default immediate mode ranks solvers by a static
# fallback.  Leg A runs the bench step once with torch.backends.cudnn.benchmark (CAMLI_MIOPEN_FIND=1) and
# MIOPEN_FIND_MODE=1 (NORMAL: every applicable solver is compiled and timed) into a user find-db under <out>/db;
# leg B re-runs the default command (immediate mode) reading that db; leg C is the default command without it.
# Usage: bash tools/miopen_find.sh <tag> [find-timeout-s]   -> gpurun_out/<tag>/
set -u
TAG=${1:-find}
TMO=${2:-1500}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT/db
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-isolated"
( timeout 300 $B > $OUT/c_default.json 2> $OUT/c_default.err )
export MIOPEN_USER_DB_PATH=$OUT/db
( CAMLI_MIOPEN_FIND=1 MIOPEN_FIND_MODE=1 CAMLI_OVERLAP=0 timeout -k 10 $TMO $B --time-budget $TMO > $OUT/a_find.json 2> $OUT/a_find.err )
echo "find leg rc=$?" >> $OUT/a_find.err
ls -la $OUT/db > $OUT/db_listing.txt 2>&1
( timeout 400 $B > $OUT/b_immediate_db.json 2> $OUT/b_immediate_db.err )
( CAMLI_MIOPEN_FIND=1 timeout 400 $B > $OUT/d_find_db.json 2> $OUT/d_find_db.err )
for f in c_default a_find b_immediate_db d_find_db; do echo "$f: $(head -c 330 $OUT/$f.json)"; done
grep -h "first step" $OUT/*.err
