#!/bin/bash
# A/B of one environment switch on the batch-4 SyncBN 1-rank-RCCL proxy of configs[3] (side_configs.ddp4): tools/ab_ddp4.sh VAR A B [rounds]
var=$1; a=$2; b=$3; rounds=${4:-3}
run() { env $var=$1 CAMLI_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((20000 + RANDOM % 20000)) HSA_ENABLE_IPC_MODE_LEGACY=0 \
        timeout 400 python bench.py --config camliraft --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --no-isolated --no-side-configs 2>/dev/null \
        | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
run $a > /dev/null
for i in $(seq $rounds); do ra="$ra $(run $a)"; rb="$rb $(run $b)"; done
python - "$var" "$a" "$b" "$ra" "$rb" <<'PY'
import sys, statistics
var, a, b, ra, rb = sys.argv[1:6]
ra, rb = [float(x) for x in ra.split()], [float(x) for x in rb.split()]
print('%s=%s: %s  median %.2f' % (var, a, ra, statistics.median(ra)))
print('%s=%s: %s  median %.2f' % (var, b, rb, statistics.median(rb)))
print('delta (%s - %s) = %+.2f ms' % (a, b, statistics.median(ra) - statistics.median(rb)))
PY
