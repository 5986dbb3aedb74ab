#!/bin/bash
# A/B of one environment switch on the headline step, alternating runs on ONE box:  tools/ab_step.sh VAR A B [rounds] [steps]
# prints ms_per_step of every run and the two medians
var=$1; a=$2; b=$3; rounds=${4:-3}; steps=${5:-20}
run() { env $var=$1 timeout 500 python bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-isolated --no-side-configs 2>/dev/null \
        | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
run $a > /dev/null          # the first run of a box pages the image in
for i in $(seq $rounds); do ra="$ra $(run $a)"; rb="$rb $(run $b)"; done
python - "$var" "$a" "$b" "$ra" "$rb" <<'PY'
import sys, statistics
var, a, b, ra, rb = sys.argv[1:6]
ra, rb = [float(x) for x in ra.split()], [float(x) for x in rb.split()]
print('%s=%s: %s  median %.2f' % (var, a, ra, statistics.median(ra)))
print('%s=%s: %s  median %.2f' % (var, b, rb, statistics.median(rb)))
print('delta (%s - %s) = %+.2f ms' % (a, b, statistics.median(ra) - statistics.median(rb)))
PY
