run() { env $1 CAMLI_OVERLAP=0 timeout 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-isolated --no-side-configs 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['host_enqueue_ms_per_step'])"; }
OLD="CAMLI_GRU_CL=0 CAMLI_CONVCL=0 CAMLI_GEMM_W128=0"
NEW="CAMLI_GRU_CL=1"
run "$NEW" > /dev/null
for i in 1 2 3; do echo "new one-lane: $(run "$NEW")"; echo "old one-lane: $(run "$OLD")"; done
