# do the auxiliary-stream branches of the image lane (round 3: -13 ms with the library's convolutions) still pay now that the
# update block's convolutions are chip-filling own kernels?
run() { env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-isolated --no-side-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('host_enqueue_ms_per_step'))"; }
for i in 1 2; do
echo "== default";                 run X=1
echo "== CAMLI_BRANCHES=0";        run CAMLI_BRANCHES=0
echo "== CAMLI_WGRAD_ASIDE=0";     run CAMLI_WGRAD_ASIDE=0
echo "== mask 1 (motion only)";    run CAMLI_BRANCH_MASK=1
echo "== mask 2 (mask head only)"; run CAMLI_BRANCH_MASK=2
echo "== mask 4 (CLFM only)";      run CAMLI_BRANCH_MASK=4
echo "== mask 8 (context only)";   run CAMLI_BRANCH_MASK=8
echo "== share one aux stream";    run CAMLI_BRANCH_SHARE=1
done
