# r6: the GRU weight gradients from the forward's kept transforms (CAMLI_GRU_KEEP_V 1 | 0)
run() { env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-isolated --no-side-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('host_enqueue_ms_per_step'))"; }
for i in 1 2 3; do
echo "== keep";     run CAMLI_GRU_KEEP_V=1
echo "== again";    run CAMLI_GRU_KEEP_V=0
done
