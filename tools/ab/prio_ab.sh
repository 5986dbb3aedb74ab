# stream-priority experiment (r6): does a HIP priority on the image lane's streams shorten the two-lane step?
run() { env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-isolated --no-side-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('host_enqueue_ms_per_step'))"; }
for i in 1 2; do
echo "== default";                      run X=1
echo "== main/aux/wgrad high";          run CAMLI_PRIO_MAIN=-1 CAMLI_PRIO_AUX=-1 CAMLI_PRIO_WGRAD=-1
echo "== main high only";               run CAMLI_PRIO_MAIN=-1
echo "== point lane high";              run CAMLI_PRIO_SIDE=-1
done
