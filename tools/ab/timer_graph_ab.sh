
for i in 1 2; do
for v in "" "CAMLI_NO_TIMER=1"; do
  echo "== $v"; env $v timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-isolated --no-side-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('host_enqueue_ms_per_step'), d['config'].get('hip_graph'))"
done; done
echo "== graph"; timeout 400 python bench.py --graph --steps 8 --warmup 3 --no-cpu-baseline --no-isolated --no-side-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('host_enqueue_ms_per_step'), d['config'].get('hip_graph'))"
