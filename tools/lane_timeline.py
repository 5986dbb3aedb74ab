"""Per-stream view of the two-lane training step (kineto trace of three steady steps): for every HIP stream the summed kernel
time and the time it has a kernel running, and for the busiest stream (the image lane) how long it sits idle while OTHER streams
run -- the time the image lane spends waiting for the point lane / the auxiliary chains.   python tools/lane_timeline.py"""
import collections
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from camliflow_amd.cores import CamLiRAFT, runtime  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

runtime.set_backend('hip')
runtime.set_deferred_param_grads(True)
runtime.set_overlap(os.environ.get('CAMLI_OVERLAP', '1') == '1')
runtime.use_tuned_gemms()
torch.manual_seed(0)
model = CamLiRAFT(bench.model_cfg(12)).cuda().train()
opt = bench.make_optimizer(model)
batch = {k: v.cuda() for k, v in bench.synthetic_batch(8, 540, 960, 8192, 1).items()}
for _ in range(4):
    bench.train_step(model, opt, batch)
torch.cuda.synchronize()
STEPS = 3
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    t0 = time.perf_counter()
    for _ in range(STEPS):
        bench.train_step(model, opt, batch)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / STEPS * 1e3
streams = collections.defaultdict(list)
names = collections.defaultdict(lambda: collections.Counter())
for e in prof.profiler.kineto_results.events():
    if str(e.device_type()).endswith('CUDA') and e.duration_ns() > 0:
        s = e.device_resource_id()
        streams[s].append((e.start_ns() / 1e3, (e.start_ns() + e.duration_ns()) / 1e3))
        names[s][e.name()[:60]] += e.duration_ns() / 1e3


def union(iv):
    iv = sorted(iv)
    out, (cs, ce) = [], iv[0]
    for s, e in iv[1:]:
        if s <= ce:
            ce = max(ce, e)
        else:
            out.append((cs, ce))
            cs, ce = s, e
    out.append((cs, ce))
    return out


def length(iv):
    return sum(e - s for s, e in iv)


def intersect(a, b):
    i = j = 0
    tot = 0.0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo:
            tot += hi - lo
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


print('wall %.1f ms per step (profiled)' % wall)
order = sorted(streams, key=lambda s: -sum(e - s_ for s_, e in streams[s]))
unions = {s: union(streams[s]) for s in streams}
everything = union([iv for s in streams for iv in streams[s]])
lo, hi = everything[0][0], everything[-1][1]
print('window %.1f ms per step, some kernel running %.1f ms per step' % ((hi - lo) / STEPS / 1e3, length(everything) / STEPS / 1e3))
for s in order:
    print('stream %s: %5d kernels per step, summed %.1f ms, busy %.1f ms per step; top: %s'
          % (s, len(streams[s]) // STEPS, sum(e - s_ for s_, e in streams[s]) / STEPS / 1e3, length(unions[s]) / STEPS / 1e3,
             ', '.join('%s %.1f' % (n, t / STEPS / 1e3) for n, t in names[s].most_common(3))))
main = order[0]
gaps = []
u = unions[main]
for (s0, e0), (s1, e1) in zip(u[:-1], u[1:]):
    gaps.append((e0, s1))
others = union([iv for s in streams if s != main for iv in streams[s]])
idle = length(gaps)
covered = intersect(gaps, others)
print('busiest stream %s idle %.1f ms per step; of that %.1f ms while another stream runs (waiting for it), %.1f ms with the whole GPU idle'
      % (main, idle / STEPS / 1e3, covered / STEPS / 1e3, (idle - covered) / STEPS / 1e3))
both = intersect(u, others)
print('busiest stream running TOGETHER with another stream: %.1f ms per step' % (both / STEPS / 1e3))
