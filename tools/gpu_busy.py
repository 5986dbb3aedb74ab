"""How much of the two-lane training step has at least one kernel running (union of kernel intervals over all streams, from a
kineto trace of two steady steps) -- the rest is the device waiting for the host.   python tools/gpu_busy.py [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from camliflow_amd.cores import CamLiRAFT, runtime  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

runtime.set_backend('hip')
runtime.set_deferred_param_grads(True)
runtime.set_overlap(os.environ.get('CAMLI_OVERLAP', '1') == '1')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
model = CamLiRAFT(bench.model_cfg(12)).cuda().train()
opt = bench.make_optimizer(model)
batch = {k: v.cuda() for k, v in bench.synthetic_batch(B, 540, 960, 8192, 1).items()}
for _ in range(4):
    bench.train_step(model, opt, batch)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(3):
    bench.train_step(model, opt, batch)
torch.cuda.synchronize()
print('unprofiled: %.1f ms per step' % ((time.perf_counter() - t0) / 3 * 1e3))
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    t0 = time.perf_counter()
    for _ in range(3):
        bench.train_step(model, opt, batch)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 3 * 1e3
iv = []
per_stream = {}
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA and e.time_range.end > e.time_range.start:
        iv.append((e.time_range.start, e.time_range.end))
iv.sort()
lo, hi = iv[0][0], iv[-1][1]
busy, cur_s, cur_e, total = 0.0, iv[0][0], iv[0][1], 0.0
for s, e in iv[1:]:
    total += 0
    if s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
busy += cur_e - cur_s
summed = sum(e - s for s, e in iv)
print('profiled: %.1f ms wall per step; trace window %.1f ms per step; >= 1 kernel running %.1f ms per step (%.1f %% of the window); '
      'summed kernel time %.1f ms per step; %d kernels per step'
      % (wall, (hi - lo) / 3e3, busy / 3e3, 100.0 * busy / (hi - lo), summed / 3e3, len(iv) // 3))
gaps = []
cur_e = iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        gaps.append(s - cur_e)
    cur_e = max(cur_e, e)
gaps.sort(reverse=True)
print('idle gaps: %d per step, largest (us): %s' % (len(gaps) // 3, [round(g, 1) for g in gaps[:12]]))
import collections
hist = collections.Counter()
for g in gaps:
    hist['<5us' if g < 5 else '<20us' if g < 20 else '<100us' if g < 100 else '>=100us'] += g
print('idle time by gap size (ms per step):', {k: round(v / 3e3, 2) for k, v in hist.items()})
