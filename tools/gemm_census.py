"""Which library GEMM / convolution calls are left in the CamLiRAFT training step, by (aten op, input shapes, calling line of
this package), with their GPU time.  torch.profiler over one steady step; backward-thread ops have no Python stack, their shapes
identify them.  Run on the GPU box:  python tools/gemm_census.py [batch]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench  # noqa: E402
from camliflow_amd.cores import CamLiRAFT, runtime  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

runtime.set_backend('hip')
runtime.set_deferred_param_grads(True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
model = CamLiRAFT(bench.model_cfg(12)).cuda().train()
opt = bench.make_optimizer(model)
batch = {k: v.cuda() for k, v in bench.synthetic_batch(B, 540, 960, 8192, 1).items()}
for _ in range(2):
    bench.train_step(model, opt, batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    bench.train_step(model, opt, batch)
    torch.cuda.synchronize()

WATCH = ('aten::mm', 'aten::bmm', 'aten::addmm', 'aten::baddbmm', 'aten::baddbmm_', 'aten::addmm_', 'aten::convolution_backward',
         'aten::miopen_convolution', 'aten::cudnn_convolution', 'aten::_conv_depthwise2d', 'aten::miopen_batch_norm',
         'aten::miopen_batch_norm_backward', 'aten::native_batch_norm', 'aten::native_batch_norm_backward', 'aten::miopen_depthwise_convolution',
         'aten::max_pool2d_with_indices', 'aten::max_pool2d_with_indices_backward', 'aten::avg_pool2d', 'aten::avg_pool2d_backward')
groups = collections.defaultdict(lambda: [0, 0.0])
per_op = collections.Counter()
for e in prof.events():
    if e.name not in WATCH:
        continue
    t = e.device_time_total if e.name == 'aten::convolution_backward' else e.self_device_time_total
    if t <= 0:
        continue
    where = ''
    for fr in (e.stack or []):
        if 'camliflow_amd' in fr or 'bench.py' in fr:
            where = fr.split('camliflow_amd/')[-1]
            break
    shapes = str([s for s in (e.input_shapes or []) if s])[:110]
    g = groups[(e.name, shapes, where)]
    g[0] += 1
    g[1] += t
    per_op[e.name] += t
print('--- GPU time per aten op (ms)')
for k, v in per_op.most_common():
    print('%-36s %8.3f' % (k, v / 1e3))
print('--- top groups (op, input shapes, calling line): calls, ms')
for (name, shapes, where), (n, t) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:110]:
    print('%-28s %5d %8.3f  %-110s %s' % (name, n, t / 1e3, shapes, where[:60]))
