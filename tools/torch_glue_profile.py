"""Which torch element-wise / copy / reduction kernels are left in the CamLiRAFT training step, by (aten op, input shapes,
calling line of this package).  torch.profiler over one steady step; backward-thread ops have no Python stack, their shapes
identify them.  Run on the GPU box:  python tools/torch_glue_profile.py [batch]"""
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

WATCH = ('aten::add', 'aten::add_', 'aten::copy_', 'aten::sum', 'aten::fill_', 'aten::zero_', 'aten::cat', 'aten::mul',
         'aten::mul_', 'aten::sub', 'aten::div', 'aten::neg', 'aten::clone', 'aten::contiguous', 'aten::index',
         'aten::where', 'aten::mean', 'aten::sqrt', 'aten::abs', 'aten::index_select', 'aten::gather', 'aten::stack')
groups = collections.defaultdict(lambda: [0, 0.0])
per_op = collections.Counter()
for e in prof.events():
    if e.name not in WATCH or e.self_device_time_total <= 0:
        continue
    where = ''
    for fr in (e.stack or []):
        if 'camliflow_amd' in fr or 'bench.py' in fr:
            where = fr.split('camliflow_amd/')[-1]
            break
    shapes = str([s for s in (e.input_shapes or []) if s])[:90]
    g = groups[(e.name, shapes, where)]
    g[0] += 1
    g[1] += e.self_device_time_total
    per_op[e.name] += e.self_device_time_total
print('--- self GPU time per aten op (ms)')
for k, v in per_op.most_common():
    print('%-22s %8.3f' % (k, v / 1e3))
print('--- top groups (op, input shapes, calling line): calls, ms')
for (name, shapes, where), (n, t) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:70]:
    print('%-16s %5d %8.3f  %-90s %s' % (name, n, t / 1e3, shapes, where[:70]))
