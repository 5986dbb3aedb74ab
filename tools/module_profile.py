"""GPU time of one CamLiRAFT training step attributed to (module scope x aten op) with torch.profiler."""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
from camliflow_amd.cores import CamLiRAFT, runtime
from torch.profiler import profile, ProfilerActivity, record_function

runtime.set_backend('hip')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
model = CamLiRAFT(bench.model_cfg(12)).cuda().train()
opt = bench.make_optimizer(model)
batch = {k: v.cuda() for k, v in bench.synthetic_batch(B, 540, 960, 8192, 1).items()}

SCOPES = ['core.branch_2d.fnet', 'core.branch_2d.cnet', 'core.branch_2d.correlation', 'core.branch_2d.motion_encoder',
          'core.branch_2d.gru', 'core.branch_2d.flow_head', 'core.branch_2d.convex_upsampler',
          'core.branch_3d.fnet', 'core.branch_3d.cnet', 'core.branch_3d.correlation', 'core.branch_3d.motion_encoder',
          'core.branch_3d.gru', 'core.branch_3d.flow_head', 'core.clfm_fnet', 'core.clfm_cnet', 'core.clfm_corr',
          'core.clfm_motion']
mods = dict(model.named_modules())
for name in SCOPES:
    m = mods[name]
    orig = m.forward
    def wrapped(*a, _orig=orig, _name=name, **k):
        with record_function('SCOPE:' + _name):
            return _orig(*a, **k)
    m.forward = wrapped

for _ in range(2):
    bench.train_step(model, opt, batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    bench.train_step(model, opt, batch)
    torch.cuda.synchronize()

# attribute each kernel to the innermost enclosing SCOPE range of its launching op (forward) ; backward ops carry
# the forward scope through autograd's sequence numbers -> use key_averages(group_by_stack_n) is unreliable, so do
# a simpler thing: forward time per scope from the ranges, backward time in bulk per aten op
events = prof.events()
scope_fwd = collections.Counter()
for e in events:
    if e.name.startswith('SCOPE:'):
        scope_fwd[e.name[6:]] += e.device_time_total
print('--- forward GPU time per scope (ms, includes children kernels)')
for k, v in scope_fwd.most_common():
    print('%-40s %8.2f' % (k, v / 1e3))
print('--- top ops by self GPU time (ms)')
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.self_device_time_total)[:45]
for e in rows:
    print('%-60s calls %6d  %8.2f' % (e.key[:60], e.count, e.self_device_time_total / 1e3))

# ---- per scope: which ops hold the forward GPU time (walk every op's parent chain up to its SCOPE range) ----
per_scope = collections.defaultdict(collections.Counter)
calls = collections.defaultdict(collections.Counter)
for e in events:
    if e.name.startswith('SCOPE:') or e.self_device_time_total <= 0:
        continue
    parent = e.cpu_parent
    while parent is not None and not parent.name.startswith('SCOPE:'):
        parent = parent.cpu_parent
    if parent is not None:
        per_scope[parent.name[6:]][e.name] += e.self_device_time_total
        calls[parent.name[6:]][e.name] += 1
want = sys.argv[2].split(',') if len(sys.argv) > 2 else ['core.clfm_corr', 'core.clfm_motion', 'core.branch_2d.motion_encoder',
                                                          'core.branch_2d.convex_upsampler', 'core.branch_3d.correlation']
for scope in want:
    print('--- %s: forward ops by self GPU time (ms, calls)' % scope)
    for name, t in per_scope[scope].most_common(14):
        print('   %-70s %8.3f %5d' % (name[:70], t / 1e3, calls[scope][name]))
