"""Kernel-only times (HIP events per C-ABI launch) of the CLFM glue kernels at the headline shapes (batch 8, 68x120, 2048 pts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from camliflow_amd.csrc import _lib, fused
g = torch.Generator().manual_seed(0)
b, c, h, w, n = 8, 128, 68, 120, 2048
feat = torch.randn(b, c, h, w, generator=g).cuda()
uv = (torch.rand(b, 2, n, generator=g) * torch.tensor([w - 1.0, h - 1.0]).view(1, 2, 1)).cuda()
data = torch.randn(b, c, n, generator=g).cuda()
scale = torch.randn(b, c, h * w, generator=g).cuda().requires_grad_(True)
idx = torch.randint(0, n, (b, h * w), generator=g).cuda()
s = torch.randn(b, c, generator=g).cuda().requires_grad_(True)
wmid = (torch.randn(c // 2, c, generator=g) * 0.1).cuda().requires_grad_(True)
wout = (torch.randn(2 * c, c // 2, generator=g) * 0.1).cuda().requires_grad_(True)
gw = torch.randn(b, c, 2, generator=g).cuda()

def run():
    fused.bilinear_sample(feat, uv)
    out = fused.gather_scale(data, scale, idx)
    torch.autograd.grad(out, scale, torch.ones_like(out))
    wgt = fused.sk_gate(s, wmid, wout)
    torch.autograd.grad(wgt, [s, wmid, wout], gw)
for _ in range(3):
    run()
torch.cuda.synchronize()
_lib.TIMER.reset(); _lib.TIMER.only = None; _lib.TIMER.enabled = True
for _ in range(20):
    run()
torch.cuda.synchronize(); _lib.TIMER.enabled = False
for k, v in _lib.TIMER.summary().items():
    print('%-30s %7.1f us  (%d launches)' % (k, v['total_ms'] / v['launches'] * 1e3, v['launches']))
