"""A/B: the all-pairs pyramid build + adjoint, hand-written MFMA kernels vs the torch composition
(matmul, division, avg_pool2d chain; autograd backward), batch 8 at 68x120 (configs[2])."""
import math
import os
import sys
import torch
from torch.nn.functional import avg_pool2d
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camliflow_amd.csrc import fused  # noqa: E402

b, c, h, w = 8, 256, 68, 120
f1 = torch.randn(b, c, h, w, device='cuda', requires_grad=True)
f2 = torch.randn(b, c, h, w, device='cuda', requires_grad=True)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def torch_build():
    vol = torch.matmul(f1.view(b, c, h * w).transpose(1, 2), f2.view(b, c, h * w)) / math.sqrt(c)
    lv = [vol.reshape(b * h * w, 1, h, w)]
    for _ in range(3):
        lv.append(avg_pool2d(lv[-1], 2, stride=2))
    return lv


lv = torch_build()
gs = [torch.randn_like(x) for x in lv]
print('torch  build fwd  %.2f ms' % timed(torch_build))
print('torch  build bwd  %.2f ms' % timed(lambda: torch.autograd.grad(torch_build(), [f1, f2], gs)))
del lv
print('camli  build fwd  %.2f ms' % timed(lambda: fused.allpairs_pyramid(f1, f2, 4)))


def camli_fb():
    pyr = fused.allpairs_pyramid(f1, f2, 4)
    pyr.grads = [g[:, 0] for g in gs]
    torch.autograd.grad(pyr.token, [f1, f2], torch.zeros(1, device='cuda'))


print('camli  build fwd+bwd  %.2f ms' % timed(camli_fb))
