"""Kernel-only times (HIP events around each C-ABI launch) of the two-channel 3x3 head at the headline shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from camliflow_amd.csrc import _lib, fused
x = torch.randn(8, 256, 68, 120, device='cuda', requires_grad=True)
w = torch.randn(2, 256, 3, 3, device='cuda', requires_grad=True)
b = torch.randn(2, device='cuda', requires_grad=True)
gy = torch.randn(8, 2, 68, 120, device='cuda')
for _ in range(3):
    torch.autograd.grad(fused.conv3x3_co2(x, w, b), [x, w, b], gy)
torch.cuda.synchronize()
_lib.TIMER.reset(); _lib.TIMER.only = None; _lib.TIMER.enabled = True
for _ in range(20):
    torch.autograd.grad(fused.conv3x3_co2(x, w, b), [x, w, b], gy)
torch.cuda.synchronize(); _lib.TIMER.enabled = False
for k, v in _lib.TIMER.summary().items():
    us = v['total_ms'] / v['launches'] * 1e3
    print('%-32s %7.1f us  %6.0f GB/s' % (k, us, v['work'] / v['launches'] / us / 1e3))
