"""Kernel-only times (HIP events per C-ABI launch) of the ResNet stem's max pooling at the headline shape
([16,64,270,480]: both frames of a batch of 8) next to torch's kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from camliflow_amd.csrc import _lib, fused
x = torch.relu(torch.randn(16, 64, 270, 480, device='cuda')).requires_grad_(True)
go = torch.randn(16, 64, 135, 240, device='cuda')

def run():
    y = fused.maxpool3x3s2(x)
    torch.autograd.grad(y, x, go)

for _ in range(3):
    run()
torch.cuda.synchronize()
_lib.TIMER.reset(); _lib.TIMER.only = None; _lib.TIMER.enabled = True
for _ in range(10):
    run()
torch.cuda.synchronize(); _lib.TIMER.enabled = False
for k, v in _lib.TIMER.summary().items():
    print('%-30s %7.1f us  (%d launches)' % (k, v['total_ms'] / v['launches'] * 1e3, v['launches']))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
y = F.max_pool2d(x, 3, 2, 1); torch.autograd.grad(y, x, go); torch.cuda.synchronize()
ev[0].record(); y = F.max_pool2d(x, 3, 2, 1); ev[1].record(); torch.autograd.grad(y, x, go); ev[2].record(); torch.cuda.synchronize()
print('torch max_pool2d fwd %.1f us, bwd %.1f us' % (1e3 * ev[0].elapsed_time(ev[1]), 1e3 * ev[1].elapsed_time(ev[2])))
