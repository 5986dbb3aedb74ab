"""Convex up-sampling forward / adjoint at the headline size (8 x 68 x 120, S = 8): us per launch.
CAMLI_UPSAMPLE_BWD=legacy selects the round-1 adjoint (one wave per row group, 18 atomics per pixel)."""
import sys; sys.path.insert(0, '/root/repo')
import torch
from camliflow_amd.csrc import _lib
from camliflow_amd.cores import runtime
from camliflow_amd.cores.geometry import convex_upsample
runtime.set_backend('hip')
flow = torch.randn(8, 2, 68, 120, device='cuda', requires_grad=True)
mask = torch.randn(8, 576, 68, 120, device='cuda', requires_grad=True)
g = torch.randn(8, 2, 544, 960, device='cuda')
def run():
    out = convex_upsample(flow, mask, scale_factor=8, mask_scale=0.25)
    torch.autograd.grad(out, (flow, mask), g)
for _ in range(3): run()
torch.cuda.synchronize(); _lib.TIMER.reset(); _lib.TIMER.only = None; _lib.TIMER.enabled = True
for _ in range(5): run()
torch.cuda.synchronize(); _lib.TIMER.enabled = False
for k, v in _lib.TIMER.summary().items(): print('%-28s %7.1f us' % (k, v['total_ms'] / v['launches'] * 1e3))
