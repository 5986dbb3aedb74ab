"""bias/activation epilogue: NCHW vs channels-last kernels on a first-stage activation [16,256,136,240] (HIP events per launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from camliflow_amd.csrc import _lib, fused
for fmt, name in ((torch.contiguous_format, 'nchw'), (torch.channels_last, 'nhwc')):
    x = torch.randn(16, 256, 136, 240, device='cuda').contiguous(memory_format=fmt)
    r = torch.randn_like(x)
    b = torch.randn(256, device='cuda', requires_grad=True)
    go = torch.randn_like(x)
    def run():
        xa = x.clone().requires_grad_(True)
        y = fused.bias_act_res(xa * 1.0, b, r, 'relu')
        torch.autograd.grad(y, [xa, b], go)
        y2 = fused.bias_act(xa * 1.0, b, 'relu')
        torch.autograd.grad(y2, [xa, b], go)
    for _ in range(2): run()
    torch.cuda.synchronize(); _lib.TIMER.reset(); _lib.TIMER.only = None; _lib.TIMER.enabled = True
    for _ in range(5): run()
    torch.cuda.synchronize(); _lib.TIMER.enabled = False
    for k, v in _lib.TIMER.summary().items():
        us = v['total_ms'] / v['launches'] * 1e3
        print('%-5s %-22s %8.1f us  %7.0f GB/s' % (name, k, us, v['work'] / v['launches'] / us / 1e3))
