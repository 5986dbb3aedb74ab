"""bias + activation epilogue alone on the shapes of the step (GPU box only): python tools/ab_biasact.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camliflow_amd.csrc import fused  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for shape in [(16, 64, 272, 480), (16, 256, 136, 240), (16, 512, 68, 120), (8, 128, 68, 120), (8, 384, 8160), (8, 128, 2048)]:
    for act in ('relu', 'leaky_relu', None):
        x = torch.randn(*shape, device='cuda')
        bias = torch.randn(shape[1], device='cuda', requires_grad=True)
        n = x.numel()

        def fwd():
            return fused.bias_act(x.requires_grad_(True) * 1.0, bias, act)    # "* 1.0": a fresh tensor the op may overwrite
        y = fwd()
        g = torch.randn_like(y)
        t_mul = timed(lambda: x * 1.0)
        t_f = timed(fwd) - t_mul
        t_fb = timed(lambda: torch.autograd.grad(fwd(), bias, g)) - t_mul
        print('%-22s %-10s fwd %7.1f us (%5.2f TB/s at 8 B/elem)   bwd %7.1f us   copy-like x*1.0 %7.1f us (%5.2f TB/s)'
              % (shape, act, t_f, 8.0 * n / t_f / 1e6, t_fb - t_f, t_mul, 8.0 * n / t_mul / 1e6))
