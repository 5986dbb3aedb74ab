"""One GRU2D half-step convolution (cat([h, motion]) -> 1x5 / 5x1, 256 -> 256 channels, batch 8 at 68x120), forward + backward:
the library on NCHW tensors (it transposes x, y, gy (twice), gx and re-transposes x for the weight gradient inside) against the
same convolution on explicitly channels-last operands (x transposed once and kept for the backward, gy transposed once).

  python tools/conv_cl_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camliflow_amd.cores import runtime  # noqa: E402
from camliflow_amd.cores.blocks import cat_conv_cl  # noqa: E402


def timeit(fn, reps=20):
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


def main():
    runtime.use_tuned_gemms()
    torch.manual_seed(0)
    b, hd, hh, ww = 8, 128, 68, 120
    for cout, ks, pad in ((256, (1, 5), (0, 2)), (256, (5, 1), (2, 0)), (128, (1, 5), (0, 2)), (128, (5, 1), (2, 0))):
        h = torch.randn(b, hd, hh, ww, device='cuda', requires_grad=True)
        m = torch.randn(b, hd, hh, ww, device='cuda', requires_grad=True)
        w = (torch.randn(cout, 2 * hd, *ks, device='cuda') * 0.02).requires_grad_(True)
        gy = torch.randn(b, cout, hh, ww, device='cuda')

        def lib():
            y = torch.nn.functional.conv2d(torch.cat([h, m], 1), w, None, padding=pad)
            return torch.autograd.grad(y, [h, m, w], gy)

        def cl():
            y = cat_conv_cl([h, m], w, pad)
            return torch.autograd.grad(y, [h, m, w], gy)

        ra, rb = lib(), cl()
        err = max(((x - y).abs().max() / y.abs().max()).item() for x, y in zip(rb, ra))
        print('cout %3d k %s: library NCHW %.0f us, channels-last explicit %.0f us (fwd+bwd), rel err %.1e'
              % (cout, ks, timeit(lib), timeit(cl), err))


def others():
    """The other k x k convolutions of an update-block iteration (single input, no cat)."""
    b, hh, ww = 8, 68, 120
    for name, cin, cout, k in (('conv_c2', 256, 192, 3), ('conv_f1', 2, 128, 7), ('conv_f2', 128, 64, 3), ('conv', 256, 126, 3),
                               ('flow.conv1', 128, 256, 3), ('mask.0', 128, 256, 3)):
        x = torch.randn(b, cin, hh, ww, device='cuda', requires_grad=True)
        w = (torch.randn(cout, cin, k, k, device='cuda') * 0.02).requires_grad_(True)
        gy = torch.randn(b, cout, hh, ww, device='cuda')
        pad = (k // 2, k // 2)

        def lib():
            return torch.autograd.grad(torch.nn.functional.conv2d(x, w, None, padding=pad), [x, w], gy)

        def cl():
            return torch.autograd.grad(cat_conv_cl([x], w, pad), [x, w], gy)

        ra, rb = lib(), cl()
        err = max(((p - q).abs().max() / q.abs().max()).item() for p, q in zip(rb, ra))
        print('%-10s %3d -> %3d %dx%d: library NCHW %.0f us, channels-last explicit %.0f us (fwd+bwd), rel err %.1e'
              % (name, cin, cout, k, k, timeit(lib), timeit(cl), err))


if __name__ == '__main__':
    main()
    others()
