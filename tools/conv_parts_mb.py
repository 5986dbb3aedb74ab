"""Device time of the three parts of the update block's library convolutions (forward, data gradient, weight gradient), each
alone, HIP events over 20 back-to-back calls, batch 8 at 68x120 -- and this repo's camli_convcl_wrw on the same shapes where it
applies.   python tools/conv_parts_mb.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camliflow_amd.csrc import fused  # noqa: E402

b, h, w = 8, 68, 120


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for (ci, co, k) in ((256, 192, 3), (128, 256, 3), (256, 126, 3), (128, 64, 3), (2, 128, 7), (128, 512, 3)):
    x = torch.randn(b, ci, h, w, device='cuda')
    wt = torch.randn(co, ci, k, k, device='cuda')
    gy = torch.randn(b, co, h, w, device='cuda')
    args = ([1, 1], [k // 2, k // 2], [1, 1], False, [0, 0], 1)
    t_f = timed(lambda: torch.ops.aten.convolution(x, wt, None, *args))
    t_d = timed(lambda: torch.ops.aten.convolution_backward(gy, x, wt, None, *args, [True, False, False]))
    t_w = timed(lambda: torch.ops.aten.convolution_backward(gy, x, wt, None, *args, [False, True, False]))
    t_b = timed(lambda: torch.ops.aten.convolution_backward(gy, x, wt, None, *args, [True, True, False]))
    flop = 2.0 * b * h * w * ci * co * k * k
    line = '%4d -> %4d %dx%d (%.1f GFLOP): forward %6.1f us, data gradient %6.1f us, weight gradient %6.1f us, both %6.1f us' % (
        ci, co, k, k, flop / 1e9, t_f, t_d, t_w, t_b)
    if (ci % 256 == 0 and co % 128 == 0) or (ci % 128 == 0 and co % 256 == 0):
        x_n, gy_n = x.permute(0, 2, 3, 1).contiguous(), gy.permute(0, 2, 3, 1).contiguous()
        taps = fused.convcl_taps(k, k, k // 2, k // 2)
        t_own = timed(lambda: fused.convcl_wrw([x_n], gy_n, taps, (k, k)))
        t_tr = timed(lambda: (fused._to_nhwc(x), fused._to_nhwc(gy)))
        line += '; own weight gradient %6.1f us (+ %5.1f us for the two NHWC transposes)' % (t_own, t_tr)
    print(line)
