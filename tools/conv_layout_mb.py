"""Update-block convolutions (raft_core.py:110-197) on MIOpen: NCHW vs channels_last activations, forward and
backward (data + weight gradient), batch 8 at 68x120 -- which layout avoids the batched_transpose wrappers and what
each convolution costs against its fp32 flop.  python tools/conv_layout_mb.py [--kernels]"""
import argparse
import sys

import torch
import torch.nn.functional as F

CASES = [  # name, cin, cout, (kh, kw)
    ('gru_zr_1x5', 256, 256, (1, 5)), ('gru_q_1x5', 256, 128, (1, 5)),
    ('gru_zr_5x1', 256, 256, (5, 1)), ('gru_q_5x1', 256, 128, (5, 1)),
    ('menc_c2_3x3', 256, 192, (3, 3)), ('menc_f1_7x7', 2, 128, (7, 7)), ('menc_f2_3x3', 128, 64, (3, 3)),
    ('menc_out_3x3', 256, 126, (3, 3)), ('head_3x3', 128, 256, (3, 3)), ('head_out_3x3', 256, 2, (3, 3)),
    ('mask_1x1', 256, 576, (1, 1)), ('menc_c1_1x1', 324, 256, (1, 1)),
]


def timeit(fn, reps=10):
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


TRUNK = [  # name, cin, cout, k, stride, H, W (input), batch: the ResNet-50 stem + stages 1-2 at 544x960, both frames in one batch
    ('stem_7x7_s2', 3, 64, 7, 2, 544, 960, 16),
    ('l1_c1_1x1', 64, 64, 1, 1, 136, 240, 16), ('l1_c2_3x3', 64, 64, 3, 1, 136, 240, 16), ('l1_c3_1x1', 64, 256, 1, 1, 136, 240, 16),
    ('l1_c1b_1x1', 256, 64, 1, 1, 136, 240, 16),
    ('l2_c1_1x1', 256, 128, 1, 1, 136, 240, 16), ('l2_c2_3x3_s2', 128, 128, 3, 2, 136, 240, 16), ('l2_c3_1x1', 128, 512, 1, 1, 68, 120, 16),
    ('l2_ds_1x1_s2', 256, 512, 1, 2, 136, 240, 16), ('l2_c1b_1x1', 512, 128, 1, 1, 68, 120, 16), ('l2_c2b_3x3', 128, 128, 3, 1, 68, 120, 16),
]


def trunk(args):
    print('%-14s %-13s %9s %9s %9s   %s' % ('conv', 'layout', 'fwd us', 'bwd us', 'TF/s f+b', 'GFLOP fwd'))
    for name, cin, cout, k, stride, h, w, b in TRUNK:
        ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
        flop = 2.0 * b * ho * wo * cin * cout * k * k
        variants = ['nchw', 'nhwc'] + (['nhwc-gemm'] if k == 1 and stride == 1 else [])
        for layout in variants:
            fmt = torch.channels_last if layout.startswith('nhwc') else torch.contiguous_format
            x = torch.randn(b, cin, h, w, device='cuda').contiguous(memory_format=fmt).requires_grad_(True)
            wt = torch.randn(cout, cin, k, k, device='cuda').contiguous(memory_format=fmt).requires_grad_(True)
            if layout == 'nhwc-gemm':      # a 1x1 convolution on channels-last data IS a plain GEMM over the pixels
                def fwd():
                    return torch.matmul(x.permute(0, 2, 3, 1).reshape(-1, cin), wt.view(cout, cin).t())
            else:
                def fwd():
                    return F.conv2d(x, wt, None, stride=stride, padding=k // 2)
            y = fwd()
            gy = torch.randn_like(y)
            t_f = timeit(fwd)
            t_b = timeit(lambda: torch.autograd.grad(y, [x, wt], gy, retain_graph=True))
            print('%-14s %-13s %9.1f %9.1f %9.1f   %.1f' % (name, layout, t_f, t_b, 3 * flop / (t_f + t_b) / 1e6, flop / 1e9))
            if args.kernels:
                from torch.profiler import ProfilerActivity, profile
                with profile(activities=[ProfilerActivity.CUDA]) as prof:
                    for _ in range(3):
                        yy = fwd()
                        torch.autograd.grad(yy, [x, wt], gy)
                    torch.cuda.synchronize()
                for ev in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:6]:
                    print('      %8.1f us x%d  %s' % (ev.device_time_total / ev.count, ev.count // 3, ev.key[:110]))
            sys.stdout.flush()
            del x, wt, y, gy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--kernels', action='store_true')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--trunk', action='store_true', help='the ResNet trunk convolutions instead of the update block')
    args = ap.parse_args()
    if args.trunk:
        return trunk(args)
    b, h, w = args.batch, 68, 120
    print('%-14s %-13s %9s %9s %9s   %s' % ('conv', 'layout', 'fwd us', 'bwd us', 'TF/s f+b', 'GFLOP fwd'))
    for name, cin, cout, (kh, kw) in CASES:
        flop = 2.0 * b * h * w * cin * cout * kh * kw
        for layout in ('nchw', 'nhwc'):
            fmt = torch.channels_last if layout == 'nhwc' else torch.contiguous_format
            x = torch.randn(b, cin, h, w, device='cuda').contiguous(memory_format=fmt).requires_grad_(True)
            wt = torch.randn(cout, cin, kh, kw, device='cuda').contiguous(memory_format=fmt).requires_grad_(True)
            pad = (kh // 2, kw // 2)
            y = F.conv2d(x, wt, None, padding=pad)
            gy = torch.randn_like(y).contiguous(memory_format=fmt)
            t_f = timeit(lambda: F.conv2d(x, wt, None, padding=pad))
            t_b = timeit(lambda: torch.autograd.grad(y, [x, wt], gy, retain_graph=True))
            print('%-14s %-13s %9.1f %9.1f %9.1f   %.1f' % (name, layout, t_f, t_b, 3 * flop / (t_f + t_b) / 1e6, flop / 1e9))
            if args.kernels:
                from torch.profiler import ProfilerActivity, profile
                with profile(activities=[ProfilerActivity.CUDA]) as prof:
                    for _ in range(3):
                        yy = F.conv2d(x, wt, None, padding=pad)
                        torch.autograd.grad(yy, [x, wt], gy)
                    torch.cuda.synchronize()
                for ev in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:6]:
                    print('      %8.1f us x%d  %s' % (ev.device_time_total / ev.count, ev.count // 3, ev.key[:110]))
            sys.stdout.flush()


if __name__ == '__main__':
    main()
