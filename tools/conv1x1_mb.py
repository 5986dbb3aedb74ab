"""1x1 convolution backward: library (MIOpen, all three gradients) vs data-gradient from the library +
weight gradient as a batched GEMM over the positions (bmm + sum over the batch)."""
import sys
import time

import torch

sys.path.insert(0, '/root/repo')


def timeit(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e6


cases = [(8, 128, 128, (68, 120)), (8, 324, 128, (68, 120)), (8, 128, 128, (2048,)), (8, 384, 128, (2048,)), (8, 256, 128, (2048,)),
         (8, 128, 256, (2048,)), (8, 32, 32, (2048, 64)), (8, 4, 32, (2048, 64)), (8, 64, 96, (8192,)), (8, 144, 125, (2048,))]
for b, ci, co, sp in cases:
    x = torch.randn(b, ci, *sp, device='cuda')
    w = torch.randn(co, ci, *([1] * len(sp)), device='cuda') * 0.05
    gy = torch.randn(b, co, *sp, device='cuda')
    nd = len(sp)
    args = ([1] * nd, [0] * nd, [1] * nd, False, [0] * nd, 1)

    def lib_all():
        return torch.ops.aten.convolution_backward(gy, x, w, None, *args, [True, True, False])

    def lib_data():
        return torch.ops.aten.convolution_backward(gy, x, w, None, *args, [True, False, False])

    def gemm_w():
        return torch.bmm(gy.flatten(2), x.flatten(2).transpose(1, 2)).sum(0)

    t_all, t_data, t_w = timeit(lib_all), timeit(lib_data), timeit(gemm_w)
    ref = lib_all()[1].flatten(1)
    got = gemm_w()
    err = ((got - ref).norm() / ref.norm()).item()
    print('B%d %4d->%-4d %-12s lib dx+dw %7.1f us | lib dx %7.1f + bmm dw %7.1f = %7.1f us   rel err %.1e'
          % (b, ci, co, 'x'.join(map(str, sp)), t_all, t_data, t_w, t_data + t_w, err), flush=True)
