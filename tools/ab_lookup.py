"""Where does the all-pairs lookup spend its time?  Times camli_allpairs_lookup_fwd per pyramid level (one launch with
L = 1 each) and with the flow field at zero (windows of neighbouring source pixels then touch neighbouring rows) on the
bench shape (B 8, 68x120).  GPU box only:  python tools/ab_lookup.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from camliflow_amd.csrc import _lib  # noqa: E402


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


def main():
    lib = _lib.load()
    b, h, w = 8, 68, 120
    p = h * w
    g = torch.Generator().manual_seed(0)
    sizes = [(68, 120), (34, 60), (17, 30), (8, 15)]
    vols = [torch.randn(b * p, hh, ww, generator=g).cuda() for hh, ww in sizes]
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
    base = torch.stack([xs, ys])[None].repeat(b, 1, 1, 1)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for label, coords in (('flow ~ N(0, 3)', (base + torch.randn(b, 2, h, w, generator=g) * 3).cuda()),
                          ('flow = 0', base.clone().cuda())):
        for levels in ([0, 1, 2, 3], [0], [1], [2], [3]):
            n = len(levels)
            ptrs = (ctypes.c_void_p * n)(*[vols[l].data_ptr() for l in levels])
            hs = (ctypes.c_int * n)(*[sizes[l][0] for l in levels])
            ws = (ctypes.c_int * n)(*[sizes[l][1] for l in levels])
            out = torch.empty(b, n * 81, h, w, device='cuda')
            # a single level l > 0 is "level 0" to the kernel: pre-scale the coordinates so the windows land where they would
            cl = coords if n > 1 else (coords * (0.5 ** levels[0])).contiguous()
            us = timed(lambda: lib.camli_allpairs_lookup_fwd(ptrs, hs, ws, n, ctypes.c_void_p(cl.data_ptr()),
                                                             ctypes.c_void_p(out.data_ptr()), b, h, w, 4, stream))
            alg = 4.0 * b * p * n * (81 + 100)
            print('%-16s levels %-12s %8.1f us   %7.1f GB/s algorithmic' % (label, levels, us, alg / us / 1e3))


if __name__ == '__main__':
    main()
