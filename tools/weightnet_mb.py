"""Micro-benchmark of camli_weightnet_fwd/bwd vs the composed path (gather + 3x conv1x1 + bias/ReLU)."""
import sys
import time

import torch

sys.path.insert(0, '/root/repo')
from camliflow_amd.cores import runtime            # noqa: E402
from camliflow_amd.cores.blocks import MLP2d       # noqa: E402
from camliflow_amd.csrc import fused, k_nearest_neighbor   # noqa: E402


def timeit(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e6


runtime.set_backend('hip')
b, n = 8, 2048
xyz = torch.rand(b, 3, n, device='cuda') * 8
knn = k_nearest_neighbor(xyz, xyz, 32)
cases = [(128, 16), (128, 32), (125, 16), (64, 32), (32, 32), (16, 16), (128, 4)]
if len(sys.argv) > 2:
    cases = [(int(sys.argv[1]), int(sys.argv[2]))]
for c, k in cases:
    mlp = MLP2d(3, [8, 32, c], act='relu').cuda()
    gout = torch.randn(b, c, n, k, device='cuda')
    with torch.no_grad():
        us_f = timeit(lambda: fused.weightnet(xyz, xyz, knn, k, mlp))

        def composed():
            off = fused.gather_points(xyz, knn[:, :, :k]) - xyz[:, :, :, None]
            return mlp(off)
        us_c = timeit(composed)
    out = fused.weightnet(xyz, xyz, knn, k, mlp)
    us_b = timeit(lambda: torch.autograd.grad(out, list(mlp.parameters()), gout, retain_graph=True))
    off = fused.gather_points(xyz, knn[:, :, :k]) - xyz[:, :, :, None]
    outc = mlp(off)
    us_cb = timeit(lambda: torch.autograd.grad(outc, list(mlp.parameters()), gout, retain_graph=True))
    byt = 4.0 * b * c * n * k
    flop = 2.0 * b * n * k * (24 + 256 + 32 * c)
    print('C%-3d k%-2d fwd %7.1f us (%5.0f GB/s write, %5.1f TFLOP/s) composed %7.1f us | bwd %7.1f us composed %7.1f us'
          % (c, k, us_f, byt / us_f / 1e3, flop / us_f / 1e6, us_c, us_b, us_cb))
