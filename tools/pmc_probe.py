"""Isolated launches of the HBM-bound kernels at the bench shapes (batch 8) plus a calibration copy of
known size, for the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in separate runs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camliflow_amd.csrc import fused

torch.manual_seed(0)
b, n = 8, 2048
# calibration: 256 MiB float4 copy (reads 256 MiB, writes 256 MiB)
src = torch.randn(64 * 1024 * 1024, device='cuda')
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)
for (c, k) in [(128, 32), (128, 16), (128, 4)]:
    feat = torch.randn(b, c, n, device='cuda', requires_grad=True)
    wgt = torch.rand(b, c, n, k, device='cuda', requires_grad=True)
    idx = torch.randint(0, n, (b, n, 32), device='cuda')
    shared = fused.SharedSetConvWeights(wgt)
    for _ in range(3):
        o = fused.pointconv_dw(feat, shared, idx, k)
    o.backward(torch.randn_like(o))
h, w = 68, 120
p = h * w
pyr = fused.AllPairsPyramid()
pyr.levels = [torch.randn(b * p, h >> l, w >> l, device='cuda') for l in range(4)]
pyr.shape = (b, h, w)
pyr.token = torch.zeros(1, device='cuda', requires_grad=True)
ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
coords = (torch.stack([xs, ys])[None].repeat(b, 1, 1, 1) + torch.randn(b, 2, h, w) * 3).cuda()
for _ in range(3):
    out = fused.allpairs_lookup(pyr, coords, 4)
out.backward(torch.randn_like(out))
torch.cuda.synchronize()
