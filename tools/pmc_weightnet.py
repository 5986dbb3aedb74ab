"""Isolated launches of camli_weightnet_fwd / _bwd at the bench shapes (batch 8, 2048 points) for the
rocprofv3 passes (kernel trace; --pmc SQ_VALU_MFMA_BUSY_CYCLES ...; --pmc WRITE_SIZE; --pmc FETCH_SIZE)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camliflow_amd.cores.blocks import MLP2d                 # noqa: E402
from camliflow_amd.csrc import fused, k_nearest_neighbor     # noqa: E402

torch.manual_seed(0)
b, n = 8, 2048
xyz = torch.rand(b, 3, n, device='cuda') * 8
knn = k_nearest_neighbor(xyz, xyz, 32)
for c, k in [(128, 16), (128, 32)]:
    mlp = MLP2d(3, [8, 32, c], act='relu').cuda()
    gout = torch.randn(b, c, n, k, device='cuda')
    for _ in range(5):
        out = fused.weightnet(xyz, xyz, knn, k, mlp)
        torch.autograd.grad(out, list(mlp.parameters()), gout)
torch.cuda.synchronize()
