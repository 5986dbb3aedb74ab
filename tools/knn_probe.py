import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camliflow_amd import csrc
g = torch.Generator(device='cpu').manual_seed(0)
for (b, m, nq, d, k) in [(8, 2048, 2048, 3, 16), (8, 2048, 2048, 3, 3), (1, 2048, 2048, 3, 16)]:
    inp = (torch.rand(b, m, d, generator=g) * 10).cuda()
    qry = (torch.rand(b, nq, d, generator=g) * 10).cuda()
    for _ in range(2):
        csrc.k_nearest_neighbor(inp, qry, k)
torch.cuda.synchronize()
