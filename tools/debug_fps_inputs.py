import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from modelutils import synthetic_inputs, oracle_boundary
from camliflow_amd.cores.camliraft import _camera_pair
from camliflow_amd.cores.geometry import persp2paral, build_pc_pyramid
from camliflow_amd.cores import runtime
inp = synthetic_inputs(1, 128, 160, 4608)
pc1, pc2 = inp['pcs'][:, :3], inp['pcs'][:, 3:]
persp, paral = _camera_pair(128, 160, inp['intrinsics'])
c1, c2 = persp2paral(pc1, persp, paral), persp2paral(pc2, persp, paral)
perspg = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in persp.items()}
g1, g2 = persp2paral(pc1.cuda(), perspg, paral), persp2paral(pc2.cuda(), perspg, paral)
print('persp2paral cpu-vs-gpu max abs diff', (c1 - g1.cpu()).abs().max().item(), 'n differing', (c1 != g1.cpu()).sum().item(), 'of', c1.numel())
with oracle_boundary():
    _, _, ic1, _ = build_pc_pyramid(c1, c2, [4096, 2048])
_, _, ig_same, _ = build_pc_pyramid(c1.cuda(), c2.cuda(), [4096, 2048])
_, _, ig_own, _ = build_pc_pyramid(g1, g2, [4096, 2048])
print('FPS identical inputs: equal =', torch.equal(ic1[1], ig_same[1].cpu()))
neq = (ic1[1] != ig_own[1].cpu())
print('FPS own-device inputs: equal =', not neq.any().item(), 'first mismatch at', neq[0].nonzero()[:1].flatten().tolist())
