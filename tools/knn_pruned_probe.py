"""Spatially pruned KNN against the brute-force families on a few cloud shapes: equality and graph-replayed time."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from camliflow_amd import csrc
from ab_knn import timed
g = torch.Generator().manual_seed(0)
def clouds(kind, b, m):
    if kind == 'uniform':
        return torch.rand(b, m, 3, generator=g) * 10
    if kind == 'clustered':
        c = torch.rand(b, 20, 3, generator=g) * 10
        pick = torch.randint(0, 20, (b, m), generator=g)
        return torch.gather(c, 1, pick[..., None].expand(-1, -1, 3)) + torch.randn(b, m, 3, generator=g) * 0.5
    ang = torch.rand(b, m, generator=g) * 2 * math.pi
    r = -8 * torch.log(torch.rand(b, m, generator=g)) + 2
    return torch.stack([r * torch.cos(ang), r * torch.sin(ang), torch.randn(b, m, generator=g) * 0.3 + 0.02 * r], dim=2)
for kind in ('uniform', 'clustered', 'disk'):
    for (m, nq, k) in [(8192, 4096, 16), (4096, 2048, 16), (16384, 4096, 16), (8192, 8192, 3)]:
        inp = clouds(kind, 8, m).cuda()
        qry = inp[:, torch.randperm(m, generator=g)[:nq]].contiguous() if nq <= m else clouds(kind, 8, nq).cuda()
        res = {}
        for mode in ('lane', 'xlane', 'pruned'):
            os.environ['CAMLI_KNN'] = mode
            out = csrc.k_nearest_neighbor(inp, qry, k)
            res[mode] = (out, timed(lambda: csrc.k_nearest_neighbor(inp, qry, k), 5))
        print('%-9s M %5d Nq %5d k %2d: lane %.1f us, xlane %.1f us, pruned %.1f us, equal %s' % (kind, m, nq, k, res['lane'][1], res['xlane'][1], res['pruned'][1], bool(torch.equal(res['lane'][0], res['pruned'][0]))))
