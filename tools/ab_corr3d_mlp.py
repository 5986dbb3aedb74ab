"""Cost MLP + neighbour sum of the point cost-volume lookup at the headline shape (B8, 2048 points, 4 levels x 16):
fused kernels (camli_corr3d_mlp_fwd/bwd) against the composed chain (2 GEMMs + 2 bias/ReLU passes + reduction), us per
forward + backward, HIP events around 10 repetitions each."""
import sys; sys.path.insert(0, '/root/repo')
import torch
from camliflow_amd.cores import runtime
from camliflow_amd.cores.blocks import MLP2d
from camliflow_amd.csrc import _lib, fused
runtime.set_backend('hip')
mlp = MLP2d(4, [32, 32], act='relu').cuda()
lookup = torch.randn(8, 4, 2048, 64, device='cuda', requires_grad=True)
gout = torch.randn(8, 128, 2048, device='cuda')
convs = [layer.conv_fn for layer in mlp.convs]
def composed():
    cost = mlp(lookup).view(8, -1, 2048, 4, 16).sum(dim=-1).permute(0, 3, 1, 2).reshape(8, -1, 2048)
    cost.backward(gout)
def fused_path():
    fused.corr3d_cost_mlp(lookup, convs[0], convs[1], 4).backward(gout)
for name, fn in (('composed', composed), ('fused', fused_path)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    print('%-9s %8.1f us per forward + backward' % (name, a.elapsed_time(b) * 100))
_lib.TIMER.reset(); _lib.TIMER.only = None; _lib.TIMER.enabled = True
for _ in range(5): fused_path()
torch.cuda.synchronize(); _lib.TIMER.enabled = False
for k, v in _lib.TIMER.summary().items(): print('%-24s %7.1f us' % (k, v['total_ms'] / v['launches'] * 1e3))
