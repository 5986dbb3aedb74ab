import sys, torch
sys.path.insert(0,'/root/repo')
from camliflow_amd.cores.blocks import MLP2d
from camliflow_amd.csrc import fused, k_nearest_neighbor
import copy
for (c,k) in [(128,16),(125,16),(64,32),(128,4),(125,16)]:
    torch.manual_seed(c+k)
    mlp = MLP2d(3,[8,32,c],act='relu').cuda()
    xyz = torch.rand(2,3,1024,device='cuda')*4
    knn = k_nearest_neighbor(xyz,xyz,32)
    gout = torch.randn(2,c,1024,k,device='cuda')
    fused.weightnet(xyz,xyz,knn,k,mlp).backward(gout)
    got=[p.grad.clone() for p in mlp.parameters()]
    mlp.zero_grad()
    offset = fused.gather_points(xyz, knn[:,:,:k]) - xyz[:,:,:,None]
    x=offset
    for conv in mlp.convs: x=torch.relu(conv.conv_fn(x))
    x.backward(gout)
    comp=[p.grad.clone() for p in mlp.parameters()]
    m64=copy.deepcopy(mlp).double(); m64.zero_grad()
    x=offset.double()
    for conv in m64.convs: x=torch.relu(conv.conv_fn(x))
    x.backward(gout.double())
    ref=[p.grad for p in m64.parameters()]
    print(c,k,'mine vs f64:',['%.1e'%((a.double()-r).norm()/r.norm()).item() for a,r in zip(got,ref)])
    print(c,k,'comp vs f64:',['%.1e'%((a.double()-r).norm()/r.norm()).item() for a,r in zip(comp,ref)])
