import sys; sys.path.insert(0,'/root/repo')
import torch
from camliflow_amd.csrc import _lib, fused
from camliflow_amd.cores import objectives, runtime
from types import SimpleNamespace as NS
runtime.set_backend('hip')
preds=[torch.randn(8,2,540,960,device='cuda',requires_grad=True) for _ in range(12)]
target=torch.cat([torch.randn(8,2,540,960),torch.ones(8,1,540,960)],1).cuda()
cfg=NS(gamma=0.8, order='l2-norm')
def run():
    loss=objectives._sequence_loss(preds,target,cfg,2); torch.autograd.grad(loss,preds)
for _ in range(2): run()
torch.cuda.synchronize(); _lib.TIMER.reset(); _lib.TIMER.only=None; _lib.TIMER.enabled=True
for _ in range(3): run()
torch.cuda.synchronize(); _lib.TIMER.enabled=False
for k,v in _lib.TIMER.summary().items(): print('%-24s %7.1f us'%(k, v['total_ms']/v['launches']*1e3))
