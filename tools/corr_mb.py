import torch, time, sys
sys.path.insert(0,'/root/repo')
from camliflow_amd import csrc
from camliflow_amd.csrc import wrapper
def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/iters*1e6
g=torch.Generator().manual_seed(0)
for (b,c,h,w) in [(1,32,144,240),(1,64,72,120),(1,96,36,60),(8,32,144,240),(8,192,17,30),(8,128,34,60),(32,128,144,240)]:
    in1=torch.randn(b,h,w,c,generator=g).cuda(); in2=torch.randn(b,h,w,c,generator=g).cuda()
    us=timeit(lambda: wrapper.CorrelationFunction.apply(in1,in2,4))
    byt=4*b*h*w*(2*c+81)
    print('%-40s %10.1f us %8.1f GB/s'%('corr2d fwd B%d C%d %dx%d'%(b,c,h,w),us,byt/us/1e3))
    go=torch.randn(b,81,h,w,device='cuda')
    a,bb=in1.clone().requires_grad_(True),in2.clone().requires_grad_(True)
    out=wrapper.CorrelationFunction.apply(a,bb,4)
    us=timeit(lambda: torch.autograd.grad(out,[a,bb],go,retain_graph=True))
    byt=4*b*h*w*(81+4*c)
    print('%-40s %10.1f us %8.1f GB/s'%('corr2d bwd B%d C%d %dx%d'%(b,c,h,w),us,byt/us/1e3))
