"""Which library kernel serves the level-0 all-pairs GEMM (8 x [8160,256] x [256,8160], alpha = 1/16)?  Run under
rocprofv3 --kernel-trace --stats: the kernel name carries the solution's macro tile, MFMA shape and pipelining parameters.

  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lg -o lg -- python tools/lib_gemm_name.py
"""
import torch

f1 = torch.randn(8, 256, 8160, device='cuda')
f2 = torch.randn(8, 256, 8160, device='cuda')
out = torch.empty(8, 8160, 8160, device='cuda')
for _ in range(5):
    torch.baddbmm(out, f1.transpose(1, 2), f2, beta=0, alpha=1 / 16.0, out=out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    torch.baddbmm(out, f1.transpose(1, 2), f2, beta=0, alpha=1 / 16.0, out=out)
b.record()
torch.cuda.synchronize()
print('library level-0 GEMM: %.1f us' % (a.elapsed_time(b) * 100))
