"""Host-side (enqueue) cost of the library convolution calls of the update block: python time per call, no synchronisation
inside the timed loops (the device runs behind).  python tools/conv_host_cost.py"""
import time
import torch

torch.backends.cudnn.benchmark = False
b, h, w = 8, 68, 120
for (ci, co, k) in ((256, 192, 3), (128, 256, 3), (256, 126, 3), (128, 64, 3), (2, 128, 7)):
    x = torch.randn(b, ci, h, w, device='cuda', requires_grad=True)
    wt = torch.randn(co, ci, k, k, device='cuda', requires_grad=True)
    gy = torch.randn(b, co, h, w, device='cuda')
    for _ in range(3):
        y = torch.nn.functional.conv2d(x, wt, padding=k // 2)
        torch.autograd.grad(y, [x, wt], gy)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    ys = [torch.nn.functional.conv2d(x, wt, padding=k // 2) for _ in range(n)]
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for y in ys:
        torch.ops.aten.convolution_backward(gy, x, wt, None, [1, 1], [k // 2, k // 2], [1, 1], False, [0, 0], 1, [True, True, False])
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print('%4d -> %4d %dx%d: forward host %6.1f us / call (device-complete %7.1f us), convolution_backward host %6.1f us / call '
          '(device-complete %7.1f us)' % (ci, co, k, k, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6, (t3 - t2) / n * 1e6, (t4 - t2) / n * 1e6))
