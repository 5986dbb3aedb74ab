"""The update block's 3x3 convolutions part by part (forward, data gradient, weight gradient): this repo's Winograd path
(camli_wino_conv3x3 / camli_wino_wrw) against the library's convolution on the same tensors, HIP events over `reps` calls.
    python tools/wino_parts_mb.py [batch] [H] [W]"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camliflow_amd.csrc import fused

b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hh = int(sys.argv[2]) if len(sys.argv) > 2 else 68
ww = int(sys.argv[3]) if len(sys.argv) > 3 else 120
reps = 20
ARGS = ([1, 1], [1, 1], [1, 1], False, [0, 0], 1)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for cin, cout in ((256, 192), (256, 126), (128, 256), (128, 64)):
    x = torch.randn(b, cin, hh, ww, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') * (9 * cin) ** -0.5
    gy = torch.randn(b, cout, hh, ww, device='cuda')
    u, ut = fused.wino_transformed_weights(w, False), fused.wino_transformed_weights(w, True)
    rows = {
        'forward': (lambda: fused.wino_conv3x3(x, u, cout), lambda: torch.ops.aten.convolution(x, w, None, *ARGS)),
        'data gradient': (lambda: fused.wino_conv3x3(gy, ut, cin),
                          lambda: torch.ops.aten.convolution_backward(gy, x, w, None, *ARGS, [True, False, False])),
        'weight gradient': (lambda: fused.wino_wrw(x, gy),
                            lambda: torch.ops.aten.convolution_backward(gy, x, w, None, *ARGS, [False, True, False])),
    }
    flop = 2.0 * b * hh * ww * cin * cout * 9
    for name, (own, lib) in rows.items():
        t_own, t_lib = timed(own), timed(lib)
        print('B=%d %3d -> %3d %dx%d %-15s own %7.1f us (%.2f of the fp32 matrix peak, direct-equivalent)   library %7.1f us (%.2f)' % (
            b, cin, cout, hh, ww, name, t_own, flop / t_own / 1e6 / 157.3, t_lib, flop / t_lib / 1e6 / 157.3))
