"""The implicit-GEMM 5-tap convolutions (camli_conv5_fwd) against the library convolution at the GRU2D shapes of the
headline step: values (fp32, relative error) and time (HIP graph replay of 10 launches each)."""
import os
import sys

os.environ.setdefault('TENSILE_STREAMK_DATA_PARALLEL', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from camliflow_amd.csrc import _lib  # noqa: E402
from tools.ab_knn import timed  # noqa: E402


def pack(w):
    """[Cout, Cin, 1, 5] | [Cout, Cin, 5, 1] -> [Cin, 5 taps, Cout]"""
    return w.reshape(w.shape[0], w.shape[1], 5).permute(1, 2, 0).contiguous()


def conv5(in0, in1, w, vertical, bias=None):
    lib = _lib.load()
    b, c0, h, wd = in0.shape
    c1 = in1.shape[1] if in1 is not None else 0
    cout = w.shape[0]
    wp = pack(w)
    out = torch.empty(b, cout, h, wd, device=in0.device)
    _lib.launch('camli_conv5_fwd', lib.camli_conv5_fwd, in0.data_ptr(), c0, in1.data_ptr() if in1 is not None else 0, c1,
                wp.data_ptr(), bias.data_ptr() if bias is not None else 0, 0, 0, 0, out.data_ptr(), 0, 0, b, cout, h, wd,
                int(vertical), 0, 0, torch.cuda.current_stream().cuda_stream)
    return out


def main():
    g = torch.Generator().manual_seed(0)
    for (b, c0, c1, cout, h, w) in [(2, 24, 8, 40, 9, 21), (1, 128, 128, 256, 68, 120), (8, 128, 128, 256, 68, 120),
                                    (8, 128, 128, 128, 68, 120), (1, 128, 128, 256, 47, 156)]:
        for vertical in (False, True):
            x0 = torch.randn(b, c0, h, w, generator=g).cuda()
            x1 = torch.randn(b, c1, h, w, generator=g).cuda()
            ks = (5, 1) if vertical else (1, 5)
            wt = (torch.randn(cout, c0 + c1, *ks, generator=g) * (5 * (c0 + c1)) ** -0.5).cuda()
            bias = torch.randn(cout, generator=g).cuda()
            pad = (2, 0) if vertical else (0, 2)
            want = F.conv2d(torch.cat([x0, x1], 1), wt, bias, padding=pad)
            got = conv5(x0, x1, wt, vertical, bias)
            err = ((got - want).norm() / want.norm()).item()
            xc = torch.cat([x0, x1], 1)
            wp = pack(wt)
            out = torch.empty_like(want)
            lib = _lib.load()
            st = torch.cuda.current_stream().cuda_stream

            def mine():
                lib.camli_conv5_fwd(x0.data_ptr(), c0, x1.data_ptr(), c1, wp.data_ptr(), bias.data_ptr(), 0, 0, 0, out.data_ptr(), 0, 0,
                                    b, cout, h, w, int(vertical), 0, 0, torch.cuda.current_stream().cuda_stream)
            t_lib = timed(lambda: F.conv2d(xc, wt, bias, padding=pad), 5, per_graph=10)
            t_cat = timed(lambda: F.conv2d(torch.cat([x0, x1], 1), wt, bias, padding=pad), 5, per_graph=10)
            t_own = timed(mine, 5, per_graph=10)
            flop = 2.0 * b * h * w * cout * (c0 + c1) * 5
            print('B%d %d+%d->%d %dx%d %s: rel err %.2e | library %.1f us (%.1f TF/s), with cat %.1f us | own %.1f us (%.1f TF/s)'
                  % (b, c0, c1, cout, h, w, '5x1' if vertical else '1x5', err, t_lib, flop / t_lib * 1e-6, t_cat, t_own,
                     flop / t_own * 1e-6), flush=True)


if __name__ == '__main__':
    main()
