"""MotionEncoder2D + FlowHead2D.conv1 + mask-head 3x3 (models/raft_core.py:142-190) forward and backward at the step's shape,
HIP events on one stream: the Winograd epilogue fusion (CAMLI_WINO_EPILOGUE, read at import) on or off.
    CAMLI_WINO_EPILOGUE=0|1 python tools/wino_epilogue_mb.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camliflow_amd.cores import runtime
from camliflow_amd.cores.blocks import conv_bias_act
from camliflow_amd.cores.raft2d import ConvexUpsampler2D, FlowHead2D, MotionEncoder2D

runtime.set_backend('hip')
runtime.set_deferred_param_grads(os.environ.get('CAMLI_DEFER_GRADS', '1') == '1')
b, hh, ww = 8, 68, 120
torch.manual_seed(0)
enc, head, up = MotionEncoder2D(4, 4).cuda(), FlowHead2D(128).cuda(), ConvexUpsampler2D(128).cuda()
flow = torch.randn(b, 2, hh, ww, device='cuda')
corr = torch.randn(b, 324, hh, ww, device='cuda', requires_grad=True)
hidden = torch.randn(b, 128, hh, ww, device='cuda', requires_grad=True)
gm = torch.randn(b, 128, hh, ww, device='cuda')
g1, g2 = torch.randn(b, 256, hh, ww, device='cuda'), torch.randn(b, 256, hh, ww, device='cuda')


def fwd():
    motion = enc(flow, corr)
    a = conv_bias_act(head.conv1, hidden, 'relu')
    m = conv_bias_act(up.mask[0], hidden, 'relu')
    return motion, a, m


def timed(reps=10):
    tf = tb = 0.0
    for i in range(reps + 3):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        motion, a, m = fwd()
        e[1].record()
        torch.autograd.backward([motion, a, m], [gm, g1, g2])
        e[2].record()
        torch.cuda.synchronize()
        if i >= 3:
            tf += e[0].elapsed_time(e[1])
            tb += e[1].elapsed_time(e[2])
    return tf / reps * 1e3, tb / reps * 1e3


f, bk = timed()
print('CAMLI_WINO_EPILOGUE=%s  forward %.0f us  backward %.0f us' % (os.environ.get('CAMLI_WINO_EPILOGUE', '1'), f, bk))
