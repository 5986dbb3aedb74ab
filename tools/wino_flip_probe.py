import sys, numpy as np, torch
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
from modelutils import hashed_fill_
from camliflow_amd.cores import runtime
from camliflow_amd.cores.raft2d import MotionEncoder2D
from camliflow_amd.cores.blocks import conv_bias_act
from camliflow_amd.csrc import fused
runtime.set_backend('hip')
g=np.load(os.path.join(R,'tests','golden','dense_update_block.npz'))
enc=hashed_fill_(MotionEncoder2D(4,4)).cuda()
corr=torch.from_numpy(g['corr']).cuda(); flow=torch.from_numpy(g['flow']).cuda()
c1=conv_bias_act(enc.conv_c1, corr, 'relu')
pre64=torch.nn.functional.conv2d(c1.double(), enc.conv_c2.weight.double(), enc.conv_c2.bias.double(), padding=1)
for tile in (2,4):
    u=fused.wino_transformed_weights(enc.conv_c2.weight, False, tile)
    y=fused.wino_conv3x3(c1, u, 192, bias=enc.conv_c2.bias)
    d=(y.double()-pre64).abs()
    flips=((y>0)!=(pre64>0))
    print('tile',tile,'max err %.2e'%d.max().item(),'flips',int(flips.sum()),'of',flips.numel(), 'values at flips', pre64[flips].tolist()[:5])
