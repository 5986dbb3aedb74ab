"""How much of the all-pairs gradient pyramid the lookups of one CamLiRAFT training step touch, as the adjoint GEMMs see it:
per level, the fraction of marked 32x32 blocks and the fraction of LIVE K steps of the two marked GEMMs
(g_f2: step = source block, tile = 4 target blocks; g_f1: step = target block, tile = 4 source blocks).

  python tools/marks_density.py [batch]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from camliflow_amd.cores import CamLiRAFT, runtime  # noqa: E402
from camliflow_amd.csrc import fused  # noqa: E402


def main():
    runtime.set_backend('hip')
    runtime.set_deferred_param_grads(True)
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    torch.manual_seed(0)
    model = CamLiRAFT(bench.model_cfg(12)).cuda().train()
    opt = bench.make_optimizer(model)
    batch = {k: v.cuda() for k, v in bench.synthetic_batch(b, 540, 960, 8192, 1).items()}
    bench.train_step(model, opt, batch)
    torch.cuda.synchronize()
    seen = []
    orig = fused._ptr_array

    def spy(tensors):
        if tensors and all(torch.is_tensor(t) and t.dtype == torch.uint8 for t in tensors):
            seen.append([t.clone() for t in tensors])
        return orig(tensors)

    fused._ptr_array = spy
    bench.train_step(model, opt, batch)
    torch.cuda.synchronize()
    fused._ptr_array = orig
    marks = seen[-1]           # the last uint8 list of the backward: the build adjoint's marks (cloned before the clearing kernel runs)
    for lvl, m in enumerate(marks):
        m = (m != 0)
        bs, sb, tb = m.shape
        pad_t = (-tb) % 4
        mt = torch.nn.functional.pad(m, (0, pad_t)).view(bs, sb, -1, 4).any(-1)          # [B, sb, tiles]: g_f2 live steps
        pad_s = (-sb) % 4
        ms = torch.nn.functional.pad(m, (0, 0, 0, pad_s)).view(bs, -1, 4, tb).any(2)      # [B, tiles, tb]: g_f1 live steps
        print('level %d: blocks %dx%d marked %.3f | g_f2 live steps %.3f (per tile min %.0f / mean %.0f / max %.0f of %d) | '
              'g_f1 live steps %.3f' % (lvl, sb, tb, m.float().mean().item(), mt.float().mean().item(),
                                        mt.float().sum(1).min().item(), mt.float().sum(1).mean().item(),
                                        mt.float().sum(1).max().item(), sb, ms.float().mean().item()))


if __name__ == '__main__':
    main()
