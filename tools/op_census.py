"""Which lines of camliflow_amd/cores issue the small aten ops of one training step (forward ops are
attributed to the innermost cores/ frame; backward ops have no Python frame and are counted per op)."""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from camliflow_amd.cores import CamLiRAFT, runtime  # noqa: E402

SKIP = ('aten.view', 'aten.reshape', 'aten._unsafe_view', 'aten.select', 'aten.slice', 'aten.detach', 'aten.alias',
        'aten.expand', 'aten.permute', 'aten.transpose', 'aten.t.', 'aten.unsqueeze', 'aten.squeeze', 'aten.split',
        'aten.as_strided', 'aten.empty', 'aten.unbind', 'aten.size', 'aten.stride', 'aten.is_', 'aten.sym_', 'aten._local_scalar')


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.fwd = collections.Counter()
        self.bwd = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            where = None
            for fr in reversed(traceback.extract_stack(limit=40)):
                if '/camliflow_amd/cores/' in fr.filename or fr.filename.endswith('bench.py'):
                    where = '%s:%d' % (os.path.basename(fr.filename), fr.lineno)
                    break
            if where:
                self.fwd[(where, name)] += 1
            else:
                self.bwd[name] += 1
        return func(*args, **(kwargs or {}))


runtime.set_backend('hip')
runtime.set_overlap(True)
torch.manual_seed(0)
model = CamLiRAFT(bench.model_cfg(12)).cuda().train()
opt = bench.make_optimizer(model, capturable=False)
batch = {k: v.cuda() for k, v in bench.synthetic_batch(2, 540, 960, 8192, seed=100).items()}
bench.train_step(model, opt, batch)
torch.cuda.synchronize()
census = Census()
with census:
    bench.train_step(model, opt, batch)
torch.cuda.synchronize()
print('--- forward ops by source line (top 60)')
for (where, name), n in census.fwd.most_common(60):
    print('%5d  %-26s %s' % (n, where, name))
print('--- ops without a cores/ frame (autograd thread, optimizer): top 25')
for name, n in census.bwd.most_common(25):
    print('%5d  %s' % (n, name))
