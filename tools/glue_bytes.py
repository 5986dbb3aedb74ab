"""Where the torch element-wise / copy glue of one CamLiRAFT training step comes from, by bytes moved: every aten op that is
not a view is attributed to the innermost camliflow_amd/ frame (forward) or to "autograd" (backward thread) with the shape
of its result.  Bytes = result bytes x (1 + tensor inputs of the same size): a proxy for the HBM time of these HBM-bound ops.

  python tools/glue_bytes.py [batch]
"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from camliflow_amd.cores import CamLiRAFT, runtime  # noqa: E402

WATCH = ('aten.add', 'aten.cat', 'aten.clone', 'aten.copy_', 'aten._to_copy', 'aten.mul', 'aten.sub', 'aten.div', 'aten.fill_',
         'aten.zero_', 'aten.zeros', 'aten.zeros_like', 'aten.new_zeros', 'aten.sum', 'aten.neg', 'aten.where', 'aten.index',
         'aten.gather', 'aten.stack', 'aten.mean', 'aten.sigmoid', 'aten.tanh', 'aten.relu', 'aten.leaky_relu', 'aten.nan_to_num',
         'aten.threshold_backward', 'aten.slice_backward', 'aten.select_backward', 'aten.constant_pad_nd', 'aten.abs',
         'aten.sqrt', 'aten.rsqrt', 'aten.pow', 'aten.exp', 'aten.log', 'aten.contiguous', 'aten.index_put', 'aten.scatter',
         'aten.masked_fill', 'aten.new_empty_strided', 'aten.upsample', 'aten.avg_pool2d', 'aten.repeat_interleave',
         'aten.expand_copy', 'aten.linalg_vector_norm', 'aten.maximum', 'aten.minimum', 'aten.clamp', 'aten.ones_like',
         'aten.le', 'aten.gt', 'aten.lt', 'aten.ge', 'aten.eq', 'aten.logical_and')


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.defaultdict(lambda: [0, 0.0])

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if name.startswith(WATCH):
            res = out if torch.is_tensor(out) else (out[0] if isinstance(out, (tuple, list)) and out and torch.is_tensor(out[0]) else None)
            if res is not None and res.is_cuda:
                flat = []
                for a in args:
                    flat.extend(a if isinstance(a, (list, tuple)) else [a])
                same = sum(1 for a in flat if torch.is_tensor(a) and a.numel() == res.numel())
                if name.startswith('aten.cat'):
                    same = 1
                nbytes = res.numel() * res.element_size() * (1 + same)
                where = 'autograd'
                for fr in reversed(traceback.extract_stack(limit=40)):
                    if '/camliflow_amd/' in fr.filename or fr.filename.endswith('bench.py'):
                        where = '%s:%d' % (os.path.basename(fr.filename), fr.lineno)
                        break
                row = self.rows[(where, name.replace('aten.', ''), tuple(res.shape))]
                row[0] += 1
                row[1] += nbytes
        return out


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
    census = Census()
    with census:
        bench.train_step(model, opt, batch)
    torch.cuda.synchronize()
    total = sum(v[1] for v in census.rows.values())
    print('total %.1f MB in %d ops' % (total / 1e6, sum(v[0] for v in census.rows.values())))
    for (where, name, shape), (n, nbytes) in sorted(census.rows.items(), key=lambda kv: -kv[1][1])[:90]:
        print('%8.1f MB %5d  %-24s %-28s %s' % (nbytes / 1e6, n, where, name, list(shape)))


if __name__ == '__main__':
    main()
