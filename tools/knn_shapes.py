"""Which KNN searches does one training step of the headline configuration issue?  Patches the launch hook and prints
(entry point arguments B, M, Nq, D, k | level sizes) -> count, plus the HIP-event time of each shape (single lane)."""
import collections
import json
import os
import sys

os.environ.setdefault('TENSILE_STREAMK_DATA_PARALLEL', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from camliflow_amd.cores import runtime  # noqa: E402
from camliflow_amd.csrc import _lib  # noqa: E402


def main():
    args = bench.NS(config='camliraft', model='camliraft', height=540, width=960, points=8192, iters=12, batch=8, mode='train')
    _lib.load()
    runtime.set_backend('hip')
    runtime.set_overlap(False)
    torch.manual_seed(0)
    model = bench.build_model(args).cuda().train()
    opt = bench.make_optimizer(model)
    batch = {k: v.cuda() for k, v in bench.synthetic_batch(8, 540, 960, 8192, seed=100).items()}
    for _ in range(2):
        bench.train_step(model, opt, batch)
    torch.cuda.synchronize()
    shapes = collections.OrderedDict()
    original = _lib.launch

    def spy(name, fn, *a, **kw):
        if name != 'camli_knn':
            return original(name, fn, *a, **kw)
        key = tuple(int(x) for x in a[-6:-1]) if fn is _lib.load().camli_knn else ('prefixes',) + tuple(int(x) for x in a[-6:-1])
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = original(name, fn, *a, **kw)
        e.record()
        shapes.setdefault(key, []).append((s, e))
        return out
    _lib.launch = spy
    import camliflow_amd.csrc.wrapper as wrapper
    import camliflow_amd.csrc.fused as fused
    bench.train_step(model, opt, batch)
    torch.cuda.synchronize()
    _lib.launch = original
    rows = []
    for key, evs in shapes.items():
        us = [s.elapsed_time(e) * 1e3 for s, e in evs]
        rows.append({'shape(B,M,Nq,D,k)': key, 'calls': len(evs), 'avg_us': round(sum(us) / len(us), 1), 'total_us': round(sum(us), 1)})
        print(rows[-1], flush=True)
    print('total', round(sum(r['total_us'] for r in rows), 1), 'us in', sum(r['calls'] for r in rows), 'launches')
    with open(os.path.join(ROOT, 'gpurun_out', 'knn_shapes.json'), 'w') as f:
        json.dump(rows, f, indent=1)


if __name__ == '__main__':
    main()
