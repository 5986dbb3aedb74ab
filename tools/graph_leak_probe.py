"""Which tensors keep an autograd graph alive between two training steps?  (torch's "AccumulateGrad node's stream does not
match ..." warning names a graph that outlives its step; a growing count here is a leak.)  Runs small CamLiRAFT steps on the
HIP path, two lanes, and lists every live tensor that still has a grad_fn after a step and who refers to it.
    python tools/graph_leak_probe.py [steps]"""
import gc
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench  # noqa: E402

LIVE = []


def describe(obj, depth, indent, seen):
    if depth == 0 or id(obj) in seen:
        return
    seen.add(id(obj))
    refs = gc.get_referrers(obj)
    for r in refs:
        if r is LIVE or r is refs or type(r).__name__ == 'frame' or id(r) in seen:
            continue
        name = type(r).__name__
        extra = ''
        if isinstance(r, dict):
            extra = ' keys ' + ','.join(str(k) for k, v in r.items() if v is obj)[:60]
        elif name in ('function', 'method'):
            extra = ' ' + getattr(r, '__qualname__', '?')
        elif name not in ('list', 'tuple', 'cell'):
            extra = ' ' + repr(r)[:70]
        print(indent + name + extra)
        describe(r, depth - 1, indent + '  ', seen)
    del refs


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    from camliflow_amd.cores import CamLiRAFT, runtime
    from camliflow_amd.csrc import _lib
    _lib.load()
    runtime.set_backend('hip')
    runtime.set_overlap(os.environ.get('CAMLI_OVERLAP', '1') == '1')
    runtime.set_deferred_param_grads(os.environ.get('CAMLI_DEFER_GRADS', '1') == '1')
    torch.manual_seed(0)
    model = CamLiRAFT(bench.model_cfg(3)).cuda().train()
    optimizer = bench.make_optimizer(model)
    batch = {k: v.cuda() for k, v in bench.synthetic_batch(2, 256, 320, 8192, seed=1).items()}
    for step in range(steps):
        bench.train_step(model, optimizer, batch)
        torch.cuda.synchronize()
        gc.collect()
        print('after step %d: %.1f MB allocated' % (step, torch.cuda.memory_allocated() / 2**20))
        del LIVE[:]
        for o in gc.get_objects():
            if isinstance(o, torch.Tensor) and o.grad_fn is not None:
                LIVE.append(o)
        o = None
        print('   %d live tensors with a grad_fn: %s' % (len(LIVE), sorted(set(type(t.grad_fn).__name__ for t in LIVE))))
        if step == steps - 1:
            shown = 0
            for t in LIVE:
                if len(gc.get_referrers(t)) <= 1 and shown < 3:       # LIVE only: nothing in Python holds it
                    print('   held from C++ only:', tuple(t.shape), type(t.grad_fn).__name__, 'refcount', sys.getrefcount(t))
                    shown += 1
            shown = 0
            for t in LIVE:
                if len(gc.get_referrers(t)) > 1 and shown < 8:
                    print('   held from Python:', tuple(t.shape), type(t.grad_fn).__name__)
                    describe(t, 4, '      ', set())
                    shown += 1
        del LIVE[:]


if __name__ == '__main__':
    main()
