"""Where does the FIRST training step of a process spend its time?  (the 'two-lane start-up stall'.)
Times the first three steps of bench.py's workload in one lane, counts the files of MIOpen's user kernel cache before and
after, and -- run twice on the same box -- shows what a second process inherits.   python tools/first_step_probe.py [lanes]"""
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
t_start = time.perf_counter()


def cache_files():
    pats = [os.path.expanduser('~/.cache/miopen/**/*'), os.path.expanduser('~/.config/miopen/**/*'), '/tmp/miopen*/**/*']
    files = [f for p in pats for f in glob.glob(p, recursive=True) if os.path.isfile(f)]
    return len(files), sum(os.path.getsize(f) for f in files)


import torch  # noqa: E402
import bench  # noqa: E402
from camliflow_amd.cores import runtime  # noqa: E402
from camliflow_amd.csrc import _lib  # noqa: E402

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 1
print('import torch + bench: %.1f s; MIOpen user cache before: %d files, %.1f MB' % ((time.perf_counter() - t_start,) + tuple(
    (cache_files()[0], cache_files()[1] / 1e6))), flush=True)
_lib.load()
runtime.set_backend('hip')
runtime.set_overlap(lanes == 2, prime_first_pass=False)
runtime.set_deferred_param_grads(True)
args = bench.NS(model='camliraft', iters=12)
torch.manual_seed(0)
model = bench.build_model(args).cuda().train()
opt = bench.make_optimizer(model)
batch = {k: v.cuda() for k, v in bench.synthetic_batch(8, 540, 960, 8192, seed=100).items()}
torch.cuda.synchronize()
for i in range(3):
    t0 = time.perf_counter()
    model(batch)
    loss = model.get_loss()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    opt.zero_grad(set_to_none=True)
    model.clear_metrics()
    torch.cuda.synchronize()
    n, size = cache_files()
    print('step %d (%d lane%s): forward %.2f s, backward %.2f s, optimizer %.2f s; MIOpen user cache now %d files, %.1f MB'
          % (i + 1, lanes, 's' if lanes > 1 else '', t1 - t0, t2 - t1, time.perf_counter() - t2, n, size / 1e6), flush=True)
