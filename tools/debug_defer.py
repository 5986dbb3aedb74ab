import sys, os, torch
sys.path.insert(0,'/root/repo')
import bench
from camliflow_amd.cores import CamLiRAFT, runtime
runtime.set_backend('hip'); runtime.set_overlap(True)
torch.manual_seed(0)
model = CamLiRAFT(bench.model_cfg(12)).cuda().train()
opt = bench.make_optimizer(model, capturable=False)
batch = {k: v.cuda() for k, v in bench.synthetic_batch(2, 540, 960, 8192, seed=100).items()}
bench.train_step(model, opt, batch)
runtime.set_deferred_param_grads(True)
orig = runtime.PARAM_GRADS.flush
def flush():
    e = runtime.PARAM_GRADS.entries
    print('flush: %d entries (%d reduce_batch)' % (len(e), sum(1 for v in e.values() if v[2])))
    orig()
runtime.PARAM_GRADS.flush = flush
import time
for i in range(3):
    torch.cuda.synchronize(); t=time.perf_counter(); bench.train_step(model, opt, batch); torch.cuda.synchronize(); print('step', (time.perf_counter()-t)*1e3)
runtime.set_deferred_param_grads(False)
for i in range(3):
    torch.cuda.synchronize(); t=time.perf_counter(); bench.train_step(model, opt, batch); torch.cuda.synchronize(); print('step (off)', (time.perf_counter()-t)*1e3)
