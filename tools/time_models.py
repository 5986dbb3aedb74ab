"""Wall time of a training step (fwd + loss + bwd) of the other BASELINE configs on one GPU."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from modelutils import MODEL_CASES, synthetic_inputs, camliraft_cfg
import camliflow_amd.cores as cores
from camliflow_amd.cores import runtime
runtime.set_backend('hip'); runtime.set_overlap(True)

def run(name, model, inputs, train=True, reps=5):
    model = model.cuda(); model.train(train)
    inputs = {k: v.cuda() for k, v in inputs.items()}
    def step():
        if train:
            model.zero_grad(set_to_none=True)
            model(inputs); model.get_loss().backward()
        else:
            with torch.no_grad(): model(inputs)
    for _ in range(2): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    print('%-58s %8.1f ms/step' % (name, (time.perf_counter() - t) / reps * 1e3), flush=True)

torch.manual_seed(0)
_, cls, cfg_fn, _ = MODEL_CASES['camlipwc']
run('config 2: CamLiPWC 960x540 + 8192 pts, B=1, train step', getattr(cores, cls)(cfg_fn()), synthetic_inputs(1, 540, 960, 8192))
run('config 2: CamLiPWC 960x540 + 8192 pts, B=1, inference', getattr(cores, cls)(cfg_fn()), synthetic_inputs(1, 540, 960, 8192), train=False)
run('CamLiRAFT 960x540 + 8192 pts, B=1, 12 iters, train step', cores.CamLiRAFT(camliraft_cfg(12)), synthetic_inputs(1, 540, 960, 8192))
run('CamLiRAFT 960x540 + 8192 pts, B=1, 20 iters, inference', cores.CamLiRAFT(camliraft_cfg(20)), synthetic_inputs(1, 540, 960, 8192), train=False)
run('config 5 shape: CamLiRAFT 1242x375 + 16384 pts, B=1, 32 iters, inference (fp32)', cores.CamLiRAFT(camliraft_cfg(32)), synthetic_inputs(1, 375, 1242, 16384, f=721.5, zmax=90.0), train=False, reps=3)
