"""Where does the CPU-vs-GPU difference of a CamLiRAFT forward come from?  (diagnostic)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from modelutils import camliraft_cfg, hashed_fill_, oracle_boundary, synthetic_inputs
from camliflow_amd.cores import CamLiRAFT, runtime

torch.manual_seed(0)
cfg = camliraft_cfg(n_iters=3)
cpu_model = hashed_fill_(CamLiRAFT(cfg)).eval()
gpu_model = CamLiRAFT(cfg); gpu_model.load_state_dict(cpu_model.state_dict()); gpu_model.cuda().eval()
inputs = synthetic_inputs(1, 128, 160, 4608)
caps = {}
def hook(store):
    def mk(name):
        def h(m, i, o):
            t = o[0] if isinstance(o, (tuple, list)) else o
            if torch.is_tensor(t) and t.is_floating_point():
                store.setdefault(name, []).append(t.detach().float().cpu())
        return h
    return mk
c_store, g_store = {}, {}
for n, m in cpu_model.named_modules():
    m.register_forward_hook(hook(c_store)(n))
for n, m in gpu_model.named_modules():
    m.register_forward_hook(hook(g_store)(n))
with oracle_boundary():
    oc = cpu_model(inputs)
backend = sys.argv[1] if len(sys.argv) > 1 else 'hip'
with runtime.use_backend(backend):
    og = gpu_model({k: v.cuda() for k, v in inputs.items()})
rows = []
for n in c_store:
    for k, (a, b) in enumerate(zip(c_store[n], g_store.get(n, []))):
        if a.shape == b.shape:
            rel = ((a - b).abs().max() / (a.abs().max() + 1e-12)).item()
            rows.append((n, k, rel, a.abs().max().item()))
seen = 0
for n, k, rel, mag in rows:
    if rel > 1e-5 and n.count('.') <= 3:
        print('%-60s call %d rel %.2e mag %.2e' % (n, k, rel, mag))
        seen += 1
        if seen > 60: break
print('final flow2d maxdiff', (oc['flow_2d'] - og['flow_2d'].cpu()).abs().max().item(), 'flow3d', (oc['flow_3d'] - og['flow_3d'].cpu()).abs().max().item())
