import sys, torch
sys.path.insert(0, '/root/repo')
from camliflow_amd import csrc
from camliflow_amd.csrc import _lib
from camliflow_amd.cores import runtime
_lib.load(); runtime.set_backend('hip')
inp = torch.rand(8, 256, 3, device='cuda') * 10; q = torch.rand(8, 2048, 3, device='cuda') * 10
for _ in range(3): csrc.k_nearest_neighbor(inp, q, 16)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
_lib.TIMER.reset(); _lib.TIMER.enabled = True
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): csrc.k_nearest_neighbor(inp, q, 16)
    torch.cuda.synchronize()
_lib.TIMER.enabled = False
evs = prof.profiler.kineto_results.events()
print(len(evs))
for e in evs:
    n = e.name()
    if 'hip' in n.lower() or 'knn' in n.lower() or 'Event' in n:
        print(n[:70], str(e.device_type()), e.correlation_id(), e.start_ns(), e.duration_ns())
