"""Host-side cost of one training step: torch.profiler CPU self times by operator (the bench step is
host-bound: ~324 of 342 ms are spent enqueueing)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from camliflow_amd.cores import CamLiRAFT, runtime  # noqa: E402

runtime.set_backend('hip')
runtime.set_overlap(True)
torch.manual_seed(0)
model = CamLiRAFT(bench.model_cfg(12)).cuda().train()
opt = bench.make_optimizer(model, capturable=False)
batch = {k: v.cuda() for k, v in bench.synthetic_batch(8, 540, 960, 8192, seed=100).items()}
for _ in range(2):
    bench.train_step(model, opt, batch)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU], record_shapes=False) as prof:
    bench.train_step(model, opt, batch)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=45, max_name_column_width=60))
