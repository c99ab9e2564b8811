"""Which torch (aten) ops still run inside a headline train step, with shapes and call sites:
the package's own work goes through the C ABI, so whatever shows up here is host-side glue."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import SparseCoderConfig, TrainConfig  # noqa: E402
from sparsify_amd.trainer import SyntheticActivations, Trainer  # noqa: E402

N, D, M, k = 16384, 4096, 131072, 192
dev = torch.device("cuda", 0)
cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k), optimizer=os.environ.get("OPT", "signsgd"), hookpoints=["h"],
                  auxk_alpha=0.0, save_every=10**9, wandb_log_frequency=10**9, save_dir="")
src = SyntheticActivations(["h"], D, N, 4, device=dev)
tr = Trainer(cfg, src, device=dev)
batches = [src.batch(s) for s in range(4)]
for s in range(3):
    tr.train_micro_batch(batches[s])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.train_micro_batch(batches[3])
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=8)
rows = []
for e in ka:
    dt = e.device_time_total
    if e.key.startswith("aten::") and dt > 15:
        st = [f for f in (e.stack or []) if "sparsify_amd" in f or "bench" in f]
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:70], st[0][-75:] if st else ""))
for r in sorted(rows, reverse=True)[:30]:
    print(f"{r[0]:9.1f} us x{r[1]:<3d} {r[2]:24s} {r[3]:70s} {r[4]}")
