"""AuxK variant of the headline step (SURVEY.md 8d second run): auxk_alpha = 1/32 with about 10 %
of the latents marked dead, k_aux = d_in / 2 = 2048. Prints ms/step and the per-entry-point times."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import SparseCoderConfig, TrainConfig, ops  # noqa: E402
from sparsify_amd.trainer import SyntheticActivations, Trainer  # noqa: E402

N, D, M, k = int(os.environ.get("TOKENS", 16384)), 4096, 131072, 192
dev = torch.device("cuda", 0)
MULTI = os.environ.get("MULTI_TOPK") == "1"        # multi-topk variant instead of AuxK (sparse_coder.py:256-262)
cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k, multi_topk=MULTI), optimizer="signsgd", hookpoints=["h"],
                  auxk_alpha=0.0 if MULTI else 1 / 32,
                  dead_feature_threshold=10**6, save_every=10**9, wandb_log_frequency=10**9, save_dir="")
steps, warm = 4, 2
src = SyntheticActivations(["h"], D, N, steps + warm, device=dev)
tr = Trainer(cfg, src, device=dev)
dead = torch.rand(M, device=dev) < 0.10
tr.num_tokens_since_fired["h"][dead] = 10**7            # these stay dead unless they fire
batches = [src.batch(s) for s in range(steps + warm)]
for s in range(warm):
    tr.train_micro_batch(batches[s])
    tr.num_tokens_since_fired["h"][dead] = 10**7
torch.cuda.synchronize()
ops.timer.enabled = True
ops.timer.reset()
t0 = time.perf_counter()
for s in range(warm, warm + steps):
    tr.train_micro_batch(batches[s])
    tr.num_tokens_since_fired["h"][dead] = 10**7
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
print(f"{'multi-topk' if MULTI else 'auxk'} step: {ms:.1f} ms  ({N / ms * 1e3:.0f} tokens/s)")
for name, v in ops.timer.summary().items():
    print(f"  {name:34s} {v['total_ms'] / steps:8.2f} ms/step in {v['calls'] / steps:.0f} calls")
