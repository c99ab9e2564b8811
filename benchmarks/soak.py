"""Longer run at the headline shape: FVU stays finite and falls, dead-latent counters and memory stay
bounded. Batches are generated inside the loop (about 2 ms per step on top of the train step). Usage: python benchmarks/soak.py [steps] [optimizer]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import SparseCoderConfig, TrainConfig  # noqa: E402
from sparsify_amd.trainer import SyntheticActivations, Trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
opt = sys.argv[2] if len(sys.argv) > 2 else "signsgd"
dev = torch.device("cuda", 0)
cfg = TrainConfig(sae=SparseCoderConfig(num_latents=131072, k=192), optimizer=opt, hookpoints=["h"], save_dir="",
                  save_every=10**9, wandb_log_frequency=20, auxk_alpha=1 / 32, dead_feature_threshold=16384 * 30)
src = SyntheticActivations(["h"], 4096, 16384, steps, device=dev)
tr = Trainer(cfg, src, device=dev)
t0 = time.perf_counter()
for s in range(steps):
    tr.train_micro_batch(src.batch(s))
    if (s + 1) % 20 == 0:
        torch.cuda.synchronize()
        info = tr.last_logged
        print(f"step {s + 1}: fvu {info.get('fvu/h'):.4f} auxk {info.get('auxk/h', 0.0):.4f} dead {info.get('dead_pct/h'):.3f} "
              f"mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB  {(time.perf_counter() - t0) / (s + 1) * 1e3:.1f} ms/step",
              flush=True)
torch.cuda.synchronize()
sae = tr.saes["h"]
sums = [float(p.detach().double().sum()) for p in (sae.encoder.weight, sae.encoder.bias, sae.W_dec, sae.b_dec)]
print("parameter checksums (two runs of the same command must print the same line):", " ".join(f"{v:.10e}" for v in sums), flush=True)
