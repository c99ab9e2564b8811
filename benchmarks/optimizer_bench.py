"""Step time of the headline workload under each optimizer wiring of the Trainer."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import SparseCoderConfig, TrainConfig  # noqa: E402
from sparsify_amd.trainer import SyntheticActivations, Trainer  # noqa: E402

N, D, M, k = 16384, 4096, 131072, 192
dev = torch.device("cuda", 0)
for opt in sys.argv[1:] or ["signsgd", "signum", "adam"]:
    cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k), optimizer=opt, hookpoints=["h"],
                      save_every=10**9, wandb_log_frequency=10**9, save_dir="")
    steps, warm = 5, 2
    src = SyntheticActivations(["h"], D, N, steps + warm, device=dev)
    tr = Trainer(cfg, src, device=dev)
    batches = [src.batch(s) for s in range(steps + warm)]
    for s in range(warm):
        tr.train_micro_batch(batches[s])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(warm, warm + steps):
        tr.train_micro_batch(batches[s])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{opt:8s} {ms:7.2f} ms/step  {N / ms * 1e3:9.0f} tokens/s", flush=True)
    del tr, src, batches
    torch.cuda.empty_cache()
