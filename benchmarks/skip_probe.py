"""Kernel times of the transcoder + skip-connection step at the headline widths (16384 tokens) next to the plain
transcoder: what the skip products cost. python benchmarks/skip_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import SparseCoderConfig, TrainConfig, ops  # noqa: E402
from sparsify_amd.trainer import SyntheticActivations, Trainer  # noqa: E402

D, M, k, N = 4096, 131072, 192, 16384
dev = torch.device("cuda", 0)
for skip in (False, True):
    cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k, transcode=True, skip_connection=skip), optimizer="signsgd",
                      hookpoints=["h"], save_every=10**9, wandb_log_frequency=10**9, save_dir="")
    src = SyntheticActivations(["h"], D, N, 7, device=dev, transcode=True)
    tr = Trainer(cfg, src, device=dev)
    batches = [src.batch(s) for s in range(7)]
    for s in range(3):
        tr.train_micro_batch(batches[s])
    torch.cuda.synchronize()
    ops.timer.enabled = True
    ops.timer.reset()
    t0 = time.perf_counter()
    for s in range(3, 7):
        tr.train_micro_batch(batches[s])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 4 * 1e3
    ops.timer.enabled = False
    kern = ops.timer.summary()
    print(f"transcode, skip={skip}: {ms:.2f} ms per step; kernels {sum(v['total_ms'] for v in kern.values()) / 4:.2f} ms", flush=True)
    for name, v in kern.items():
        if v["total_ms"] / 4 > 0.05:
            print(f"  {name:40s} {v['total_ms'] / 4:7.2f} ms/step in {v['calls'] // 4} calls", flush=True)
    del tr, src, batches
    torch.cuda.empty_cache()
