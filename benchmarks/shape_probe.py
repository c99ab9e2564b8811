"""A few steps of the Trainer at shapes other than the headline's (widths that are not a power of
two, slice counts that do not divide by the 8 XCDs): FVU must stay finite and near its start (on
noise inputs it first rises for some tens of steps, as at the headline shape and in the oracle). Usage:
python benchmarks/shape_probe.py [d_in num_latents k tokens] ..."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import SparseCoderConfig, TrainConfig  # noqa: E402
from sparsify_amd.trainer import SyntheticActivations, Trainer  # noqa: E402

shapes = [(2304, 36864, 64, 8192), (1600, 51200, 32, 16384), (5120, 81920, 128, 8192), (1000, 8000, 50, 4096)]
if len(sys.argv) > 1:
    a = [int(v) for v in sys.argv[1:]]
    shapes = [tuple(a[i:i + 4]) for i in range(0, len(a), 4)]
dev = torch.device("cuda", 0)
for D, M, k, N in shapes:
    for opt in ("signsgd", "signum"):
        cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k), optimizer=opt, hookpoints=["h"], save_dir="",
                          save_every=10**9, wandb_log_frequency=1)
        steps = 12
        src = SyntheticActivations(["h"], D, N, steps, device=dev)
        tr = Trainer(cfg, src, device=dev)
        fv = []
        batches = [src.batch(s) for s in range(steps)]
        for s in range(steps):
            if s == steps // 2:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            tr.train_micro_batch(batches[s])
            fv.append(tr.last_logged.get("fvu/h"))
        torch.cuda.synchronize()
        ok = all(v == v and 0.0 < v < 1.5 for v in fv)
        print(f"d_in={D} M={M} k={k} N={N} {opt:8s}: fvu {fv[0]:.4f} -> {fv[-1]:.4f}  {(time.perf_counter() - t0) / (steps - steps // 2) * 1e3:.2f} ms/step  "
              f"{'ok' if ok else 'BAD'}", flush=True)
        del tr, src
        torch.cuda.empty_cache()
