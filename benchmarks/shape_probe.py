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
    for opt in os.environ.get("OPTS", "signsgd,signum").split(","):
        transcode = os.environ.get("TRANSCODE") == "1"
        cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k, transcode=transcode, skip_connection=os.environ.get("SKIP") == "1"),
                          optimizer=opt, hookpoints=["h"], save_dir="", auxk_alpha=float(os.environ.get("AUXK", "0")),
                          save_every=10**9, wandb_log_frequency=1)
        steps = 12
        src = SyntheticActivations(["h"], D, N, steps, device=dev, transcode=transcode)
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
        # (a transcoder with a skip connection under plain SignSGD oscillates between FVU ~1 and ~10 on these
        # inputs in the CPU oracle as well: sign steps on a dense d_in x d_in matrix; only finiteness is checked there)
        ok = all(v == v and 0.0 < v < (1e6 if transcode else 1.5) for v in fv)
        if os.environ.get("KERNELS") == "1":
            from sparsify_amd import ops
            ops.timer.enabled = True
            ops.timer.reset()
            tr.train_micro_batch(batches[-1])
            torch.cuda.synchronize()
            ops.timer.enabled = False
            print("   ", {n: round(v["total_ms"], 3) for n, v in ops.timer.summary().items() if v["total_ms"] > 0.05})
        print(f"d_in={D} M={M} k={k} N={N} {opt:8s}: fvu {fv[0]:.4f} -> {fv[-1]:.4f}  {(time.perf_counter() - t0) / (steps - steps // 2) * 1e3:.2f} ms/step  "
              f"{'ok' if ok else 'BAD'}", flush=True)
        del tr, src
        torch.cuda.empty_cache()
