"""The headline train step on the heavy-tailed workload (SyntheticActivations(skew="zipf"), coder planted with the
dictionary) next to the uniform one: ms per step, per-entry-point times, the latent usage histogram, and the
parameter checksum of two identical runs (bitwise repeatability). Usage: python benchmarks/skew_bench.py [steps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import SparseCoderConfig, TrainConfig, ops  # noqa: E402
from sparsify_amd.trainer import SyntheticActivations, Trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
warm = 3
N, D, M, k = (int(os.environ.get(n, d)) for n, d in (("N", 16384), ("D", 4096), ("M", 131072), ("K", 192)))
dev = torch.device("cuda", 0)
optimizer = os.environ.get("OPT", "signsgd")


def run(skew):
    cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k), optimizer=optimizer, hookpoints=["h"], save_every=10**9,
                      wandb_log_frequency=10**9, save_dir="")
    src = SyntheticActivations(["h"], D, N, steps + warm, device=dev, skew=skew, atoms=M, draws=k)
    tr = Trainer(cfg, src, device=dev)
    if skew:
        src.plant(tr.saes["h"])
    batches = [src.batch(s) for s in range(steps + warm)]
    for s in range(warm):
        tr.train_micro_batch(batches[s])
    torch.cuda.synchronize()
    ops.timer.enabled = True
    ops.timer.reset()
    t0 = time.perf_counter()
    for s in range(warm, warm + steps):
        tr.train_micro_batch(batches[s])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    ops.timer.enabled = False
    kern = {n: round(v["total_ms"] / steps, 3) for n, v in ops.timer.summary().items()}
    sae = tr.saes["h"]
    with torch.no_grad():
        idx = sae.encode(batches[-1]["h"]).top_indices
    cnt = torch.bincount(idx.flatten(), minlength=M)
    f = cnt.float() / N
    usage = {"latents_over_10pct": int((f > 0.1).sum()), "latents_over_1pct": int((f > 0.01).sum()), "never": int((cnt == 0).sum()),
             "max_share": float(f.max()), "entries_in_segments_over_512": float(cnt[cnt > 512].sum() / cnt.sum())}
    chk = [float(p.detach().double().sum()) for p in sae.parameters()]
    fvu = float(tr._metrics["fvu/h"])
    del tr, batches, src
    torch.cuda.empty_cache()
    return ms, kern, usage, chk, fvu


runs = {"": (None, "zipf", "zipf"), "zipf": ("zipf",), "uniform": (None,)}[os.environ.get("ONLY", "")]
for skew in runs:
    ms, kern, usage, chk, fvu = run(skew)
    print(json.dumps({"skew": skew, "ms_per_step": round(ms, 3), "usage": usage, "checksum": chk, "fvu_sum": fvu,
                      "kernels_ms": {n: v for n, v in sorted(kern.items(), key=lambda kv: -kv[1])[:12]}}), flush=True)
