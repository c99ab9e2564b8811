"""Robustness matrix at the headline widths: every step variant x tokens per step, two steps each; prints ms per step
and the FVU (must be finite). python benchmarks/matrix_probe.py [tokens ...]"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import SparseCoderConfig, TrainConfig  # noqa: E402
from sparsify_amd.trainer import SyntheticActivations, Trainer  # noqa: E402

tokens = [int(a) for a in sys.argv[1:]] or [4096, 32768, 65536]
D, M, k = 4096, 131072, 192
dev = torch.device("cuda", 0)
for N in tokens:
    for kind in ("plain", "auxk", "multitopk", "signum", "adam", "skewed", "transcode+skip"):
        cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k, multi_topk=(kind == "multitopk"),
                                                transcode=kind.startswith("transcode"), skip_connection=kind.endswith("skip")),
                          optimizer={"signum": "signum", "adam": "adam"}.get(kind, "signsgd"), hookpoints=["h"],
                          auxk_alpha=1 / 32 if kind == "auxk" else 0.0, dead_feature_threshold=10**6, save_every=10**9,
                          wandb_log_frequency=1, save_dir="")
        skew = dict(skew="zipf", atoms=M, draws=k) if kind == "skewed" else {}
        src = SyntheticActivations(["h"], D, N, 3, device=dev, transcode=kind.startswith("transcode"), **skew)
        tr = Trainer(cfg, src, device=dev)
        if kind == "skewed":
            src.plant(tr.saes["h"])
        dead = torch.rand(M, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) < 0.10
        batches = [src.batch(s) for s in range(3)]       # generated ahead: the zipf source costs more than the step
        t0 = 0.0
        for s in range(3):
            if s == 1:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            if kind == "auxk":
                tr.num_tokens_since_fired["h"][dead] = 10**7
            tr.train_micro_batch(batches[s])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 2 * 1e3
        fvu = tr.last_logged.get("fvu/h")
        ok = fvu is not None and math.isfinite(fvu)
        print(f"N={N:6d} {kind:15s} {ms:8.2f} ms/step  {N / ms * 1e3:10.0f} tokens/s  fvu {fvu:.4f}  {'ok' if ok else 'NOT FINITE'}", flush=True)
        if not ok:
            raise SystemExit(1)
        del tr, src, batches
        torch.cuda.empty_cache()
