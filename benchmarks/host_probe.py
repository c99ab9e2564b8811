"""Where the host spends its time while it enqueues one train step (no synchronisation between
the sections): a section that takes as long as the GPU work in front of it blocks on the device."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import SparseCoderConfig, TrainConfig, ops  # noqa: E402
from sparsify_amd.trainer import SyntheticActivations, Trainer  # noqa: E402

dev = torch.device("cuda", 0)
D, M, k, N, steps = 4096, 131072, 192, 16384, 7
cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k), optimizer="signsgd", hookpoints=["h"], save_dir="",
                  save_every=10**9, wandb_log_frequency=10**9)
src = SyntheticActivations(["h"], D, N, steps, device=dev)
tr = Trainer(cfg, src, device=dev)
sae, opt = tr.saes["h"], tr.optimizers[0]
batches = [src.batch(s)["h"] for s in range(steps)]
for s in range(steps):
    if s == 3:
        torch.cuda.synchronize()
        # per-op host time inside the forward, once
        real = {}
        for name in ("center_cast", "encoder_preacts", "topk", "gather_rows_weighted", "residual_sumsq",
                     "total_variance", "scale_by_ratio", "build_segments", "segmented_weight_grad", "vec_mat",
                     "colsum", "signsgd_step_", "decoder_project_signsgd_"):
            if hasattr(ops, name):
                f = getattr(ops, name)

                def wrap(f=f, name=name):
                    def g(*a, **kw):
                        t = time.perf_counter()
                        r = f(*a, **kw)
                        print(f"      {name}: {(time.perf_counter() - t) * 1e3:.2f} ms host", flush=True)
                        return r
                    return g
                real[name] = f
                setattr(ops, name, wrap())
    t0 = time.perf_counter()
    out = sae(x=batches[s], y=batches[s])
    t1 = time.perf_counter()
    out.fvu.backward()
    t2 = time.perf_counter()
    opt.step()
    opt.zero_grad()
    t3 = time.perf_counter()
    print(f"step {s}: forward {1e3 * (t1 - t0):.2f}  backward {1e3 * (t2 - t1):.2f}  optimizer {1e3 * (t3 - t2):.2f} ms host",
          flush=True)
    if s == 3:
        for name, f in real.items():
            setattr(ops, name, f)
torch.cuda.synchronize()
