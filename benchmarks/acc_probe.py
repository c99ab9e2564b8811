"""Gradient accumulation at the headline shape (grad_acc_steps = 4, 16384 tokens per micro-step): ms per micro-step
with the weight-gradient kernels adding into .grad themselves (TrainConfig.accumulate_in_kernel) and with autograd's
own accumulation pass. python benchmarks/acc_probe.py [optimizer]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import SparseCoderConfig, TrainConfig, ops  # noqa: E402
from sparsify_amd.trainer import SyntheticActivations, Trainer  # noqa: E402

opt = sys.argv[1] if len(sys.argv) > 1 else "signsgd"
D, M, k, N, ACC = 4096, 131072, 192, 16384, 4
dev = torch.device("cuda", 0)
for in_kernel in (True, False, True, False):
    ops.clear_grad_sinks()
    cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k), optimizer=opt, hookpoints=["h"], grad_acc_steps=ACC,
                      accumulate_in_kernel=in_kernel, save_every=10**9, wandb_log_frequency=10**9, save_dir="")
    src = SyntheticActivations(["h"], D, N, 3 * ACC, device=dev)
    tr = Trainer(cfg, src, device=dev)
    batches = [src.batch(s) for s in range(3 * ACC)]
    t0 = 0.0
    for s, b in enumerate(batches):
        if s == ACC:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        tr.train_micro_batch(b)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / (2 * ACC) * 1e3
    print(f"{opt} accumulate_in_kernel={in_kernel}: {ms:.2f} ms per micro-step ({N / ms * 1e3:.0f} tokens/s)", flush=True)
    del tr, src, batches
    torch.cuda.empty_cache()
