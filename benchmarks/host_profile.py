"""Host time of a launch-bound step (BASELINE configs[0] widths, 4096 tokens): cProfile over 300 steps, the 35 functions
with the most own time. python benchmarks/host_profile.py"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import SparseCoderConfig, TrainConfig  # noqa: E402
from sparsify_amd.trainer import SyntheticActivations, Trainer  # noqa: E402

D, M, k, N, STEPS = 768, 24576, 32, 4096, 300
dev = torch.device("cuda", 0)
cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k), optimizer="signsgd", hookpoints=["h"], save_every=10**9,
                  wandb_log_frequency=10**9, save_dir="")
src = SyntheticActivations(["h"], D, N, 8, device=dev)
tr = Trainer(cfg, src, device=dev)
batches = [src.batch(s) for s in range(8)]
for s in range(20):
    tr.train_micro_batch(batches[s % 8])
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(STEPS):
    tr.train_micro_batch(batches[s % 8])
torch.cuda.synchronize()
print(f"unprofiled: {(time.perf_counter() - t0) / STEPS * 1e3:.3f} ms per step", flush=True)
pr = cProfile.Profile()
pr.enable()
for s in range(STEPS):
    tr.train_micro_batch(batches[s % 8])
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
