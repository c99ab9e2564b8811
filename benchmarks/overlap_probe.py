"""Host-side timing of the overlapped gradient exchange on one GPU (one-rank RCCL group):
how long Optimizer.step() holds the host, and the step time with the exchange on the compute
stream / on the exchange stream. Usage: python benchmarks/overlap_probe.py [gloo|nccl]"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import SparseCoderConfig, TrainConfig, parallel  # noqa: E402
from sparsify_amd.trainer import SyntheticActivations, Trainer  # noqa: E402

backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend, init_method="tcp://127.0.0.1:29544", rank=0, world_size=1,
                        **({"device_id": dev} if backend == "nccl" else {}))
parallel.SignExchange.FORCE = True
D, M, k, N, steps = 4096, 131072, 192, 16384, 8
modes = [m == "1" for m in sys.argv[2]] if len(sys.argv) > 2 else [False, True]
for overlap in modes:
    cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k), optimizer="signsgd", hookpoints=["h"], save_dir="",
                      save_every=10**9, wandb_log_frequency=10**9, overlap_exchange=overlap)
    src = SyntheticActivations(["h"], D, N, steps, device=dev)
    tr = Trainer(cfg, src, device=dev)
    batches = [src.batch(s) for s in range(steps)]
    opt = tr.optimizers[0]
    host = []
    real_step = opt.step

    def timed_step(*a, **kw):
        t = time.perf_counter()
        real_step(*a, **kw)
        host.append((time.perf_counter() - t) * 1e3)

    opt.step = timed_step
    for s in range(3):
        tr.train_micro_batch(batches[s])
    torch.cuda.synchronize()
    host.clear()
    t0 = time.perf_counter()
    for s in range(3, steps):
        tr.train_micro_batch(batches[s])
    t_enq = (time.perf_counter() - t0) * 1e3 / (steps - 3)
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) * 1e3 / (steps - 3)
    print(f"overlap={overlap}: {t_all:.2f} ms/step, host enqueue {t_enq:.2f} ms/step, "
          f"host inside optimizer.step {sum(host) / len(host):.2f} ms", flush=True)
    del tr, src, batches
    torch.cuda.empty_cache()
dist.destroy_process_group()
