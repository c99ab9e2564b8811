"""Times the compact AuxK selection (k_aux = 2048 of ~13 k dead latents' pre-activations per token)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import ops  # noqa: E402

N, Md, k = 16384, 13120, 2048
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
pre = torch.randn(N, Md, device=dev, generator=g).clamp_min(0).to(torch.bfloat16)
for tag in ("resident", "sweeps"):
    if tag == "sweeps":
        os.environ["SPHIP_TOPK_NO_RESIDENT"] = "1"
    v, i = ops.topk(pre, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        v, i = ops.topk(pre, k)
    e1.record()
    torch.cuda.synchronize()
    print(f"{tag}: {e0.elapsed_time(e1) / 5:.3f} ms  checksum {int(i.sum())}", flush=True)
