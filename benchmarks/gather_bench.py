"""Times sphip_gather_rows_weighted (decode) at the headline shape for both kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import ops  # noqa: E402

N, D, M, k = 16384, 4096, 131072, 192
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
W = torch.randn(M, D, device=dev, generator=g)
idx = torch.randint(0, M, (N, k), device=dev, generator=g)
acts = torch.rand(N, k, device=dev, generator=g).to(torch.bfloat16)
bias = torch.randn(D, device=dev, generator=g)
outs = {}
for mode in sys.argv[1:] or ["full", "sliced"]:
    os.environ["SPHIP_GATHER"] = mode
    for _ in range(2):
        out = ops.gather_rows_weighted(idx, acts, W, bias)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = ops.gather_rows_weighted(idx, acts, W, bias)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    outs[mode] = out
    print(f"{mode}: {ms:.3f} ms  {N * (k + 1) * D * 4 / ms / 1e6:.0f} GB/s algorithmic", flush=True)
if len(outs) == 2:
    a, b = outs.values()
    print("bitwise equal:", bool(torch.equal(a, b)))
