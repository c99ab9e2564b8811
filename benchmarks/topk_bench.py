"""Times sphip_topk on the block-maxima path at the headline shape and checks the selection
against the full-row kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import ops  # noqa: E402

N, D, M, k = 16384, 4096, 131072, 192
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, D, device=dev, generator=g).to(torch.bfloat16)
W = (torch.randn(M, D, device=dev, generator=g) / 64).to(torch.bfloat16)
b = torch.zeros(M, device=dev, dtype=torch.bfloat16)
pre, tmax = ops.encoder_preacts(x, W, b, want_tile_max=True)
del x, W
res = []
for rnd in range(3):
    v, i = ops.topk(pre, k, tile_max=tmax)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        v, i = ops.topk(pre, k, tile_max=tmax)
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 10)
v2, i2 = ops.topk(pre, k)
same = torch.equal(v.sort(-1).values, v2.sort(-1).values) and torch.equal(i.sort(-1).values, i2.sort(-1).values)
print(f"top-k (block maxima): median {sorted(res)[1]:.3f} ms; same selection as the full-row kernel: {same}", flush=True)
