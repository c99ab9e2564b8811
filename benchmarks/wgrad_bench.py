"""Times sphip_segmented_weight_grad with dcoeff (the W_dec gradient) at the headline shape."""
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
rows = torch.randn(N, D, device=dev, generator=g)
seg = ops.build_latent_segments(idx, M)
outs = {}
for mode in sys.argv[1:] or ["full", "sliced", "l2"]:   # sliced = the wide kernel when dcoeff is wanted
    os.environ["SPHIP_WGRAD"] = mode
    for _ in range(2):
        out = ops.segmented_weight_grad(seg, acts, rows, W, want_bias=False, want_dcoeff=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = ops.segmented_weight_grad(seg, acts, rows, W, want_bias=False, want_dcoeff=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    outs[mode] = out
    print(f"{mode}: {ms:.3f} ms  {N * k * D * 4 / ms / 1e6:.0f} GB/s of row reads", flush=True)
if len(outs) >= 2:
    a, b = list(outs.values())[:2]
    print("gW bitwise equal:", bool(torch.equal(a[0], b[0])), " dcoeff max abs diff:",
          float((a[2] - b[2]).abs().max()), "of scale", float(a[2].abs().max()))
