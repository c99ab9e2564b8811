"""Times the fused encoder weight gradient + SignSGD step (sphip_segmented_weight_grad_signsgd,
bf16 rows, L2-sliced kernel) at the headline shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import ops  # noqa: E402

N, D, M, k = 16384, 4096, 131072, 192
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
W = torch.randn(M, D, device=dev, generator=g)
shadow = W.to(torch.bfloat16)
idx = torch.randint(0, M, (N, k), device=dev, generator=g)
gv = torch.randn(N, k, device=dev, generator=g).to(torch.bfloat16)
rows = torch.randn(N, D, device=dev, generator=g).to(torch.bfloat16)
shift = torch.randn(D, device=dev, generator=g)
seg = ops.build_latent_segments(idx, M)
cs = torch.randn(M, device=dev, generator=g)          # the b_dec partial sums ride along, as in the train step
res = []
for rnd in range(3):
    out = ops.segmented_weight_grad_signsgd_(seg, gv, rows, W, 1e-4, shadow, row_shift=shift, coeff_sums=cs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = ops.segmented_weight_grad_signsgd_(seg, gv, rows, W, 1e-4, shadow, row_shift=shift, coeff_sums=cs)
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 5)
res.sort()
print(f"fused encoder gradient: median {res[1]:.3f} ms (min {res[0]:.3f})  {N * k * D * 2 / res[1] / 1e6:.0f} GB/s of row reads",
      flush=True)
print("checksums:", float(W.double().sum()), float(shadow.double().sum()), float(out[1].double().sum()), flush=True)
