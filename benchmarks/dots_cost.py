"""What the dot part of the W_dec gradient costs at the headline shape: sphip_segmented_weight_grad on fp32 rows with
and without dcoeff (the L2-sliced kernel either way), next to the decode gather, which is what a token-major pass
computing the same dots would have to repeat (the same 51 GB of W_dec row pieces). python benchmarks/dots_cost.py"""
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


def timed(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for rnd in range(2):
    a = timed(lambda: ops.segmented_weight_grad(seg, acts, rows, W, want_dcoeff=True))
    b = timed(lambda: ops.segmented_weight_grad(seg, acts, rows, None, want_dcoeff=False))
    c = timed(lambda: ops.gather_rows_weighted(idx, acts, W, None, table_owner=W))
    print(f"gradient + dots {a:.3f} ms   gradient alone {b:.3f} ms   dots cost {a - b:.3f} ms   "
          f"a token-major pass over the same rows (the decode) {c:.3f} ms", flush=True)
