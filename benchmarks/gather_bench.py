"""Times the decode gather at the headline shape: token-major kernels (full, sliced) and the
L2-resident form (l2)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import ops  # noqa: E402

N, D, M, k = int(os.environ.get("TOKENS", "16384")), 4096, 131072, 192
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
W = torch.randn(M, D, device=dev, generator=g)
idx = torch.randint(0, M, (N, k), device=dev, generator=g)
acts = torch.rand(N, k, device=dev, generator=g).to(torch.bfloat16)
bias = torch.randn(D, device=dev, generator=g)
outs = {}
for mode in sys.argv[1:] or ["full", "sliced", "l2", "l2rm"]:
    # l2: L2-resident form reading the slice-major copy of the table (made once, outside the timed
    # loop, as in training where it is refreshed once per optimizer step); l2rm: row-major table
    os.environ["SPHIP_GATHER"] = "l2" if mode == "l2rm" else mode
    os.environ["SPHIP_GATHER_SM"] = "0" if mode == "l2rm" else "1"
    for _ in range(2):
        out = ops.gather_rows_weighted(idx, acts, W, bias, table_owner=W)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = ops.gather_rows_weighted(idx, acts, W, bias, table_owner=W)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    outs[mode] = out
    print(f"{mode}: {ms:.3f} ms  {N * (k + 1) * D * 4 / ms / 1e6:.0f} GB/s algorithmic", flush=True)
if "full" in outs and "sliced" in outs:
    print("full / sliced bitwise equal:", bool(torch.equal(outs["full"], outs["sliced"])))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    W.add_(0.0)                                  # bumps the version: the copy is remade
    ops.slice_major_copy(W)
e1.record()
torch.cuda.synchronize()
print(f"slice-major copy of the table (+ a 2 x 2 GiB torch pass): {e0.elapsed_time(e1) / 5:.3f} ms", flush=True)
if "l2" in outs and len(outs) > 1:
    ref = outs["sliced"] if "sliced" in outs else outs["full"]
    print("l2 vs token-major: max abs diff", float((outs["l2"] - ref).abs().max()), "of scale", float(ref.abs().max()))
