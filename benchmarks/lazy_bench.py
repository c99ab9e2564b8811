"""Encoder forward at the headline shape: ordinary path (GEMM writing the (N, M) matrix + top-k)
against ops.encode_topk_lazy (candidate emission, csrc/lazy.hip). Prints ms per call and the
per-entry-point HIP-event timings."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from sparsify_amd import ops

N, D, M, k = 16384, 4096, 131072, 192
if len(sys.argv) > 1:
    N = int(sys.argv[1])
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(N, D, device="cuda", generator=g).bfloat16()
W = (torch.randn(M, D, device="cuda", generator=g) / D ** 0.5).bfloat16()
b = (torch.randn(M, device="cuda", generator=g) * 0.1).bfloat16()


def eager():
    pre, tm = ops.encoder_preacts(x, W, b, want_tile_max=True)
    return ops.topk(pre, k, tile_max=tm)


def lazy():
    return ops.encode_topk_lazy(x, W, b, k)


def lazy_nofb():
    return ops.encode_topk_lazy(x, W, b, k, fallback=False)


def run(fn, reps=6):
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
    for name, fn in (("eager", eager), ("lazy", lazy), ("lazy_no_fallback", lazy_nofb)):
        print(f"{name:18s} {run(fn):8.3f} ms", flush=True)
for Ma in (16384, 24576, 32768, 40960, 49152):
    ops.lazy_split = (lambda M, k=None, Ma=Ma: Ma)
    print(f"first {Ma} latents the ordinary way: {run(lazy_nofb):8.3f} ms", flush=True)
ops.lazy_split = (lambda M, k=None: 32768)
ops.timer.enabled = True
for _ in range(4):
    lazy()
torch.cuda.synchronize()
print({k_: round(v["mean_ms"], 3) for k_, v in ops.timer.summary().items()})
ws = ops._workspace(x.device, 1, "lazy")
cnt = ws[256:256 + 4 * N].view(torch.int32)
print("chunks per token: mean %.0f max %d" % (cnt.float().mean().item(), cnt.max().item()))
