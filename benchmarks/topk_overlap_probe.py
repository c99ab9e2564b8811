"""Does the latency-bound top-k hide under the encoder GEMM? The token batch is cut in two halves;
the top-k of the first half runs on a second stream while the GEMM of the second half runs.
Prints sequential (one GEMM + one top-k) against split-sequential and split-overlapped times."""
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
pre = torch.empty(N, M, dtype=torch.bfloat16, device=dev)
tmax = torch.empty(N // 64, M // 128, 64, dtype=torch.bfloat16, device=dev)
side = torch.cuda.Stream()
PARTS = int(os.environ.get("PARTS", "2"))
h = N // PARTS


def seq():
    ops.encoder_preacts_block(x, W, b, pre, tmax, 0, M)
    return ops.topk(pre, k, tile_max=tmax)


def split(overlap):
    main = torch.cuda.current_stream()
    outs = []
    for p in range(PARTS):
        sl, tl = slice(p * h, (p + 1) * h), slice(p * h // 64, (p + 1) * h // 64)
        ops.encoder_preacts_block(x[sl], W, b, pre[sl], tmax[tl], 0, M)
        if overlap and p < PARTS - 1:
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                outs.append(ops.topk(pre[sl], k, tile_max=tmax[tl]))
        else:
            outs.append(ops.topk(pre[sl], k, tile_max=tmax[tl]))
    if overlap:
        main.wait_stream(side)
    return outs


def t(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5


for rnd in range(3):
    print(f"sequential {t(seq):.3f} ms   split {t(lambda: split(False)):.3f} ms   split + overlap {t(lambda: split(True)):.3f} ms", flush=True)
v0, i0 = seq()
outs = split(True)
torch.cuda.synchronize()
print("identical:", torch.equal(torch.cat([o[0] for o in outs]), v0) and torch.equal(torch.cat([o[1] for o in outs]), i0))
