"""Do two torch streams run concurrently here? A compute-bound chain (bf16 matmuls) on one
stream and a memory-bound chain (copies) on another, timed alone and together."""
import sys
import time

import torch

dev = torch.device("cuda", 0)
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
src = torch.empty(1 << 30, device=dev, dtype=torch.uint8)
dst = torch.empty_like(src)


def chain_mm(n=20):
    for _ in range(n):
        torch.mm(a, b)


def chain_copy(n=20):
    for _ in range(n):
        dst.copy_(src)


def timed(f):
    torch.cuda.synchronize()
    t = time.perf_counter()
    f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3


for _ in range(2):
    chain_mm(); chain_copy()
print(f"matmul chain alone {timed(chain_mm):.2f} ms, copy chain alone {timed(chain_copy):.2f} ms", flush=True)
for label, main in (("null stream + pool stream", None), ("two pool streams", torch.cuda.Stream())):
    side = torch.cuda.Stream()

    def both():
        with torch.cuda.stream(side):
            chain_copy()
        if main is None:
            chain_mm()
        else:
            with torch.cuda.stream(main):
                chain_mm()

    print(f"{label}: together {timed(both):.2f} ms", flush=True)
