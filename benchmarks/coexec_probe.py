"""Does the decoder's optimizer pass (HBM-bound, 8 GiB) hide under the next step's encoder GEMM
(matrix-pipe / L2-bound) when the two run on different streams?  Headline shape; prints the GEMM
alone, the decoder step alone, the two back to back on one stream and the two on two streams.
  python benchmarks/coexec_probe.py [reps] [side-stream priority: 0 | -1]"""
import sys
import time

import torch

from sparsify_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
prio = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N, D, M = 16384, 4096, 131072
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, D, device=dev, generator=g).to(torch.bfloat16)
W = (torch.randn(M, D, device=dev, generator=g) / D ** 0.5).to(torch.bfloat16)
b = torch.zeros(M, device=dev, dtype=torch.bfloat16)
Wd = torch.randn(M, D, device=dev, generator=g)
ops.unit_norm_rows_(Wd)
G = torch.randn(M, D, device=dev, generator=g)
sm = torch.empty(M * D, device=dev)

main = torch.cuda.current_stream()
side = torch.cuda.Stream(priority=prio)


def gemm():
    return ops.encoder_preacts(x, W, b, want_tile_max=True)


def dstep():
    ops.decoder_project_signsgd_(Wd, G, 1e-6, renorm=True, slice_major=sm)


def both_serial():
    dstep()
    gemm()


def both_streams():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        dstep()
    gemm()
    main.wait_stream(side)


def both_streams_gemm_first():
    side.wait_stream(main)
    gemm()
    with torch.cuda.stream(side):
        dstep()
    main.wait_stream(side)


def timeit(fn, label):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"{label:28s} {ms:8.3f} ms", flush=True)
    return ms


for rnd in range(2):
    a = timeit(gemm, "gemm alone")
    d = timeit(dstep, "decoder step alone")
    s = timeit(both_serial, "one stream")
    c = timeit(both_streams, f"two streams (prio {prio})")
    c2 = timeit(both_streams_gemm_first, "two streams, GEMM first")
    print(f"round {rnd}: sum {a + d:.3f}  serial {s:.3f}  concurrent {c:.3f} / {c2:.3f}  hidden {s - c:+.3f} ms", flush=True)
