"""SCHED 0 against SCHED 1 of the 256-tile encoder GEMM (SPHIP_GEMM_SCHED): same MFMA chains in
the same order, so the outputs and block maxima must be bit-identical, at ragged and full shapes,
over repeated launches (a staging race shows as rare differing tiles)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
os.environ["SPHIP_GEMM"] = "256"
bad_total = 0
for (N, D, M) in [(256, 64, 256), (300, 128, 520), (1000, 192, 700), (512, 4096, 1024), (4096, 1024, 4096),
                  (257, 256, 255), (2048, 4096, 16384), (16384, 4096, 32768)]:
    torch.manual_seed(N + D + M)
    x = torch.randn(N, D, device=dev).bfloat16()
    W = (torch.randn(M, D, device=dev) / D**0.5).bfloat16()
    b = (torch.randn(M, device=dev) * 0.1).bfloat16()
    for relu in (True, False):
        os.environ["SPHIP_GEMM_SCHED"] = "0"
        ref, refmax = ops.encoder_preacts(x, W, b, relu=relu, want_tile_max=True)
        rows = torch.arange(N, device=dev)
        bad = 0
        for rep in range(8 if N * M < 1 << 28 else 3):
            os.environ["SPHIP_GEMM_SCHED"] = "1"
            got, gotmax = ops.encoder_preacts(x, W, b, relu=relu, want_tile_max=True)
            bad += int((got.view(torch.int16) != ref.view(torch.int16)).sum())
            bad += int((gotmax[rows // 64, :, rows % 64].view(torch.int16) != refmax[rows // 64, :, rows % 64].view(torch.int16)).sum())
        print(f"N={N} D={D} M={M} relu={relu}: {bad} differing elements", flush=True)
        bad_total += bad
print("OK" if bad_total == 0 else "MISMATCH", flush=True)
sys.exit(0 if bad_total == 0 else 1)
