"""Times sphip_encoder_preacts alone at the headline shape (bf16, N=16384, D=4096, M=131072).
Usage: python benchmarks/gemm_bench.py [diag ...]   (diag = SPHIP_GEMM_DIAG values to sweep; non-zero
values need a library built with `make -C sparsify_amd/csrc clean all DIAG=1`; "lib" = the vendor
library's GEMM alone through torch, F.linear without bias or ReLU, "libref" = what the reference
launches, relu(F.linear(x, W, b)), both for comparison only; "s0" / "s1" = SPHIP_GEMM_SCHED 0 / 1 of
the shipped kernel, whose outputs are also compared bit for bit over repeated launches)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import ops  # noqa: E402

N, D, M = 16384, 4096, 131072
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, D, device=dev, generator=g).to(torch.bfloat16)
W = (torch.randn(M, D, device=dev, generator=g) / 64).to(torch.bfloat16)
b = torch.zeros(M, device=dev, dtype=torch.bfloat16)
diags = sys.argv[1:] or ["0"]
res = {dg: [] for dg in diags}


def launch(dg):
    if dg in ("s0", "s1"):
        os.environ["SPHIP_GEMM_DIAG"] = "0"
        os.environ["SPHIP_GEMM_SCHED"] = dg[1]
        return ops.encoder_preacts(x, W, b)
    if dg == "lib":
        return torch.nn.functional.linear(x, W)
    if dg == "libref":
        return torch.relu(torch.nn.functional.linear(x, W, b))
    return ops.encoder_preacts(x, W, b)


for rnd in range(3):                      # interleaved rounds: variants share the device's state
    for dg in diags:
        if not dg.startswith("lib") and dg not in ("s0", "s1"):
            os.environ["SPHIP_GEMM_DIAG"] = dg
        launch(dg)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            launch(dg)
        e1.record()
        torch.cuda.synchronize()
        res[dg].append(e0.elapsed_time(e1) / 5)
for dg, v in res.items():
    v = sorted(v)
    ms = v[len(v) // 2]
    print(f"diag={dg} median {ms:.3f} ms (min {v[0]:.3f})  {2.0 * N * D * M / ms / 1e9:.0f} TFLOP/s", flush=True)
if "s0" in diags and "s1" in diags:
    b.copy_((torch.randn(M, device=dev, generator=g) * 0.1).to(torch.bfloat16))
    ref = launch("s0")
    bad = 0
    for i in range(6):
        got = launch("s1" if i % 2 == 0 else "s0")
        bad += int((got.view(torch.int16) != ref.view(torch.int16)).sum())
    print(f"s0 / s1 bitwise over 6 launches: {bad} differing elements", flush=True)
