"""Per-segment cycle stamps of the 256-tile GEMM (needs a library built with
`make -C sparsify_amd/csrc clean all DIAG=1`; SPHIP_GEMM_DIAG & 32 selects the stamped kernel)."""
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
dbg = torch.zeros(8 * 8 * 10, dtype=torch.int32, device=dev)
os.environ["SPHIP_GEMM_DBG"] = str(dbg.data_ptr())
names = ["L0 work+vmwait", "L0 barrier", "M0 dma+lgkm", "M0 mfma", "M0 barrier",
         "L1 work+vmwait", "L1 barrier", "M1 dma+lgkm", "M1 mfma", "M1 barrier"]
for dg in sys.argv[1:] or ["32"]:
    os.environ["SPHIP_GEMM_DIAG"] = dg
    for _ in range(2):
        ops.encoder_preacts(x, W, b)
    dbg.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.encoder_preacts(x, W, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    t = dbg.view(8, 8, 10).float().cpu() / 64.0 / 5       # cycles per K tile
    tot = float(t.sum(dim=2).mean())
    print(f"diag={dg}: {ms:.2f} ms/launch; cycles per K tile {tot:.0f}; "
          f"implied clock {tot * 64 * 128 / (ms * 1e-3) / 1e9:.2f} GHz (if tiles fill the launch time)")
    for grp, sl in (("group0 (waves 0-3)", slice(0, 4)), ("group1 (waves 4-7)", slice(4, 8))):
        m = t[:, sl].mean(dim=(0, 1))
        print(f"  {grp}: total {m.sum():.0f}")
        for n_, v in zip(names, m.tolist()):
            print(f"     {n_:16s} {v:7.0f}")
