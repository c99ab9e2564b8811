"""Times the optimizer-tail kernels on a (131072, 4096) fp32 matrix."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import ops  # noqa: E402

M, D = 131072, 4096
dev = torch.device("cuda", 0)
W = torch.randn(M, D, device=dev)
G = torch.randn(M, D, device=dev)
sh = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
sg = ops.sign_i8(G)


def t(name, fn, gbytes):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name:34s} {ms:.3f} ms  {gbytes / ms:.0f} GB/s", flush=True)


GiB = 2**30 / 1e6
t("signsgd + shadow", lambda: ops.signsgd_step_(W, G, 1e-9, sh), 7 * 2 * GiB)
t("signsgd", lambda: ops.signsgd_step_(W, G, 1e-9), 6 * 2 * GiB)
t("decoder project+sign+renorm", lambda: ops.decoder_project_signsgd_(W, G, 1e-9, True), 6 * 2 * GiB)
smb = torch.empty(M * D, device=dev)
t("decoder project+sign+renorm+slice-major", lambda: ops.decoder_project_signsgd_(W, G, 1e-9, True, smb), 8 * 2 * GiB)
t("unit_norm", lambda: ops.unit_norm_rows_(W), 4 * 2 * GiB)
t("apply_signs + shadow", lambda: ops.apply_signs_(W, sg, 1e-9, sh), (4 + 1 + 4 + 2) * 0.5 * GiB)
t("decoder apply_signs + renorm", lambda: ops.decoder_apply_signs_(W, sg, 1e-9, True), (4 + 1 + 4) * 0.5 * GiB)
t("sign_i8", lambda: ops.sign_i8(G, sg), (4 + 1) * 0.5 * GiB)
