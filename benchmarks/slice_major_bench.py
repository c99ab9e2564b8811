"""Times sphip_slice_major_f32 on the headline decoder matrix (131072 x 4096 fp32)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import _lib, ops  # noqa: E402

M, D = 131072, 4096
W = torch.randn(M, D, device="cuda")
out = torch.empty(D // 32, M, 32, device="cuda")
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    _lib.check(lib.sphip_slice_major_f32(W.data_ptr(), out.data_ptr(), M, D, st), "slice_major")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    lib.sphip_slice_major_f32(W.data_ptr(), out.data_ptr(), M, D, st)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
ok = torch.equal(out, W.view(M, D // 32, 32).transpose(0, 1))
print(f"slice-major copy: {ms:.3f} ms  {2 * M * D * 4 / ms / 1e6:.0f} GB/s  correct: {ok}", flush=True)
