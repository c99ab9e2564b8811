"""The allocation-free entry point sphip_total_variance_static (include/sparsify_hip.h): called
through the C ABI and compared, bit for bit, with sphip_total_variance on two calls in a row, the
second on a smaller input (the built-in scratch keeps the first call's partial sums). Collected
last on purpose (file name)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(y):
    from sparsify_amd import _lib, ops
    from sparsify_amd._lib import BF16, F32, check
    lib = _lib.load()
    N, D = y.shape
    dt = BF16 if y.dtype == torch.bfloat16 else F32
    stream = torch.cuda.current_stream(y.device).cuda_stream
    got = torch.full((1,), float("nan"), device=y.device)
    check(lib.sphip_total_variance_static(y.data_ptr(), dt, N, D, got.data_ptr(), stream), "sphip_total_variance_static")
    want = ops.total_variance(y)
    torch.cuda.synchronize()
    return got.cpu(), want.cpu()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_total_variance_static_equals_workspace_form(device, dtype):
    g = torch.Generator(device="cuda").manual_seed(1)
    big = (torch.randn(4096, 4096, device="cuda", generator=g) * 3 + torch.randn(4096, device="cuda", generator=g)).to(dtype)
    small = (torch.randn(37, 200, device="cuda", generator=g) - 2).to(dtype)        # ragged: D % 4 == 0 only
    odd = (torch.randn(5, 51, device="cuda", generator=g)).to(dtype)                # scalar-path columns
    for y in (big, small, odd, big[:100].contiguous()):
        got, want = _both(y)
        assert torch.isfinite(want).all() and torch.equal(got, want), (tuple(y.shape), got, want)
    ref = (small.float() - small.float().mean(0)).pow(2).sum()
    torch.testing.assert_close(_both(small)[0][0], ref.cpu(), rtol=2e-2 if dtype == torch.bfloat16 else 1e-5, atol=0)


def test_total_variance_static_rejects_what_does_not_fit(device):
    from sparsify_amd import _lib
    lib = _lib.load()
    y = torch.zeros(4, 8200, device="cuda")
    out = torch.zeros(1, device="cuda")
    rc = lib.sphip_total_variance_static(y.data_ptr(), _lib.F32, 4, 8200, out.data_ptr(), None)
    assert rc == 2 and b"8192" in lib.sphip_last_error()
    rc = lib.sphip_total_variance_static(y.data_ptr(), _lib.F32, 4, 8192, out.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
