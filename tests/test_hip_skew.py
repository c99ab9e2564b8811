"""The heavy-tailed workload (trainer.SyntheticActivations(skew="zipf"), coder planted with the
dictionary): latent usage follows a power law instead of the Poisson(N k / M) of Gaussian rows, so the
long-segment path of the weight gradients (wgrad.hip: latents with more than 128 entries are cut into
128-entry chunks) does a sizeable share of the work. Reference behaviour being matched: the sorted,
load-balanced segmented reduction of sparsify/xformers.py:106-136 and the index_add_ scatter of
sparsify/fused_encoder.py:67-84, whatever the distribution of the indices.

  * reduced shape, against the CPU oracle (pinned to the reference): forward, losses and every gradient;
  * the headline shape (N = 16384, D = 4096, M = 131072, k = 192): the decode / weight-gradient adjoint
    identity, the checksum over all latents, the fused encoder step against gradient + separate step,
    and the same bits on a second run.
"""
import pytest
import torch

from oracle import sae_oracle as O
from test_hip_api import run_coder, same_topk_sets

pytestmark = pytest.mark.gpu


def usage(idx, M):
    cnt = torch.bincount(idx.reshape(-1), minlength=M)
    return cnt, float(cnt[cnt > 128].sum() / cnt.sum())


def test_skewed_step_vs_oracle(device):
    """N = 4096 tokens of width 1024 (8 MiB of bf16 rows: the training-size dispatch), 8192 latents,
    k = 32, bf16 autocast. A tenth of the entries and more sit in long segments."""
    from sparsify_amd import SparseCoder, SparseCoderConfig
    from sparsify_amd.trainer import SyntheticActivations
    N, D, M, k = 4096, 1024, 8192, 32
    src = SyntheticActivations(["h"], D, N, 1, device="cuda", skew="zipf", atoms=M, draws=k)
    torch.manual_seed(0)
    sae = SparseCoder(D, SparseCoderConfig(num_latents=M, k=k)).cuda()
    src.plant(sae)
    x = src.batch(0)["h"]
    assert x.dtype == torch.bfloat16
    with torch.no_grad():
        sae.b_dec.copy_(x.float().mean(0))
    P = {"W_enc": sae.encoder.weight.detach().cpu().clone(), "b_enc": sae.encoder.bias.detach().cpu().clone(),
         "W_dec": sae.W_dec.detach().cpu().clone(), "b_dec": sae.b_dec.detach().cpu().clone()}
    out = run_coder(sae, "bf16", x, None, None)
    out.fvu.backward()
    cnt, long_share = usage(out.latent_indices, M)
    assert int(cnt.max()) > 1000 and long_share > 0.1, (int(cnt.max()), long_share)
    forced = {"top": (out.latent_acts.detach().cpu(), out.latent_indices.cpu())}
    cfg = O.OracleConfig(k=k)
    ref, G = O.forward_backward(P, x.cpu(), None, cfg, bf16=True, forced=forced)
    pre_cpu = ref["pre_acts"]
    same_topk_sets(out.latent_indices.cpu(), O.select_latents(pre_cpu, k, "topk")[1], pre_cpu, k, margin=2**-7)
    torch.testing.assert_close(out.sae_out.detach().cpu(), ref["sae_out"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(float(out.fvu.detach()), float(ref["fvu"]), rtol=2e-4, atol=1e-6)
    grads = {"W_enc": sae.encoder.weight.grad, "b_enc": sae.encoder.bias.grad, "W_dec": sae.W_dec.grad, "b_dec": sae.b_dec.grad}
    for pname, gr in grads.items():
        rel = (gr.cpu() - G[pname]).norm() / G[pname].norm().clamp_min(1e-12)
        assert rel < 2e-3, f"{pname}: relative error {rel:.3e}"
    # the rows of the densest latents (thousands of terms each, summed in chunks) on their own
    dense = cnt.topk(8).indices.cpu()
    for pname in ("W_enc", "W_dec"):
        rel = (grads[pname].cpu()[dense] - G[pname][dense]).norm() / G[pname][dense].norm()
        assert rel < 2e-3, f"{pname}, dense rows: relative error {rel:.3e}"


@pytest.fixture(scope="module")
def skewed():
    from sparsify_amd import ops
    from sparsify_amd.trainer import SyntheticActivations
    N, D, M, K = 16384, 4096, 131072, 192
    src = SyntheticActivations(["h"], D, N, 1, device="cuda", skew="zipf", atoms=M, draws=K)
    x = src.batch(0)["h"]
    xc = (x.float() - x.float().mean(0)).to(torch.bfloat16)
    Wb = src.dictionary.to(torch.bfloat16)
    pre, tmax = ops.encoder_preacts(xc, Wb, torch.zeros(M, device="cuda", dtype=torch.bfloat16), want_tile_max=True)
    vals, idx = ops.topk(pre, K, tile_max=tmax)
    del pre, tmax, Wb
    return dict(N=N, D=D, M=M, K=K, x=x, W=src.dictionary, vals=vals, idx=idx,
                gen=torch.Generator(device="cuda").manual_seed(3))


def test_skewed_usage_is_heavy_tailed(skewed):
    cnt, long_share = usage(skewed["idx"], skewed["M"])
    f = cnt.float() / skewed["N"]
    assert int((f > 0.1).sum()) >= 30 and float(f.max()) > 0.3, (int((f > 0.1).sum()), float(f.max()))
    assert int((f < 1e-3).sum()) > skewed["M"] // 2
    assert long_share > 0.15, long_share


def test_skewed_decode_adjoint_and_checksum_at_full_size(skewed):
    """<g, decode(c)> = sum_e c_e dcoeff_e and sum_j gW[j, :] = sum_n (sum_s c[n, s]) g[n, :] with long
    segments of up to ~7000 entries; the no-dot form writes the same gradient; a second run the same bits."""
    from sparsify_amd import ops
    N, D, M, K = (skewed[n] for n in "NDMK")
    g, idx, W = skewed["gen"], skewed["idx"], skewed["W"]
    a = skewed["vals"]                                               # bf16 activations: the 4-byte entry lists
    y = ops.gather_rows_weighted(idx, a, W)
    grad = torch.randn(N, D, device="cuda", generator=g)
    seg = ops.build_latent_segments(idx, M)
    gW, _, dc = ops.segmented_weight_grad(seg, a, grad, W, want_dcoeff=True)
    lhs = (grad.double() * y.double()).sum()
    rhs = (a.double() * dc.double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-6 * float(lhs.abs() + rhs.abs()) + 1e-3
    chk = gW.double().sum(dim=0)
    ref = (a.double().sum(dim=1, keepdim=True) * grad.double()).sum(dim=0)
    torch.testing.assert_close(chk, ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
    # the densest latent's row and dots against fp64
    cnt = torch.bincount(idx.reshape(-1), minlength=M)
    j = int(cnt.argmax())
    hit = (idx == j)
    cj = (a.double() * hit).sum(1)
    row = (cj[:, None] * grad.double()).sum(0)
    torch.testing.assert_close(gW[j].double(), row, rtol=1e-5, atol=1e-5 * float(row.abs().max()))
    want_dc = grad[hit.any(1)].double() @ W[j].double()
    torch.testing.assert_close(dc[hit].double(), want_dc, rtol=1e-5, atol=1e-5 * float(want_dc.abs().max()))
    gW2, _, dc2 = ops.segmented_weight_grad(seg, a, grad, W, want_dcoeff=True)
    assert torch.equal(gW, gW2) and torch.equal(dc, dc2)
    del gW2, dc2
    gW3, _, _ = ops.segmented_weight_grad(seg, a, grad)                      # no dcoeff: the other kernel form
    assert torch.equal(gW, gW3)


def test_skewed_fused_encoder_update_at_full_size(skewed):
    """Fused encoder step = gradient + separate SignSGD step, bit for bit, dense latents included (their
    chunk partials are combined by the same kernel in both paths); identical on a second run."""
    from sparsify_amd import ops
    N, D, M, K = (skewed[n] for n in "NDMK")
    g, idx, rows, W = skewed["gen"], skewed["idx"], skewed["x"], skewed["W"]
    gv = torch.randn(N, K, device="cuda", generator=g).to(torch.bfloat16)
    shift = torch.randn(D, device="cuda", generator=g)
    lr = 1e-3
    seg = ops.build_latent_segments(idx, M)
    gW, gb, _ = ops.segmented_weight_grad(seg, gv, rows, want_bias=True, row_shift=shift)
    assert torch.equal(ops.segment_coeff_sums(seg, gv), gb)
    cnt = torch.bincount(idx.reshape(-1), minlength=M)
    j = int(cnt.argmax())
    cj = (gv.double() * (idx == j)).sum(1)
    torch.testing.assert_close(gb[j].double(), cj.sum(), rtol=1e-5, atol=1e-3)
    row = (cj[:, None] * rows.double()).sum(0) - cj.sum() * shift.double()
    torch.testing.assert_close(gW[j].double(), row, rtol=1e-5, atol=2e-5 * float(row.abs().max()))
    sep = W.clone()
    sh_sep = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    ops.signsgd_step_(sep, gW, lr, sh_sep)
    del gW
    fused = W.clone()
    sh_fused = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    _, part = ops.segmented_weight_grad_signsgd_(seg, gv, rows, fused, lr, sh_fused, row_shift=shift, coeff_sums=gb)
    assert torch.equal(fused, sep) and torch.equal(sh_fused.view(torch.int16), sh_sep.view(torch.int16))
    want_bdec = ops.vec_mat(gb, W, -1.0)
    torch.testing.assert_close(ops.colsum(part, -1.0), want_bdec, rtol=1e-4, atol=1e-5 * float(want_bdec.abs().max()) + 1e-6)
    fused.copy_(W)
    _, part2 = ops.segmented_weight_grad_signsgd_(seg, gv, rows, fused, lr, sh_fused, row_shift=shift, coeff_sums=gb)
    assert torch.equal(part, part2) and torch.equal(fused, sep)
