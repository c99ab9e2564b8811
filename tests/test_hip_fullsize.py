"""Size-independent properties at BASELINE.json's headline sizes (N = 16384 tokens, D = 4096,
M = 131072 latents, k = 192), where the CPU oracle cannot run: the kernels that the training
shapes select (256-tile GEMM with block maxima, block-maxima top-k, Infinity-Cache-sliced gather,
wide-sliced and L2-sliced weight gradients, row-resident decoder step) are checked through
identities that hold for any correct implementation."""
import pytest
import torch

pytestmark = pytest.mark.gpu

N, D, M, K = 16384, 4096, 131072, 192


@pytest.fixture(scope="module")
def state():
    from sparsify_amd import ops
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(7)
    x = (torch.randn(N, D, device=dev, generator=g) + torch.randn(D, device=dev, generator=g)).to(torch.bfloat16)
    W = (torch.rand(M, D, device=dev, generator=g) * 2 - 1) / D**0.5
    b = torch.zeros(M, device=dev, dtype=torch.bfloat16)
    Wb = W.to(torch.bfloat16)
    pre, tmax = ops.encoder_preacts(x, Wb, b, want_tile_max=True)
    vals, idx = ops.topk(pre, K, tile_max=tmax)
    return dict(x=x, W=W, Wb=Wb, pre=pre, tmax=tmax, vals=vals, idx=idx, gen=g)


def test_gemm_rows_against_fp64_and_block_maxima(state):
    """Sampled rows of the 256-tile GEMM against an fp64 product of the same bf16 operands; the
    block maxima are exactly the maxima of the stored pre-activations."""
    rows = torch.tensor([0, 1, 255, 256, 8191, 16383], device="cuda")
    ref = torch.relu(state["x"][rows].double() @ state["Wb"].double().T)
    got = state["pre"][rows].double()
    # one bf16 rounding of an fp32-accumulated sum of 4096 products
    torch.testing.assert_close(got, ref, rtol=2**-7, atol=2e-3)
    pm = state["pre"].view(N, M // 128, 128).amax(dim=2)                      # (N, T)
    tm = state["tmax"].permute(0, 2, 1).reshape(N, M // 128)                  # [n//64, t, n%64] -> (N, T)
    assert torch.equal(pm, tm)


def test_topk_is_the_exact_selection(state):
    """Every selected value is its row's entry at the selected index, indices are distinct, and the
    k-th largest selected value is the row's k-th largest entry (torch.kthvalue as the checker)."""
    pre, vals, idx = state["pre"], state["vals"], state["idx"]
    assert torch.equal(pre.gather(1, idx), vals)
    assert int((idx.sort(dim=1).values.diff(dim=1) == 0).sum()) == 0
    kth = pre.float().kthvalue(M - K + 1, dim=1).values
    assert torch.equal(vals.float().min(dim=1).values, kth)
    # nothing outside the selection exceeds the k-th value
    above = (pre.float() > kth[:, None]).sum(dim=1)
    assert int(above.max()) <= K - 1
    # the plain full-row kernel returns the same thing
    from sparsify_amd import ops
    v2, i2 = ops.topk(pre, K)
    assert torch.equal(v2, vals) and torch.equal(i2, idx)


def test_decode_is_linear_and_adjoint_to_the_weight_gradient(state):
    """decode(a1 + a2) = decode(a1) + decode(a2); <g, decode(c)> = sum_e c_e dcoeff_e (the dot of
    the W_dec gradient pass is the adjoint of the decode gather); sum_j gW[j, :] = sum_n (sum_s
    c[n, s]) g[n, :] (checksum over all latents); the L2-sliced kernel writes the same gW."""
    from sparsify_amd import ops
    g, idx, W = state["gen"], state["idx"], state["W"]
    a1 = torch.rand(N, K, device="cuda", generator=g)
    a2 = torch.rand(N, K, device="cuda", generator=g)
    y1, y2, y12 = (ops.gather_rows_weighted(idx, a, W) for a in (a1, a2, a1 + a2))
    torch.testing.assert_close(y12, y1 + y2, rtol=2e-5, atol=2e-5)
    grad = torch.randn(N, D, device="cuda", generator=g)
    seg = ops.build_latent_segments(idx, M)
    gW, _, dc = ops.segmented_weight_grad(seg, a1, grad, W, want_dcoeff=True)
    lhs = (grad.double() * y1.double()).sum()
    rhs = (a1.double() * dc.double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-6 * float(lhs.abs() + rhs.abs()) + 1e-3
    chk = gW.double().sum(dim=0)
    ref = (a1.double().sum(dim=1, keepdim=True) * grad.double()).sum(dim=0)
    # 131072 fp32-rounded rows per column sum
    torch.testing.assert_close(chk, ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
    gW2, _, _ = ops.segmented_weight_grad(seg, a1, grad)                     # no dcoeff: L2-sliced kernel
    assert torch.equal(gW, gW2)
    # segments: a stable permutation of all entries, sorted by latent
    perm = seg.perm.to(torch.int64)
    assert torch.equal(perm.sort().values, torch.arange(N * K, device="cuda"))
    lat = idx.reshape(-1)[perm]
    assert bool((lat.diff() >= 0).all())
    assert bool(((lat.diff() > 0) | (perm.diff() > 0)).all())


def test_fused_encoder_update_at_full_size(state):
    """The fused encoder-gradient kernel (sorted lane-group assignment, 7 waves, whole-line
    parameter stores, b_dec partial sums) at the headline shape: bit for bit gradient + separate
    SignSGD step and bf16 shadow; the column sums of its per-block partials equal gb^T W_old
    (sphip_vec_mat) to fp32 summation order; a second run gives identical bits."""
    from sparsify_amd import ops
    g = state["gen"]
    idx, rows = state["idx"], state["x"]
    gv = torch.randn(N, K, device="cuda", generator=g).to(torch.bfloat16)
    shift = torch.randn(D, device="cuda", generator=g)
    lr = 1e-3
    seg = ops.build_latent_segments(idx, M)
    gW, gb, _ = ops.segmented_weight_grad(seg, gv, rows, want_bias=True, row_shift=shift)
    assert torch.equal(ops.segment_coeff_sums(seg, gv), gb)
    want_bdec = ops.vec_mat(gb, state["W"], -1.0)
    sep = state["W"].clone()
    sh_sep = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    ops.signsgd_step_(sep, gW, lr, sh_sep)
    del gW
    fused = state["W"].clone()
    sh_fused = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    _, part = ops.segmented_weight_grad_signsgd_(seg, gv, rows, fused, lr, sh_fused, row_shift=shift, coeff_sums=gb)
    assert torch.equal(fused, sep) and torch.equal(sh_fused.view(torch.int16), sh_sep.view(torch.int16))
    got_bdec = ops.colsum(part, -1.0)
    torch.testing.assert_close(got_bdec, want_bdec, rtol=1e-4, atol=1e-5 * float(want_bdec.abs().max()) + 1e-6)
    fused.copy_(state["W"])
    _, part2 = ops.segmented_weight_grad_signsgd_(seg, gv, rows, fused, lr, sh_fused, row_shift=shift, coeff_sums=gb)
    assert torch.equal(part, part2) and torch.equal(fused, sep)


def test_decoder_step_and_unit_norm_are_idempotent_where_they_should_be(state):
    from sparsify_amd import ops
    W = state["W"].clone()
    ops.unit_norm_rows_(W)
    torch.testing.assert_close(W.norm(dim=1), torch.ones(M, device="cuda"), rtol=1e-5, atol=0)
    W2 = W.clone()
    ops.unit_norm_rows_(W2)
    torch.testing.assert_close(W2, W, rtol=0, atol=3e-7)                     # idempotent up to 1/(1 + eps)
    G = torch.randn(M, D, device="cuda", generator=state["gen"])
    Gp = G.clone()
    ops.remove_parallel_grad_(Gp, W)
    assert float((Gp * W).sum(dim=1).abs().max()) < 1e-3                     # projected rows are orthogonal
    Gpp = Gp.clone()
    ops.remove_parallel_grad_(Gpp, W)
    torch.testing.assert_close(Gpp, Gp, rtol=0, atol=1e-5)                   # projecting twice changes nothing
    Wa, Wb = W.clone(), W.clone()
    ops.decoder_project_signsgd_(Wa, G, 1e-3, True)
    ops.decoder_project_signsgd_(Wb, Gp, 1e-3, True)                         # same step from the projected gradient
    assert float(((Wa - Wb).abs() > 1e-6).float().mean()) < 1e-3


def test_auxk_dense_products_against_row_gather(state):
    """AuxK decode of k_aux = D/2 = 2048 dead latents per token (10 % of the latents dead): the
    dense matrix-core path (dead_decode.py) and the row-gather path give the same reconstruction
    and the same W_dec / b_dec gradients up to fp32 summation order; rows of W_dec.grad that
    belong to live latents stay exactly zero."""
    from sparsify_amd import ops
    from sparsify_amd.dead_decode import DeadLatentDecode
    from sparsify_amd.utils import SparseDecode
    g = state["gen"]
    dead = torch.rand(M, device="cuda", generator=g) < 0.10
    n_dead = int(dead.sum())
    acts, idx = ops.topk(state["pre"], D // 2, keep_mask=dead)
    assert bool(dead[idx].all())
    slot_of, dead_list = ops.dead_compact(dead, n_dead)
    # the compact selection (recomputed pre-activations of the dead latents) picks the same set
    from sparsify_amd.dead_decode import auxk_select_compact
    c_acts, c_slots = auxk_select_compact(state["x"], state["Wb"], None, dead_list, n_dead, D // 2)
    assert torch.equal(dead_list[c_slots].sort(-1).values, idx.sort(-1).values)
    assert torch.equal(c_acts.float().sort(-1).values, acts.float().sort(-1).values)
    W = state["W"]
    bias = torch.randn(D, device="cuda", generator=g)
    gy = torch.randn(N, D, device="cuda", generator=g)
    res = {}
    for name in ("dense", "gather"):
        Wd, bd = W.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        if name == "dense":
            out = DeadLatentDecode.apply(c_slots, c_acts, Wd, bd, dead_list, None, n_dead)
        else:
            out = SparseDecode.apply(idx, Wd, acts.float(), bd)
        (out * gy).sum().backward()
        res[name] = (out.detach(), Wd.grad, bd.grad)
        del out
    for a, b_, what in zip(res["dense"], res["gather"], ("decode", "W_dec.grad", "b_dec.grad")):
        scale = float(b_.abs().max())
        err = float((a - b_).abs().max())
        assert err <= 2e-5 * scale, f"{what}: max abs difference {err:.3e} at scale {scale:.3e}"
    assert not bool(res["dense"][1][~dead].any())


def test_k_decay_start_k_40960(state):
    """The first steps of a run with k-decay use k = min(M, 10 d_in) = 40960 latents per token
    (sparsify/trainer.py:176-177, 229-235): top-k, decode and both weight gradients at that k on a
    small batch of the headline dictionary. Checked against dense torch expressions of the same
    quantities (a (tokens x M) matrix of the selected activations times the table)."""
    from sparsify_amd import ops
    n, k = 48, 40960
    pre = state["pre"][:n].contiguous()
    vals, idx = ops.topk(pre, k, tile_max=state["tmax"][:1].contiguous())
    # exact selection: the k-th largest value, nothing above it left out, values gathered from the row
    assert torch.equal(pre.gather(1, idx), vals)
    assert int((idx.sort(dim=1).values.diff(dim=1) == 0).sum()) == 0
    kth = pre.float().kthvalue(M - k + 1, dim=1).values
    assert torch.equal(vals.float().min(dim=1).values, kth)
    assert int(((pre.float() > kth[:, None]).sum(dim=1)).max()) <= k - 1
    # ties at the boundary (relu zeros): the lowest indices win
    for r in (0, n - 1):
        row = pre[r].float()
        below = (row == kth[r]).nonzero().flatten()
        need = k - int((row > kth[r]).sum())
        chosen = idx[r][vals[r].float() == kth[r]].sort().values
        assert torch.equal(chosen, below[:need])
    # decode + gradients against the dense form
    W = state["W"]
    acts = vals.float()
    A = torch.zeros(n, M, device="cuda").scatter_(1, idx, acts)
    out = ops.gather_rows_weighted(idx, acts, W)
    ref = A.double() @ W.double()
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-4)
    g = torch.randn(n, D, device="cuda", generator=state["gen"])
    seg = ops.build_latent_segments(idx, M)
    gW, gb, dc = ops.segmented_weight_grad(seg, acts, g, W, want_bias=True, want_dcoeff=True)
    torch.testing.assert_close(gW.double(), A.double().T @ g.double(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gb.double(), A.double().sum(0), rtol=1e-6, atol=1e-6)
    dense_dc = g.double() @ W.double().T                                   # (n, M)
    torch.testing.assert_close(dc.double(), dense_dc.gather(1, idx), rtol=1e-5, atol=1e-5)


def test_reference_encoder_test_at_its_own_size():
    """The reference's own encoder test, line for line, at the size it is written for (tests/test_encode.py:7-57:
    N = 8192, D = 1024, M = 131072, k = 32, fp32, seed 42): fused_encoder against relu(linear) + topk + autograd of
    plain torch on the same device. `assert_close(indices, indices_naive)` there relies on both sides computing the
    very same pre-activations; two fp32 GEMMs with different summation orders differ by ~1e-5 relative, so the index
    sets are compared with the north-star margin rule and the values / gradients on the rows whose sets agree."""
    import torch.nn.functional as F
    from sparsify_amd import fused_encoder
    from test_hip_api import same_topk_sets
    torch.manual_seed(42)
    N, D, M, k = 8192, 1024, 131_072, 32
    x = torch.randn(N, D, requires_grad=True, device="cuda")
    W = torch.randn(M, D, requires_grad=True, device="cuda")
    b = torch.randn(M, requires_grad=True, device="cuda")
    output = F.relu(F.linear(x, W, b))
    values_naive, indices_naive = torch.topk(output, k, dim=1, sorted=False)
    values_naive.sum().backward()
    x_grad_naive, W_grad_naive, b_grad_naive = x.grad.clone(), W.grad.clone(), b.grad.clone()
    x.grad = W.grad = b.grad = None
    values, indices, _ = fused_encoder(x, W, b, k, "topk")
    values.sum().backward()
    torch.cuda.synchronize()
    same_topk_sets(indices.cpu(), indices_naive.cpu(), output.detach().cpu(), k, margin=1e-5)
    si, order = indices.sort(-1)
    sn, order_n = indices_naive.sort(-1)
    same = (si == sn).all(-1)
    assert float(same.float().mean()) > 0.999
    torch.testing.assert_close(values.detach().gather(1, order)[same], values_naive.detach().gather(1, order_n)[same],
                               rtol=1e-5, atol=2e-4)
    torch.testing.assert_close(values.sum(), values_naive.sum(), rtol=1e-6, atol=0)
    torch.testing.assert_close(x.grad[same], x_grad_naive[same], rtol=1e-4, atol=1e-3)
    # W.grad / b.grad sum over all rows: the few boundary rows move single entries of a few latents
    moved = torch.zeros(M, dtype=torch.bool, device="cuda")
    moved[indices[~same].reshape(-1)] = True
    moved[indices_naive[~same].reshape(-1)] = True
    torch.testing.assert_close(W.grad[~moved], W_grad_naive[~moved], rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(b.grad[~moved], b_grad_naive[~moved], rtol=1e-5, atol=1e-5)
