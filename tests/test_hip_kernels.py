"""GPU parity tests, kernel by kernel: HIP path (through the C ABI) vs the CPU oracle on the
same seeded inputs. Integer/index results must be bit-exact; floating point tolerances are
written next to each assert."""
import numpy as np
import pytest
import torch

from oracle import sae_oracle as O

pytestmark = pytest.mark.gpu


def dev(t):
    return t.cuda() if isinstance(t, torch.Tensor) else t


def sort_by_index(vals, idx):
    order = idx.argsort(dim=-1)
    return vals.gather(-1, order), idx.gather(-1, order)


# ------------------------------------------------------------------ encoder GEMM
@pytest.mark.parametrize("N,D,M", [(64, 48, 384), (128, 64, 512), (100, 72, 200), (257, 256, 1000),
                                    (512, 768, 1536), (300, 4096, 640)])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_encoder_preacts(device, N, D, M, dtype):
    from sparsify_amd import ops
    if dtype == "bf16" and D % 8:
        pytest.skip("bf16 path needs D % 8 == 0")
    torch.manual_seed(N + D + M)
    x = torch.randn(N, D)
    W = torch.randn(M, D) / D**0.5
    b = torch.randn(M) * 0.1
    if dtype == "bf16":
        xb, Wb, bb = x.bfloat16(), W.bfloat16(), b.bfloat16()
        got = ops.encoder_preacts(dev(xb), dev(Wb), dev(bb)).cpu()
        assert got.dtype == torch.bfloat16
        # exact reference of the bf16 operands in fp64, then one rounding to bf16
        exact = (xb.double() @ Wb.double().T + bb.double()).clamp_min(0)
        # fp32 accumulation + a single bf16 rounding: within 1 bf16 ulp (2^-8 relative)
        torch.testing.assert_close(got.double(), exact, rtol=2**-7, atol=1e-3)
        ref = O.pre_activations(x, W, b, True)
        mism = (got != ref).float().mean().item()
        assert mism < 0.02, f"{mism:.4f} of bf16 outputs differ from the CPU bf16 linear"
        raw = ops.encoder_preacts(dev(xb), dev(Wb), None, relu=False).cpu()
        assert (raw < 0).any()
    else:
        got = ops.encoder_preacts(dev(x), dev(W), dev(b)).cpu()
        exact = (x.double() @ W.double().T + b.double()).clamp_min(0)
        # exact-fp32 MFMA chain over D terms
        torch.testing.assert_close(got.double(), exact, rtol=1e-5, atol=2e-6 * D**0.5)


@pytest.mark.parametrize("N,D,M", [(256, 64, 256), (300, 128, 520), (1000, 192, 700), (512, 4096, 1024),
                                    (4096, 1024, 4096), (257, 256, 255)])
@pytest.mark.parametrize("relu", [True, False])
def test_encoder_preacts_256_tile_kernel(device, monkeypatch, N, D, M, relu):
    """The large-shape LDS-DMA kernel (forced with SPHIP_GEMM=256) against the exact product and
    against the 128x128 kernel: both accumulate in fp32 and round once, so they may differ
    by one bf16 ulp on a few outputs only."""
    from sparsify_amd import ops
    torch.manual_seed(N + D + M)
    xb = torch.randn(N, D).bfloat16()
    Wb = (torch.randn(M, D) / D**0.5).bfloat16()
    bb = (torch.randn(M) * 0.1).bfloat16()
    monkeypatch.setenv("SPHIP_GEMM", "256")
    got = ops.encoder_preacts(dev(xb), dev(Wb), dev(bb), relu=relu).cpu()
    monkeypatch.setenv("SPHIP_GEMM", "128")
    base = ops.encoder_preacts(dev(xb), dev(Wb), dev(bb), relu=relu).cpu()
    exact = xb.double() @ Wb.double().T + bb.double()
    if relu:
        exact = exact.clamp_min(0)
    torch.testing.assert_close(got.double(), exact, rtol=2**-7, atol=1e-3)
    assert (got != base).float().mean() < 0.01
    # integer-valued operands: every product and partial sum is exact -> must be bit-identical
    xi = torch.randint(-3, 4, (N, D)).bfloat16()
    Wi = torch.randint(-3, 4, (M, D)).bfloat16()
    monkeypatch.setenv("SPHIP_GEMM", "256")
    goti = ops.encoder_preacts(dev(xi), dev(Wi), None, relu=False).cpu()
    assert torch.equal(goti.float(), (xi.float() @ Wi.float().T).bfloat16().float())


@pytest.mark.parametrize("N,D,M,cuts", [(300, 128, 1024, (0, 256, 1024)), (512, 256, 2048, (0, 512, 1024, 1536, 2048)),
                                          (257, 64, 640, (0, 128, 640))])
@pytest.mark.parametrize("relu", [True, False])
def test_encoder_preacts_by_latent_blocks(device, monkeypatch, N, D, M, cuts, relu):
    """sphip_encoder_preacts_block over disjoint latent blocks fills the same pre and block maxima
    as one sphip_encoder_preacts call, bit for bit (same kernel, same per-tile arithmetic)."""
    from sparsify_amd import ops
    torch.manual_seed(N + M)
    xb = dev(torch.randn(N, D).bfloat16())
    Wb = dev((torch.randn(M, D) / D**0.5).bfloat16())
    bb = dev((torch.randn(M) * 0.1).bfloat16())
    monkeypatch.setenv("SPHIP_GEMM", "256")
    whole, whole_max = ops.encoder_preacts(xb, Wb, bb, relu=relu, want_tile_max=True)
    pre = torch.full_like(whole, float("nan"))
    tmax = torch.full_like(whole_max, float("nan"))
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        ops.encoder_preacts_block(xb, Wb, bb, pre, tmax, lo, hi, relu=relu)
    assert torch.equal(pre.view(torch.int16), whole.view(torch.int16))
    rows = torch.arange(N, device=pre.device)
    got = tmax[rows // 64, :, rows % 64].view(torch.int16)          # rows past N are undefined
    assert torch.equal(got, whole_max[rows // 64, :, rows % 64].view(torch.int16))
    with pytest.raises(ValueError):
        ops.encoder_preacts_block(xb, Wb, bb, pre, tmax, 64, 256)    # not on a 128-latent boundary
    assert ops.encoder_preacts_block(xb, Wb, None, pre, None, 0, 128, relu=relu) is None


# ------------------------------------------------------------------ top-k
def check_topk(pre_cpu, k, vals, idx, keep=None):
    """Set semantics of torch.topk: selected values are the k largest, indices are valid,
    distinct and point at the returned values; ties may differ in index only."""
    N, M = pre_cpu.shape
    vals, idx = vals.cpu(), idx.cpu()
    assert idx.dtype == torch.int64 and idx.shape == (N, k)
    assert int(idx.min()) >= 0 and int(idx.max()) < M
    assert (idx.sort(-1).values.diff(dim=-1) > 0).all(), "duplicate index in a row"
    masked = pre_cpu.float()
    if keep is not None:
        masked = torch.where(keep[None], masked, torch.tensor(-torch.inf))
    torch.testing.assert_close(vals.float(), masked.gather(1, idx), rtol=0, atol=0)
    ref_vals = masked.topk(k, dim=-1).values
    torch.testing.assert_close(vals.float().sort(-1, descending=True).values, ref_vals, rtol=0, atol=0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,M,k", [(7, 384, 10), (5, 512, 16), (3, 1000, 32), (4, 4096, 32), (2, 24576, 32),
                                    (3, 131072, 192), (2, 2051, 7), (2, 6000, 1500), (2, 300, 300), (1, 262144, 768),
                                    (2, 131072, 2048), (2, 40000, 4096), (1, 20000, 5000), (2, 262144, 1500)])
def test_topk_random(device, dtype, N, M, k):
    from sparsify_amd import ops
    torch.manual_seed(M + k)
    pre = torch.randn(N, M).clamp_min(0).to(dtype)
    vals, idx = ops.topk(dev(pre), k)
    check_topk(pre, k, vals, idx)
    if dtype == torch.float32:
        # distinct values almost surely among the positives: index sets must match torch exactly
        ref = pre.topk(k, dim=-1)
        pos = ref.values.min(-1).values > 0
        got_sorted = idx.cpu().sort(-1).values
        assert torch.equal(got_sorted[pos], ref.indices.sort(-1).values[pos])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_topk_ties_and_degenerate_rows(device, dtype):
    from sparsify_amd import ops
    M, k = 8192, 64
    rows = [
        torch.zeros(M),                                            # all equal
        torch.cat([torch.arange(10.0) + 1, torch.zeros(M - 10)]),  # fewer than k positive
        torch.arange(M, dtype=torch.float32),                      # ascending (worst case for the bound)
        torch.arange(M, 0, -1, dtype=torch.float32),               # descending
        torch.full((M,), 3.0),                                     # constant non-zero
        torch.randint(0, 4, (M,)).float(),                         # heavy ties
        -torch.rand(M),                                            # all negative (raw affine output)
    ]
    pre = torch.stack(rows).to(dtype)
    vals, idx = ops.topk(dev(pre), k)
    check_topk(pre, k, vals, idx)
    # deterministic tie rule: lowest indices win
    assert torch.equal(idx[0].cpu().sort().values, torch.arange(k))
    # repeatability (bitwise)
    v2, i2 = ops.topk(dev(pre), k)
    assert torch.equal(v2, vals) and torch.equal(i2, idx)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,k,mult", [(8192, 64, 4), (1000, 17, 4), (5000, 1100, 4), (4096, 300, 2), (640, 160, 4), (64, 7, 1)])
def test_exclude_selected_splits_a_wider_topk(device, dtype, M, k, mult):
    """sphip_exclude_selected (multi-top-k, sparse_coder.py:256-262): the top-k set is a subset of the top-(mult k)
    set of the same row by the selection rule (ties to the lowest index); exactly its members are marked -1,
    whatever the order either kernel path emitted (value order, or index order on the exact path), ties included."""
    from sparsify_amd import ops
    g = torch.Generator().manual_seed(M + k)
    rows = [torch.randn(9, M, generator=g), torch.randn(9, M, generator=g).clamp_min(0),
            torch.randint(0, 4, (9, M), generator=g).float(),                        # ties at the boundary of both
            torch.randn(9, M, generator=g).clamp_min(0) * (torch.rand(9, M, generator=g) < 0.02),   # mostly zeros
            torch.zeros(2, M), torch.full((2, M), 2.5)]
    pre = dev(torch.cat(rows).to(dtype))
    kw = mult * k
    v, i = ops.topk(pre, k)
    vw, iw = ops.topk(pre, kw)
    rest = ops.exclude_selected(vw, iw, v, i).cpu()
    i, iw = i.cpu(), iw.cpu()
    marked = rest == -1
    assert torch.equal(marked.sum(1), torch.full((pre.shape[0],), k))
    assert torch.equal(rest[~marked], iw[~marked])
    for n in range(pre.shape[0]):
        assert set(iw[n][marked[n]].tolist()) == set(i[n].tolist()), n
    # an inactive slot of the wider list stays as it is, one of the narrower list selects nothing
    iw2 = iw.clone()
    iw2[:, -1] = -1
    i2 = i.clone()
    i2[:, 0] = -1
    rest2 = ops.exclude_selected(vw, dev(iw2), v, dev(i2)).cpu()
    assert bool((rest2[:, -1] == -1).all())
    assert bool(((rest2 == -1).sum(1) <= k + 1).all())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,k", [(13120, 2048), (16384, 1024), (8192, 512), (4096, 2048), (2056, 300), (8, 8)])
def test_topk_register_resident_exact_path(device, monkeypatch, dtype, M, k):
    """Short rows with a large k (the compact AuxK selection) take topk_resident_kernel when the row
    fits the workgroup's registers: values and indices position for position those of the sweeping
    exact path of topk_kernel (SPHIP_TOPK_NO_RESIDENT=1), random rows and rows full of ties."""
    from sparsify_amd import ops
    torch.manual_seed(M + k)
    rows = [torch.randn(M).clamp_min(0) for _ in range(5)] + [
        torch.zeros(M), torch.full((M,), 3.0), torch.randint(0, 4, (M,)).float(), -torch.rand(M),
        torch.arange(M, dtype=torch.float32) / 64, torch.randn(M).clamp_min(0) * (torch.rand(M) < 0.01)]
    pre = torch.stack(rows).to(dtype)
    vals, idx = ops.topk(dev(pre), k)
    check_topk(pre, k, vals, idx)
    monkeypatch.setenv("SPHIP_TOPK_NO_RESIDENT", "1")
    v0, i0 = ops.topk(dev(pre), k)
    assert torch.equal(i0, idx) and torch.equal(v0.view(torch.int16 if dtype == torch.bfloat16 else torch.int32),
                                                vals.view(torch.int16 if dtype == torch.bfloat16 else torch.int32))


def _block_maxima(pre, tile=128):
    """tile_max in the layout sphip_encoder_preacts writes: [n // 64, t, n % 64]."""
    N, M = pre.shape
    T = -(-M // tile)
    padded = torch.full((-(-N // 64) * 64, T * tile), float("-inf"), dtype=torch.float32)
    padded[:N, :M] = pre.float()
    tm = padded.reshape(-1, 64, T, tile).amax(-1)              # [N/64, 64, T]
    return tm.permute(0, 2, 1).contiguous().to(pre.dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("M,k", [(8192, 64), (32768, 192), (4096, 32)])
def test_topk_block_maxima_path_with_ties(device, monkeypatch, dtype, wide, M, k):
    """Degenerate rows through the block-maxima path, four waves per row (k <= 256) and 1024
    threads per row: ties overflow the
    candidate list and must fall through to the exact path with the lowest-index rule."""
    from sparsify_amd import ops
    torch.manual_seed(M + k)
    rows = [
        torch.zeros(M),
        torch.cat([torch.arange(10.0) + 1, torch.zeros(M - 10)]),
        torch.arange(M, dtype=torch.float32) / 64,
        torch.full((M,), 3.0),
        torch.randint(0, 4, (M,)).float(),
        torch.randn(M).clamp_min(0),
        torch.randn(M).clamp_min(0) * (torch.rand(M) < 0.001),      # fewer positives than k
    ]
    pre = torch.stack(rows).to(dtype)
    if wide:
        monkeypatch.setenv("SPHIP_TOPK_WIDE", "1")
    vals, idx = ops.topk(dev(pre), k, tile_max=dev(_block_maxima(pre)))
    check_topk(pre, k, vals, idx)
    assert torch.equal(idx[0].cpu().sort().values, torch.arange(k))
    v0, i0 = ops.topk(dev(pre), k)                                   # full-row sweep: same selection
    assert torch.equal(i0.sort(-1).values, idx.sort(-1).values)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,k,frac", [(4096, 24, 0.33), (4096, 2048, 0.33), (24576, 384, 0.1), (512, 100, 0.1),
                                      (131072, 2048, 0.1), (131072, 2048, 0.01)])
def test_topk_masked_auxk(device, dtype, M, k, frac):
    """where(dead_mask, pre, -inf).topk(k_aux) -- sparse_coder.py:240-243."""
    from sparsify_amd import ops
    torch.manual_seed(M + k)
    pre = torch.randn(6, M).clamp_min(0).to(dtype)
    keep = torch.rand(M) < frac
    vals, idx = ops.topk(dev(pre), k, keep_mask=dev(keep))
    check_topk(pre, k, vals, idx, keep)
    if int(keep.sum()) >= k:
        assert keep[idx.cpu()].all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_groupmax(device, dtype):
    from sparsify_amd import ops
    torch.manual_seed(3)
    for N, M, k in [(9, 384, 12), (4, 512, 16), (3, 4096, 32), (2, 24576, 192)]:
        pre = torch.randn(N, M).clamp_min(0).to(dtype)
        vals, idx = ops.groupmax(dev(pre), k)
        rv, ri = O.select_latents(pre.float(), k, "groupmax")
        assert torch.equal(idx.cpu(), ri)
        torch.testing.assert_close(vals.cpu().float(), rv, rtol=0, atol=0)


# ------------------------------------------------------------------ gathers
@pytest.mark.parametrize("N,k,M,D", [(2, 10, 128, 48), (2, 10, 128, 64), (96, 24, 640, 200), (33, 7, 100, 50),
                                      (64, 192, 4096, 4096), (5, 3, 17, 1028)])
@pytest.mark.parametrize("cdtype", [torch.float32, torch.bfloat16])
def test_gather_rows_weighted(device, N, k, M, D, cdtype):
    from sparsify_amd import ops
    torch.manual_seed(D)
    table = torch.randn(M, D)
    idx = torch.stack([torch.randperm(M)[:k] for _ in range(N)])
    coeff = torch.randn(N, k).to(cdtype)
    bias = torch.randn(D)
    got = ops.gather_rows_weighted(dev(idx), dev(coeff), dev(table), dev(bias)).cpu()
    ref = O.gather_sum(idx, coeff.float(), table) + bias
    # fp32 fma chain of k terms vs the oracle's mul+add chain
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5 * k**0.5)
    got2 = ops.gather_rows_weighted(dev(idx), dev(coeff), dev(table), None).cpu()
    torch.testing.assert_close(got2, ref - bias, rtol=1e-5, atol=1e-5 * k**0.5)


@pytest.mark.parametrize("N,D,M,k", [(300, 64, 1024, 32), (256, 128, 24576, 32), (64, 64, 131072, 192),
                                      (40, 64, 1000, 17), (33, 64, 4104, 1024), (16, 64, 262144, 64)])
@pytest.mark.parametrize("gemm", ["128", "256"])
@pytest.mark.parametrize("relu", [True, False])
def test_topk_with_block_maxima(device, monkeypatch, N, D, M, k, gemm, relu):
    """Block maxima from the GEMM (epilogue of the 256 kernel / side kernel otherwise) are exact,
    and top-k through them returns bit for bit what the full-row kernel returns -- including
    rows of ties (relu zeros), partial last blocks and k up to the thread count."""
    from sparsify_amd import ops
    monkeypatch.setenv("SPHIP_GEMM", gemm)
    torch.manual_seed(M + k)
    x = torch.randn(N, D).bfloat16()
    W = (torch.randn(M, D) / D**0.5).bfloat16()
    b = (torch.randn(M) * 0.1 - (1.0 if relu else 0.0)).bfloat16()     # relu: most entries are 0 (ties)
    pre, tmax = ops.encoder_preacts(dev(x), dev(W), dev(b), relu=relu, want_tile_max=True)
    T = -(-M // 128)
    padded = torch.full((N, T * 128), float("-inf"))
    padded[:, :M] = pre.float().cpu()
    NB = -(-N // 64)
    got_tmax = tmax.float().cpu().permute(0, 2, 1).reshape(NB * 64, T)[:N]      # [n // 64, t, n % 64] -> [n, t]
    assert torch.equal(got_tmax, padded.view(N, T, 128).amax(dim=2))
    v0, i0 = ops.topk(pre, k)
    v1, i1 = ops.topk(pre, k, tile_max=tmax)
    assert torch.equal(i0, i1) and torch.equal(v0, v1)
    # fp32 rows with block maxima (side kernel)
    pre32 = pre.float() + dev(torch.randn(N, M)) * 1e-3
    tm32 = torch.nn.functional.pad(pre32, (0, T * 128 - M), value=float("-inf")).view(N, T, 128).amax(dim=2)
    tm32 = torch.nn.functional.pad(tm32, (0, 0, 0, NB * 64 - N)).view(NB, 64, T).permute(0, 2, 1).contiguous()
    v2, i2 = ops.topk(pre32, k)
    v3, i3 = ops.topk(pre32, k, tile_max=tm32)
    assert torch.equal(i2, i3) and torch.equal(v2, v3)


@pytest.mark.parametrize("N,k,M,D,cdtype,bias", [(2048, 192, 49152, 4096, torch.bfloat16, True),
                                                 (2077, 200, 40000, 4096, torch.float32, False)])
def test_gather_l2_resident_form(device, monkeypatch, N, k, M, D, cdtype, bias):
    """The L2-resident decode (gather_l2.hip: latent blocks, bucketed slot lists) against the
    token-major kernel on the same inputs; sums differ in order only. Ragged N, a last latent block
    that is not full, inactive (negative) slots."""
    from sparsify_amd import _lib, ops
    assert _lib.load().sphip_gather_l2_workspace_bytes(N, k, M, D) > 0
    assert _lib.load().sphip_gather_l2_workspace_bytes(N, 8, M, D) == 0          # too few slots per block
    assert _lib.load().sphip_gather_l2_workspace_bytes(N, k, 4096, D) == 0       # table fits the caches
    g = torch.Generator(device="cuda").manual_seed(N + k)
    table = torch.randn(M, D, device="cuda", generator=g)
    idx = torch.randint(0, M, (N, k), device="cuda", generator=g)
    idx[::7, ::5] = -1                                                            # slots of another shard
    idx[3] = torch.arange(k, device="cuda") + (M - k)                            # one block only
    idx[5] = -1                                                                   # a token without slots
    coeff = torch.randn(N, k, device="cuda", generator=g).to(cdtype)
    b = torch.randn(D, device="cuda", generator=g) if bias else None
    monkeypatch.setenv("SPHIP_GATHER", "l2")
    out = ops.gather_rows_weighted(idx, coeff, table, b)                          # row-major table
    again = ops.gather_rows_weighted(idx, coeff, table, b)
    assert torch.equal(out, again)                                                # deterministic
    via_copy = ops.gather_rows_weighted(idx, coeff, table, b, table_owner=table)  # slice-major copy
    assert torch.equal(out, via_copy)                                             # same sums, same order
    sm = ops.slice_major_copy(table)
    assert sm.shape == (D // 32, M, 32) and torch.equal(sm, table.view(M, D // 32, 32).transpose(0, 1))
    table[11] += 1.0                                                              # version bump -> refreshed copy
    assert torch.equal(ops.slice_major_copy(table)[:, 11], table[11].view(D // 32, 32))
    table[11] -= 1.0
    monkeypatch.setenv("SPHIP_GATHER", "sliced")
    ref = ops.gather_rows_weighted(idx, coeff, table, b)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)
    if bias:
        assert torch.equal(out[5], b)
    else:
        assert not out[5].any()


@pytest.mark.parametrize("N,k,M,D", [(7, 13, 300, 256), (130, 192, 4096, 1024), (33, 8, 64, 4096)])
@pytest.mark.parametrize("cdtype", [torch.float32, torch.bfloat16])
def test_gather_sliced_kernel_matches_full(device, monkeypatch, N, k, M, D, cdtype):
    """The Infinity-Cache-sliced kernel (picked for the training shapes) against the oracle and,
    bit for bit, against the token-per-workgroup kernel; negative indices are inactive slots."""
    from sparsify_amd import ops
    torch.manual_seed(N)
    table = torch.randn(M, D)
    idx = torch.randint(0, M, (N, k))
    idx[::3, ::4] = -1
    coeff = torch.randn(N, k).to(cdtype)
    bias = torch.randn(D)
    outs = {}
    for mode in ("full", "sliced"):
        monkeypatch.setenv("SPHIP_GATHER", mode)
        outs[mode] = ops.gather_rows_weighted(dev(idx), dev(coeff), dev(table), dev(bias)).cpu()
    assert torch.equal(outs["full"], outs["sliced"])
    live = (idx >= 0)
    ref = O.gather_sum(idx.clamp(min=0), coeff.float() * live, table) + bias
    torch.testing.assert_close(outs["sliced"], ref, rtol=1e-5, atol=1e-5 * k**0.5)


# ------------------------------------------------------------------ segments
@pytest.mark.parametrize("N,k,M", [(2, 10, 128), (96, 24, 640), (1000, 32, 24576), (300, 192, 131072),
                                    (17, 5, 7), (64, 16, 1 << 20), (4096, 32, 2049)])
def test_latent_segments_bit_exact(device, N, k, M):
    from sparsify_amd import ops
    torch.manual_seed(N + k)
    if k <= M:
        idx = torch.stack([torch.randperm(M)[:k] for _ in range(N)]) if M < 5000 else torch.randint(0, M, (N, k))
    else:
        idx = torch.randint(0, M, (N, k))
    if M > 100:
        idx[:, 0] = 5          # one very long segment
    seg = ops.build_latent_segments(dev(idx), M)
    flat = idx.reshape(-1).numpy()
    ref_perm = np.argsort(flat, kind="stable").astype(np.int64)
    counts = np.bincount(flat, minlength=M)
    ref_begin = np.concatenate([[0], np.cumsum(counts)])
    perm = seg.perm.cpu().numpy().view(np.uint32).astype(np.int64)
    assert np.array_equal(perm, ref_perm)
    assert np.array_equal(seg.seg_begin.cpu().numpy().astype(np.int64), ref_begin)


@pytest.mark.parametrize("N,k,M,D", [(2, 10, 128, 48), (96, 24, 640, 200), (256, 32, 2048, 1024),
                                      (128, 16, 512, 2048), (64, 48, 1024, 4096), (40, 8, 64, 8200)])
@pytest.mark.parametrize("rdtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cdtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("variant", ["full", "sliced", "wide"])
def test_segmented_weight_grad(device, monkeypatch, N, k, M, D, rdtype, cdtype, variant):
    from sparsify_amd import ops
    monkeypatch.setenv("SPHIP_WGRAD", variant)     # kernel used when no dcoeff is requested
    torch.manual_seed(N + D)
    idx = torch.stack([torch.randperm(M)[:k] for _ in range(N)])
    for n in range(N // 2):                                # a long segment, rows stay distinct
        if 3 not in idx[n]:
            idx[n, 0] = 3
    coeff = torch.randn(N, k).to(cdtype)
    rows = torch.randn(N, D).to(rdtype)
    W = torch.randn(M, D)
    seg = ops.build_latent_segments(dev(idx), M)
    gW, gb, dc = ops.segmented_weight_grad(seg, dev(coeff), dev(rows), dev(W), want_bias=True, want_dcoeff=True)
    ref_gW = O.scatter_rows(idx, coeff.float(), rows.float(), M)
    ref_gb = O.scatter_scalar(idx, coeff.float(), M)
    ref_dc = torch.einsum("nd,nsd->ns", rows.float(), W[idx])
    # fp32 sums of <= N/2 terms (segment) / D terms (dot), different order than the oracle
    torch.testing.assert_close(gW.cpu(), ref_gW, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gb.cpu(), ref_gb, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dc.cpu(), ref_dc, rtol=1e-4, atol=1e-4 * D**0.5)
    # rows given un-shifted + row_shift == shifted rows
    shift = torch.randn(D)
    gWs, gbs, _ = ops.segmented_weight_grad(seg, dev(coeff), dev(rows), want_bias=True, row_shift=dev(shift))
    ref_s = O.scatter_rows(idx, coeff.float(), rows.float() - shift, M)
    torch.testing.assert_close(gWs.cpu(), ref_s, rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(gbs.cpu(), ref_gb, rtol=1e-4, atol=1e-4)
    # the sliced and the full-row kernel run the same sequential fma chains: bitwise equal
    gWn, gbn, _ = ops.segmented_weight_grad(seg, dev(coeff), dev(rows), want_bias=True)
    assert torch.equal(gWn, gW) and torch.equal(gbn, gb)
    # bitwise reproducible
    gW2, gb2, dc2 = ops.segmented_weight_grad(seg, dev(coeff), dev(rows), dev(W), want_bias=True, want_dcoeff=True)
    assert torch.equal(gW, gW2) and torch.equal(gb, gb2) and torch.equal(dc, dc2)
    # every row written: empty segments are exact zeros
    empty = torch.ones(M, dtype=torch.bool)
    empty[idx.reshape(-1)] = False
    assert (gW.cpu()[empty] == 0).all()


# ------------------------------------------------------------------ loss pieces
@pytest.mark.parametrize("N,D", [(64, 48), (128, 64), (1000, 200), (513, 4096), (37, 50)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_loss_reductions(device, N, D, dtype):
    from sparsify_amd import ops
    torch.manual_seed(N)
    y = (torch.randn(N, D) + torch.randn(D)).to(dtype)
    p = torch.randn(N, D)
    tv = ops.total_variance(dev(y)).cpu()
    yd = y.double()
    # bf16 inputs: mean and difference are rounded to bf16 like the reference's autocast does
    torch.testing.assert_close(tv.double()[0], O.total_variance(y).double(), rtol=5e-5, atol=0)
    if dtype == torch.float32:
        torch.testing.assert_close(tv.double()[0], (yd - yd.mean(0)).pow(2).sum(), rtol=2e-5, atol=0)
    e, sq = ops.residual_sumsq(dev(y), dev(p))
    torch.testing.assert_close(e.cpu(), y.float() - p, rtol=0, atol=0)
    torch.testing.assert_close(sq.cpu().double()[0], (yd - p.double()).pow(2).sum(), rtol=2e-5, atol=0)
    cs = ops.colsum(dev(y), 1.0 / N).cpu()
    torch.testing.assert_close(cs.double(), yd.mean(0), rtol=1e-5, atol=1e-6)
    num = torch.tensor([3.0]); den = torch.tensor([7.0])
    sc = ops.scale_by_ratio(e, -2.0, dev(num), dev(den)).cpu()
    torch.testing.assert_close(sc, e.cpu() * (-2.0 * 3.0 / 7.0), rtol=1e-6, atol=0)
    v = torch.randn(N)
    if dtype == torch.float32:
        vm = ops.vec_mat(dev(v), dev(y), -1.0).cpu()
        torch.testing.assert_close(vm.double(), -(v.double() @ yd), rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------ dense updates (golden: real reference)
def test_dense_updates_vs_reference_golden(device):
    from conftest import load_golden
    from sparsify_amd import ops
    g = load_golden("dense_ops")
    W = dev(g["W0"].clone())
    ops.unit_norm_rows_(W)
    torch.testing.assert_close(W.cpu(), g["W1"], rtol=1e-6, atol=1e-7)
    G = dev(g["G0"].clone())
    ops.remove_parallel_grad_(G, dev(g["W1"]))
    torch.testing.assert_close(G.cpu(), g["G1"], rtol=1e-5, atol=1e-6)
    p = dev(g["p0"].clone())
    shadow = torch.empty_like(p, dtype=torch.bfloat16)
    for t in range(10):
        ops.signsgd_step_(p, dev(g["grads"][t].contiguous()), g["lr"], shadow)
        if t == 0:
            torch.testing.assert_close(p.cpu(), g["p1"], rtol=0, atol=1e-7)
    torch.testing.assert_close(p.cpu(), g["p10"], rtol=0, atol=1e-6)
    assert torch.equal(shadow.cpu(), p.cpu().bfloat16())
    # fused projected-gradient + sign step == the two reference steps
    W2 = dev(g["W1"].clone())
    ops.decoder_project_signsgd_(W2, dev(g["G0"]), 1e-3)
    ref = O.signsgd_update(g["W1"], g["G1"], 1e-3)
    # sign() of a projected component that is ~0 may flip; everything else is exact
    bad = (W2.cpu() - ref).abs() > 1e-7
    assert bad.float().mean() < 1e-3
    assert (g["G1"][bad].abs() < 1e-5).all()


@pytest.mark.parametrize("M,D", [(64, 48), (100, 1024), (37, 1028), (50, 4096), (9, 8192), (5, 8196), (3, 16384)])
def test_decoder_row_kernels_fused_vs_separate(device, M, D):
    """unit-norm / projected sign step / both fused: the row-resident kernels (D <= 8192) and the
    two-sweep fallbacks must agree bit for bit with the separate reference steps."""
    from sparsify_amd import ops
    torch.manual_seed(M + D)
    W0 = torch.randn(M, D)
    G = torch.randn(M, D)
    W1 = dev(W0.clone())
    ops.unit_norm_rows_(W1)
    torch.testing.assert_close(W1.cpu(), O.unit_norm_rows(W0), rtol=2e-6, atol=1e-7)
    # separate: project+sign, then normalise
    Wa = W1.clone()
    ops.decoder_project_signsgd_(Wa, dev(G), 3e-3)
    ref = O.signsgd_update(W1.cpu(), O.remove_parallel_component(G, W1.cpu()), 3e-3)
    bad = (Wa.cpu() - ref).abs() > 1e-7          # sign() of a ~0 projected component may flip
    assert bad.float().mean() < 2e-3
    ops.unit_norm_rows_(Wa)
    # fused
    Wb = W1.clone()
    ops.decoder_project_signsgd_(Wb, dev(G), 3e-3, renorm=True)
    assert torch.equal(Wa, Wb)
    torch.testing.assert_close(Wb.cpu().norm(dim=1), torch.ones(M), rtol=1e-5, atol=0)


@pytest.mark.parametrize("M,D", [(64, 48), (100, 1024), (37, 1028), (50, 4096), (5, 8196), (3, 50)])
def test_sign_exchange_kernels(device, M, D):
    """sign -> int8 -> apply is bit for bit the fused SignSGD / decoder step (the data-parallel
    sign exchange must produce the single-GPU update)."""
    from sparsify_amd import ops
    torch.manual_seed(M * D)
    W0 = O.unit_norm_rows(torch.randn(M, D))
    G = torch.randn(M, D)
    G[0, : D // 2] = 0.0                                        # exact zeros keep sign 0
    sg = ops.sign_i8(dev(G))
    assert torch.equal(sg.cpu(), torch.sign(G).to(torch.int8))
    # plain parameter (+ bf16 shadow)
    pa, pb = dev(W0.clone()), dev(W0.clone())
    sha, shb = torch.empty_like(pa, dtype=torch.bfloat16), torch.empty_like(pb, dtype=torch.bfloat16)
    ops.signsgd_step_(pa, dev(G), 3e-3, sha)
    ops.apply_signs_(pb, sg, 3e-3, shb)
    assert torch.equal(pa, pb) and torch.equal(sha, shb)
    # decoder rows: projected sign by row blocks, then apply (+ renorm)
    half = M // 2
    Wd = dev(W0)
    ps = torch.cat([ops.decoder_project_sign_i8(Wd[:half].contiguous(), dev(G[:half].contiguous())),
                    ops.decoder_project_sign_i8(Wd[half:].contiguous(), dev(G[half:].contiguous()))])
    for renorm in (False, True):
        wa, wb = dev(W0.clone()), dev(W0.clone())
        ops.decoder_project_signsgd_(wa, dev(G), 3e-3, renorm)
        ops.decoder_apply_signs_(wb, ps, 3e-3, renorm)
        assert torch.equal(wa, wb)


def test_dead_latent_bookkeeping(device):
    from sparsify_amd import ops
    M = 1000
    torch.manual_seed(0)
    counts = torch.randint(0, 50, (M,), dtype=torch.int64)
    fired = torch.zeros(M, dtype=torch.bool)
    idx = torch.randint(0, M, (64, 8))
    d_counts, d_fired = dev(counts.clone()), dev(fired.clone())
    ops.mark_fired_(d_fired, dev(idx))
    ref_fired = fired.clone(); ref_fired[idx.flatten()] = True           # trainer.py:422
    assert torch.equal(d_fired.cpu(), ref_fired)
    ops.dead_counter_step_(d_counts, d_fired, 512)
    ref_counts = counts + 512; ref_counts[ref_fired] = 0                 # trainer.py:531-533
    assert torch.equal(d_counts.cpu(), ref_counts)
    assert not d_fired.any()                                             # trainer.py:537-538
    mask, n_dead = ops.dead_mask(d_counts, 530)
    assert torch.equal(mask.cpu(), ref_counts > 530)                     # trainer.py:415
    assert int(n_dead.item()) == int((ref_counts > 530).sum())


def test_center_cast(device):
    from sparsify_amd import ops
    torch.manual_seed(1)
    for N, D in [(5, 48), (33, 50), (64, 4096)]:
        for dt in (torch.float32, torch.bfloat16):
            x = torch.randn(N, D).to(dt)
            b = torch.randn(D)
            f, h = ops.center_cast(dev(x), dev(b), True, True)
            ref = x - b                       # promotes to fp32 like sparse_coder.py:192
            assert torch.equal(f.cpu(), ref)
            assert torch.equal(h.cpu(), ref.bfloat16())
    w = torch.randn(1000, 3)
    assert torch.equal(ops.cast_bf16(dev(w)).cpu(), w.bfloat16())


# ------------------------------------------------------------------ AuxK as dense products
@pytest.mark.parametrize("N,K,M", [(256, 64, 256), (300, 192, 520), (1000, 1024, 48), (77, 128, 4096)])
def test_gemm_bf16_f32acc(device, N, K, M):
    """fp32-output form of the 256x256 kernel: exact on integer operands, accumulate adds."""
    from sparsify_amd import ops
    torch.manual_seed(N + K + M)
    xi = torch.randint(-4, 5, (N, K)).bfloat16()
    Wi = torch.randint(-4, 5, (M, K)).bfloat16()
    out = torch.full((N, M), float("nan"), device="cuda")
    ops.gemm_bf16_f32acc(dev(xi), dev(Wi), out, accumulate=False)
    exact = xi.float() @ Wi.float().T
    assert torch.equal(out.cpu(), exact)
    ops.gemm_bf16_f32acc(dev(xi), dev(Wi), out, accumulate=True)
    assert torch.equal(out.cpu(), 2 * exact)
    x = torch.randn(N, K).bfloat16()
    W = torch.randn(M, K).bfloat16()
    got = ops.gemm_bf16_f32acc(dev(x), dev(W), torch.empty(N, M, device="cuda"), accumulate=False).cpu()
    torch.testing.assert_close(got.double(), x.double() @ W.double().T, rtol=1e-5, atol=2e-6 * K**0.5 * 4)
    with pytest.raises(RuntimeError):
        ops.gemm_bf16_f32acc(dev(x[:, :K - 8].contiguous()), dev(W[:, :K - 8].contiguous()), out, accumulate=False)


def test_split_bf16_transposed_is_exact(device):
    from sparsify_amd import ops
    torch.manual_seed(0)
    R, C = 333, 200
    src = torch.randn(R, C) * torch.logspace(-6, 6, C)
    src[5, 7] = 0.0
    rows = torch.randperm(R)[:150]
    for lst, n_rows, ldr in ((None, R, 384), (rows, 150, 192), (None, R, 333), (rows, 150, 150), (rows, 129, 130)):
        parts = ops.split_bf16_transposed(dev(src), None if lst is None else dev(lst), n_rows, ldr).cpu()
        want = (src if lst is None else src[lst[:n_rows]]).T                         # (C, n_rows)
        total = parts[0].double() + parts[1].double() + parts[2].double()
        assert torch.equal(total[:, :n_rows].float(), want)                         # 8 + 8 + 8 bits: exact
        assert torch.equal(parts[0][:, :n_rows], want.bfloat16())
        assert (parts[:, :, n_rows:] == 0).all()


@pytest.mark.parametrize("N,D,M,k,n_dead", [(200, 48, 640, 24, 100), (256, 256, 2048, 128, 300), (130, 64, 512, 32, 32)])
def test_dead_latent_decode_matches_gather(device, N, D, M, k, n_dead):
    """DeadLatentDecode (dense products) against the row-gather decode on the same selection:
    forward and both gradients agree to fp32 summation-order accuracy."""
    from sparsify_amd import ops
    from sparsify_amd.dead_decode import DeadLatentDecode
    from sparsify_amd.utils import SparseDecode
    torch.manual_seed(N + n_dead)
    dead = torch.zeros(M, dtype=torch.bool)
    dead[torch.randperm(M)[:n_dead]] = True
    pre = torch.randn(N, M).relu().bfloat16()
    kk = min(k, n_dead)
    acts, idx = ops.topk(dev(pre), kk, keep_mask=dev(dead))
    slot_of, dead_list = ops.dead_compact(dev(dead), n_dead)
    assert torch.equal(dead_list.cpu(), dead.nonzero().flatten())
    want_slots = torch.full((M,), -1, dtype=torch.int32)
    want_slots[dead] = torch.arange(n_dead, dtype=torch.int32)
    assert torch.equal(slot_of.cpu(), want_slots)
    W = torch.randn(M, D)
    b = torch.randn(D)
    gy = dev(torch.randn(N, D))
    res = {}
    for name in ("gather", "dense"):
        Wd, bd = dev(W).requires_grad_(True), dev(b).requires_grad_(True)
        if name == "gather":
            out = SparseDecode.apply(idx, Wd, acts.float(), bd)
        else:
            out = DeadLatentDecode.apply(idx, acts, Wd, bd, dead_list, slot_of, n_dead)
        (out * gy).sum().backward()
        res[name] = (out.detach().cpu(), Wd.grad.cpu(), bd.grad.cpu())
    for a, b_, what in zip(res["dense"], res["gather"], ("decode", "W_dec.grad", "b_dec.grad")):
        torch.testing.assert_close(a, b_, rtol=2e-5, atol=2e-5 * float(b_.abs().max()), msg=lambda m, w=what: f"{w}: {m}")
    assert (res["dense"][1][~dead] == 0).all()


@pytest.mark.parametrize("N,D,M,n_dead,k", [(512, 256, 32768, 3000, 128), (300, 64, 1024, 200, 32), (256, 128, 2048, 77, 77)])
def test_auxk_compact_selection_equals_masked_topk(device, N, D, M, n_dead, k):
    """auxk_select_compact recomputes the dead latents' pre-activations as a compact matrix and
    selects there: the same latents with the same values as the masked top-k over the full row
    (sparse_coder.py:240-243), for the 256-tile and the 128-tile GEMM alike."""
    from sparsify_amd import ops
    from sparsify_amd.dead_decode import auxk_select_compact
    torch.manual_seed(M + n_dead)
    x = dev(torch.randn(N, D).bfloat16())
    W = dev((torch.randn(M, D) / D**0.5).bfloat16())
    b = dev((torch.randn(M) * 0.1).bfloat16())
    dead = torch.zeros(M, dtype=torch.bool)
    dead[torch.randperm(M)[:n_dead]] = True
    pre = ops.encoder_preacts(x, W, b)
    want_acts, want_idx = ops.topk(pre, k, keep_mask=dev(dead))
    slot_of, dead_list = ops.dead_compact(dev(dead), n_dead)
    acts, slots = auxk_select_compact(x, W, b, dead_list, n_dead, k)
    got_idx = dead_list[slots]
    order_a, order_b = got_idx.argsort(-1), want_idx.argsort(-1)
    assert torch.equal(got_idx.gather(-1, order_a), want_idx.gather(-1, order_b))
    assert torch.equal(acts.gather(-1, order_a).view(torch.int16), want_acts.gather(-1, order_b).view(torch.int16))


@pytest.mark.parametrize("R,D,M", [(1, 4, 1), (37, 72, 200), (1000, 4096, 4096)])
def test_scatter_rows_and_scatter_add_rows(device, R, D, M):
    """sphip_scatter_rows: zeros with the listed rows set; sphip_scatter_add_rows: the listed
    (distinct) rows incremented in place, everything else untouched -- both exact."""
    from sparsify_amd import ops
    torch.manual_seed(R + D + M)
    rows = dev(torch.randperm(M)[:R].contiguous())
    src = dev(torch.randn(R + 3, D))                                  # more rows than R: the tail is ignored
    dense = ops.scatter_rows(src, rows, R, M)
    want = torch.zeros(M, D, device="cuda")
    want[rows] = src[:R]
    assert torch.equal(dense, want)
    base = dev(torch.randn(M, D))
    got = ops.scatter_add_rows_(base.clone(), src, rows, R)
    want = base.clone()
    want[rows] += src[:R]
    assert torch.equal(got, want)
    with pytest.raises(ValueError):
        ops.scatter_add_rows_(base.clone(), src[:, : D // 2 or 1].contiguous() if D > 1 else src[:0], rows, R)


@pytest.mark.parametrize("skewed", [False, True])
@pytest.mark.parametrize("rdtype", [torch.bfloat16, torch.float32])
def test_weight_grad_long_segments_take_the_wide_kernel(device, monkeypatch, skewed, rdtype):
    """Without a dot part the L2-sliced kernel leaves latents with long segments (dense latents,
    more than 512 entries) to the chunked long-segment kernels, launched next to it over a device-side
    list of them: 256-entry chunks summed in entry order, chunk partials added in order. The rows of
    all other latents equal the full-row kernel's bit for bit; a long latent's row equals it to fp32
    summation-order accuracy and is the same on every run; the fused SignSGD form
    (sphip_segmented_weight_grad_signsgd) equals gradient + separate step bit for bit."""
    from sparsify_amd import ops
    torch.manual_seed(5)
    N, D, M, k, lr = 4096, 1024, 4096, 32, 3e-3          # N * D * 2 B = 8 MiB: the training-size dispatch
    idx = torch.stack([torch.randperm(M)[:k] for _ in range(N)])
    if skewed:
        idx[:, :3] = torch.tensor([7, 1000, 4095])      # three latents fire on every token: segments of 4096
    idx, coeff = dev(idx), dev(torch.randn(N, k).to(torch.bfloat16))
    rows = dev(torch.randn(N, D).to(rdtype))
    shift = dev(torch.randn(D))
    seg = ops.build_latent_segments(idx, M)
    longest = int((seg.seg_begin[1:] - seg.seg_begin[:-1]).max())
    assert longest >= N if skewed else longest < 128           # 512 is where the wide kernel takes over
    gW, gb, _ = ops.segmented_weight_grad(seg, coeff, rows, want_bias=True, row_shift=shift)
    monkeypatch.setenv("SPHIP_WGRAD", "full")
    gW_full, gb_full, _ = ops.segmented_weight_grad(seg, coeff, rows, want_bias=True, row_shift=shift)
    monkeypatch.delenv("SPHIP_WGRAD")
    long_rows = torch.zeros(M, dtype=torch.bool, device="cuda")
    if skewed:
        long_rows[torch.tensor([7, 1000, 4095])] = True
    assert torch.equal(gW[~long_rows], gW_full[~long_rows]) and torch.equal(gb, gb_full)
    if skewed:
        # 4096 terms of magnitude ~1: the two summation orders differ by a few ulp of the partial sums
        torch.testing.assert_close(gW[long_rows], gW_full[long_rows], rtol=1e-5, atol=1e-3)
        c7 = (coeff.double() * (idx == 7)).sum(1)
        ref64 = (c7[:, None] * rows.double()).sum(0) - c7.sum() * shift.double()
        assert (gW[7].double() - ref64).abs().max() <= 2 * (gW_full[7].double() - ref64).abs().max() + 1e-4
        gW_again, _, _ = ops.segmented_weight_grad(seg, coeff, rows, want_bias=True, row_shift=shift)
        assert torch.equal(gW, gW_again)
    # fused with the optimizer step
    p0 = dev(torch.randn(M, D))
    sep, fused = p0.clone(), p0.clone()
    sh_sep, sh_fused = torch.empty(M, D, dtype=torch.bfloat16, device="cuda"), torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    ops.signsgd_step_(sep, gW, lr, sh_sep)
    ops.segmented_weight_grad_signsgd_(seg, coeff, rows, fused, lr, sh_fused, row_shift=shift)
    assert torch.equal(sep, fused) and torch.equal(sh_sep.view(torch.int16), sh_fused.view(torch.int16))
    assert torch.equal(ops.segment_coeff_sums(seg, coeff), gb)
    # ... and, given the coefficient sums, it leaves per block of 32 latents the partial sums of
    # gb^T W_old (b_dec's encoder-side gradient, otherwise a vec_mat pass before the update):
    # same update bit for bit, the partials against fp64 and identical on a second run
    fused2, fused3 = p0.clone(), p0.clone()
    _, part = ops.segmented_weight_grad_signsgd_(seg, coeff, rows, fused2, lr, sh_fused, row_shift=shift, coeff_sums=gb)
    _, part3 = ops.segmented_weight_grad_signsgd_(seg, coeff, rows, fused3, lr, None, row_shift=shift, coeff_sums=gb)
    assert torch.equal(fused2, sep) and torch.equal(sh_sep.view(torch.int16), sh_fused.view(torch.int16))
    assert part.shape == (M // 32, D) and torch.equal(part, part3)
    want = (gb.double()[:, None] * p0.double()).view(M // 32, 32, D).sum(1)
    torch.testing.assert_close(part.double(), want, rtol=1e-5, atol=1e-4 * float(want.abs().max()))
    torch.testing.assert_close(ops.colsum(part, -1.0), ops.vec_mat(gb, p0, -1.0), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("N,D,M,k", [(300, 200, 1000, 8), (257, 72, 95, 5), (1024, 328, 4100, 16)])
@pytest.mark.parametrize("rdtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("sfree", [False, True])
def test_fused_weight_grad_step_ragged_shapes(device, N, D, M, k, rdtype, sfree):
    """The fused weight-gradient + optimizer kernel on shapes that are not whole blocks of 32
    latents nor whole 128-byte column slices (the lane exchange before the parameter store, the
    partial last slice, the per-block partial sums of gb^T W): bit for bit gradient + separate step."""
    from sparsify_amd import ops
    torch.manual_seed(N + D + M)
    idx = dev(torch.stack([torch.randperm(M)[:k] for _ in range(N)]))
    coeff = dev(torch.randn(N, k).to(torch.bfloat16))
    rows = dev(torch.randn(N, D).to(rdtype))
    shift = dev(torch.randn(D))
    lr, beta, c = 3e-3, 0.95, 0.25
    seg = ops.build_latent_segments(idx, M)
    gW, gb, _ = ops.segmented_weight_grad(seg, coeff, rows, want_bias=True, row_shift=shift)
    p0, z0 = dev(torch.randn(M, D)), dev(torch.randn(M, D))
    sep, fused, zs, zf = p0.clone(), p0.clone(), z0.clone(), z0.clone()
    sh_sep = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    sh_fused = torch.empty_like(sh_sep)
    if sfree:
        ops.schedulefree_signsgd_step_(sep, zs, gW, lr, beta, c, sh_sep)
        _, part = ops.segmented_weight_grad_signsgd_(seg, coeff, rows, fused, lr, sh_fused, row_shift=shift,
                                                     schedule_free=(zf, beta, c), coeff_sums=gb)
        assert torch.equal(zs, zf)
    else:
        ops.signsgd_step_(sep, gW, lr, sh_sep)
        _, part = ops.segmented_weight_grad_signsgd_(seg, coeff, rows, fused, lr, sh_fused, row_shift=shift, coeff_sums=gb)
    assert torch.equal(sep, fused) and torch.equal(sh_sep.view(torch.int16), sh_fused.view(torch.int16))
    nb = -(-M // 32)
    pad = torch.zeros(nb * 32, D, dtype=torch.float64, device="cuda")
    pad[:M] = gb.double()[:, None] * p0.double()
    want = pad.view(nb, 32, D).sum(1)
    assert part.shape == (nb, D)
    torch.testing.assert_close(part.double(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))


@pytest.mark.parametrize("skewed", [False, True])
def test_weight_grad_l2_dot_path(device, monkeypatch, skewed):
    """W_dec gradient + d top_acts through the L2-sliced kernel with per-slice dot partials
    (SPHIP_WGRAD=l2; long segments go to the chunked long-segment kernels over the device-side list,
    and the sum of the partials picks the right slab per entry): gW bit for bit the full-row kernel's
    (long latents: to summation-order accuracy), the dots equal up to summation order."""
    from sparsify_amd import ops
    torch.manual_seed(11)
    N, D, M, k = 4096, 1024, 4096, 32
    idx = torch.stack([torch.randperm(M)[:k] for _ in range(N)])
    if skewed:
        idx[:, :2] = torch.tensor([5, 4090])             # two latents fire on every token
        idx[::3, 2] = 17                                  # one on every third (1366 entries > 512)
    idx, coeff = dev(idx), dev(torch.randn(N, k).to(torch.bfloat16))
    rows, W = dev(torch.randn(N, D)), dev(torch.randn(M, D))
    seg = ops.build_latent_segments(idx, M)
    monkeypatch.setenv("SPHIP_WGRAD", "l2")
    gW, gb, dc = ops.segmented_weight_grad(seg, coeff, rows, W, want_bias=True, want_dcoeff=True)
    gW2, _, dc2 = ops.segmented_weight_grad(seg, coeff, rows, W, want_bias=True, want_dcoeff=True)
    assert torch.equal(gW, gW2) and torch.equal(dc, dc2)                      # deterministic
    monkeypatch.setenv("SPHIP_WGRAD", "full")
    gW_f, gb_f, dc_f = ops.segmented_weight_grad(seg, coeff, rows, W, want_bias=True, want_dcoeff=True)
    long_rows = torch.zeros(M, dtype=torch.bool, device="cuda")
    if skewed:
        long_rows[torch.tensor([5, 4090, 17])] = True    # chunked sums: fp32 summation-order accuracy (see above)
        torch.testing.assert_close(gW[long_rows], gW_f[long_rows], rtol=1e-5, atol=1e-3)
    assert torch.equal(gW[~long_rows], gW_f[~long_rows]) and torch.equal(gb, gb_f)
    ref = (rows[:, None, :] * W[idx]).sum(-1) if N * k * D <= (1 << 28) else None
    torch.testing.assert_close(dc, dc_f, rtol=1e-4, atol=1e-3)
    if ref is not None:
        torch.testing.assert_close(dc, ref, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("rdtype,dot", [(torch.float32, True), (torch.float32, False), (torch.bfloat16, False)])
@pytest.mark.parametrize("skewed", [False, True])
def test_weight_grad_accumulates_over_token_blocks(device, rdtype, dot, skewed):
    """sphip_segmented_weight_grad_acc: a second block of tokens adds its sums to the rows the first one wrote --
    bit for bit the fp32 sum of the two blocks' own outputs (gW and gbias; long segments through the chunked path
    included), dcoeff per block untouched; and ops.weight_grad_in_token_blocks over a batch too long for the
    L2-sliced kernels (the reference's index_add_ / atomics take any number of rows, fused_encoder.py:67-84,
    xformers.py:110-136) against a float64 sum over the whole batch."""
    from sparsify_amd import ops
    torch.manual_seed(5)
    N, D, M, k = 8192, 1024, 2048, 16
    idx = torch.stack([torch.randperm(M)[:k] for _ in range(N)])
    if skewed:
        idx[:, 0] = 9                                     # fires on every token: thousands of entries per block
        idx[::5, 1] = 1200
    idx = dev(idx)
    coeff = dev(torch.randn(N, k).to(torch.bfloat16))
    rows = dev(torch.randn(N, D).to(rdtype))
    W = dev(torch.randn(M, D)) if dot else None
    shift = None if dot else dev(torch.randn(D))
    kw = dict(want_bias=True, want_dcoeff=dot, row_shift=shift)
    h = N // 2
    segs = [ops.build_latent_segments(idx[lo:hi], M) for lo, hi in ((0, h), (h, N))]
    a = ops.segmented_weight_grad(segs[0], coeff[:h], rows[:h], W, **kw)
    b = ops.segmented_weight_grad(segs[1], coeff[h:], rows[h:], W, **kw)
    gW, gb = a[0].clone(), a[1].clone()
    dc2 = torch.full((N - h, k), float("nan"), device="cuda") if dot else None
    ops.segmented_weight_grad(segs[1], coeff[h:], rows[h:], W, into=(gW, gb), dcoeff_out=dc2, **kw)
    assert torch.equal(gW, a[0] + b[0]) and torch.equal(gb, a[1] + b[1])
    if dot:
        assert torch.equal(dc2, b[2])
    # the driver: blocks of at most TOKEN_BLOCK tokens
    old = ops.TOKEN_BLOCK
    ops.TOKEN_BLOCK = 3000
    try:
        assert ops.token_blocks(N)[-1][1] == N and all(hi - lo <= 3000 for lo, hi in ops.token_blocks(N))
        tW, tb, tdc = ops.weight_grad_in_token_blocks(idx, M, coeff, rows, W, **kw)
    finally:
        ops.TOKEN_BLOCK = old
    hit = torch.zeros(N, M, dtype=torch.float64, device="cuda")
    hit.scatter_add_(1, idx, coeff.double())
    want_b = hit.sum(0)
    want_W = hit.T @ rows.double()
    if shift is not None:
        want_W -= want_b[:, None] * shift.double()
    torch.testing.assert_close(tW.double(), want_W, rtol=1e-5, atol=2e-5 * float(want_W.abs().max()))
    torch.testing.assert_close(tb.double(), want_b, rtol=1e-5, atol=1e-4)
    if dot:
        whole = ops.segmented_weight_grad(ops.build_latent_segments(idx, M), coeff, rows, W, **kw)
        torch.testing.assert_close(tdc, whole[2], rtol=1e-4, atol=1e-3)
    tW2, tb2, _ = ops.weight_grad_in_token_blocks(idx, M, coeff, rows, W, **kw)      # default blocks: one call
    torch.testing.assert_close(tW2, tW, rtol=1e-5, atol=2e-5 * float(want_W.abs().max()))


def test_weight_grad_token_blocks_fall_back_where_the_kernel_cannot_add(device):
    """A shape whose dcoeff path is not the L2-sliced kernel (small rows): the accumulate call reports
    SPHIP_EUNSUPPORTED without touching gW, and the driver runs the whole batch in one call instead."""
    from sparsify_amd import _lib, ops
    torch.manual_seed(6)
    N, D, M, k = 300, 64, 200, 4
    idx = dev(torch.stack([torch.randperm(M)[:k] for _ in range(N)]))
    coeff, rows, W = dev(torch.randn(N, k)), dev(torch.randn(N, D)), dev(torch.randn(M, D))
    seg = ops.build_latent_segments(idx, M)
    gW0, gb0, dc0 = ops.segmented_weight_grad(seg, coeff, rows, W, want_bias=True, want_dcoeff=True)
    gW = gW0.clone()
    with pytest.raises(_lib.HipLibraryError) as err:
        ops.segmented_weight_grad(seg, coeff, rows, W, want_bias=True, want_dcoeff=True, into=(gW, gb0.clone()))
    assert err.value.status == _lib.SPHIP_EUNSUPPORTED and torch.equal(gW, gW0)
    old = ops.TOKEN_BLOCK
    ops.TOKEN_BLOCK = 128
    try:
        tW, tb, tdc = ops.weight_grad_in_token_blocks(idx, M, coeff, rows, W, want_bias=True, want_dcoeff=True)
    finally:
        ops.TOKEN_BLOCK = old
    assert torch.equal(tW, gW0) and torch.equal(tb, gb0) and torch.equal(tdc, dc0)


@pytest.mark.parametrize("R,D", [(1000, 96), (33, 4096), (4099, 288), (64, 32)])
def test_slice_major_copy_ragged(device, R, D):
    """sphip_slice_major_f32 on row counts that are not whole 32-row tiles and slice counts that are
    not whole groups of 8."""
    from sparsify_amd import ops
    torch.manual_seed(R + D)
    t = dev(torch.randn(R, D))
    sm = ops.slice_major_copy(t)
    assert torch.equal(sm, t.view(R, D // 32, 32).transpose(0, 1))


@pytest.mark.parametrize("D,renorm", [(256, True), (4096, True), (1056, False), (8192, True)])
def test_decoder_step_writes_the_slice_major_copy(device, D, renorm):
    """sphip_decoder_project_signsgd with a slice-major buffer: the parameter is bit for bit what the
    call without the buffer leaves, and the buffer equals sphip_slice_major_f32 of the updated
    parameter (what the L2-resident decode would otherwise have to copy)."""
    from sparsify_amd import _lib, ops
    from sparsify_amd._lib import check
    M = 1000
    g = torch.Generator(device="cuda").manual_seed(D)
    W0 = torch.randn(M, D, device="cuda", generator=g)
    W0 /= W0.norm(dim=1, keepdim=True)
    G = torch.randn(M, D, device="cuda", generator=g)
    plain = ops.decoder_project_signsgd_(W0.clone(), G, 3e-3, renorm)
    buf = torch.full((D // 32, M, 32), float("nan"), device="cuda")
    got = ops.decoder_project_signsgd_(W0.clone(), G, 3e-3, renorm, slice_major=buf)
    assert torch.equal(got, plain)
    want = torch.empty_like(buf)
    check(_lib.load().sphip_slice_major_f32(got.data_ptr(), want.data_ptr(), M, D, None), "sphip_slice_major_f32")
    torch.cuda.synchronize()
    assert torch.equal(buf, want)
    assert torch.equal(buf, got.view(M, D // 32, 32).permute(1, 0, 2).contiguous())


@pytest.mark.parametrize("M,D", [(1000, 256), (777, 4096), (64, 8192), (333, 96), (50, 12288)])
def test_plain_signsgd_row_step_writes_the_slice_major_copy(device, M, D):
    """sphip_signsgd_step_rows (a transcoder's decoder: SignSGD.step, sign_sgd.py:15-23, no projection, no
    normalisation): the parameter bit for bit sphip_signsgd_step's, zero gradients leaving their elements alone, and
    the buffer the slice-major form of the updated matrix -- rows wider than the row-resident kernel takes included."""
    from sparsify_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + D)
    W0 = torch.randn(M, D, device="cuda", generator=g)
    G = torch.randn(M, D, device="cuda", generator=g)
    G[::7] = 0.0
    plain = ops.signsgd_step_(W0.clone(), G, 3e-3)
    buf = torch.full((D // 32, M, 32), float("nan"), device="cuda")
    got = ops.signsgd_step_rows_(W0.clone(), G, 3e-3, buf)
    assert torch.equal(got, plain) and torch.equal(got[::7], W0[::7])
    assert torch.equal(buf, got.view(M, D // 32, 32).permute(1, 0, 2).contiguous())
    assert torch.equal(ops.signsgd_step_rows_(W0.clone(), G, 3e-3, None), plain)


@pytest.mark.parametrize("M,D", [(1000, 256), (777, 4096), (64, 8192), (333, 96)])
def test_plain_schedulefree_row_step_writes_the_slice_major_copy(device, M, D):
    """sphip_schedulefree_signsgd_step_rows: y and z bit for bit the flat schedule-free step's, the buffer the
    slice-major form of the new y."""
    from sparsify_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + D + 1)
    W0 = torch.randn(M, D, device="cuda", generator=g)
    Z0 = W0 + 0.01 * torch.randn(M, D, device="cuda", generator=g)
    G = torch.randn(M, D, device="cuda", generator=g)
    G[::5] = 0.0
    Wr, Zr = W0.clone(), Z0.clone()
    ops.schedulefree_signsgd_step_(Wr, Zr, G, 3e-3, 0.95, 0.25, None)
    buf = torch.full((D // 32, M, 32), float("nan"), device="cuda")
    Wf, Zf = W0.clone(), Z0.clone()
    ops.schedulefree_signsgd_step_rows_(Wf, Zf, G, 3e-3, 0.95, 0.25, buf)
    assert torch.equal(Wf, Wr) and torch.equal(Zf, Zr)
    assert torch.equal(buf, Wf.view(M, D // 32, 32).permute(1, 0, 2).contiguous())


@pytest.mark.parametrize("D", [64, 1024, 2560, 4096])
@pytest.mark.parametrize("renorm", [False, True])
def test_schedulefree_decoder_step_writes_the_slice_major_copy(device, D, renorm):
    """sphip_decoder_schedulefree_signsgd_step with a slice-major buffer: y and z bit for bit what
    the call without the buffer leaves, the buffer the slice-major form of the new y."""
    from sparsify_amd import ops
    M = 777
    g = torch.Generator(device="cuda").manual_seed(D + 1)
    W0 = torch.randn(M, D, device="cuda", generator=g)
    W0 /= W0.norm(dim=1, keepdim=True)
    Z0 = W0 + 0.01 * torch.randn(M, D, device="cuda", generator=g)
    G = torch.randn(M, D, device="cuda", generator=g)
    Wp, Zp = W0.clone(), Z0.clone()
    ops.decoder_schedulefree_signsgd_step_(Wp, Zp, G, 3e-3, 0.95, 0.25, renorm)
    Wg, Zg = W0.clone(), Z0.clone()
    buf = torch.full((D // 32, M, 32), float("nan"), device="cuda")
    ops.decoder_schedulefree_signsgd_step_(Wg, Zg, G, 3e-3, 0.95, 0.25, renorm, slice_major=buf)
    assert torch.equal(Wg, Wp) and torch.equal(Zg, Zp)
    assert torch.equal(buf, Wg.view(M, D // 32, 32).permute(1, 0, 2).contiguous())


@pytest.mark.parametrize("schedule_free", [False, True])
def test_apply_signs_by_row_blocks_writes_the_slice_major_copy(device, schedule_free):
    """The data-parallel decoder step (sphip_decoder_apply_signs / ..._schedulefree_apply_signs) run
    row block by row block with the slice-major buffer of the whole matrix: parameters bit for
    bit those of the call without the buffer, the buffer the slice-major form of the result."""
    from sparsify_amd import ops
    M, D = 1000, 1024
    g = torch.Generator(device="cuda").manual_seed(3)
    W0 = torch.randn(M, D, device="cuda", generator=g)
    W0 /= W0.norm(dim=1, keepdim=True)
    Z0 = W0 + 0.01 * torch.randn(M, D, device="cuda", generator=g)
    signs = torch.randint(-1, 2, (M, D), device="cuda", generator=g, dtype=torch.int8)
    Wp, Zp = W0.clone(), Z0.clone()
    if schedule_free:
        ops.decoder_schedulefree_apply_signs_(Wp, Zp, signs, 3e-3, 0.95, 0.25, True)
    else:
        ops.decoder_apply_signs_(Wp, signs, 3e-3, True)
    Wg, Zg = W0.clone(), Z0.clone()
    buf = torch.full((D // 32, M, 32), float("nan"), device="cuda")
    for lo, hi in ((0, 256), (256, 700), (700, M)):
        if schedule_free:
            ops.decoder_schedulefree_apply_signs_(Wg[lo:hi], Zg[lo:hi], signs[lo:hi], 3e-3, 0.95, 0.25, True,
                                                  slice_major=buf, row0=lo)
        else:
            ops.decoder_apply_signs_(Wg[lo:hi], signs[lo:hi], 3e-3, True, slice_major=buf, row0=lo)
    assert torch.equal(Wg, Wp) and torch.equal(Zg, Zp)
    assert torch.equal(buf, Wg.view(M, D // 32, 32).permute(1, 0, 2).contiguous())
    with pytest.raises(ValueError):
        ops.decoder_apply_signs_(Wg[0:10], signs[0:10], 3e-3, True, slice_major=buf, row0=M - 5)


@pytest.mark.parametrize("M", [1, 15, 16, 17, 1000, 16400, 131071, 131072, 300_000])
def test_dead_compact_any_length_and_alignment(device, M):
    """sphip_dead_compact: slot_of / dead_list of a mask of any length, dense, sparse, empty and full,
    also through a mask that does not start on a 16-byte boundary (the vector loads fall back)."""
    from sparsify_amd import ops
    gen = torch.Generator().manual_seed(M)
    for frac in (0.0, 0.1, 0.5, 1.0):
        for shift in (0, 1):
            base = torch.zeros(M + shift, dtype=torch.bool)
            base[shift:] = torch.rand(M, generator=gen) < frac
            dead = dev(base)[shift:]
            want_list = base[shift:].nonzero().flatten()
            n = int(want_list.numel())
            slot_of, dead_list = ops.dead_compact(dead, n)
            want_slot = torch.where(base[shift:], base[shift:].cumsum(0) - 1, torch.tensor(-1)).to(torch.int32)
            assert torch.equal(slot_of.cpu(), want_slot) and torch.equal(dead_list.cpu(), want_list)


@pytest.mark.parametrize("N,k,n_dead,M", [(100, 8, 40, 512), (64, 16, 64, 64), (257, 32, 700, 4096), (1, 4, 5, 32),
                                          (9, 16, 32760, 40000),      # a row of A fills the 64 KiB of LDS exactly
                                          (9, 16, 33000, 40000)])     # one more tile of slots: the scatter form
@pytest.mark.parametrize("mapped", [False, True])
def test_dead_scatter_acts_composed_rows_equal_the_scatter(device, monkeypatch, N, k, n_dead, M, mapped):
    """sphip_dead_scatter_acts: A (N, Kd) and A^T (Kd, ceil64 N) composed row by row in LDS and
    transposed by tiles, against a dense torch construction and against the element-wise scatter
    form (SPHIP_DEAD_SCATTER=1), bit for bit, padding included."""
    from sparsify_amd import ops
    gen = torch.Generator().manual_seed(N + k)
    dead = torch.zeros(M, dtype=torch.bool)
    dead[torch.randperm(M, generator=gen)[:n_dead]] = True
    dead_ids = dead.nonzero().flatten()
    pick = torch.stack([torch.randperm(n_dead, generator=gen)[:k] for _ in range(N)])       # distinct slots per token
    acts = torch.randn(N, k, generator=gen).to(torch.bfloat16)
    Kd = -(-n_dead // 64) * 64
    if mapped:
        slot_of, _ = ops.dead_compact(dev(dead), n_dead)
        idx = dead_ids[pick]
        if N > 2:
            idx[1, 0] = -1                                    # inactive slot
    else:
        slot_of, idx = None, pick.clone()
    want = torch.zeros(N, Kd, dtype=torch.bfloat16)
    for n in range(N):
        for e in range(k):
            if idx[n, e] >= 0:
                want[n, pick[n, e]] = acts[n, e]
    A, At = ops.dead_scatter_acts(dev(idx), dev(acts), slot_of, Kd, num_valid=n_dead)
    assert torch.equal(A.cpu().view(torch.int16), want.view(torch.int16))
    ldT = -(-N // 64) * 64
    wantT = torch.zeros(Kd, ldT, dtype=torch.bfloat16)
    wantT[:, :N] = want.T
    assert torch.equal(At.cpu().view(torch.int16), wantT.view(torch.int16))
    monkeypatch.setenv("SPHIP_DEAD_SCATTER", "1")
    A2, At2 = ops.dead_scatter_acts(dev(idx), dev(acts), slot_of, Kd, num_valid=n_dead)
    assert torch.equal(A2.view(torch.int16), A.view(torch.int16)) and torch.equal(At2.view(torch.int16), At.view(torch.int16))
