"""Top-k without the (N, M) pre-activation matrix (csrc/lazy.hip, ops.encode_topk_lazy) against
the ordinary path (sphip_encoder_preacts + sphip_topk), which the oracle tests pin: the same
values on the same indices for every row, bit for bit (integer / index work: exact).
Reference: sparsify/fused_encoder.py:29-33, 50."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _force_256(monkeypatch):
    # the candidate-emitting GEMM is the 256x256 kernel; the ordinary path of these small shapes
    # would otherwise take the 128x128 kernel, which rounds a few outputs differently
    monkeypatch.setenv("SPHIP_GEMM", "256")


def _operands(N, D, M, seed, scale=1.0, integer=False):
    g = torch.Generator().manual_seed(seed)
    if integer:
        x = torch.randint(-2, 3, (N, D), generator=g).float()
        W = torch.randint(-2, 3, (M, D), generator=g).float()
        b = torch.randint(-2, 3, (M,), generator=g).float()
    else:
        x = torch.randn(N, D, generator=g)
        W = torch.randn(M, D, generator=g) * scale / D ** 0.5
        b = torch.randn(M, generator=g) * 0.1
    return x.cuda().bfloat16(), W.cuda().bfloat16(), b.cuda().bfloat16()


def _eager(x, W, b, k):
    from sparsify_amd import ops

    pre, tm = ops.encoder_preacts(x, W, b, want_tile_max=True)
    return ops.topk(pre, k, tile_max=tm)


def _same_selection(a, b):
    (va, ia), (vb, ib) = a, b
    oa, ob = ia.argsort(dim=1), ib.argsort(dim=1)
    assert torch.equal(ia.gather(1, oa), ib.gather(1, ob))
    assert torch.equal(va.gather(1, oa).view(torch.int16), vb.gather(1, ob).view(torch.int16))


@pytest.mark.parametrize("N,D,M,k,bias", [(512, 256, 16384, 32, True), (300, 128, 16392, 64, True),
                                          (256, 64, 4096, 17, False), (1024, 512, 32768, 192, True),
                                          (64, 64, 2048, 256, True)])
def test_lazy_topk_equals_the_ordinary_path(N, D, M, k, bias):
    from sparsify_amd import ops

    x, W, b = _operands(N, D, M, seed=N + M + k)
    b = b if bias else None
    vals, idx, flag = ops.encode_topk_lazy(x, W, b, k, fallback=False)
    assert int(flag.item()) == 0
    _same_selection((vals, idx), _eager(x, W, b, k))
    # value-descending, ties by ascending index: the order sphip_topk's list path leaves
    v = vals.float()
    assert bool((v[:, :-1] >= v[:, 1:]).all())
    tie = v[:, :-1] == v[:, 1:]
    assert bool((idx[:, :-1][tie] < idx[:, 1:][tie]).all())


def test_lazy_topk_with_rows_full_of_ties():
    """Small-integer operands: a handful of distinct values per row, hundreds of ties at the
    k-th value. The selection is still the lowest-index one; rows whose ties exceed the sort
    buffer raise the flag and are redone by the predicated ordinary path."""
    from sparsify_amd import ops

    N, D, M, k = 256, 64, 16384, 48
    x, W, b = _operands(N, D, M, seed=5, integer=True)
    want = _eager(x, W, b, k)
    vals, idx, flag = ops.encode_topk_lazy(x, W, b, k, fallback=True)
    _same_selection((vals, idx), want)
    # and without the fallback: either exact, or flagged
    vals2, idx2, flag2 = ops.encode_topk_lazy(x, W, b, k, fallback=False)
    if int(flag2.item()) == 0:
        _same_selection((vals2, idx2), want)


@pytest.mark.parametrize("redo_rows", [2048, 96])
def test_lazy_topk_overflow_takes_the_predicated_fallback(monkeypatch, redo_rows):
    """The first M/4 latents are all inactive (threshold 0), every other chunk holds a positive
    value: 6144 chunks per token against a capacity of 4096. The flag goes up and the
    predicated launches behind it produce the ordinary path's result -- in one piece, or 96 tokens
    at a time through the fixed-size workspace (ops.LAZY_REDO_ROWS; 256 tokens = 2 pieces + a tail)."""
    from sparsify_amd import ops
    monkeypatch.setattr(ops, "LAZY_REDO_ROWS", redo_rows)

    N, D, M, k = 256, 64, 65536, 32
    x, W, b = _operands(N, D, M, seed=9)
    Ma = ops.lazy_split(M)
    b = b.clone()
    b[:Ma] = -100.0
    b[Ma:] += 1.0
    _, _, flag = ops.encode_topk_lazy(x, W, b, k, fallback=False)
    assert int(flag.item()) == 1
    vals, idx, flag = ops.encode_topk_lazy(x, W, b, k, fallback=True)
    assert int(flag.item()) == 1
    _same_selection((vals, idx), _eager(x, W, b, k))
    # the next call starts from cleared counters
    x2, W2, b2 = _operands(N, D, M, seed=10)
    vals, idx, flag = ops.encode_topk_lazy(x2, W2, b2, k, fallback=True)
    assert int(flag.item()) == 0
    _same_selection((vals, idx), _eager(x2, W2, b2, k))


def test_lazy_topk_few_positive_values():
    """Fewer than k positive pre-activations in a row: the selection is filled with zeros of the
    lowest indices, exactly as the ordinary path does."""
    from sparsify_amd import ops

    N, D, M, k = 128, 64, 8192, 64
    x, W, b = _operands(N, D, M, seed=11)
    b = torch.full_like(b, -3.5)           # a few dozen positives per row at most
    want = _eager(x, W, b, k)
    assert int((want[0] == 0).sum()) > 0
    vals, idx, flag = ops.encode_topk_lazy(x, W, b, k, fallback=False)
    assert int(flag.item()) == 0
    _same_selection((vals, idx), want)


def test_lazy_topk_full_size():
    """BASELINE shape: N = 16384 tokens, D = 4096, M = 131072, k = 192."""
    from sparsify_amd import ops

    os.environ.pop("SPHIP_GEMM", None)
    N, D, M, k = 16384, 4096, 131072, 192
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(N, D, device="cuda", generator=g).bfloat16()
    W = (torch.randn(M, D, device="cuda", generator=g) / D ** 0.5).bfloat16()
    b = (torch.randn(M, device="cuda", generator=g) * 0.1).bfloat16()
    assert ops.lazy_eligible(N, D, M, k)
    vals, idx, flag = ops.encode_topk_lazy(x, W, b, k, fallback=False)
    assert int(flag.item()) == 0
    want = _eager(x, W, b, k)
    _same_selection((vals, idx), want)
    assert torch.equal(idx, want[1])        # same order too: both are value-descending, ties by index


# ---------------------------------------------------------------- the operator surface
def _coder_inputs(N, D, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(N, D, generator=g).cuda()


def test_fused_encoder_lazy_pre_acts():
    """fused_encoder(..., lazy_pre_acts=True): EncoderOutput is still the reference's 3-tuple
    (fused_encoder.py:21-30, 50); values, indices and gradients equal the storing path's, and
    pre_acts materialises to the same matrix when something reads it."""
    from sparsify_amd.fused_encoder import LazyPreActs, fused_encoder

    N, D, M, k = 512, 128, 8192, 32
    x = _coder_inputs(N, D)
    g = torch.Generator().manual_seed(1)
    W = (torch.randn(M, D, generator=g) / D ** 0.5).cuda().requires_grad_()
    b = (torch.randn(M, generator=g) * 0.1).cuda().requires_grad_()
    gv = torch.randn(N, k, generator=g).cuda()
    outs = []
    for lazy in (False, True):
        W.grad = b.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = fused_encoder(x, W, b, k, "topk", lazy_pre_acts=lazy)
        assert len(out) == 3
        out.top_acts.backward(gv.to(out.top_acts.dtype))
        outs.append((out, W.grad.clone(), b.grad.clone()))
    (e, gWe, gbe), (l, gWl, gbl) = outs
    assert isinstance(l.pre_acts, LazyPreActs) and not isinstance(e.pre_acts, LazyPreActs)
    assert torch.equal(l.top_indices, e.top_indices) and torch.equal(l.top_acts, e.top_acts)
    assert torch.equal(gWl, gWe) and torch.equal(gbl, gbe)
    assert l.pre_acts.shape == e.pre_acts.shape and l.pre_acts.dtype == e.pre_acts.dtype
    assert not l.pre_acts.is_materialized
    mask = torch.rand(M, device="cuda") < 0.5
    masked = torch.where(mask[None], l.pre_acts, -torch.inf)          # sparse_coder.py:240
    assert l.pre_acts.is_materialized
    assert torch.equal(masked, torch.where(mask[None], e.pre_acts, -torch.inf))
    assert torch.equal(l.pre_acts.reshape(-1, M), e.pre_acts)

    # read after the weights moved: refused, not silently different
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = fused_encoder(x, W, b, k, "topk", lazy_pre_acts=True)
    with torch.no_grad():
        W.mul_(1.0)
    with pytest.raises(RuntimeError, match="parameters changed"):
        out.pre_acts.materialize()


@pytest.mark.parametrize("transcode", [False, True])
def test_sparse_coder_lazy_equals_storing(transcode, monkeypatch):
    from sparsify_amd import SparseCoder, SparseCoderConfig
    from sparsify_amd.fused_encoder import LazyPreActs

    N, D, M, k = 512, 128, 8192, 32
    torch.manual_seed(0)
    sae = SparseCoder(D, SparseCoderConfig(num_latents=M, k=k, transcode=transcode)).cuda()
    x = _coder_inputs(N, D, 3)
    y = _coder_inputs(N, D, 4) if transcode else None
    from sparsify_amd import ops
    calls = []
    real = ops.encode_topk_lazy
    monkeypatch.setattr(ops, "encode_topk_lazy", lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
    res = []
    for lazy in (False, True):
        sae.lazy_pre_acts = lazy
        sae.zero_grad(set_to_none=True)
        out = sae(x, y)
        out.fvu.backward()
        res.append((out, {n: p.grad.clone() for n, p in sae.named_parameters() if p.grad is not None}))
        assert len(calls) == int(lazy)
    (e, ge), (l, gl) = res
    assert torch.equal(l.latent_indices, e.latent_indices) and torch.equal(l.latent_acts, e.latent_acts)
    assert torch.equal(l.sae_out, e.sae_out) and torch.equal(l.fvu, e.fvu)
    assert ge.keys() == gl.keys()
    for n in ge:
        assert torch.equal(ge[n], gl[n]), n
    # AuxK reads pre_acts: the forward stores them whatever the switch says
    sae.lazy_pre_acts = True
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert isinstance(sae.encode(x).pre_acts, LazyPreActs)
    assert len(calls) == 2
    dead = torch.zeros(M, dtype=torch.bool, device="cuda")
    dead[::7] = True
    a = sae(x, y, dead_mask=dead)
    assert len(calls) == 2
    sae.lazy_pre_acts = False
    b = sae(x, y, dead_mask=dead)
    assert torch.equal(a.auxk_loss, b.auxk_loss) and float(a.auxk_loss.detach()) > 0


def test_trainer_lazy_pre_acts_same_run(tmp_path, monkeypatch):
    """TrainConfig.lazy_pre_acts: ten SignSGD steps end in bit-identical parameters."""
    from sparsify_amd import SparseCoder, SparseCoderConfig, TrainConfig
    from sparsify_amd.trainer import Trainer

    class Source:
        def __init__(self, batches, d):
            self.batches, self.widths = batches, {"h0": d}

        def __len__(self):
            return len(self.batches)

        def __iter__(self):
            for b in self.batches:
                yield {"h0": b}

    N, D, M, k, steps = 512, 128, 8192, 32, 10
    batches = [_coder_inputs(N, D, 100 + i) for i in range(steps)]
    from sparsify_amd import ops
    calls = []
    real = ops.encode_topk_lazy
    monkeypatch.setattr(ops, "encode_topk_lazy", lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
    finals = []
    for lazy in (False, True):
        cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k), optimizer="signsgd", lr=1e-3, save_dir=str(tmp_path),
                          save_every=10**9, lazy_pre_acts=lazy)
        torch.manual_seed(0)
        sae0 = SparseCoder(D, cfg.sae).cuda()
        tr = Trainer(cfg, Source(batches, D), device="cuda", saes={"h0": sae0})
        for b in batches:
            tr.train_micro_batch({"h0": b})
        finals.append({n: p.detach().clone() for n, p in tr.saes["h0"].named_parameters()})
        assert len(calls) == (steps if lazy else 0)
    for n in finals[0]:
        assert torch.equal(finals[0][n], finals[1][n]), n
