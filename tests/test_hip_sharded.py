"""Latent-dim sharding on one GPU: two "ranks" run as two threads of this process with an
in-process stand-in for the two collectives (all-gather of candidates, all-reduce of partial
reconstructions); everything else is the real HIP path. Contract: identical
to the unsharded SparseCoder."""
import threading

import pytest
import torch

from oracle import sae_oracle as O

pytestmark = pytest.mark.gpu


def own_stream(torch):
    """A stream of its own for a thread-rank: the package's scratch buffers are per (device, stream),
    as they must be for real ranks (one process each); two thread-ranks on one stream would share them
    and interleave their multi-kernel reductions."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.default_stream())
    torch.cuda.set_stream(s)


class FakeWorld:
    """Thread-per-rank collectives with the signatures of sparsify_amd.parallel."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.local = threading.local()
        self.slots = [None] * world
        self.lock = threading.Lock()
        self.bag = []

    def rank(self):
        return self.local.rank

    def world_size(self):
        return self.world

    def shard_range(self, M, r=None, w=None):
        per = M // self.world
        r = self.rank() if r is None else r
        return r * per, (r + 1) * per

    def _exchange(self, value):
        self.slots[self.rank()] = value
        self.barrier.wait()
        got = list(self.slots)
        self.barrier.wait()
        return got

    def all_reduce_(self, x, op="mean"):
        torch.cuda.synchronize()
        parts = self._exchange(x.clone())
        tot = sum(parts[1:], parts[0].clone())
        x.copy_(tot / self.world if op == "mean" else tot)
        return x

    def gather_topk_candidates(self, lv, li, lo):
        torch.cuda.synchronize()
        parts = self._exchange((lv.clone(), (li + lo).clone()))
        return torch.cat([p[0] for p in parts], 1), torch.cat([p[1] for p in parts], 1)


@pytest.mark.parametrize("transcode", [True, False])
def test_sharded_equals_unsharded(device, monkeypatch, transcode):
    from sparsify_amd import SparseCoder, SparseCoderConfig, parallel
    from sparsify_amd.sharded import ShardedSparseCoder

    W = 2
    N, D, M, k = 96, 64, 512, 12
    fw = FakeWorld(W)
    for name in ("rank", "world_size", "shard_range", "all_reduce_", "gather_topk_candidates"):
        monkeypatch.setattr(parallel, name, getattr(fw, name))

    cfg = SparseCoderConfig(num_latents=M, k=k, transcode=transcode)
    torch.manual_seed(0)
    ref = SparseCoder(D, cfg).cuda()
    with torch.no_grad():
        ref.encoder.bias.normal_(0, 0.05); ref.b_dec.normal_(0, 0.3)
        if transcode:
            ref.W_dec.normal_(0, D**-0.5)
    torch.manual_seed(1)
    x = (torch.randn(N, D) + torch.randn(D)).cuda()
    y = (x @ (torch.randn(D, D, device="cuda") / D**0.5)) if transcode else None

    with torch.autocast("cuda", enabled=False):
        out_ref = type(ref).forward.__wrapped__(ref, x, y)
    out_ref.fvu.backward()

    results = [None] * W
    errors = []

    def worker(r):
        try:
            fw.local.rank = r
            torch.cuda.set_device(0)
            own_stream(torch)
            lo, hi = fw.shard_range(M)
            sh = ShardedSparseCoder(D, cfg, device="cuda")
            with torch.no_grad():
                sh.encoder.weight.copy_(ref.encoder.weight[lo:hi]); sh.encoder.bias.copy_(ref.encoder.bias[lo:hi])
                sh.W_dec.copy_(ref.W_dec[lo:hi]); sh.b_dec.copy_(ref.b_dec)
            with torch.autocast("cuda", enabled=False):
                out = type(sh).forward.__wrapped__(sh, x, y)
            out.fvu.backward()
            torch.cuda.synchronize()
            results[r] = (out, sh, lo, hi)
        except Exception as e:      # pragma: no cover
            errors.append(e)
            fw.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
    [t.start() for t in threads]
    [t.join(120) for t in threads]
    assert not errors, errors
    # reduce_shared_grads(): b_dec.grad summed over the shards (done here on the main thread:
    # the two "ranks" share one autograd device thread, so backward itself has no collective)
    bdec_grad = sum(r[1].b_dec.grad for r in results)
    for out, sh, lo, hi in results:
        assert torch.equal(out.latent_indices.sort(-1).values, out_ref.latent_indices.sort(-1).values)
        torch.testing.assert_close(out.sae_out, out_ref.sae_out, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(out.fvu, out_ref.fvu, rtol=1e-5, atol=1e-7)
        tol = dict(rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(sh.encoder.weight.grad, ref.encoder.weight.grad[lo:hi], **tol)
        torch.testing.assert_close(sh.encoder.bias.grad, ref.encoder.bias.grad[lo:hi], **tol)
        torch.testing.assert_close(sh.W_dec.grad, ref.W_dec.grad[lo:hi], **tol)
        torch.testing.assert_close(bdec_grad, ref.b_dec.grad, **tol)


def _run_sharded(fw, W, cfg, D, M, P, x, y, autocast):
    """W thread-ranks, each with its slice of the dictionary P; returns [(out, coder, lo, hi)]."""
    from sparsify_amd.sharded import ShardedSparseCoder
    results, errors = [None] * W, []

    def worker(r):
        try:
            fw.local.rank = r
            torch.cuda.set_device(0)
            own_stream(torch)
            lo, hi = fw.shard_range(M)
            sh = ShardedSparseCoder(D, cfg, device="cuda")
            with torch.no_grad():
                sh.encoder.weight.copy_(P["W_enc"][lo:hi]); sh.encoder.bias.copy_(P["b_enc"][lo:hi])
                sh.W_dec.copy_(P["W_dec"][lo:hi]); sh.b_dec.copy_(P["b_dec"])
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                out = type(sh).forward.__wrapped__(sh, x, y)
            out.fvu.backward()
            torch.cuda.synchronize()
            results[r] = (out, sh, lo, hi)
        except Exception as e:      # pragma: no cover
            errors.append(e)
            fw.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
    [t.start() for t in threads]
    [t.join(120) for t in threads]
    assert not errors, errors
    return results


@pytest.mark.parametrize("transcode", [True, False])
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("W", [2, 4])
def test_sharded_vs_oracle(device, monkeypatch, transcode, mode, W):
    """BASELINE configs[4] against the CPU oracle of the UNSHARDED reference coder (the contract of
    latent sharding): fp32 -- identical index sets, values / reconstruction /
    gradients to fp32 summation-order tolerance; bf16 autocast -- sets equal except at near ties
    of the k-th boundary (margin 2^-7, the north-star rule), everything else against the oracle
    evaluated on the selection the shards agreed on."""
    from test_hip_api import same_topk_sets
    from sparsify_amd import SparseCoderConfig, parallel

    N, D, M, k = 160, 64, 1024, 12
    fw = FakeWorld(W)
    for name in ("rank", "world_size", "shard_range", "all_reduce_", "gather_topk_candidates"):
        monkeypatch.setattr(parallel, name, getattr(fw, name))
    cfg = SparseCoderConfig(num_latents=M, k=k, transcode=transcode)
    P = O.init_params(D, M, transcode=transcode, seed=5)
    g = torch.Generator().manual_seed(6)
    P["b_enc"] = 0.05 * torch.randn(M, generator=g)
    P["b_dec"] = 0.3 * torch.randn(D, generator=g)
    if transcode:
        P["W_dec"] = torch.randn(M, D, generator=g) / D**0.5
    x = torch.randn(N, D, generator=g) + torch.randn(D, generator=g)
    y = x @ (torch.randn(D, D, generator=g) / D**0.5) if transcode else None
    bf16 = mode == "bf16"
    if bf16:
        x, y = x.bfloat16(), (None if y is None else y.bfloat16())

    results = _run_sharded(fw, W, cfg, D, M, {n: t.cuda() for n, t in P.items()}, x.cuda(),
                           None if y is None else y.cuda(), autocast=bf16)
    out0 = results[0][0]
    ocfg = O.OracleConfig(k=k, transcode=transcode)
    forced = {"top": (out0.latent_acts.detach().cpu(), out0.latent_indices.cpu())} if bf16 else None
    ref, G = O.forward_backward(P, x, y, ocfg, bf16=bf16, forced=forced)
    ref_idx = O.select_latents(ref["pre_acts"], k, "topk")[1]
    bdec_grad = sum(r[1].b_dec.grad for r in results).cpu()
    vtol = dict(rtol=2**-7, atol=1e-3) if bf16 else dict(rtol=2e-4, atol=2e-5)
    for out, sh, lo, hi in results:
        got_idx = out.latent_indices.cpu()
        assert torch.equal(got_idx, out0.latent_indices.cpu())          # every shard holds the same selection
        if bf16:
            same_topk_sets(got_idx, ref_idx, ref["pre_acts"], k, margin=2**-7)
        else:
            assert torch.equal(got_idx.sort(-1).values, ref_idx.sort(-1).values)
        order_g, order_r = got_idx.argsort(-1), ref["top_indices"].argsort(-1)
        torch.testing.assert_close(out.latent_acts.detach().float().cpu().gather(-1, order_g),
                                   ref["top_acts"].float().gather(-1, order_r), **vtol)
        torch.testing.assert_close(out.sae_out.detach().cpu(), ref["sae_out"], rtol=2e-4, atol=1e-4)
        torch.testing.assert_close(float(out.fvu.detach()), float(ref["fvu"]), rtol=2e-4, atol=1e-6)
        grads = {"W_enc": sh.encoder.weight.grad.cpu(), "b_enc": sh.encoder.bias.grad.cpu(), "W_dec": sh.W_dec.grad.cpu()}
        for name, gr in grads.items():
            want = G[name][lo:hi]
            rel = (gr - want).norm() / want.norm().clamp_min(1e-12)
            assert rel < (2e-3 if bf16 else 2e-4), f"{name}[{lo}:{hi}]: relative error {rel:.3e}"
        want = G["b_dec"]
        assert (bdec_grad - want).norm() / want.norm() < (2e-3 if bf16 else 2e-4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_full_width_merge_equals_unsharded_topk(device, dtype):
    """BASELINE configs[4] at its full width: 262144 latents as 8 shards of 32768, k = 192. The
    selection the shards agree on -- local top-k per shard, candidates concatenated shard-major as
    the all-gather delivers them, sharded.merge_candidates -- is the same SET, value for value, as
    sphip_topk over the whole 262144-wide row (ties included: both take the lowest indices)."""
    from sparsify_amd import ops
    from sparsify_amd.sharded import merge_candidates
    W, per, k = 8, 32768, 192
    N = 2048 if dtype == torch.bfloat16 else 512
    g = torch.Generator(device="cuda").manual_seed(0)
    pre = torch.randn(N, W * per, device="cuda", generator=g).clamp_min_(0).to(dtype)       # relu output: half zeros
    pre[7, :] = 0                                         # a row of ties only
    pre[8, 5::1000] = 3.0                                 # ties across shard boundaries above the rest
    want_v, want_i = ops.topk(pre, k)
    cv, ci = [], []
    for r in range(W):
        lv, li = ops.topk(pre[:, r * per:(r + 1) * per].contiguous(), k)
        cv.append(lv); ci.append(li + r * per)
    cand_v, cand_i = torch.cat(cv, 1), torch.cat(ci, 1)
    for r in (0, 3, 7):
        mv, gidx, local = merge_candidates(cand_v, cand_i, k, r * per, (r + 1) * per)
        order_a, order_b = gidx.argsort(-1), want_i.argsort(-1)
        assert torch.equal(gidx.gather(-1, order_a), want_i.gather(-1, order_b))
        assert torch.equal(mv.gather(-1, order_a), want_v.gather(-1, order_b))
        mine = (gidx >= r * per) & (gidx < (r + 1) * per)
        assert torch.equal(local[mine], gidx[mine] - r * per) and (local[~mine] == -1).all()
    assert torch.equal(want_i[7].sort().values.cpu(), torch.arange(k))


def test_synthetic_transcoder_run_follows_the_oracle_into_its_collapse(device):
    """The bench's synthetic transcoder workload (BASELINE configs[4] inputs: y = x R + noise,
    zero-initialised decoder, plain SignSGD at the reference's learning rate) at a reduced shape.
    On these inputs the REFERENCE's own arithmetic (the pinned CPU oracle) drives the dictionary
    onto a few hundred latents within ten steps and the FVU above 1: sign steps on a decoder that
    starts at zero overshoot. The HIP Trainer follows the oracle step for step while the run is
    well conditioned (first eight steps: FVU within 0.5 %, number of latents that fired within 2 %)
    and shows the same collapse afterwards (the two runs decorrelate through bf16 tie picks)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchmarks"))
    import collapse_probe
    rows = collapse_probe.run(D=256, M=16384, k=16, N=2048, steps=20, transcode=True, verbose=False)
    for step, got, want, got_active, want_active in rows[:8]:
        assert abs(got - want) <= 5e-3 * want, rows[step]
        assert abs(got_active - want_active) <= 0.02 * want_active + 2, rows[step]
    late = rows[12:]
    got_fvu = sum(r[1] for r in late) / len(late)
    want_fvu = sum(r[2] for r in late) / len(late)
    assert want_fvu > 1.0 and got_fvu > 1.0, (got_fvu, want_fvu)            # the reference's arithmetic diverges too
    assert abs(got_fvu - want_fvu) <= 0.10 * want_fvu, (got_fvu, want_fvu)
    first_active = rows[0][4]
    for step, _, _, got_active, want_active in late:
        assert want_active < 0.1 * first_active and got_active < 0.1 * first_active, rows[step]
        assert abs(got_active - want_active) <= 0.3 * want_active, rows[step]


@pytest.mark.parametrize("mark", [-1, "past_M"])
def test_inactive_slots_are_skipped(device, mark):
    """idx = -1 (or any index outside [0, M), sparsify_hip.h) marks a slot owned by another shard:
    no contribution anywhere, no read outside the table."""
    from sparsify_amd import ops
    torch.manual_seed(0)
    N, k, M, D = 40, 8, 64, 128
    idx = torch.stack([torch.randperm(M)[:k] for _ in range(N)])
    drop = torch.rand(N, k) < 0.5
    off = torch.full_like(idx, -1) if mark == -1 else M + torch.randint(0, 1 << 40, idx.shape)
    idx_m = torch.where(drop, off, idx).cuda()
    coeff = torch.randn(N, k)
    table = torch.randn(M, D)
    g = torch.randn(N, D)
    got = ops.gather_rows_weighted(idx_m, coeff.cuda(), table.cuda()).cpu()
    ref = ((coeff * ~drop)[:, :, None] * table[idx]).sum(1)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)
    seg = ops.build_latent_segments(idx_m, M)
    assert int(seg.seg_begin[M]) == int((~drop).sum())
    gW, gb, dc = ops.segmented_weight_grad(seg, coeff.cuda(), g.cuda(), table.cuda(), want_bias=True, want_dcoeff=True)
    ref_gW = torch.zeros(M, D)
    ref_gW.index_add_(0, idx.flatten(), ((coeff * ~drop)[:, :, None] * g[:, None, :]).reshape(-1, D))
    torch.testing.assert_close(gW.cpu(), ref_gW, rtol=1e-4, atol=1e-5)
    ref_dc = torch.einsum("nd,nsd->ns", g, table[idx]) * ~drop
    torch.testing.assert_close(dc.cpu(), ref_dc, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("W", [2, 4])
@pytest.mark.parametrize("variant", ["auxk", "auxk_few_dead", "multi_topk", "skip", "all"])
def test_sharded_auxk_multitopk_skip_vs_oracle(device, monkeypatch, variant, W):
    """The other terms of SparseCoder.forward on the latent-sharded coder (sparse_coder.py:219-222,
    230-251, 256-262; no reference code for the sharding itself), fp32, against the CPU oracle of
    the unsharded reference coder: identical selections (main, AuxK, multi-top-k), losses and
    every gradient -- W_skip's replicated, b_dec's summed over the shards."""
    from sparsify_amd import SparseCoderConfig, parallel
    from sparsify_amd.sharded import ShardedSparseCoder

    N, D, M, k = 160, 64, 1024, 12
    alpha = 1 / 32
    auxk = variant in ("auxk", "auxk_few_dead", "all")
    multi = variant in ("multi_topk", "all")
    skip = variant in ("skip", "all")
    transcode = skip                                   # the reference ties the skip connection to transcoders
    fw = FakeWorld(W)
    for name in ("rank", "world_size", "shard_range", "all_reduce_", "gather_topk_candidates"):
        monkeypatch.setattr(parallel, name, getattr(fw, name))
    cfg = SparseCoderConfig(num_latents=M, k=k, transcode=transcode, multi_topk=multi, skip_connection=skip)
    P = O.init_params(D, M, transcode=transcode, seed=7)
    g = torch.Generator().manual_seed(8)
    P["b_enc"] = 0.05 * torch.randn(M, generator=g)
    P["b_dec"] = 0.3 * torch.randn(D, generator=g)
    if transcode:
        P["W_dec"] = torch.randn(M, D, generator=g) / D**0.5
    if skip:
        P["W_skip"] = torch.randn(D, D, generator=g) * 0.05
    x = torch.randn(N, D, generator=g) + torch.randn(D, generator=g)
    y = x @ (torch.randn(D, D, generator=g) / D**0.5) if transcode else None
    dead = None
    if auxk:
        dead = torch.zeros(M, dtype=torch.bool)
        if variant == "auxk_few_dead":
            dead[torch.randperm(M, generator=g)[:20]] = True          # fewer than k_aux = D / 2 = 32
            dead[M // W - 1] = dead[M // W] = True                    # on both sides of a shard boundary
        else:
            dead[torch.rand(M, generator=g) < 0.3] = True
    ocfg = O.OracleConfig(k=k, transcode=transcode, multi_topk=multi)
    ref, G = O.forward_backward(P, x, y, ocfg, dead_mask=dead, auxk_alpha=alpha if auxk else 0.0)

    results, errors = [None] * W, []
    Pg = {n: t.cuda() for n, t in P.items()}
    xg, yg, dg = x.cuda(), (None if y is None else y.cuda()), (None if dead is None else dead.cuda())

    def worker(r):
        try:
            fw.local.rank = r
            torch.cuda.set_device(0)
            own_stream(torch)
            lo, hi = fw.shard_range(M)
            sh = ShardedSparseCoder(D, cfg, device="cuda")
            with torch.no_grad():
                sh.encoder.weight.copy_(Pg["W_enc"][lo:hi]); sh.encoder.bias.copy_(Pg["b_enc"][lo:hi])
                sh.W_dec.copy_(Pg["W_dec"][lo:hi]); sh.b_dec.copy_(Pg["b_dec"])
                if skip:
                    sh.W_skip.copy_(Pg["W_skip"])
            with torch.autocast("cuda", enabled=False):
                out = type(sh).forward.__wrapped__(sh, xg, yg, dead_mask=dg)
            (out.fvu + alpha * out.auxk_loss + out.multi_topk_fvu / 8).backward()
            torch.cuda.synchronize()
            results[r] = (out, sh, lo, hi)
        except Exception as e:      # pragma: no cover
            errors.append(e)
            fw.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
    [t.start() for t in threads]
    [t.join(120) for t in threads]
    assert not errors, errors

    bdec_grad = sum(r[1].b_dec.grad for r in results).cpu()
    for out, sh, lo, hi in results:
        got_idx = out.latent_indices.cpu()
        assert torch.equal(got_idx.sort(-1).values, ref["latent_indices"].sort(-1).values)
        order_g, order_r = got_idx.argsort(-1), ref["latent_indices"].argsort(-1)
        torch.testing.assert_close(out.latent_acts.detach().cpu().gather(-1, order_g),
                                   ref["latent_acts"].gather(-1, order_r), rtol=2e-4, atol=2e-5)
        torch.testing.assert_close(out.sae_out.detach().cpu(), ref["sae_out"], rtol=2e-4, atol=1e-4)
        for key in ("fvu", "auxk_loss", "multi_topk_fvu"):
            torch.testing.assert_close(float(getattr(out, key).detach()), float(ref[key]), rtol=2e-4, atol=1e-6)
        if auxk:
            assert float(ref["auxk_loss"]) > 0
        grads = {"W_enc": sh.encoder.weight.grad.cpu(), "b_enc": sh.encoder.bias.grad.cpu(), "W_dec": sh.W_dec.grad.cpu()}
        for name, gr in grads.items():
            want = G[name][lo:hi]
            rel = (gr - want).norm() / want.norm().clamp_min(1e-12)
            assert rel < 2e-4, f"{name}[{lo}:{hi}]: relative error {rel:.3e}"
        if skip:
            rel = (sh.W_skip.grad.cpu() - G["W_skip"]).norm() / G["W_skip"].norm()
            assert rel < 2e-4, f"W_skip: relative error {rel:.3e}"
    assert (bdec_grad - G["b_dec"]).norm() / G["b_dec"].norm() < 2e-4
