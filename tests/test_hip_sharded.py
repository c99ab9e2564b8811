"""Latent-dim sharding on one GPU: two "ranks" run as two threads of this process with an
in-process stand-in for the two collectives (all-gather of candidates, all-reduce of partial
reconstructions); everything else is the real HIP path. Contract: identical
to the unsharded SparseCoder."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


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


def test_inactive_slots_are_skipped(device):
    """idx = -1 marks a slot owned by another shard: no contribution anywhere."""
    from sparsify_amd import ops
    torch.manual_seed(0)
    N, k, M, D = 40, 8, 64, 128
    idx = torch.stack([torch.randperm(M)[:k] for _ in range(N)])
    drop = torch.rand(N, k) < 0.5
    idx_m = torch.where(drop, torch.full_like(idx, -1), idx).cuda()
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
