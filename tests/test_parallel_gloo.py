"""World-size-2 tests of the multi-GPU plumbing (sparsify_amd/parallel.py) on CPU with gloo.
They cover what the N>1 bench path and the Trainer use: reductions with the reference's op
names, the distribute_modules plan and activation gathers, the once-per-step async gradient
averaging, and the candidate exchange of the latent-sharded top-k."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def launch(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_run, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _reductions(rank, world):
    from sparsify_amd import parallel as P
    x = torch.full((4,), float(rank + 1))
    mean = P.all_reduce_(x.clone(), "mean")
    tot = P.all_reduce_(x.clone(), "sum")
    fired = torch.zeros(6, dtype=torch.bool)
    fired[rank] = True
    fired[5] = rank == 1
    P.all_reduce_(fired, "max")
    rows = torch.arange(6, dtype=torch.float32).reshape(2, 3) + 10 * rank
    gathered = P.all_gather_rows(rows)
    mask = P.all_gather_rows(torch.tensor([[True, False]]) if rank == 0 else torch.tensor([[False, False]]))
    with pytest.raises(ValueError):
        P.all_reduce_(x, "median")
    # gather-to-owner (distribute_modules: only the hookpoint's owner receives the batches)
    to1 = P.gather_rows(rows, 1)
    mask_to0 = P.gather_rows(torch.tensor([[rank == 0, False]]), 0)
    return dict(mean=mean, tot=tot, fired=fired, gathered=gathered, mask=mask, lo_hi=P.shard_range(8), to1=to1,
                mask_to0=mask_to0)


def test_reductions_and_gathers():
    out = launch(_reductions)
    for r in out:
        assert torch.equal(r["mean"], torch.full((4,), 1.5))
        assert torch.equal(r["tot"], torch.full((4,), 3.0))
        assert r["fired"].tolist() == [True, True, False, False, False, True]
        assert torch.equal(r["gathered"], torch.tensor([[0., 1, 2], [3, 4, 5], [10, 11, 12], [13, 14, 15]]))
        assert r["mask"].dtype == torch.bool and r["mask"].tolist() == [[True, False], [False, False]]
    assert out[0]["lo_hi"] == (0, 4) and out[1]["lo_hi"] == (4, 8)
    assert out[0]["to1"] is None and torch.equal(out[1]["to1"], out[1]["gathered"])
    assert out[1]["mask_to0"] is None and out[0]["mask_to0"].dtype == torch.bool
    assert out[0]["mask_to0"].tolist() == [[True, False], [False, False]]


def test_module_plan():
    from sparsify_amd import parallel as P
    hooks = [f"layers.{i}" for i in range(8)]
    assert P.module_plan(hooks, 4) == [hooks[0:2], hooks[2:4], hooks[4:6], hooks[6:8]]   # trainer.py:641-644
    with pytest.raises(AssertionError):
        P.module_plan(hooks, 3)
    plan = P.module_plan(hooks, 4)
    assert [P.module_owner(plan, h) for h in hooks] == [0, 0, 1, 1, 2, 2, 3, 3]


def _grad_reducer(rank, world):
    from sparsify_amd import parallel as P
    torch.manual_seed(0)
    model = torch.nn.Linear(5, 3)
    P.broadcast_parameters(model.parameters())
    acc = 2
    red = P.GradReducer(model.parameters())
    g = torch.Generator().manual_seed(100)
    data = torch.randn(world * acc, 4, 5, generator=g)
    for micro in range(acc):
        x = data[rank * acc + micro]
        (model(x).pow(2).mean() / acc).backward()      # accumulates locally, no collective
    red.finish()
    return [p.grad.clone() for p in model.parameters()], data


def _grad_reducer_async(rank, world):
    """The same with the weight counted as a large gradient: its all-reduce leaves from the
    post-accumulate hook of the LAST micro-step's backward (armed), the bias goes into the flat
    bucket of finish()."""
    from sparsify_amd import parallel as P
    torch.manual_seed(0)
    model = torch.nn.Linear(5, 3)
    P.broadcast_parameters(model.parameters())
    acc = 2
    red = P.GradReducer(model.parameters(), bucket_bytes=40)        # weight 60 B: large; bias 12 B: bucket
    g = torch.Generator().manual_seed(100)
    data = torch.randn(world * acc, 4, 5, generator=g)
    launched = []
    for micro in range(acc):
        red.armed = micro == acc - 1
        x = data[rank * acc + micro]
        (model(x).pow(2).mean() / acc).backward()
        launched.append(len(red._inflight))
    red.finish()
    assert launched == [0, 1] and not red._inflight and not red.armed
    return [p.grad.clone() for p in model.parameters()], data


@pytest.mark.parametrize("fn", [_grad_reducer, _grad_reducer_async])
def test_grad_reducer_matches_global_average(fn):
    out = launch(fn)
    (g0, data), (g1, _) = out
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)               # replicas hold identical gradients
    torch.manual_seed(0)
    model = torch.nn.Linear(5, 3)
    loss = sum(model(data[i]).pow(2).mean() / 2 for i in range(4)) / 2
    loss.backward()
    for a, p in zip(g0, model.parameters()):
        torch.testing.assert_close(a, p.grad, rtol=1e-6, atol=1e-7)


def _sharded_topk(rank, world):
    from sparsify_amd import parallel as P
    torch.manual_seed(7)
    N, M, k = 16, 64, 5
    pre = torch.rand(N, M)                         # same on both ranks (replicated tokens)
    lo, hi = P.shard_range(M)
    lv, li = pre[:, lo:hi].topk(k, dim=-1)
    vals, gidx = P.gather_topk_candidates(lv, li, lo)
    assert vals.shape == (N, world * k)
    mv, mpos = vals.topk(k, dim=-1)                # merge (ops.topk on the GPU path)
    assert gidx.shape == (N, world, k)             # the (token, shard, slot) view of the gathered 8-byte records
    midx = P.candidate_indices(gidx, mpos)
    ref = pre.topk(k, dim=-1)
    return torch.equal(mv, ref.values) and torch.equal(midx.sort(-1).values, ref.indices.sort(-1).values)


@pytest.mark.parametrize("world", [2, 4])
def test_latent_sharded_topk_merge_equals_global_topk(world):
    assert all(launch(_sharded_topk, world=world))


def _sign_exchange(rank, world):
    """parallel.SignExchange with torch stand-ins for the two HIP kernels: the update must be
    the SignSGD step on the rank-averaged gradient (plain and decoder-projected)."""
    from sparsify_amd import parallel as P
    torch.manual_seed(0)
    M, D, lr = 8, 131072, 1e-2                      # >= MIN_NUMEL, M divisible by the world size
    W = torch.randn(M, D)
    W /= W.norm(dim=1, keepdim=True)
    grads = [torch.randn(M, D, generator=torch.Generator().manual_seed(10 + r)) for r in range(world)]
    mean = sum(grads) / world
    assert P.SignExchange.eligible(W) and not P.SignExchange.eligible(torch.zeros(7, D)) \
        and not P.SignExchange.eligible(torch.zeros(D))
    # plain
    p = W.clone()
    P.SignExchange.step(grads[rank], lambda s, lo, hi: torch.sign(s).to(torch.int8),
                        lambda signs, lo, hi: p[lo:hi].sub_(lr * signs.float()))
    # (beyond two ranks the reduction's summation order may differ from sum(): a ~0 mean can flip its sign)
    ok_plain = torch.equal(p, W - lr * torch.sign(mean)) if world == 2 else \
        float((p != W - lr * torch.sign(mean)).float().mean()) < 1e-5
    # the same update exchanged in row blocks (what the overlapped optimizer step does)
    p2 = W.clone()
    order = []
    P.SignExchange.step(grads[rank], lambda s, lo, hi: torch.sign(s).to(torch.int8),
                        lambda signs, lo, hi: p2[lo:hi].sub_(lr * signs.float()), blocks=2,
                        on_block=lambda lo, hi: order.append((lo, hi)))
    ok_plain = ok_plain and torch.equal(p2, p) and order == [(0, M // 2), (M // 2, M)]
    ok_plain = ok_plain and P.SignExchange.row_blocks(M, 8) == M // world and P.SignExchange.row_blocks(M, 8, 256) == 1
    # decoder: sign of the projected gradient, taken on this rank's row block only
    q = W.clone()
    seen = []

    def proj_sign(shard, lo, hi):
        seen.append((lo, hi))
        w = W[lo:hi]
        return torch.sign(shard - (shard * w).sum(1, keepdim=True) * w).to(torch.int8)

    P.SignExchange.step(grads[rank], proj_sign, lambda signs, lo, hi: q[lo:hi].sub_(lr * signs.float()))
    ref = W - lr * torch.sign(mean - (mean * W).sum(1, keepdim=True) * W)
    frac_bad = float((q != ref).float().mean())      # summation order of the reduce may flip a ~0 sign
    return ok_plain, frac_bad, seen[0] == (rank * M // world, (rank + 1) * M // world)


@pytest.mark.parametrize("world", [2, 4])
def test_sign_exchange_equals_step_on_mean_gradient(world):
    for ok_plain, frac_bad, block_ok in launch(_sign_exchange, world=world):
        assert ok_plain and block_ok
        assert frac_bad < 1e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_candidate_records_round_trip(dtype):
    """The latent-shard exchange ships one 8-byte record per candidate (value bits | global index):
    packing and unpacking is exact for every value (negative, -inf for masked-out latents, zeros) and for
    indices up to the widest dictionary."""
    from sparsify_amd import parallel as P
    g = torch.Generator().manual_seed(0)
    vals = torch.randn(37, 11, generator=g).to(dtype)
    vals[0, :3] = torch.tensor([float("-inf"), 0.0, -0.0]).to(dtype)
    idx = torch.randint(0, 2**31 - 1, (37, 11), generator=g)
    idx[1, :2] = torch.tensor([0, 2**32 - 1])
    rec = P.pack_candidates(vals, idx)
    assert rec.dtype == torch.int64 and rec.element_size() == 8 and rec.shape == vals.shape
    v2, i2 = P.unpack_candidates(rec, dtype)
    assert v2.dtype == dtype and torch.equal(v2.view(torch.int16 if dtype == torch.bfloat16 else torch.int32),
                                             vals.view(torch.int16 if dtype == torch.bfloat16 else torch.int32))
    assert torch.equal(i2, idx)
    pos = torch.randint(0, 2 * 11, (37, 5), generator=g)
    flat = torch.cat([idx, idx + 1], 1)                                   # two "shards", shard-major per token
    view = torch.stack([idx, idx + 1], 0).permute(1, 0, 2)                # the gathered (W, N, k) records seen as (N, W, k)
    assert not view.is_contiguous() and torch.equal(P.candidate_indices(view, pos), flat.gather(1, pos))
