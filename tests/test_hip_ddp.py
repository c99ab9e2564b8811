"""Data-parallel Trainer with two ranks on ONE GPU (gloo rendezvous, device tensors staged
through the host): the N > 1 code of bench.py / Trainer runs end to end with the HIP kernels,
replicas stay identical, and the sign exchange (reduce-scatter + all-gather of int8 signs),
on the compute stream or overlapped with the next forward, gives bit for bit the parameters
of the plain gradient all-reduce. RCCL itself ("nccl") runs here as a one-rank group, which
exercises its stream semantics under the overlapped exchange; more ranks need a multi-GPU
node (the driver's scaling bench)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


MODES = ("allreduce", "exchange", "overlap")
OPTIMIZERS = ("signsgd", "signum")       # plain SignSGD and the reference's default (schedule-free)


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sparsify_amd import SparseCoderConfig, TrainConfig
        from sparsify_amd.trainer import SyntheticActivations, Trainer
        D, M, k, N, steps = 256, 4096, 16, 512, 4
        out = {}
        for mode in [(o, m) for o in OPTIMIZERS for m in MODES]:
            optimizer, how = mode
            exchange, overlap = how != "allreduce", how == "overlap"
            cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k), optimizer=optimizer, hookpoints=["h0"],
                              save_dir="", save_every=10**9, wandb_log_frequency=1, exchange_signs=exchange,
                              overlap_exchange=overlap)
            src = SyntheticActivations(["h0"], D, N, steps, device="cuda", rank=rank)
            torch.manual_seed(0)
            tr = Trainer(cfg, src, device="cuda")
            for s in range(steps):
                tr.train_micro_batch(src.batch(s))
            sae = tr.saes["h0"]
            torch.cuda.synchronize()
            out[mode] = {n: p.detach().cpu() for n, p in sae.named_parameters()}
            out[mode]["fvu"] = tr.last_logged["fvu/h0"]
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def test_two_rank_trainer_sign_exchange_equals_all_reduce(device):
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=90)
    for p in procs:
        if p.is_alive():
            p.kill()
            pytest.fail("a rank did not finish (collective mismatch?)")
        assert p.exitcode == 0
    r0, r1 = ret[0], ret[1]
    for mode in r0.keys():
        for name in r0[mode]:
            if name == "fvu":
                assert r0[mode]["fvu"] == pytest.approx(r1[mode]["fvu"], rel=1e-6)     # logged mean over ranks
            else:
                assert torch.equal(r0[mode][name], r1[mode][name]), f"replicas diverged: {name} ({mode})"
    # two ranks: a + b is order independent, so every exchange sees the same mean gradient
    for opt in OPTIMIZERS:
        for how in MODES[1:]:
            for name in r0[(opt, "allreduce")]:
                if name != "fvu":
                    assert torch.equal(r0[(opt, "allreduce")][name], r0[(opt, how)][name]), \
                        f"{opt}/{how} != all-reduce: {name}"
        assert r0[(opt, "overlap")]["fvu"] < 1.0
    assert not torch.equal(r0[("signsgd", "overlap")]["W_dec"], r0[("signum", "overlap")]["W_dec"])


def test_eight_rank_replicas_stay_identical(device):
    """The same run with eight ranks on one GPU (gloo): the row-block and shard arithmetic of the sign
    exchange at the world size of a full node (131072-row headline matrices split 8 x 8; here 4096 rows,
    where the overlapped form falls back to two row blocks), early exchange of the decoder rows, the
    bucketed bias reduction. Every rank must hold bit-identical parameters in every mode; the three
    modes differ from one another only by the order in which eight summands are added."""
    world = 8
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    for p in procs:
        if p.is_alive():
            p.kill()
            pytest.fail("a rank did not finish (collective mismatch?)")
        assert p.exitcode == 0
    r0 = ret[0]
    for r in range(1, world):
        for mode in r0.keys():
            for name in r0[mode]:
                if name == "fvu":
                    assert r0[mode]["fvu"] == pytest.approx(ret[r][mode]["fvu"], rel=1e-6)
                else:
                    assert torch.equal(r0[mode][name], ret[r][mode][name]), f"rank {r} diverged: {name} ({mode})"
    for opt in OPTIMIZERS:
        for how in MODES:
            assert r0[(opt, how)]["fvu"] < 1.0
        # same update rule on the same mean gradient up to summation order: the runs stay close
        a, b = r0[(opt, "allreduce")]["W_dec"], r0[(opt, "overlap")]["W_dec"]
        assert ((a - b).abs() > 1e-6).float().mean() < 0.02


def _rccl_one_rank(port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from sparsify_amd import SparseCoderConfig, TrainConfig, ops, parallel
        from sparsify_amd.trainer import SyntheticActivations, Trainer
        D, M, k, N, steps = 1024, 16384, 32, 4096, 6
        out = {}
        for mode in [(o, m) for o in OPTIMIZERS for m in MODES]:
            parallel.SignExchange.FORCE = mode[1] != "allreduce"    # one rank: no exchange unless forced
            cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k), optimizer=mode[0], hookpoints=["h0"],
                              save_dir="", save_every=10**9, wandb_log_frequency=1, exchange_signs=True,
                              overlap_exchange=mode[1] == "overlap")
            src = SyntheticActivations(["h0"], D, N, steps, device="cuda")
            torch.manual_seed(0)
            tr = Trainer(cfg, src, device="cuda")
            ops.timer.enabled = True
            ops.timer.reset()
            for s in range(steps):
                tr.train_micro_batch(src.batch(s))
            parallel.wait_for_updates()
            torch.cuda.synchronize()
            ops.timer.enabled = False
            out[mode] = {n: p.detach().cpu() for n, p in tr.saes["h0"].named_parameters()}
            out[mode]["fvu"] = tr.last_logged["fvu/h0"]
            out[mode]["block_launches"] = len(ops.timer.events.get("sphip_encoder_preacts_block", []))
            out[mode]["early"] = getattr(tr.optimizers[0], "early_exchanges", 0)
        ret[0] = out
    finally:
        dist.destroy_process_group()


def test_overlapped_exchange_over_rccl_one_rank(device):
    """The overlapped exchange on RCCL's own streams (a one-rank group: its reduce-scatter and
    all-gather are copies, the stream and event choreography is the real one): six steps give
    bit for bit the parameters of the plain single-stream SignSGD step, and from the second
    step on the encoder GEMM ran block by block behind the arriving updates."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    proc = ctx.Process(target=_rccl_one_rank, args=(_free_port(), ret))
    proc.start()
    proc.join(timeout=100)
    if proc.is_alive():
        proc.kill()
        pytest.fail("the rank did not finish")
    assert proc.exitcode == 0
    out = ret[0]
    for opt in OPTIMIZERS:
        plain = out[(opt, "allreduce")]
        for how in MODES[1:]:
            for name, ref in plain.items():
                if name not in ("fvu", "block_launches", "early"):
                    assert torch.equal(ref, out[(opt, how)][name]), f"{opt}/{how}: {name} differs from the plain step"
            assert out[(opt, how)]["fvu"] == plain["fvu"]
        assert plain["block_launches"] == 0 and out[(opt, "exchange")]["block_launches"] == 0
        assert out[(opt, "overlap")]["block_launches"] == 5 * 8    # steps 2..6, SignSGD.OVERLAP_BLOCKS each
        # plain SignSGD: the decoder matrix's first rows leave from a hook inside every backward
        assert out[(opt, "overlap")]["early"] == (6 if opt == "signsgd" else 0)
        assert out[(opt, "exchange")]["early"] == 0


def _shard_worker(rank, world, port, ret, save_dir, variants=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = _train_small(shard=True, save_dir=save_dir, variants=variants)
    finally:
        dist.destroy_process_group()


def _train_small(shard, save_dir="", variants=False):
    """variants: AuxK (a latent that did not fire in the last step counts as dead: from the second step on) and multi-top-k."""
    from sparsify_amd import SparseCoderConfig, TrainConfig
    from sparsify_amd.trainer import SyntheticActivations, Trainer
    D, M, k, N, steps = 128, 2048, 16, 512, 8 if variants else 5
    extra = dict(auxk_alpha=1 / 32, dead_feature_threshold=100) if variants else {}
    cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k, multi_topk=variants), optimizer="signsgd", hookpoints=["h0"],
                      save_dir=save_dir, run_name="run", save_every=10**9, wandb_log_frequency=1, shard_latents=shard, **extra)
    src = SyntheticActivations(["h0"], D, N, steps, device="cuda", rank=0)      # every rank sees the same tokens
    torch.manual_seed(0)
    tr = Trainer(cfg, src, device="cuda")
    fvus, aux = [], []
    for s in range(steps):
        tr.train_micro_batch(src.batch(s))
        fvus.append(tr.last_logged["fvu/h0"])
        if variants:
            aux.append((tr.last_logged["auxk/h0"], tr.last_logged["multi_topk_fvu/h0"]))
    if save_dir:
        tr.save()
        # save_best is a collective for latent shards too (every rank joins the gathers); it
        # deadlocked when only rank 0 entered the checkpoint
        tr.save_best({"h0": 0.0})
    torch.cuda.synchronize()
    sae = tr.saes["h0"]
    return {"fvu": fvus, "aux": aux, "rows": tuple(sae.encoder.weight.shape), "W_dec": sae.W_dec.detach().cpu(),
            "W_enc": sae.encoder.weight.detach().cpu(), "b_dec": sae.b_dec.detach().cpu()}


@pytest.mark.parametrize("variants", [False, True])
def test_latent_sharded_trainer_matches_unsharded(device, tmp_path, variants):
    """TrainConfig.shard_latents with two ranks on one GPU: each rank trains half of the latents of
    one dictionary on the same tokens. The FVU curve and the parameters follow the unsharded
    Trainer (the reconstruction is a sum of two partial sums, so a gradient at the rounding level
    may take the other sign), and the checkpoint rank 0 writes from the gathered shards is an
    ordinary unsharded one. variants: with the AuxK and multi-top-k terms (their losses follow too)."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, ret, str(tmp_path), variants)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=90)
    for p in procs:
        if p.is_alive():
            p.kill()
            pytest.fail("a rank did not finish (collective mismatch?)")
        assert p.exitcode == 0
    ref = _train_small(shard=False, variants=variants)
    r0, r1 = ret[0], ret[1]
    assert r0["rows"] == (1024, 128) and r1["rows"] == (1024, 128)
    assert r0["fvu"] == r1["fvu"]                                   # replicated loss, identical on the ranks
    for a, b in zip(r0["fvu"], ref["fvu"]):
        assert abs(a - b) <= 2e-3 * b, (r0["fvu"], ref["fvu"])
    if variants:
        assert r0["aux"] == r1["aux"]
        assert ref["aux"][-1][0] > 0 and ref["aux"][0][0] == 0      # AuxK switches on when latents have died
        for (a_aux, a_multi), (b_aux, b_multi) in zip(r0["aux"], ref["aux"]):
            assert abs(a_aux - b_aux) <= 5e-3 * max(b_aux, 1e-6), (r0["aux"], ref["aux"])
            assert abs(a_multi - b_multi) <= 2e-3 * b_multi, (r0["aux"], ref["aux"])
    assert ref["fvu"][-1] < 1.0
    assert torch.equal(r0["b_dec"], r1["b_dec"])                    # the one shared parameter stays replicated
    for name in ("W_dec", "W_enc"):
        whole = torch.cat([r0[name], r1[name]])
        assert (whole != ref[name]).float().mean() < 0.01, name
    from sparsify_amd import SparseCoder
    best = SparseCoder.load_from_disk(tmp_path / "run" / "best" / "h0" / "h0")
    assert torch.equal(best.W_dec.detach(), torch.cat([r0["W_dec"], r1["W_dec"]]))
    back = SparseCoder.load_from_disk(tmp_path / "run" / "h0")
    assert back.num_latents == 2048 and back.d_in == 128
    assert torch.equal(back.W_dec.detach(), torch.cat([r0["W_dec"], r1["W_dec"]]))
    assert torch.equal(back.encoder.weight.detach(), torch.cat([r0["W_enc"], r1["W_enc"]]))


def _sign_exchange_worker(rank, world, port, backend, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sparsify_amd import SignSGD, parallel
        parallel.SignExchange.FORCE = True
        ret[(backend, rank)] = _sign_exchange_steps(SignSGD, parallel, exchange=True)
    finally:
        dist.destroy_process_group()


def _sign_exchange_steps(SignSGD, parallel, exchange):
    """Two optimizer steps on an encoder-like and a decoder-like matrix with the same gradients on
    every rank. The gradients are bf16-representable, so the sum of eight equal copies and its
    division by eight are exact: the rank-averaged gradient is the gradient itself, whatever the
    order a backend adds in."""
    M, D, lr = 16384, 256, 3e-3
    gen = torch.Generator().manual_seed(5)
    W_enc = torch.nn.Parameter(torch.randn(M, D, generator=gen).cuda())
    W_dec = torch.randn(M, D, generator=gen)
    W_dec = torch.nn.Parameter((W_dec / W_dec.norm(dim=1, keepdim=True)).cuda())
    opt = SignSGD([dict(params=[W_dec], project_rows=True, renorm_rows=True), dict(params=[W_enc])], lr=lr,
                  exchange_signs=exchange, overlap=exchange)
    blocks = []
    for _ in range(2):
        W_enc.grad = torch.randn(M, D, generator=gen).bfloat16().float().cuda()
        W_dec.grad = torch.randn(M, D, generator=gen).bfloat16().float().cuda()
        opt.step()
        blocks.append(len(parallel.pending_blocks(W_enc) or []))
        parallel.wait_for_updates()
        torch.cuda.synchronize()
    return {"W_enc": W_enc.detach().cpu(), "W_dec": W_dec.detach().cpu(), "blocks": blocks,
            "reduces": bool(opt.reduces(W_enc)) and bool(opt.reduces(W_dec))}


def test_sign_exchange_rccl_one_rank_equals_eight_gloo_ranks(device):
    """parallel.SignExchange on the same tensors over a real RCCL group of one rank (the overlapped
    form: exchange stream, eight row blocks for the encoder matrix) and over eight gloo ranks (the
    shard arithmetic of a full node: every rank reduces 1/8 of every row block, takes its signs, all
    ranks apply all signs): bit-equal parameters on every rank, equal to the plain one-GPU SignSGD step
    (projected gradient, sign step, re-normalisation for the decoder matrix)."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    for backend, world in (("nccl", 1), ("gloo", 8)):
        port = _free_port()
        procs = [ctx.Process(target=_sign_exchange_worker, args=(r, world, port, backend, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=200)
        for p in procs:
            if p.is_alive():
                p.kill()
                pytest.fail(f"a {backend} rank did not finish")
            assert p.exitcode == 0
    from sparsify_amd import SignSGD, parallel
    plain = _sign_exchange_steps(SignSGD, parallel, exchange=False)
    one = ret[("nccl", 0)]
    assert one["reduces"] and one["blocks"] == [8, 8] and not plain["reduces"]
    for name in ("W_enc", "W_dec"):
        assert torch.equal(one[name], plain[name]), f"{name}: one-rank RCCL exchange != plain step"
        for r in range(8):
            got = ret[("gloo", r)]
            assert got["reduces"] and torch.equal(got[name], one[name]), f"{name}: gloo rank {r} != RCCL one-rank"
    assert ret[("gloo", 0)]["blocks"] == [8, 8]
