"""Muon on the HIP path (sparsify_amd/muon.py, csrc/muon.hip, the axpy epilogue of the bf16 GEMM
kernels) against the reference's own outputs (tests/golden/muon.npz, written by
sparsify/muon.py through tools/make_golden.py) and against the CPU oracle."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden
from oracle import sae_oracle as O

pytestmark = pytest.mark.gpu


def test_newton_schulz_vs_reference(device):
    """quintic_newtonschulz (muon.py:11-40) of a wide and of a tall matrix."""
    from sparsify_amd.muon import quintic_newtonschulz
    g = load_golden("muon")
    for name in ("A", "B"):
        got = quintic_newtonschulz(g[name].cuda(), 5)
        assert got.dtype == torch.bfloat16 and got.shape == g[name].shape
        torch.testing.assert_close(got.float().cpu(), g["ns" + name], rtol=2e-2, atol=2e-2)


def test_muon_steps_vs_reference(device):
    """Three single-process Muon steps (weight decay, momentum, Nesterov, orthogonalisation,
    0.2 sqrt(max(shape)) scale; muon.py:115-181) on the reference's gradients.

    Tolerance: the orthogonalised update is five bf16 Newton-Schulz iterations; the MFMA kernels sum
    their fp32 products in another order than the CPU's bf16 matmul, single bf16 roundings flip and
    propagate, which is why the reference's own outputs are matched to 2e-2 in nsA / nsB above (the
    CPU oracle meets 1e-5 on p1 / p3 only because it calls the very matmul the reference called).
    A parameter moves by step = lr 0.2 sqrt(max(shape)) times the update, so p is held to
    2e-2 step per optimizer step taken, and on average to a quarter of that."""
    from sparsify_amd.muon import Muon
    g = load_golden("muon")
    p = torch.nn.Parameter(g["p0"].clone().cuda())
    opt = Muon([p], lr=g["lr"], ddp=False)
    step = g["lr"] * 0.2 * max(g["p0"].shape) ** 0.5
    for t in range(3):
        p.grad = g["grads"][t].clone().cuda()
        opt.step()
        if t == 0:
            torch.testing.assert_close(p.detach().cpu(), g["p1"], rtol=0, atol=2e-2 * step + 1e-6)
            assert float((p.detach().cpu() - g["p1"]).abs().mean()) < 5e-3 * step
            # and the step moved the parameter by an orthogonal-ish update of that size
            moved = (p.detach().cpu() - g["p0"] * (1 - g["lr"] * 0.1)).norm()
            want = (g["p1"] - g["p0"] * (1 - g["lr"] * 0.1)).norm()
            assert abs(float(moved / want) - 1) < 5e-3
    torch.testing.assert_close(p.detach().cpu(), g["p3"], rtol=0, atol=3 * 2e-2 * step + 1e-6)
    assert float((p.detach().cpu() - g["p3"]).abs().mean()) < 3 * 5e-3 * step


def test_momentum_pass_bitwise(device):
    """sphip_muon_momentum against torch's own lerp_ / bfloat16() / norm on the same values
    (muon.py:157-160, :24, :31): buffer to the last bit or two, bf16 update, the denominator as torch
    computes it on a bf16 tensor (bf16(norm) + 1e-7 -> bf16)."""
    from sparsify_amd import ops
    gen = torch.Generator().manual_seed(5)
    g, buf = torch.randn(96, 50, generator=gen), torch.randn(96, 50, generator=gen)
    for beta, nesterov in ((0.95, True), (0.9, False), (0.0, True)):
        b_gpu = buf.clone().cuda()
        u, norm = ops.muon_momentum(g.cuda(), b_gpu if beta > 0 else None, beta, nesterov)
        if beta > 0:
            b_ref = buf.clone().lerp_(g, 1 - beta)
            u_ref = g.clone().lerp_(b_ref, beta) if nesterov else b_ref
            torch.testing.assert_close(b_gpu.cpu(), b_ref, rtol=0, atol=3e-7)       # (fma or mul + add: one rounding apart)
        else:
            u_ref = g
        x = u_ref.bfloat16()
        assert (u.cpu() != x).float().mean() < 1e-3                                  # ... a last-bit move may cross a bf16 boundary
        den = (u.cpu().norm() + 1e-7)
        assert den.dtype == torch.bfloat16
        assert float(norm[1]) == float(den), (float(norm[1]), float(den))
        torch.testing.assert_close(float(norm[0]), float(u.cpu().float().pow(2).sum()), rtol=1e-5, atol=0)
        xn = ops.muon_normalise_(u.clone(), norm).cpu()
        assert torch.equal(xn, u.cpu() / den)


def _rnd(t):
    return t.bfloat16().float()


@pytest.mark.parametrize("kernel", ["128", "256"])
@pytest.mark.parametrize("shape", [(512, 256, 384), (300, 64, 260)])
def test_gemm_bf16_axpy_rounding_chain(device, monkeypatch, kernel, shape):
    """out = rnd(rnd(alpha rnd(x W^T)) + rnd(beta C)) on both GEMM kernels against the same chain
    evaluated from an fp64 product: equal up to one bf16 ulp where the fp32 accumulation order
    moves a value across a rounding boundary."""
    from sparsify_amd import ops
    monkeypatch.setenv("SPHIP_GEMM", kernel)
    N, K, M = shape
    g = torch.Generator().manual_seed(N)
    x = torch.randn(N, K, generator=g).bfloat16()
    W = (torch.randn(M, K, generator=g) / K**0.5).bfloat16()
    C = torch.randn(N, M, generator=g).bfloat16()
    prod = (x.double() @ W.double().T).float()
    for alpha, beta, useC in ((1.0, 0.0, False), (2.0315, -4.775, True), (1.0, 3.4445, True)):
        got = ops.gemm_bf16_axpy(x.cuda(), W.cuda(), C.cuda() if useC else None, alpha=alpha, beta=beta).float().cpu()
        t = _rnd(prod)
        if alpha != 1.0:
            t = _rnd(alpha * t)
        want = _rnd(t + _rnd(beta * C.float())) if useC else t
        torch.testing.assert_close(got, want, rtol=2**-6, atol=2**-7)
        assert (got != want).float().mean() < 0.05


@pytest.mark.parametrize("kernel", ["128", "256"])
@pytest.mark.parametrize("shape", [(2048, 512), (256, 1024)])
def test_newton_schulz_vs_oracle_both_kernels(device, monkeypatch, kernel, shape):
    from sparsify_amd.muon import quintic_newtonschulz
    monkeypatch.setenv("SPHIP_GEMM", kernel)
    G = torch.randn(*shape, generator=torch.Generator().manual_seed(3))
    got = quintic_newtonschulz(G.cuda(), 5).float().cpu()
    want = O.newton_schulz5(G).float()
    torch.testing.assert_close(got, want, rtol=2e-2, atol=2e-2)
    # an orthogonalised matrix: singular values pushed into the quintic's band around 1
    sv = torch.linalg.svdvals(got)
    assert 0.6 < float(sv.min()) and float(sv.max()) < 1.3


def test_transpose_and_apply(device):
    from sparsify_amd import ops
    g = torch.Generator().manual_seed(0)
    a = torch.randn(130, 70, generator=g).bfloat16()
    assert torch.equal(ops.transpose_bf16(a.cuda()).cpu(), a.T.contiguous())
    p = torch.randn(77, 33, generator=g)
    o = torch.randn(77, 33, generator=g).bfloat16()
    sh = torch.empty(77, 33, dtype=torch.bfloat16, device="cuda")
    got = ops.muon_apply_(p.clone().cuda(), o.cuda(), 0.9998, 3e-3, sh)
    want = p * 0.9998 - 3e-3 * o.float()
    torch.testing.assert_close(got.cpu(), want, rtol=1e-6, atol=1e-7)
    assert torch.equal(sh, got.bfloat16())


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _muon_data():
    g = torch.Generator().manual_seed(21)
    return (torch.randn(1024, 128, generator=g), torch.randn(128, 128, generator=g),
            [(torch.randn(1024, 128, generator=g), torch.randn(128, 128, generator=g)) for _ in range(3)])


def _run_muon(ddp):
    from sparsify_amd.muon import Muon
    p0, q0, grads = _muon_data()
    p, q = torch.nn.Parameter(p0.clone().cuda()), torch.nn.Parameter(q0.clone().cuda())
    opt = Muon([p, q], lr=2e-3, ddp=ddp)
    for gp, gq in grads:
        p.grad, q.grad = gp.clone().cuda(), gq.clone().cuda()
        opt.step()
    torch.cuda.synchronize()
    return p.detach().cpu(), q.detach().cpu(), tuple(opt.state[p]["exp_avg"].shape)


def _muon_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = _run_muon(ddp=True)
    finally:
        dist.destroy_process_group()


def test_row_sharded_muon_two_ranks(device):
    """Data-parallel Muon: the rows of a tall matrix split over two ranks (Gram matrix summed over
    ranks, update rows all-gathered), the square matrix orthogonalised on both. Replicas identical;
    parameters equal to the single-rank step up to the rounding of the split Gram sum."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_muon_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
    for p in procs:
        if p.is_alive():
            p.kill()
            pytest.fail("a rank did not finish (collective mismatch?)")
        assert p.exitcode == 0
    one = _run_muon(ddp=False)
    assert ret[0][2] == (512, 128) and one[2] == (1024, 128)           # 1/W of the momentum buffer per rank
    for a, b in zip(ret[0][:2], ret[1][:2]):
        assert torch.equal(a, b)
    # (bf16 roundings of the Newton-Schulz iterates flip with the summation order of the Gram matrix)
    step = 2e-3 * 0.2 * 1024 ** 0.5
    torch.testing.assert_close(ret[0][0], one[0], rtol=0, atol=3 * 2e-2 * step)
    assert float((ret[0][0] - one[0]).abs().mean()) < 3 * 5e-3 * step
    assert torch.equal(ret[0][1], one[1])                               # not sharded: the same kernels on the same data


def _muon_resume_worker(rank, world, port, ret, save_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sparsify_amd import SparseCoderConfig, TrainConfig
        from sparsify_amd.trainer import SyntheticActivations, Trainer
        D, M, k, N, steps, cut = 256, 4096, 16, 512, 5, 3

        def make():
            cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k), optimizer="muon", hookpoints=["h0"], lr_warmup_steps=2,
                              save_dir=save_dir, run_name="run", save_every=10**9, wandb_log_frequency=1)
            src = SyntheticActivations(["h0"], D, N, steps, device="cuda", rank=rank)
            torch.manual_seed(0)
            return Trainer(cfg, src, device="cuda"), src

        tr, src = make()                                   # the uninterrupted run
        for s in range(steps):
            tr.train_micro_batch(src.batch(s))
        torch.cuda.synchronize()
        straight = {n: p.detach().cpu() for n, p in tr.saes["h0"].named_parameters()}

        tr, src = make()                                   # the same run, saved after `cut` steps ...
        for s in range(cut):
            tr.train_micro_batch(src.batch(s))
        muon = tr.optimizers[0]
        w = tr.saes["h0"].encoder.weight
        rows_before = muon.state[w]["exp_avg"].detach().cpu().clone()
        tr.save()
        tr2, src = make()                                  # ... and resumed by a fresh Trainer
        tr2.load_state(f"{save_dir}/run")
        assert tr2.global_step == cut
        w2 = tr2.saes["h0"].encoder.weight
        rows_after = tr2.optimizers[0].state[w2]["exp_avg"].detach().cpu().clone()
        for s in range(cut, steps):
            tr2.train_micro_batch(src.batch(s))
        torch.cuda.synchronize()
        resumed = {n: p.detach().cpu() for n, p in tr2.saes["h0"].named_parameters()}
        ret[rank] = dict(straight=straight, resumed=resumed, rows_before=rows_before, rows_after=rows_after,
                         range=muon._rows_of(w))
    finally:
        dist.destroy_process_group()


def test_row_sharded_muon_resume_two_ranks(device, tmp_path):
    """The momentum of a row-sharded data-parallel Muon run survives a checkpoint: rank 0 writes
    optimizer_0.pt in the reference's format (state key `exp_avg`, the whole parameter's shape:
    muon.py:154-157, trainer.py:663-668) from the gathered row shards, every rank cuts its own rows back
    out on load -- rank 1's are its own, not rank 0's -- and the resumed run ends on bit for bit the
    parameters of the uninterrupted one."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_muon_resume_worker, args=(r, 2, port, ret, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    for p in procs:
        if p.is_alive():
            p.kill()
            pytest.fail("a rank did not finish (collective mismatch?)")
        assert p.exitcode == 0
    sd = torch.load(tmp_path / "run" / "optimizer_0.pt", map_location="cpu", weights_only=True)
    bufs = [st["exp_avg"] for st in sd["state"].values()]
    assert sorted(tuple(b.shape) for b in bufs) == [(4096, 256), (4096, 256)]          # encoder.weight and W_dec, whole
    assert all("momentum_buffer" not in st for st in sd["state"].values())
    for r in (0, 1):
        lo, hi, shared = ret[r]["range"]
        assert shared and (lo, hi) == (2048 * r, 2048 * (r + 1))
        assert torch.equal(ret[r]["rows_before"], ret[r]["rows_after"]), f"rank {r} resumed with other momentum rows"
        assert any(torch.equal(b[lo:hi], ret[r]["rows_before"]) for b in bufs)
        for name, want in ret[r]["straight"].items():
            assert torch.equal(ret[r]["resumed"][name], want), f"rank {r}: {name} differs after the resume"
    assert not torch.equal(ret[0]["rows_before"], ret[1]["rows_before"])
    for name, a in ret[0]["resumed"].items():
        assert torch.equal(a, ret[1]["resumed"][name])
