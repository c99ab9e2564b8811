"""Trainer (hook body + step tail, trainer.py:340-447, 510-538) on the GPU against the CPU
oracle's train loop, plus optimizer variants, dead-latent bookkeeping, k-decay and
checkpoint round trips."""
import pytest
import torch

from oracle import sae_oracle as O

pytestmark = pytest.mark.gpu


class ListSource:
    def __init__(self, batches, d_in, hook="h0"):
        self.batches, self.widths, self.hook = batches, {hook: d_in}, hook

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        for b in self.batches:
            yield {self.hook: b}


def make_batches(n, N, D, transcode=False, seed=0):
    g = torch.Generator().manual_seed(seed)
    dictionary = torch.randn(128, D, generator=g)
    dictionary /= dictionary.norm(dim=1, keepdim=True)
    off = torch.randn(D, generator=g)
    R = torch.randn(D, D, generator=g) / D**0.5
    out = []
    for _ in range(n):
        codes = torch.rand(N, 128, generator=g) * (torch.rand(N, 128, generator=g) < 0.05)
        x = codes @ dictionary + 0.05 * torch.randn(N, D, generator=g) + off
        out.append((x, x @ R) if transcode else x)
    return out


@pytest.mark.parametrize("transcode", [False, True])
def test_trainer_matches_oracle_loop(device, transcode, tmp_path):
    from sparsify_amd import SparseCoderConfig, TrainConfig
    from sparsify_amd.trainer import Trainer
    D, M, k, N, steps = 64, 1024, 8, 512, 30
    batches = make_batches(steps, N, D, transcode)
    gpu_batches = [tuple(t.cuda() for t in b) if transcode else b.cuda() for b in batches]
    cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k, transcode=transcode), optimizer="signsgd", lr=2e-3,
                      save_dir=str(tmp_path), save_every=10**9, wandb_log_frequency=1)
    from sparsify_amd import SparseCoder
    torch.manual_seed(0)       # same host-side init stream as the oracle (the Trainer itself
    sae0 = SparseCoder(D, cfg.sae).cuda()      # initialises on the device, like the reference)
    tr = Trainer(cfg, ListSource(gpu_batches, D), device="cuda", saes={"h0": sae0})
    P = O.init_params(D, M, transcode=transcode, seed=0)
    assert torch.equal(tr.saes["h0"].encoder.weight.detach().cpu(), P["W_enc"])
    ocfg = O.OracleConfig(k=k, transcode=transcode)
    fired_since = torch.zeros(M, dtype=torch.long)
    for step in range(steps):
        b = batches[step]
        x, y = (b if transcode else (b, None))
        want = O.train_step_signsgd(P, x, y, ocfg, 2e-3, step=step, bf16=True)
        tr.train_micro_batch({"h0": gpu_batches[step]})
        got = tr.last_logged["fvu/h0"]
        tol = 2e-3 if step == 0 else 8e-2          # bf16 tie picks decorrelate the runs slowly
        if transcode and step >= 10:               # zero-initialised decoder + sign steps: the
            continue                               # transcoder run is chaotic beyond the start
        assert abs(got - float(want["fvu"])) <= tol * float(want["fvu"]), (step, got, float(want["fvu"]))
    assert tr.global_step == steps
    # dead-latent counters are exact integers: rebuild them from the last step's selection
    sae = tr.saes["h0"]
    counts = tr.num_tokens_since_fired["h0"].cpu()
    assert counts.min() == 0 and counts.max() <= steps * N and (counts % N == 0).all()
    # checkpoint round trip (sae.safetensors + cfg.json, sparse_coder.py:167-179)
    tr.save()
    from sparsify_amd import SparseCoder
    back = SparseCoder.load_from_disk(tmp_path / "unnamed" / "h0", device="cuda")
    for a, b in zip(back.state_dict().values(), sae.state_dict().values()):
        assert torch.equal(a, b)
    assert back.cfg == sae.cfg and back.d_in == D
    tr2 = Trainer(cfg, ListSource(gpu_batches, D), device="cuda")
    assert not torch.equal(tr2.saes["h0"].W_dec, sae.W_dec)
    tr2.load_state(str(tmp_path / "unnamed"))
    assert tr2.global_step == steps
    assert torch.equal(tr2.num_tokens_since_fired["h0"], tr.num_tokens_since_fired["h0"])
    assert torch.equal(tr2.saes["h0"].W_dec, sae.W_dec)


@pytest.mark.parametrize("optimizer", ["signum", "adam", "muon", "signsgd"])
def test_trainer_optimizers_reduce_fvu(device, optimizer, tmp_path):
    from sparsify_amd import SparseCoderConfig, TrainConfig
    from sparsify_amd.trainer import Trainer
    D, M, k, N, steps = 64, 512, 8, 512, 60
    batches = [b.cuda() for b in make_batches(steps, N, D)]
    cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k), optimizer=optimizer, save_dir="", save_every=10**9,
                      lr=1e-3 if optimizer == "adam" else None, lr_warmup_steps=5, auxk_alpha=1 / 32,
                      dead_feature_threshold=4 * N)
    tr = Trainer(cfg, ListSource(batches, D), device="cuda")
    curve = []
    for b in batches:
        tr.train_micro_batch({"h0": b})
        curve.append(tr.last_logged["fvu/h0"])
    assert all(torch.isfinite(torch.tensor(curve)))
    assert sum(curve[-5:]) / 5 < 0.97 * curve[0], (optimizer, curve[0], curve[-5:])
    assert "auxk/h0" in tr.last_logged and "dead_pct/h0" in tr.last_logged


@pytest.mark.parametrize("transcode,dtype", [(False, torch.bfloat16), (False, torch.float32), (True, torch.bfloat16)])
def test_fused_encoder_update_equals_separate_step(device, transcode, dtype, monkeypatch):
    """TrainConfig.fuse_encoder_update: the SignSGD step of encoder.weight applied inside the
    weight-gradient kernel gives bit for bit the parameters of gradient + optimizer pass, and the
    (M, D) gradient is never materialised."""
    from sparsify_amd import SparseCoderConfig, TrainConfig, ops
    from sparsify_amd.trainer import Trainer
    D, M, k, N, steps = 128, 2048, 16, 1024, 6
    batches = make_batches(steps, N, D, transcode)
    gpu = [tuple(t.cuda().to(dtype) for t in b) if transcode else b.cuda().to(dtype) for b in batches]
    res = {}
    calls = {"fused": 0, "grads": 0}
    fused_op, step_op = ops.segmented_weight_grad_signsgd_, ops.signsgd_step_

    def count_fused(*a, **kw):
        calls["fused"] += 1
        return fused_op(*a, **kw)

    def count_step(p, g, *a, **kw):
        calls["grads"] += p.shape == (M, D)
        return step_op(p, g, *a, **kw)

    monkeypatch.setattr(ops, "segmented_weight_grad_signsgd_", count_fused)
    monkeypatch.setattr(ops, "signsgd_step_", count_step)
    for fuse in (False, True):
        cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k, transcode=transcode), optimizer="signsgd",
                          save_dir="", save_every=10**9, wandb_log_frequency=1, fuse_encoder_update=fuse)
        torch.manual_seed(0)
        tr = Trainer(cfg, ListSource(gpu, D), device="cuda")
        fvus = []
        calls["fused"] = calls["grads"] = 0
        for b in gpu:
            tr.train_micro_batch({"h0": b})
            fvus.append(tr.last_logged["fvu/h0"])
        # fused: every step updates encoder.weight inside the gradient kernel and no (M, D) gradient
        # reaches the optimizer (a transcoder's W_dec, not projected, still takes the plain step)
        assert calls["fused"] == (steps if fuse else 0)
        assert calls["grads"] == (steps if transcode else 0) + (0 if fuse else steps)
        res[fuse] = ({n: p.detach().clone() for n, p in tr.saes["h0"].named_parameters()}, fvus)
    for name, ref in res[False][0].items():
        assert torch.equal(ref, res[True][0][name]), name
    assert res[False][1] == res[True][1]


@pytest.mark.parametrize("transcode", [False, True])
def test_fused_encoder_update_schedulefree_equals_separate_step(device, transcode, monkeypatch):
    """The same for the reference's default optimizer "signum" (schedule-free SignSGD): the step of
    (encoder.weight = y, z) applied inside the weight-gradient kernel leaves bit for bit the
    parameters, the z buffers and the averaged weights of gradient + one-kernel optimizer step."""
    from sparsify_amd import SparseCoderConfig, TrainConfig, ops
    from sparsify_amd.trainer import Trainer
    D, M, k, N, steps = 128, 2048, 16, 1024, 6
    batches = make_batches(steps, N, D, transcode)
    gpu = [tuple(t.cuda().bfloat16() for t in b) if transcode else b.cuda().bfloat16() for b in batches]
    calls = {"fused": 0}
    fused_op = ops.segmented_weight_grad_signsgd_

    def count_fused(*a, **kw):
        calls["fused"] += kw.get("schedule_free") is not None
        return fused_op(*a, **kw)

    monkeypatch.setattr(ops, "segmented_weight_grad_signsgd_", count_fused)
    res = {}
    for fuse in (False, True):
        cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k, transcode=transcode), optimizer="signum",
                          save_dir="", save_every=10**9, wandb_log_frequency=1, fuse_encoder_update=fuse)
        torch.manual_seed(0)
        tr = Trainer(cfg, ListSource(gpu, D), device="cuda")
        calls["fused"] = 0
        fvus = []
        for b in gpu:
            tr.train_micro_batch({"h0": b})
            fvus.append(tr.last_logged["fvu/h0"])
        assert calls["fused"] == (steps if fuse else 0)
        opt, sae = tr.optimizers[0], tr.saes["h0"]
        y = {n: p.detach().clone() for n, p in sae.named_parameters()}
        z = {n: opt.state[p]["z"].clone() for n, p in sae.named_parameters()}
        opt.eval()
        x = {n: p.detach().clone() for n, p in sae.named_parameters()}
        res[fuse] = (y, z, x, fvus, [g["k"] for g in opt.param_groups])
    for part in range(3):
        for name, ref in res[False][part].items():
            assert torch.equal(ref, res[True][part][name]), (part, name)
    assert res[False][3] == res[True][3] and res[False][4] == res[True][4] == [steps] * len(res[True][4])


def test_trainer_save_best(device, tmp_path):
    """trainer.py:540-544, 701-724: at every save_every-th step the per-coder best FVU is
    tracked and the coder written under {run}/best/{name}; a worse loss overwrites nothing."""
    from sparsify_amd import SparseCoder, SparseCoderConfig, TrainConfig
    from sparsify_amd.trainer import Trainer
    D, M, k, N, steps = 32, 256, 4, 256, 6
    batches = [b.cuda() for b in make_batches(steps, N, D)]
    cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k), optimizer="signsgd", save_dir=str(tmp_path),
                      run_name="run", save_every=2, save_best=True)
    tr = Trainer(cfg, ListSource(batches, D), device="cuda")
    best = tmp_path / "run" / "best" / "h0"
    seen = []
    for b in batches:
        tr.train_micro_batch({"h0": b})
        seen.append(tr.last_logged["fvu/h0"])
    # step = (global_step + 1) // grad_acc_steps and saves happen when (step + 1) % save_every == 0
    # (trainer.py:511, 540): with save_every = 2 that is the 1st, 3rd and 5th batch
    saved_at = [seen[i] for i in (0, 2, 4)]
    assert tr.best_loss["h0"] == min(saved_at)
    assert (best / "h0" / "sae.safetensors").exists() and (best / "state.pt").exists()
    kept = SparseCoder.load_from_disk(best / "h0", device="cuda")
    if saved_at[-1] == min(saved_at):                            # the last save was the best one
        assert torch.equal(kept.W_dec, tr.saes["h0"].W_dec)
    tr.best_loss["h0"] = -1.0                                    # nothing beats this
    stamp = (best / "h0" / "sae.safetensors").stat().st_mtime_ns
    tr.save_best({"h0": 0.5})
    assert (best / "h0" / "sae.safetensors").stat().st_mtime_ns == stamp and tr.best_loss["h0"] == -1.0
    tr2 = Trainer(cfg, ListSource(batches, D), device="cuda")
    tr2.load_state(str(tmp_path / "run"))
    # the last periodic save ran inside the 5th batch, before global_step was advanced
    assert tr2.best_loss["h0"] == min(saved_at) and tr2.global_step == 4


def test_resume_signum_continues_like_uninterrupted_run(device, tmp_path):
    """Resume with the default optimizer (schedule-free SignSGD): the checkpoint holds the averaged
    iterate x and train_mode = False (trainer.py:652-654, 675-677); load_state must put the
    parameters back on y, after which the next steps equal those of a run that never stopped."""
    from sparsify_amd import SparseCoderConfig, TrainConfig
    from sparsify_amd.trainer import Trainer
    D, M, k, N, steps, stop = 64, 512, 8, 256, 7, 4
    batches = [b.cuda() for b in make_batches(steps, N, D)]

    def cfg():
        return TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k), optimizer="signum", save_dir=str(tmp_path),
                           run_name="run", save_every=10**9, wandb_log_frequency=1)

    torch.manual_seed(0)
    whole = Trainer(cfg(), ListSource(batches, D), device="cuda")
    curve = []
    for b in batches:
        whole.train_micro_batch({"h0": b})
        curve.append(whole.last_logged["fvu/h0"])
    torch.manual_seed(0)
    first = Trainer(cfg(), ListSource(batches, D), device="cuda")
    for b in batches[:stop]:
        first.train_micro_batch({"h0": b})
    first.save()
    y_before = first.saes["h0"].W_dec.detach().clone()       # back on y after the save
    torch.manual_seed(1)
    second = Trainer(cfg(), ListSource(batches, D), device="cuda")
    second.load_state(str(tmp_path / "run"))
    assert second.global_step == stop and second.optimizers[0].train_mode
    torch.testing.assert_close(second.saes["h0"].W_dec.detach(), y_before, rtol=0, atol=2e-7)   # x -> y -> x -> y lerps
    resumed = []
    for b in batches[stop:]:
        second.train_micro_batch({"h0": b})                  # raised "call .train()" before the fix
        resumed.append(second.last_logged["fvu/h0"])
    for got, want in zip(resumed, curve[stop:]):
        assert abs(got - want) <= 2e-3 * want, (resumed, curve[stop:])


def test_bf16_shadow_and_slice_major_caches_follow_the_tensor_not_the_address(device, tmp_path):
    """Coders loaded one after the other land at the same addresses with the same shapes and the
    same version counters (caching allocator + load_model's copy_): the bf16 shadow of
    encoder.weight and the slice-major copy of W_dec must not survive their parameter."""
    import gc

    import importlib

    from sparsify_amd import SparseCoder, SparseCoderConfig, ops
    fe = importlib.import_module("sparsify_amd.fused_encoder")    # (the package re-exports the function of that name)
    D, M, k, N = 256, 16384, 16, 1024          # large enough for the L2 decode (slice-major copy)
    cfg = SparseCoderConfig(num_latents=M, k=k)
    x = (torch.randn(N, D, generator=torch.Generator().manual_seed(0)) + 1.0).cuda()
    want, paths = [], []
    for i in range(3):
        torch.manual_seed(10 + i)
        sae = SparseCoder(D, cfg)
        sae.save_to_disk(tmp_path / f"c{i}")
        paths.append(tmp_path / f"c{i}")
        sae = sae.cuda()
        with torch.no_grad():
            out = sae(x)
        want.append((out.sae_out.clone(), out.latent_indices.clone()))
        del sae, out
        gc.collect()
        fe._shadows.clear(); ops._slice_major.clear()      # references computed without any cache
    for i, path in enumerate(paths):
        sae = SparseCoder.load_from_disk(path, device="cuda")
        with torch.no_grad():
            out = sae(x)
        assert torch.equal(out.latent_indices, want[i][1]), f"coder {i} encoded with another coder's weights"
        assert torch.equal(out.sae_out, want[i][0]), f"coder {i} decoded from another coder's W_dec"
        del sae, out
        gc.collect()
    assert not fe._shadows and not ops._slice_major          # entries die with their parameters


def test_trainer_grad_accumulation_and_k_decay(device):
    from sparsify_amd import SparseCoderConfig, TrainConfig
    from sparsify_amd.trainer import Trainer
    D, M, k, N = 32, 256, 4, 128
    batches = [b.cuda() for b in make_batches(8, N, D)]
    cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k), optimizer="signsgd", grad_acc_steps=2, k_decay_steps=4,
                      save_dir="", save_every=10**9)
    tr = Trainer(cfg, ListSource(batches, D), device="cuda")
    ks, stepped = [], []
    w0 = tr.saes["h0"].encoder.weight.detach().clone()
    for i, b in enumerate(batches):
        ks.append(tr.get_current_k())
        stepped.append(tr.train_micro_batch({"h0": b}))
        if i == 0:
            assert torch.equal(tr.saes["h0"].encoder.weight, w0)      # first micro-step only accumulates
    assert stepped == [False, True] * 4                                 # divmod(global_step+1, 2), trainer.py:511
    assert ks[0] == min(M, 10 * D) and ks[-1] == k and ks == sorted(ks, reverse=True)   # trainer.py:229-235
    assert not torch.equal(tr.saes["h0"].encoder.weight, w0)


@pytest.mark.parametrize("optimizer,auxk", [("signsgd", 0.0), ("adam", 0.0), ("signsgd", 1 / 32)])
def test_grad_accumulation_in_the_kernel_equals_autograd_accumulation(device, optimizer, auxk):
    """TrainConfig.accumulate_in_kernel (trainer.py:444-447 + 510-520: loss / acc_steps per micro-step, one
    optimizer step per grad_acc_steps): from the second micro-step on the weight-gradient kernels add into
    encoder.weight.grad / W_dec.grad themselves. The same fp32 addition autograd would make, so two optimizer steps
    of three micro-steps each must leave bit-identical parameters with the switch on and off -- AuxK rows (added to
    the dense W_dec gradient by the main decode's backward) included."""
    from sparsify_amd import SparseCoderConfig, TrainConfig, ops
    from sparsify_amd.trainer import Trainer
    D, M, k, N = 512, 2048, 16, 4096
    batches = [b.cuda().to(torch.bfloat16) for b in make_batches(6, N, D)]
    finals = []
    for in_kernel in (True, False):
        ops.clear_grad_sinks()
        torch.manual_seed(0)
        cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k), optimizer=optimizer, lr=1e-3, grad_acc_steps=3,
                          auxk_alpha=auxk, dead_feature_threshold=N, accumulate_in_kernel=in_kernel, save_dir="",
                          save_every=10**9)
        tr = Trainer(cfg, ListSource(batches, D), device="cuda")
        hits0 = ops.GRAD_SINK_HITS
        for b in batches:
            tr.train_micro_batch({"h0": b})
        torch.cuda.synchronize()
        # two matrices x two later micro-steps x two optimizer steps
        assert ops.GRAD_SINK_HITS - hits0 == (8 if in_kernel else 0)
        finals.append({n: p.detach().clone() for n, p in tr.saes["h0"].named_parameters()})
    ops.clear_grad_sinks()
    for name in finals[0]:
        assert torch.equal(finals[0][name], finals[1][name]), name


@pytest.mark.parametrize("D", [64, 1024, 4096, 5120, 8192])
def test_schedulefree_fused_step_equals_generic_path(device, monkeypatch, D):
    """ScheduleFreeWrapper(SignSGD): the one-kernel step (plain and decoder groups with projection
    + re-normalisation) against the generic multi-pass path of the same class."""
    from sparsify_amd.schedule_free import ScheduleFreeWrapper
    from sparsify_amd.sign_sgd import SignSGD
    M, steps = 96, 6
    torch.manual_seed(D)
    W0 = O.unit_norm_rows(torch.randn(M, D))
    E0, b0 = torch.randn(M, D), torch.randn(M)
    grads = [(torch.randn(M, D), torch.randn(M, D), torch.randn(M)) for _ in range(steps)]

    def run(fused: bool):
        Wd, We, b = (torch.nn.Parameter(t.clone().cuda()) for t in (W0, E0, b0))
        opt = ScheduleFreeWrapper(SignSGD([dict(params=[Wd], lr=3e-3, project_rows=True, renorm_rows=True),
                                           dict(params=[We, b], lr=3e-3)]), momentum=0.95)
        opt.train()
        if not fused:
            monkeypatch.setattr(ScheduleFreeWrapper, "_fusable", lambda self: False)
        else:
            monkeypatch.undo()
            assert opt._fusable() or steps == 0 or Wd.grad is None
        for gd, ge, gb in grads:
            Wd.grad, We.grad, b.grad = gd.clone().cuda(), ge.clone().cuda(), gb.clone().cuda()
            opt.step()
            assert id(Wd) in opt.renormalised
            opt.renormalised.discard(id(Wd))
        z = [opt.state[p]["z"].cpu() for p in (Wd, We, b)]
        opt.eval()
        x = [p.detach().cpu().clone() for p in (Wd, We, b)]
        return x, z

    xf, zf = run(True)
    xg, zg = run(False)
    for a, c in zip(xf[1:] + zf[1:], xg[1:] + zg[1:]):
        torch.testing.assert_close(a, c, rtol=2e-6, atol=2e-7)
    # decoder: sign() of a ~0 projected component may differ between the two reduction orders
    for a, c in ((xf[0], xg[0]), (zf[0], zg[0])):
        bad = (a - c).abs() > 1e-5
        assert bad.float().mean() < 2e-3


@pytest.mark.parametrize("D", [64, 1000, 4096, 5120])
def test_schedulefree_kernels_vs_independent_oracle(device, D):
    """sphip_schedulefree_signsgd_step / sphip_decoder_schedulefree_signsgd_step over ten steps against
    oracle/schedule_free_oracle.py, a float64 statement of the published algorithm that shares no
    code with sparsify_amd/schedule_free.py (PARITY UNPINNED: the `schedulefree` package the
    reference imports is not available; see the oracle's header). Decoder rows: projection w.r.t.
    y, step, re-normalisation of y."""
    from oracle.schedule_free_oracle import ScheduleFreeSignSGD
    from sparsify_amd import ops
    M, steps, lr, beta = 80, 10, 3e-3, 0.95
    gen = torch.Generator().manual_seed(D)
    W0 = O.unit_norm_rows(torch.randn(M, D, generator=gen))
    E0 = torch.randn(M, D, generator=gen)
    ora_d, ora_e = ScheduleFreeSignSGD(W0, lr, beta), ScheduleFreeSignSGD(E0, lr, beta)
    Wy, Wz = W0.clone().cuda(), W0.clone().cuda()
    Ey, Ez = E0.clone().cuda(), E0.clone().cuda()
    shadow = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    for t in range(steps):
        gd, ge = torch.randn(M, D, generator=gen), torch.randn(M, D, generator=gen)
        c = ora_d.step(gd, project_rows=True, renorm_rows=True)
        assert ora_e.step(ge) == c and abs(c - 1.0 / (t + 1)) < 1e-12            # constant lr: c = 1 / (t + 1)
        ops.decoder_schedulefree_signsgd_step_(Wy, Wz, gd.cuda(), lr, beta, c, renorm=True)
        ops.schedulefree_signsgd_step_(Ey, Ez, ge.cuda(), lr, beta, c, shadow)
        # plain parameters: every element to fp32 rounding of a value of size ~1, every step
        torch.testing.assert_close(Ey.cpu().double(), ora_e.y, rtol=0, atol=2e-6)
        torch.testing.assert_close(Ez.cpu().double(), ora_e.z, rtol=0, atol=2e-6)
        assert torch.equal(shadow, Ey.bfloat16())
        # decoder rows: the sign of a projected component within rounding of zero may differ, which
        # moves that element by 2 lr; anything else is fp32 rounding
        for got, want in ((Wy, ora_d.y), (Wz, ora_d.z)):
            bad = (got.cpu().double() - want).abs() > 2e-6
            assert bad.double().mean() < 1e-3, (t, float(bad.double().mean()))
        # keep the two runs on the same trajectory (a flipped sign is legitimate, not an error to compound)
        ora_d.y, ora_d.z = Wy.cpu().double(), Wz.cpu().double()
    torch.testing.assert_close(Wy.norm(dim=1).cpu(), torch.ones(M), rtol=0, atol=1e-6)
    # .eval(): x = lerp(y, z, 1 - 1/beta) as schedule_free.ScheduleFreeWrapper.eval computes it
    x = torch.lerp(Ey, Ez, 1 - 1 / beta)
    torch.testing.assert_close(x.cpu().double(), ora_e.eval_weights(), rtol=0, atol=5e-6)


@pytest.mark.parametrize("betas", [(0.9, 0.999), (0.3, 0.95)])       # beta1 <= 0.5: torch.lerp's other branch
@pytest.mark.parametrize("n", [1, 7, 4096 + 3, 300_001])
def test_adam_step_vs_torch_adam(device, n, betas):
    """sparsify_amd.adam.Adam (sphip_adam_step) against torch.optim.Adam (single-tensor, fp32, on the
    CPU and on the GPU) over ten steps with a changing learning rate; the state dictionaries are
    interchangeable, and the bf16 shadow is the rounded parameter, bit for bit. Tolerance: a few
    ulp of fp32 per step (the only freedom is fused multiply-adds inside torch's own kernels)."""
    import copy
    from sparsify_amd import ops
    from sparsify_amd.adam import Adam
    gen = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=gen)
    grads = [torch.randn(n, generator=gen) * (10.0 ** (i % 3 - 1)) for i in range(10)]
    ours = torch.nn.Parameter(p0.clone().cuda())
    refs = [torch.nn.Parameter(p0.clone()), torch.nn.Parameter(p0.clone().cuda())]
    opt = Adam([ours], lr=1e-3, betas=betas)
    ropts = [torch.optim.Adam([r], lr=1e-3, betas=betas, foreach=False, fused=False) for r in refs]
    for i, g in enumerate(grads):
        for o in [opt] + ropts:
            o.param_groups[0]["lr"] = 1e-3 * (i + 1) / 10
        ours.grad = g.cuda()
        opt.step()
        for r, ro in zip(refs, ropts):
            r.grad = g.to(r.device)
            ro.step()
            torch.testing.assert_close(ours.detach().cpu(), r.detach().cpu(), rtol=2e-6, atol=2e-7)
    st, rst = opt.state[ours], ropts[0].state[refs[0]]
    assert float(st["step"]) == float(rst["step"]) == 10
    torch.testing.assert_close(st["exp_avg"].cpu(), rst["exp_avg"], rtol=2e-6, atol=1e-9)
    torch.testing.assert_close(st["exp_avg_sq"].cpu(), rst["exp_avg_sq"], rtol=2e-6, atol=1e-12)
    # torch's optimizer continues from our state dictionary, and ours from torch's
    cont = torch.nn.Parameter(ours.detach().clone())
    copt = torch.optim.Adam([cont], lr=1e-3, betas=betas, foreach=False, fused=False)
    copt.load_state_dict(copy.deepcopy(opt.state_dict()))      # (load_state_dict aliases tensors already in place)
    back = Adam([torch.nn.Parameter(ours.detach().clone())], lr=1e-3, betas=betas)
    back.load_state_dict(copy.deepcopy(copt.state_dict()))
    g = torch.randn(n, generator=gen).cuda()
    ours.grad, cont.grad = g, g.clone()
    back.param_groups[0]["params"][0].grad = g.clone()
    opt.step(), copt.step(), back.step()
    torch.testing.assert_close(ours.detach(), cont.detach(), rtol=2e-6, atol=2e-7)
    assert torch.equal(back.param_groups[0]["params"][0].detach(), ours.detach())
    # shadow refreshed in the same pass
    if n % 4 == 3:
        p = p0.clone().cuda()
        m, v, sh = torch.zeros_like(p), torch.zeros_like(p), torch.empty(n, dtype=torch.bfloat16, device="cuda")
        ops.adam_step_(p, grads[0].cuda(), m, v, 1e-3, 0.9, 0.999, 1e-8, 1, sh)
        assert torch.equal(sh, p.bfloat16())
        with pytest.raises(Exception):
            ops.adam_step_(p, grads[0].cuda(), m, v, 1e-3, 0.9, 0.999, 1e-8, 0)


def test_fused_encoder_update_refuses_a_second_backward(device):
    """With `fuse_encoder_update` the SignSGD step of encoder.weight happens inside
    FusedEncoder.backward. A second backward through the same (retained) graph would step twice on a
    gradient taken at the old weights: autograd must refuse it (the update bumps the parameter's version,
    which invalidates the tensors the graph saved), and the parameter must hold exactly one step."""
    from sparsify_amd import SparseCoderConfig, TrainConfig
    from sparsify_amd.trainer import Trainer
    D, M, k, N = 64, 512, 8, 256
    x = make_batches(1, N, D)[0].cuda()
    cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k), optimizer="signsgd", lr=1e-3, save_dir="", save_every=10**9)
    tr = Trainer(cfg, ListSource([x], D), device="cuda")
    sae = tr.saes["h0"]
    w0 = sae.encoder.weight.detach().clone()
    out = sae(x=x, y=x)
    out.fvu.backward(retain_graph=True)
    assert sae.encoder.weight.grad is None, "the fused update was expected to run inside the backward"
    w1 = sae.encoder.weight.detach().clone()
    moved = (w1 - w0).abs()
    assert float(moved.max()) == pytest.approx(1e-3, rel=1e-3) and bool(((moved == 0) | ((moved - 1e-3).abs() < 1e-6)).all())
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        out.fvu.backward()
    assert torch.equal(sae.encoder.weight.detach(), w1), "a refused backward must not have touched the parameter"
