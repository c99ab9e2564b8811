"""BASELINE.json's configurations at their own widths, against the CPU oracle (pinned to the
reference by tests/golden/*, tests/test_oracle_golden.py).

configs[0]: SparseCoder train step, d_in = 768, 32x -> 24576 latents, k = 32, fp32 (the reference's
            CPU-runnable case; the oracle runs it at full size): one Trainer step against the
            oracle, then the north-star check at this size -- 1000 SignSGD steps, the mean FVU of
            the last 100 within 1 % of the oracle's loop (sparsify/trainer.py:340-447, 510-538).
configs[1]: Pythia-160m's shape (GPT-NeoX, 12 layers x 768, 12 heads; random init: no checkpoint
            is reachable offline), residual-stream SAEs on all 12 layers, 32x, k = 32, bf16 model,
            on-the-fly activations through ModelTrainer; one hookpoint is followed by the
            oracle's loop on the activations the model produced (tests/test_driver.py's method).
"""
import pytest
import torch

from oracle import sae_oracle as O
from test_hip_api import run_coder, same_topk_sets

pytestmark = pytest.mark.gpu

D0, M0, K0, N0 = 768, 24576, 32, 1024


def dictionary_batches(D, atoms, seed, noise=0.02):
    """Sparse codes (16 of `atoms` active per token) over a fixed unit-norm dictionary + noise + offset: a
    run on these trains (FVU 0.69 -> 0.1 in 150 SignSGD steps of 1024 tokens at lr 1e-3 with 512 atoms)
    and is well conditioned, unlike i.i.d. noise."""
    gen = torch.Generator().manual_seed(seed)
    dictionary = torch.randn(atoms, D, generator=gen)
    dictionary /= dictionary.norm(dim=1, keepdim=True)
    offset = torch.randn(D, generator=gen)

    def batch(step, N):
        g = torch.Generator().manual_seed(1000 + step)
        codes = torch.rand(N, atoms, generator=g) * (torch.rand(N, atoms, generator=g) < 16.0 / atoms)
        return codes @ dictionary + noise * torch.randn(N, D, generator=g) + offset

    return batch


def test_config0_train_step_vs_oracle(device):
    """One forward / backward at d_in = 768, M = 24576, k = 32, N = 1024, fp32. Index sets equal to the
    oracle's except where the k-th / (k+1)-th pre-activations are within 1e-5 relative (fp32 sums in
    another order); values, reconstruction, FVU and every gradient against the oracle on the same
    selection: rtol 5e-4."""
    from sparsify_amd import SparseCoder, SparseCoderConfig
    batch = dictionary_batches(D0, 512, seed=7)
    P = O.init_params(D0, M0, seed=0)
    torch.manual_seed(0)
    sae = SparseCoder(D0, SparseCoderConfig(expansion_factor=32, k=K0))
    assert sae.num_latents == M0 and torch.equal(sae.encoder.weight.detach(), P["W_enc"])
    sae = sae.cuda()
    x = batch(0, N0)
    P["b_dec"] = x.mean(0)
    with torch.no_grad():
        sae.b_dec.copy_(P["b_dec"])
    out = run_coder(sae, "fp32", x.cuda(), None, None)
    out.fvu.backward()
    cfg = O.OracleConfig(k=K0)
    free, _ = O.forward_backward(P, x, None, cfg, need_grads=False)
    same_topk_sets(out.latent_indices.cpu(), free["latent_indices"], free["pre_acts"], K0, margin=1e-5)
    n_same = int((out.latent_indices.cpu().sort(-1).values == free["latent_indices"].sort(-1).values).all(-1).sum())
    assert n_same >= N0 - 2, f"only {n_same} of {N0} rows carry the oracle's set"
    forced = {"top": (out.latent_acts.detach().cpu(), out.latent_indices.cpu())}
    ref, G = O.forward_backward(P, x, None, cfg, forced=forced)
    torch.testing.assert_close(out.latent_acts.detach().cpu(), free["pre_acts"].gather(1, out.latent_indices.cpu()),
                               rtol=5e-4, atol=2e-5)
    torch.testing.assert_close(out.sae_out.detach().cpu(), ref["sae_out"], rtol=5e-4, atol=2e-5)
    torch.testing.assert_close(float(out.fvu.detach()), float(ref["fvu"]), rtol=5e-4, atol=1e-7)
    got = {"W_enc": sae.encoder.weight.grad, "b_enc": sae.encoder.bias.grad, "W_dec": sae.W_dec.grad, "b_dec": sae.b_dec.grad}
    for name, gr in got.items():
        scale = float(G[name].abs().max())
        torch.testing.assert_close(gr.cpu(), G[name], rtol=5e-4, atol=5e-5 * scale, msg=lambda m, n=name: f"{n}: {m}")


def test_config0_fvu_after_1k_steps_within_1pct(device, monkeypatch, capsys):
    """north_star at configs[0]'s own size and dtype (fp32, the reference's CPU eager path): 1000 steps of
    the Trainer -- plain SignSGD with the fused encoder and decoder steps, its forward run without
    autocast -- next to the oracle's fp32 loop on the same seeded batches; mean FVU of the last 100
    steps within 1 %. (The same oracle loop with its initialisation perturbed by 1e-6 relative ends
    within 0.05 %: the run is well conditioned.)"""
    from sparsify_amd import SparseCoder, SparseCoderConfig, TrainConfig
    from sparsify_amd.trainer import Trainer
    steps, lr = 1000, 1e-3
    batch = dictionary_batches(D0, 512, seed=11)
    P = O.init_params(D0, M0, seed=0)
    cfg = TrainConfig(sae=SparseCoderConfig(expansion_factor=32, k=K0), optimizer="signsgd", lr=lr, save_dir="",
                      save_every=10**9, wandb_log_frequency=1)
    torch.manual_seed(0)
    sae0 = SparseCoder(D0, cfg.sae).cuda()
    fwd = SparseCoder.forward.__wrapped__                 # the undecorated forward: no bf16 autocast

    def fp32_forward(self, *a, **kw):
        with torch.autocast("cuda", enabled=False):
            return fwd(self, *a, **kw)

    monkeypatch.setattr(SparseCoder, "forward", fp32_forward)

    class Src:
        widths = {"h0": D0}

        def __len__(self):
            return steps

    tr = Trainer(cfg, Src(), device="cuda", saes={"h0": sae0})
    assert torch.equal(sae0.encoder.weight.detach().cpu(), P["W_enc"])
    ocfg = O.OracleConfig(k=K0)
    got, want = [], []
    for step in range(steps):
        x = batch(step, N0)
        want.append(float(O.train_step_signsgd(P, x, None, ocfg, lr, step=step, bf16=False)["fvu"]))
        tr.train_micro_batch({"h0": x.cuda()})
        got.append(tr.last_logged["fvu/h0"])
        if step < 5:
            assert abs(got[step] - want[step]) <= 2e-4 * want[step], (step, got[step], want[step])
        if step % 100 == 99:
            # minutes of CPU oracle work: a progress line past pytest's capture, so the run is not taken for hung
            with capsys.disabled():
                print(f"[config0 1k steps] step {step + 1}: fvu {got[-1]:.5f} (oracle {want[-1]:.5f})", flush=True)
    assert sae0.encoder.weight.grad is None, "the encoder step was expected to run inside the gradient kernel"
    got, want = torch.tensor(got), torch.tensor(want)
    assert want[-100:].mean() < 0.6 * want[0], "the run must actually train"
    rel = abs(got[-100:].mean() - want[-100:].mean()) / want[-100:].mean()
    assert rel <= 0.01, f"FVU after 1k steps differs by {rel:.4%} ({float(got[-100:].mean()):.5f} vs {float(want[-100:].mean()):.5f})"


def pythia160m_shape(vocab=50304):
    from transformers import GPTNeoXConfig, GPTNeoXForCausalLM
    torch.manual_seed(0)
    cfg = GPTNeoXConfig(vocab_size=vocab, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                        intermediate_size=3072, max_position_embeddings=2048, rotary_pct=0.25)
    return GPTNeoXForCausalLM(cfg).eval()


def test_config1_pythia160m_shape_all_layers(device):
    """configs[1]: 12 residual-stream SAEs (32x = 24576 latents, k = 32) trained on the fly on a bf16
    GPT-NeoX of Pythia-160m's shape, 10 steps of 8 x 128 tokens. Every coder's FVU falls; the coder of
    `layers.6`, given the oracle's initialisation, follows the oracle's loop on the captured activations."""
    from sparsify_amd import SparseCoderConfig, TrainConfig
    from sparsify_amd.driver import ModelTrainer
    from test_driver import _Capture
    model = pythia160m_shape().to(torch.bfloat16).cuda()
    S, B, steps, lr = 128, 8, 10, 1e-3
    data = torch.randint(0, 50304, (B * steps, S), generator=torch.Generator().manual_seed(5))
    cfg = TrainConfig(sae=SparseCoderConfig(expansion_factor=32, k=32), optimizer="signsgd", lr=lr, batch_size=B,
                      save_dir="", save_every=10**9, wandb_log_frequency=1)
    mt = ModelTrainer(cfg, data, model)
    points = [f"layers.{i}" for i in range(12)]
    assert mt.cfg.hookpoints == points and all(mt.saes[p].num_latents == 24576 and mt.saes[p].d_in == 768 for p in points)
    follow = "layers.6"
    P = O.init_params(768, 24576, seed=0)
    sae = mt.saes[follow]
    with torch.no_grad():
        sae.encoder.weight.copy_(P["W_enc"]); sae.encoder.bias.copy_(P["b_enc"])
        sae.W_dec.copy_(P["W_dec"]); sae.b_dec.copy_(P["b_dec"])
    for p in sae.parameters():
        torch.autograd.graph.increment_version(p)
    curves = {p: [] for p in points}
    flush = mt._flush_metrics

    def spy(step):
        flush(step)
        for p in points:
            curves[p].append(mt.last_logged[f"fvu/{p}"])

    mt._flush_metrics = spy
    mt.fit()
    assert mt.global_step == steps and all(len(c) == steps for c in curves.values())
    for p, c in curves.items():
        assert all(v == v and v < 1e3 for v in c) and c[-1] < c[0], (p, c)
    cap = _Capture(model, [follow], False)
    ocfg = O.OracleConfig(k=32)
    for s in range(steps):
        x = data[s * B: (s + 1) * B].cuda()
        acts = cap(x)[follow].flatten(0, 1).cpu()
        assert acts.dtype == torch.bfloat16
        want = float(O.train_step_signsgd(P, acts, None, ocfg, lr, step=s, bf16=True)["fvu"])
        tol = 2e-3 if s == 0 else 5e-2          # bf16 tie picks decorrelate the runs slowly
        assert abs(curves[follow][s] - want) <= tol * want, (s, curves[follow][s], want)


@pytest.mark.parametrize("variant", ["plain", "auxk", "multitopk"])
def test_step_on_more_tokens_than_one_block_vs_oracle(device, variant):
    """N = 26000 tokens in one step: more than the L2-sliced weight-gradient kernels take at once (24576), so both
    backward passes go through ops.weight_grad_in_token_blocks (two blocks, the second adding to the first). The
    reference's index_add_ / atomics take any number of rows (fused_encoder.py:67-84, xformers.py:110-136; its
    default step is 65536 tokens); forward, losses and every gradient against the oracle on the same selections.
    auxk: a tenth of the latents dead -- the dense AuxK path parks its gradient rows for the main decode's backward,
    which adds them to the blocks' sum. multitopk: the 4k decode's backward (all 4k slots, in blocks too) parks its
    dense gradient and the main decode's blocks add into it."""
    from oracle import sae_oracle as O
    from sparsify_amd import SparseCoder, SparseCoderConfig, ops
    from test_hip_api import run_coder, same_topk_sets
    N, D, M, k = 26000, 512, 2048, 16
    alpha = 1 / 32 if variant == "auxk" else 0.0
    assert N > ops.TOKEN_BLOCK_LIMIT and len(ops.token_blocks(N)) == 2
    torch.manual_seed(0)
    sae = SparseCoder(D, SparseCoderConfig(num_latents=M, k=k, multi_topk=(variant == "multitopk"))).cuda()
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(N, D, generator=g) + torch.randn(D, generator=g)).to(torch.bfloat16).cuda()
    dead = (torch.rand(M, generator=g) < 0.1) if variant == "auxk" else None
    with torch.no_grad():
        sae.b_dec.copy_(x.float().mean(0))
    P = {"W_enc": sae.encoder.weight.detach().cpu().clone(), "b_enc": sae.encoder.bias.detach().cpu().clone(),
         "W_dec": sae.W_dec.detach().cpu().clone(), "b_dec": sae.b_dec.detach().cpu().clone()}
    ops.timer.enabled = True
    ops.timer.reset()
    try:
        out = run_coder(sae, "bf16", x, None, None if dead is None else dead.cuda())
        (out.fvu + alpha * out.auxk_loss + out.multi_topk_fvu / 8).backward()
        torch.cuda.synchronize()
        calls = ops.timer.summary()
    finally:
        ops.timer.enabled = False
    # two blocks x (W_dec, encoder.weight), sorted once per block; the 4k decode's gradient (812 entries per latent
    # here) takes the wave-per-latent kernel, which cannot add: the whole batch in one call, one more sort
    blocks_4k = all(ops.weight_grad_accumulates(hi - lo, 4 * k, M, D, torch.float32, False) for lo, hi in ops.token_blocks(N))
    extra = 0 if variant != "multitopk" else (2 if blocks_4k else 1)
    assert calls["sphip_segmented_weight_grad"]["calls"] == 4 + extra
    assert calls["sphip_build_latent_segments"]["calls"] == 2 + extra
    with torch.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():
        enc = sae.encode(x)
    forced = {"top": (enc.top_acts.cpu(), enc.top_indices.cpu())}
    if variant == "multitopk":
        forced["multi"] = (out.latent_acts.detach().cpu(), out.latent_indices.cpu())
    if dead is not None:
        assert "sphip_gemm_bf16_f32acc" in calls                            # the dense AuxK path ran
        a_acts, a_idx = ops.topk(enc.pre_acts, D // 2 if int(dead.sum()) >= D // 2 else int(dead.sum()), keep_mask=dead.cuda())
        forced["aux"] = (a_acts.cpu(), a_idx.cpu())
    cfg = O.OracleConfig(k=k, multi_topk=(variant == "multitopk"))
    ref, G = O.forward_backward(P, x.cpu(), None, cfg, dead_mask=dead, auxk_alpha=alpha, bf16=True, forced=forced)
    same_topk_sets(enc.top_indices.cpu(), O.select_latents(ref["pre_acts"], k, "topk")[1], ref["pre_acts"], k, margin=2**-7)
    torch.testing.assert_close(out.sae_out.detach().cpu(), ref["sae_out"], rtol=1e-4, atol=1e-4)
    for key, val in (("fvu", out.fvu), ("auxk_loss", out.auxk_loss), ("multi_topk_fvu", out.multi_topk_fvu)):
        torch.testing.assert_close(float(val.detach()), float(ref[key]), rtol=2e-4, atol=1e-6)
    grads = {"W_enc": sae.encoder.weight.grad, "b_enc": sae.encoder.bias.grad, "W_dec": sae.W_dec.grad, "b_dec": sae.b_dec.grad}
    for pname, gr in grads.items():
        rel = (gr.cpu() - G[pname]).norm() / G[pname].norm().clamp_min(1e-12)
        assert rel < 2e-3, f"{pname}: relative error {rel:.3e}"
