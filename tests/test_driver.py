"""On-the-fly activation driver (sparsify_amd/driver.py; reference trainer.py:38-90, :333-508,
utils.py:20-61). Hookpoint / width resolution runs on CPU; the train loop needs the GPU."""
import pytest
import torch


def tiny_lm(layers=3, hidden=64, vocab=128):
    from transformers import GPTNeoXConfig, GPTNeoXForCausalLM
    torch.manual_seed(0)
    cfg = GPTNeoXConfig(vocab_size=vocab, hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=4,
                        intermediate_size=2 * hidden, max_position_embeddings=64)
    return GPTNeoXForCausalLM(cfg).eval()


def test_hookpoint_and_width_resolution():
    from sparsify_amd import SparseCoderConfig, TrainConfig
    from sparsify_amd.driver import get_layer_list, resolve_hookpoints, resolve_widths
    m = tiny_lm(layers=12)
    name, mods = get_layer_list(m)
    assert name == "layers" and len(mods) == 12
    # wildcards, natural order (layers.2 before layers.10), trainer.py:41-48
    pts = resolve_hookpoints(m, TrainConfig(sae=SparseCoderConfig(), hookpoints=["layers.*.mlp"]))
    assert pts == [f"layers.{i}.mlp" for i in range(12)]
    # layers -> hookpoints, all layers by default, stride (trainer.py:50-59)
    assert resolve_hookpoints(m, TrainConfig(sae=SparseCoderConfig(), layers=[1, 5])) == ["layers.1", "layers.5"]
    assert resolve_hookpoints(m, TrainConfig(sae=SparseCoderConfig(), layer_stride=4)) == ["layers.0", "layers.4", "layers.8"]
    with pytest.raises(AssertionError):
        resolve_hookpoints(m, TrainConfig(sae=SparseCoderConfig(), hookpoints=["layers.0"], layers=[0]))
    widths = resolve_widths(m, ["layers.3", "layers.3.mlp.dense_h_to_4h", "embed_in"])
    assert widths == {"layers.3": 64, "layers.3.mlp.dense_h_to_4h": 128, "embed_in": 64}


class _Capture:
    """The same activations the driver sees, captured with plain hooks (for the base Trainer)."""

    def __init__(self, model, points, transcode):
        self.model, self.points, self.transcode = model, points, transcode

    @torch.no_grad()
    def __call__(self, x):
        got = {}

        def mk(name):
            def hook(mod, inputs, outputs):
                i = inputs[0] if isinstance(inputs, tuple) else inputs
                o = outputs[0] if isinstance(outputs, tuple) else outputs
                got[name] = (i.detach().clone(), o.detach().clone()) if self.transcode else o.detach().clone()
            return hook

        hs = [self.model.base_model.get_submodule(p).register_forward_hook(mk(p)) for p in self.points]
        try:
            self.model(x)
        finally:
            for h in hs:
                h.remove()
        return got


@pytest.mark.gpu
@pytest.mark.parametrize("transcode", [False, True])
def test_model_trainer_equals_trainer_on_captured_activations(device, transcode):
    from sparsify_amd import SparseCoderConfig, TrainConfig
    from sparsify_amd.driver import ModelTrainer
    from sparsify_amd.trainer import Trainer
    model = tiny_lm().cuda()
    S, B, steps, vocab = 32, 8, 6, 128
    g = torch.Generator().manual_seed(1)
    data = torch.randint(0, vocab, (B * steps + 3, S), generator=g)         # 3 leftover sequences are dropped
    pats = ["layers.*.mlp"] if transcode else ["layers.[02]"]

    def cfg():
        return TrainConfig(sae=SparseCoderConfig(num_latents=512, k=8, transcode=transcode), optimizer="signsgd", lr=1e-3,
                           hookpoints=list(pats), batch_size=B, exclude_tokens=[0, 5], init_seeds=[0, 7],
                           save_dir="", save_every=10**9, wandb_log_frequency=1)

    mt = ModelTrainer(cfg(), data, model)
    points = mt.cfg.hookpoints
    assert points == (["layers.0.mlp", "layers.1.mlp", "layers.2.mlp"] if transcode else ["layers.0", "layers.2"])
    assert set(mt.saes) == {f"{p}/seed{s}" for p in points for s in (0, 7)}          # trainer.py:83-87
    mt.fit()
    assert mt.global_step == steps

    class Src:
        widths = {p: 64 for p in points}

        def __len__(self):
            return steps

    c2 = cfg()
    c2.hookpoints = list(points)
    base = Trainer(c2, Src(), device="cuda")
    cap = _Capture(model, points, transcode)
    fvus = []
    for s in range(steps):
        x = data[s * B: (s + 1) * B].cuda()
        mask = torch.isin(x, torch.tensor([0, 5], device="cuda"), invert=True)
        base.train_micro_batch(cap(x), mask)
        fvus.append(base.last_logged[f"fvu/{points[0]}/seed0"])
    for name, sae in mt.saes.items():
        for (n1, p1), (n2, p2) in zip(sae.named_parameters(), base.saes[name].named_parameters()):
            assert torch.equal(p1, p2), f"{name}.{n1}"
    assert torch.equal(mt.num_tokens_since_fired[f"{points[0]}/seed0"], base.num_tokens_since_fired[f"{points[0]}/seed0"])
    assert fvus[-1] < fvus[0]
    assert mt.last_logged[f"fvu/{points[0]}/seed0"] == fvus[-1]


@pytest.mark.gpu
@pytest.mark.parametrize("transcode", [False, True])
def test_model_trainer_follows_the_oracle_loop(device, transcode):
    """The driver against the CPU oracle (pinned to the reference): the coders of one hookpoint are
    given the oracle's initialisation, the model's activations are captured with plain hooks, and
    the oracle's train loop (trainer.py:392-447, 510-522 restated) runs on exactly those tokens
    (excluded tokens dropped, trainer.py:383-390). FVU step by step, as
    tests/test_hip_trainer.py::test_trainer_matches_oracle_loop."""
    from oracle import sae_oracle as O
    from sparsify_amd import SparseCoderConfig, TrainConfig
    from sparsify_amd.driver import ModelTrainer
    model = tiny_lm().cuda()
    S, B, steps, vocab, M, k, lr = 32, 16, 8, 128, 512, 8, 2e-3
    data = torch.randint(0, vocab, (B * steps, S), generator=torch.Generator().manual_seed(4))
    point = "layers.1.mlp" if transcode else "layers.1"
    cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k, transcode=transcode), optimizer="signsgd", lr=lr,
                      hookpoints=[point], batch_size=B, exclude_tokens=[0, 5], save_dir="", save_every=10**9,
                      wandb_log_frequency=1)
    mt = ModelTrainer(cfg, data, model)
    P = O.init_params(64, M, transcode=transcode, seed=0)
    sae = mt.saes[point]
    with torch.no_grad():
        sae.encoder.weight.copy_(P["W_enc"]); sae.encoder.bias.copy_(P["b_enc"])
        sae.W_dec.copy_(P["W_dec"]); sae.b_dec.copy_(P["b_dec"])
    curve = []
    flush = mt._flush_metrics

    def spy(step):
        flush(step)
        curve.append(mt.last_logged[f"fvu/{point}"])

    mt._flush_metrics = spy
    mt.fit()
    assert len(curve) == steps
    cap = _Capture(model, [point], transcode)
    ocfg = O.OracleConfig(k=k, transcode=transcode)
    for s in range(steps):
        x = data[s * B: (s + 1) * B].cuda()
        keep = torch.isin(x, torch.tensor([0, 5], device="cuda"), invert=True).reshape(-1).cpu()
        got = cap(x)[point]
        xin, y = ((got[0].flatten(0, 1).cpu()[keep], got[1].flatten(0, 1).cpu()[keep]) if transcode
                  else (got.flatten(0, 1).cpu()[keep], None))
        want = float(O.train_step_signsgd(P, xin, y, ocfg, lr, step=s, bf16=True)["fvu"])
        tol = 2e-3 if s == 0 else 8e-2          # bf16 tie picks decorrelate the runs slowly
        assert abs(curve[s] - want) <= tol * want, (s, curve[s], want)


@pytest.mark.gpu
@pytest.mark.parametrize("loss_fn", ["ce", "kl"])
def test_end_to_end_loss_coder_gradients_match_the_oracle(device, loss_fn):
    """loss_fn 'ce' / 'kl': whatever gradient the language model sends into a coder's output, the
    coder's parameter gradients are those of the CPU oracle (oracle.backward_from_output_grad: the
    reference's decoder / cast / fused-encoder backward, pinned by the goldens) for the same
    dL/d sae_out, the same tokens and the same selected latents -- to the bf16 tolerance of the
    other oracle comparisons (2e-3 of the gradient's norm), both hookpoints, the second of which
    receives its input through the first coder's spliced reconstruction."""
    from oracle import sae_oracle as O
    from sparsify_amd import SparseCoderConfig, TrainConfig
    from sparsify_amd.driver import ModelTrainer
    model = tiny_lm().cuda()
    S, B, vocab = 32, 8, 128
    data = torch.randint(0, vocab, (B, S), generator=torch.Generator().manual_seed(3))
    cfg = TrainConfig(sae=SparseCoderConfig(num_latents=512, k=8), optimizer="signsgd", lr=1e-3, loss_fn=loss_fn,
                      hookpoints=["layers.0", "layers.2"], batch_size=B, grad_acc_steps=2,      # no optimizer step yet
                      save_dir="", save_every=10**9, wandb_log_frequency=1)
    mt = ModelTrainer(cfg, data, model)
    seen = {}
    orig = mt.train_on_activations

    def spy(name, inputs, outputs, mask=None, e2e=False):
        out, n = orig(name, inputs, outputs, mask, e2e=e2e)
        rec = {"x": outputs.detach().reshape(-1, outputs.shape[-1]).float().cpu(), "idx": out.latent_indices.detach().cpu(),
               "acts": out.latent_acts.detach().cpu(),
               "P": {"W_enc": mt.saes[name].encoder.weight.detach().cpu().clone(),
                     "b_enc": mt.saes[name].encoder.bias.detach().cpu().clone(),
                     "W_dec": mt.saes[name].W_dec.detach().cpu().clone(), "b_dec": mt.saes[name].b_dec.detach().cpu().clone()}}
        out.sae_out.register_hook(lambda g, rec=rec: rec.__setitem__("g", g.detach().reshape(-1, g.shape[-1]).float().cpu()))
        seen[name] = rec
        return out, n

    mt.train_on_activations = spy
    mt.fit()
    assert set(seen) == {"layers.0", "layers.2"} and all("g" in r for r in seen.values())
    ocfg = O.OracleConfig(k=8)
    for name, r in seen.items():
        want, _ = O.backward_from_output_grad(r["P"], r["x"] - r["P"]["b_dec"], r["idx"], r["acts"], r["g"], ocfg)
        sae = mt.saes[name]
        got = {"W_enc": sae.encoder.weight.grad, "b_enc": sae.encoder.bias.grad, "W_dec": sae.W_dec.grad, "b_dec": sae.b_dec.grad}
        for pname, gr in got.items():
            rel = (gr.cpu() - want[pname]).norm() / want[pname].norm().clamp_min(1e-20)
            assert rel < 2e-3, f"{name}.{pname}: relative error {rel:.3e}"


@pytest.mark.gpu
@pytest.mark.parametrize("loss_fn", ["ce", "kl"])
def test_end_to_end_loss_gradients_match_torch_autograd(device, loss_fn):
    """loss_fn 'ce' / 'kl' (trainer.py:425-430, 486-500): the model's loss is back-propagated through
    the spliced reconstructions into the coders, including the encoder's grad-input path for the
    second hookpoint. Checked against torch autograd on a plain-torch restatement of the coder
    (same bf16 autocast, same selected latents)."""
    import torch.nn.functional as F
    from sparsify_amd import SparseCoderConfig, TrainConfig
    from sparsify_amd.driver import ModelTrainer
    model = tiny_lm().cuda()
    S, B, vocab = 32, 8, 128
    data = torch.randint(0, vocab, (B, S), generator=torch.Generator().manual_seed(3))
    cfg = TrainConfig(sae=SparseCoderConfig(num_latents=512, k=8), optimizer="signsgd", lr=1e-3, loss_fn=loss_fn,
                      hookpoints=["layers.0", "layers.2"], batch_size=B, grad_acc_steps=2,      # no optimizer step yet
                      save_dir="", save_every=10**9, wandb_log_frequency=1)
    mt = ModelTrainer(cfg, data, model)
    init = {n: {k: v.detach().clone() for k, v in sae.state_dict().items()} for n, sae in mt.saes.items()}
    picked = {}
    orig = mt.train_on_activations

    def spy(name, inputs, outputs, mask=None, e2e=False):
        out, n = orig(name, inputs, outputs, mask, e2e=e2e)
        picked[name] = out.latent_indices.detach().clone()
        return out, n

    mt.train_on_activations = spy
    mt.fit()
    assert mt.global_step == 1 and set(picked) == {"layers.0", "layers.2"}

    # ---- plain-torch restatement, driven by autograd
    x = data.cuda()
    ref = {}
    for name, sd in init.items():
        ref[name] = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        # step-0 initialisation the hook applied before the forward (trainer.py:403-408)
    acts0 = {}

    def mk(name):
        def hook(mod, inputs, outputs):
            h, *aux = outputs if isinstance(outputs, tuple) else (outputs,)
            flat = h.flatten(0, 1)
            P = ref[name]
            if name not in acts0:
                acts0[name] = True
                with torch.no_grad():
                    P["b_dec"].copy_(flat.detach().float().mean(0))
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pre = F.relu(F.linear(flat - P["b_dec"], P["encoder.weight"], P["encoder.bias"]))
                top = pre.gather(1, picked[name])
            rec = (P["W_dec"][picked[name]] * top.float().unsqueeze(-1)).sum(1) + P["b_dec"]
            out = rec.type_as(h).reshape(h.shape)
            return (out, *aux) if isinstance(outputs, tuple) else out
        return hook

    hs = [model.base_model.get_submodule(n).register_forward_hook(mk(n)) for n in ref]
    try:
        if loss_fn == "ce":
            loss = model(x, labels=x).loss
        else:
            for h in hs:
                h.remove()
            with torch.no_grad():
                clean = model(x).logits.softmax(-1)
            hs = [model.base_model.get_submodule(n).register_forward_hook(mk(n)) for n in ref]
            loss = -torch.sum(clean * model(x).logits.log_softmax(-1), dim=-1).mean()
        (loss / 2).backward()
    finally:
        for h in hs:
            h.remove()
    for name, sae in mt.saes.items():
        got = dict(sae.named_parameters())
        for pname, p_ref in ref[name].items():
            g_ref, g_got = p_ref.grad, got[pname].grad
            assert g_got is not None and g_ref is not None, f"{name}.{pname}"
            cos = F.cosine_similarity(g_got.flatten().double(), g_ref.flatten().double(), dim=0)
            rel = (g_got.double() - g_ref.double()).norm() / g_ref.double().norm()
            # bf16 autocast on both sides, different accumulation orders
            assert cos > 0.999 and rel < 4e-2, f"{name}.{pname}: cos {cos:.5f} rel {rel:.3e}"


def test_memmap_dataset(tmp_path):
    import numpy as np
    from torch.utils.data import DataLoader
    from sparsify_amd.data import MemmapDataset
    toks = (np.arange(10 * 16 + 5) % 50000).astype(np.uint16)       # 5 trailing tokens do not fill a row
    path = tmp_path / "tokens.bin"
    toks.tofile(path)
    ds = MemmapDataset(str(path), ctx_len=16)
    assert len(ds) == 10 and ds[3]["input_ids"].dtype == torch.int64
    assert torch.equal(ds[3]["input_ids"], torch.arange(48, 64))
    assert len(MemmapDataset(str(path), 16, max_examples=4)) == 4
    assert torch.equal(ds.select(range(2, 5))[0]["input_ids"], ds[2]["input_ids"]) and len(ds.select(range(2, 5))) == 3
    parts = [ds.shard(3, i) for i in range(3)]
    assert [len(p) for p in parts] == [4, 3, 3] and torch.equal(parts[1][0]["input_ids"], ds[4]["input_ids"])
    batch = next(iter(DataLoader(ds, batch_size=4)))
    assert batch["input_ids"].shape == (4, 16)


@pytest.mark.parametrize("name,kw,model_max", [("rows16", dict(max_seq_len=16), 64),
                                               ("rows16_tail", dict(max_seq_len=16, return_final_batch=True), 64),
                                               ("model_max_24", dict(max_seq_len=2048), 24)])
def test_chunk_and_tokenize_vs_reference(name, kw, model_max):
    """sparsify_amd.data.chunk_and_tokenize against the rows the reference's own function produced
    (sparsify/data.py:14-101 through tools/make_golden.py -> tests/golden/tokenize.json) on the same
    word-level tokenizer and documents: same rows, same order, `input_ids` the only column."""
    import json
    from datasets import Dataset
    from conftest import GOLDEN, toy_documents, toy_tokenizer
    from sparsify_amd.data import chunk_and_tokenize
    want = json.loads((GOLDEN / "tokenize.json").read_text())[name]
    docs = toy_documents()
    ds = chunk_and_tokenize(Dataset.from_dict({"text": docs, "other": list(range(len(docs)))}), toy_tokenizer(model_max),
                            num_proc=1, load_from_cache_file=False, **kw)
    assert ds.column_names == ["input_ids"]
    got = [row["input_ids"] for row in ds]
    assert all(isinstance(r, torch.Tensor) and r.dtype == torch.int64 for r in got)
    assert [r.tolist() for r in got] == want
    row_len = min(model_max, kw["max_seq_len"])
    assert all(len(r) == row_len for r in want[:-1]) and want[0][0] == 0          # starts with the end-of-text token
    assert (len(want[-1]) < row_len) == bool(kw.get("return_final_batch")) or len(want[-1]) == row_len


def test_chunk_and_tokenize_too_little_data():
    from datasets import Dataset
    from conftest import toy_tokenizer
    from sparsify_amd.data import chunk_and_tokenize
    with pytest.raises(ValueError, match="Not enough data"):
        chunk_and_tokenize(Dataset.from_dict({"text": ["the of and"]}), toy_tokenizer(), num_proc=1, max_seq_len=16,
                           load_from_cache_file=False)
    ds = chunk_and_tokenize(Dataset.from_dict({"text": ["the of and"]}), toy_tokenizer(), num_proc=1, max_seq_len=16,
                            return_final_batch=True, load_from_cache_file=False)
    assert [r["input_ids"].tolist() for r in ds] == [[0, 2, 3, 4]]
