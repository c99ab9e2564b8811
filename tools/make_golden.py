#!/usr/bin/env python
"""Generate tests/golden/*.npz from the REAL reference, imported from /root/reference.

Runs only in the build container (the reference does not exist on the GPU box); the
fixtures it writes are data only: inputs and the reference's outputs. The loading
recipe follows SURVEY.md (Appendix): the reference's Python files are imported as they
lie on disk; ``simple_parsing`` / ``natsort`` (absent, config sugar only) are replaced
by two trivial shims and ``sparsify/__init__.py`` (-> trainer -> schedulefree) is
bypassed. SPARSIFY_DISABLE_TRITON=1 selects the reference's own eager decoder, which
is what its tests/test_decode.py treats as ground truth for the Triton kernels.

    python tools/make_golden.py            # rewrites every fixture
"""
from __future__ import annotations

import dataclasses
import os
import sys
import types
from pathlib import Path

import numpy as np

os.environ["SPARSIFY_DISABLE_TRITON"] = "1"
sys.dont_write_bytecode = True

REF = "/root/reference"
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"


def _load_reference():
    sp = types.ModuleType("simple_parsing")

    class Serializable:
        def to_dict(self):
            return dataclasses.asdict(self)

        @classmethod
        def from_dict(cls, d, drop_extra_fields=False):
            names = {f.name for f in dataclasses.fields(cls)}
            return cls(**{k: v for k, v in d.items() if k in names})

    sp.Serializable = Serializable
    sp.list_field = lambda *a, **k: dataclasses.field(default_factory=lambda: list(a))
    sys.modules["simple_parsing"] = sp
    ns = types.ModuleType("natsort")
    ns.natsorted = lambda x, key=None: sorted(x, key=key)
    sys.modules["natsort"] = ns
    pkg = types.ModuleType("sparsify")
    pkg.__path__ = [f"{REF}/sparsify"]
    sys.modules["sparsify"] = pkg


_load_reference()
import torch  # noqa: E402
from sparsify.config import SparseCoderConfig  # noqa: E402
from sparsify.fused_encoder import fused_encoder  # noqa: E402
from sparsify.muon import Muon, quintic_newtonschulz  # noqa: E402
from sparsify.sign_sgd import SignSGD  # noqa: E402
from sparsify.sparse_coder import SparseCoder  # noqa: E402
from sparsify.utils import eager_decode  # noqa: E402

torch.set_num_threads(8)


def npy(t):
    if isinstance(t, torch.Tensor):
        t = t.detach()
        if t.dtype == torch.bfloat16:
            t = t.float()
        return t.cpu().numpy()
    return np.asarray(t)


def sort_by_index(vals, idx):
    order = idx.argsort(dim=-1)
    return vals.gather(-1, order), idx.gather(-1, order)


def save(name, **arrays):
    OUT.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT / f"{name}.npz", **{k: npy(v) for k, v in arrays.items()})
    size = (OUT / f"{name}.npz").stat().st_size
    print(f"  {name}.npz  {size/1024:.0f} KiB")


# ------------------------------------------------------------------ 1. fused encoder
def gen_encoder():
    for tag, (N, D, M, k), act in [
        ("s1_topk", (64, 48, 384, 10), "topk"),
        ("s2_topk", (128, 64, 512, 16), "topk"),
        ("s1_groupmax", (64, 48, 384, 12), "groupmax"),
        ("s2_groupmax", (128, 64, 512, 16), "groupmax"),
    ]:
        torch.manual_seed(42)
        x = torch.randn(N, D, requires_grad=True)
        W = torch.randn(M, D, requires_grad=True)
        b = torch.randn(M, requires_grad=True)
        c = torch.randn(M)          # per-latent loss coefficient (order independent)
        r = torch.randn(N)          # per-token loss coefficient
        vals, idx, pre = fused_encoder(x, W, b, k, act)
        loss = (r[:, None] * vals * c[idx]).sum()
        loss.backward()
        sv, si = sort_by_index(vals, idx)
        save(f"encoder_{tag}", x=x, W=W, b=b, c=c, r=r, k=k, values=sv, indices=si,
             pre_acts=pre, loss=loss, gx=x.grad, gW=W.grad, gb=b.grad)
    # the reference's own known-answer test (tests/test_encode.py:7-57) at CPU size:
    torch.manual_seed(42)
    N, D, M, k = 256, 64, 2048, 32
    x = torch.randn(N, D, requires_grad=True)
    W = torch.randn(M, D, requires_grad=True)
    b = torch.randn(M, requires_grad=True)
    out = torch.nn.functional.relu(torch.nn.functional.linear(x, W, b))
    nv, ni = torch.topk(out, k, dim=1, sorted=False)
    nv.sum().backward()
    naive = [t.grad.clone() for t in (x, W, b)]
    for t in (x, W, b):
        t.grad = None
    vals, idx, _ = fused_encoder(x, W, b, k, "topk")
    vals.sum().backward()
    torch.testing.assert_close(vals, nv)
    torch.testing.assert_close(idx, ni)
    for a, t in zip(naive, (x, W, b)):
        torch.testing.assert_close(t.grad, a)
    sv, si = sort_by_index(vals, idx)
    save("encoder_kat", x=x, W=W, b=b, k=k, values=sv, indices=si, gx=x.grad, gb=b.grad,
         gW_rowsum=W.grad.sum(1), gW_colsum=W.grad.sum(0))


# ------------------------------------------------------------------ 2. decoder
def gen_decode():
    for d_in in (48, 64):      # tests/test_decode.py:7-21
        torch.manual_seed(d_in)
        batch, d_sae, k = 2, 128, 10
        latents = torch.rand(batch, d_sae)
        W_dec = torch.randn(d_sae, d_in, requires_grad=True)
        top_vals, top_idx = latents.topk(k)
        top_vals = top_vals.clone().requires_grad_(True)
        res = eager_decode(top_idx, top_vals, W_dec.mT)
        gy = torch.randn_like(res)
        (res * gy).sum().backward()
        save(f"decode_{d_in}", W_dec=W_dec, top_vals=top_vals, top_idx=top_idx, out=res, gy=gy,
             gW=W_dec.grad, gvals=top_vals.grad)
    # a larger, ragged-width case with repeated latents across tokens
    torch.manual_seed(7)
    N, M, D, k = 96, 640, 200, 24
    W_dec = torch.randn(M, D, requires_grad=True)
    lat = torch.rand(N, M)
    lat[:, :40] += 1.0             # a few very popular latents -> long segments
    top_vals, top_idx = lat.topk(k)
    top_vals = top_vals.clone().requires_grad_(True)
    res = eager_decode(top_idx, top_vals, W_dec.mT)
    gy = torch.randn_like(res)
    (res * gy).sum().backward()
    save("decode_ragged", W_dec=W_dec, top_vals=top_vals, top_idx=top_idx, out=res, gy=gy,
         gW=W_dec.grad, gvals=top_vals.grad)


# ------------------------------------------------------------------ 3. SparseCoder fwd/bwd
def make_sae(D, M, k, *, transcode, skip, multi_topk, activation="topk", seed=0):
    torch.manual_seed(seed)
    cfg = SparseCoderConfig(activation=activation, num_latents=M, k=k, multi_topk=multi_topk,
                            skip_connection=skip, transcode=transcode)
    sae = SparseCoder(D, cfg)
    with torch.no_grad():     # move away from the trivial init so every term is exercised
        sae.encoder.bias.normal_(0, 0.05)
        sae.b_dec.normal_(0, 0.3)
        if transcode:
            sae.W_dec.normal_(0, 1.0 / D**0.5)
        if skip:
            sae.W_skip.normal_(0, 0.5 / D**0.5)
        if not transcode:
            sae.set_decoder_norm_to_unit_norm()
    return sae


def gen_sparse_coder():
    alpha = 1 / 32
    shapes = {"s1": (64, 48, 384, 10), "s2": (128, 64, 512, 16)}
    variants = {
        "plain": dict(),
        "auxk": dict(dead=True),
        "multitopk": dict(multi_topk=True),
        "skip": dict(skip=True),
        "auxk_multitopk": dict(dead=True, multi_topk=True),
        "groupmax": dict(activation="groupmax"),
        "fewdead": dict(dead=True, few=True),
    }
    for sname, (N, D, M, k) in shapes.items():
        for kind in ("sae", "transcoder"):
            for mode in ("fp32", "bf16"):
                for vname, v in variants.items():
                    if sname == "s2" and vname not in ("plain", "auxk"):
                        continue
                    if vname == "groupmax" and (kind != "sae" or mode != "fp32"):
                        continue
                    kk = 12 if (vname == "groupmax" and sname == "s1") else k
                    sae = make_sae(D, M, kk, transcode=(kind == "transcoder"),
                                   skip=v.get("skip", False), multi_topk=v.get("multi_topk", False),
                                   activation=v.get("activation", "topk"))
                    torch.manual_seed(123)
                    offs = torch.randn(D)
                    x = torch.randn(N, D) + offs
                    if kind == "transcoder":
                        R = torch.randn(D, D) / D**0.5
                        y = x @ R + 0.1 * torch.randn(N, D)
                    else:
                        y = None
                    dead = None
                    if v.get("dead"):
                        dead = torch.zeros(M, dtype=torch.bool)
                        if v.get("few"):
                            dead[5:M:37] = True      # fewer dead latents than d_in // 2
                        else:
                            dead[::3] = True
                    if mode == "bf16":
                        with torch.autocast("cpu", dtype=torch.bfloat16):
                            out = sae(x, y, dead_mask=dead)
                    else:
                        out = sae(x, y, dead_mask=dead)
                    loss = out.fvu + alpha * out.auxk_loss + out.multi_topk_fvu / 8
                    loss.backward()
                    acts, idx = sort_by_index(out.latent_acts, out.latent_indices)
                    arrays = dict(
                        x=x, k=kk, alpha=alpha,
                        W_enc=sae.encoder.weight, b_enc=sae.encoder.bias, W_dec=sae.W_dec,
                        b_dec=sae.b_dec,
                        sae_out=out.sae_out, latent_acts=acts, latent_indices=idx, fvu=out.fvu,
                        auxk_loss=out.auxk_loss, multi_topk_fvu=out.multi_topk_fvu,
                        g_W_enc=sae.encoder.weight.grad, g_b_enc=sae.encoder.bias.grad,
                        g_W_dec=sae.W_dec.grad, g_b_dec=sae.b_dec.grad,
                    )
                    if y is not None:
                        arrays["y"] = y
                    if dead is not None:
                        arrays["dead_mask"] = dead
                    if sae.W_skip is not None:
                        arrays["W_skip"] = sae.W_skip
                        arrays["g_W_skip"] = sae.W_skip.grad
                    save(f"coder_{sname}_{kind}_{mode}_{vname}", **arrays)


# ------------------------------------------------------------------ 4. dense updates
def gen_dense():
    torch.manual_seed(5)
    sae = make_sae(48, 384, 10, transcode=False, skip=False, multi_topk=False)
    with torch.no_grad():
        sae.W_dec.mul_(torch.rand(384, 1) * 3 + 0.1)      # un-normalise
        sae.W_dec[7].zero_()                             # a zero row exercises the eps
    W0 = sae.W_dec.detach().clone()
    sae.set_decoder_norm_to_unit_norm()
    W1 = sae.W_dec.detach().clone()
    sae.W_dec.grad = torch.randn(384, 48)
    G0 = sae.W_dec.grad.clone()
    sae.remove_gradient_parallel_to_decoder_directions()
    G1 = sae.W_dec.grad.clone()
    # SignSGD, 10 steps on fixed pseudo-gradients (including exact zeros)
    torch.manual_seed(6)
    p = torch.nn.Parameter(torch.randn(33, 17))
    p0 = p.detach().clone()
    opt = SignSGD([p], lr=3e-3)
    grads = torch.randn(10, 33, 17)
    grads[:, 3, :5] = 0.0
    snaps = []
    for t in range(10):
        p.grad = grads[t].clone()
        opt.step()
        snaps.append(p.detach().clone())
    save("dense_ops", W0=W0, W1=W1, G0=G0, G1=G1, p0=p0, grads=grads, p1=snaps[0], p10=snaps[9],
         lr=3e-3)


# ------------------------------------------------------------------ 5. Muon
def gen_muon():
    torch.manual_seed(11)
    A = torch.randn(64, 256)
    B = torch.randn(256, 64)
    nsA = quintic_newtonschulz(A, 5)
    nsB = quintic_newtonschulz(B, 5)
    p = torch.nn.Parameter(torch.randn(96, 48))
    p0 = p.detach().clone()
    opt = Muon([p], lr=2e-3, ddp=False)
    grads = torch.randn(3, 96, 48)
    snaps = []
    for t in range(3):
        p.grad = grads[t].clone()
        opt.step()
        snaps.append(p.detach().clone())
    save("muon", A=A, B=B, nsA=nsA, nsB=nsB, p0=p0, grads=grads, p1=snaps[0], p3=snaps[2], lr=2e-3)


# ------------------------------------------------------------------ 6. FVU trajectory
def gen_trajectory():
    rng = np.random.RandomState(0)
    D, M, k, N, steps = 64, 2048, 8, 512, 200
    n_data = 2048
    dictionary = rng.randn(256, D).astype(np.float32)
    dictionary /= np.linalg.norm(dictionary, axis=1, keepdims=True)
    codes = rng.rand(n_data, 256).astype(np.float32) * (rng.rand(n_data, 256) < 0.03)
    data = codes @ dictionary + 0.05 * rng.randn(n_data, D).astype(np.float32) \
        + rng.randn(D).astype(np.float32)
    data_t = torch.from_numpy(data.astype(np.float32))
    lr = 5e-3 / (M / 2**14) ** 0.5          # trainer.py:152
    curves = {}
    for mode in ("fp32", "bf16"):
        torch.manual_seed(0)
        cfg = SparseCoderConfig(num_latents=M, k=k)
        sae = SparseCoder(D, cfg)
        opt = SignSGD(sae.parameters(), lr=lr)
        fvus = []
        for step in range(steps):
            xb = data_t[(step * N) % n_data:(step * N) % n_data + N]
            if step == 0:                    # trainer.py:394-404
                sae.b_dec.data = xb.mean(0).to(sae.dtype)
            sae.set_decoder_norm_to_unit_norm()     # trainer.py:406-408
            if mode == "bf16":
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    out = sae(xb)
            else:
                out = sae(xb)
            out.fvu.backward()
            sae.remove_gradient_parallel_to_decoder_directions()   # trainer.py:513-515
            opt.step()
            opt.zero_grad()
            fvus.append(float(out.fvu))
        curves[mode] = np.array(fvus, dtype=np.float32)
        print(f"  trajectory {mode}: fvu {fvus[0]:.4f} -> {fvus[-1]:.4f}")
    save("trajectory", data=data_t, lr=lr, D=D, M=M, k=k, N=N, fvu_fp32=curves["fp32"],
         fvu_bf16=curves["bf16"])


# ------------------------------------------------------------------ 7. checkpoints
def gen_checkpoint():
    """Checkpoints written by the reference's own save_to_disk (sparse_coder.py:167-179):
    the files a user of the reference holds. tests/ load them through the build's
    load_from_disk. The reverse direction (the build writes, the reference loads) can only be
    checked here, where the reference is importable; it is asserted below."""
    import json
    import tempfile
    D, M, k, N = 48, 384, 10, 64
    for tag, kw in (("sae", dict(transcode=False, skip=False)),
                    ("transcoder_skip", dict(transcode=True, skip=True))):
        sae = make_sae(D, M, k, multi_topk=False, seed=3, **kw)
        path = OUT / f"ckpt_{tag}"
        sae.save_to_disk(path)
        torch.manual_seed(321)
        x = torch.randn(N, D) + torch.randn(D)
        y = x @ (torch.randn(D, D) / D**0.5) if kw["transcode"] else None
        back = SparseCoder.load_from_disk(path)
        out = back(x, y)
        acts, idx = sort_by_index(out.latent_acts, out.latent_indices)
        arrays = dict(x=x, sae_out=out.sae_out, latent_acts=acts, latent_indices=idx, fvu=out.fvu)
        if y is not None:
            arrays["y"] = y
        save(f"ckpt_{tag}_io", **arrays)
        for f in sorted(path.iterdir()):
            print(f"  {path.name}/{f.name}  {f.stat().st_size/1024:.0f} KiB")

        # reverse: a checkpoint written by the build must load in the reference, bit for bit
        sys.path.insert(0, str(OUT.parent.parent))
        from sparsify_amd import SparseCoder as Ours
        ours = Ours.load_from_disk(path)
        with tempfile.TemporaryDirectory() as tmp:
            ours.save_to_disk(tmp)
            theirs = SparseCoder.load_from_disk(tmp)
            with open(f"{tmp}/cfg.json") as fa, open(path / "cfg.json") as fb:
                ca, cb = json.load(fa), json.load(fb)
            assert ca == cb, (ca, cb)
        ref_sd, got_sd = sae.state_dict(), theirs.state_dict()
        assert ref_sd.keys() == got_sd.keys()
        for key in ref_sd:
            assert torch.equal(ref_sd[key], got_sd[key]), key
        print(f"  reverse interop ({tag}): build-written checkpoint loads in the reference, equal")


def gen_tokenize():
    """sparsify/data.py:14-101 on the toy tokenizer and documents of tests/conftest.py: the rows the
    reference's chunk_and_tokenize produces, for three settings."""
    import json

    from datasets import Dataset
    from sparsify.data import chunk_and_tokenize
    sys.path.insert(0, str(OUT.parent))
    from conftest import toy_documents, toy_tokenizer
    docs = toy_documents()
    out = {}
    for name, kw, max_len in (("rows16", dict(max_seq_len=16), 64), ("rows16_tail", dict(max_seq_len=16, return_final_batch=True), 64),
                              ("model_max_24", dict(max_seq_len=2048), 24)):
        ds = chunk_and_tokenize(Dataset.from_dict({"text": docs, "other": list(range(len(docs)))}), toy_tokenizer(max_len),
                                num_proc=1, load_from_cache_file=False, **kw)
        # (a fast tokenizer's `overflow_to_sample_mapping` stays as a hidden column: the format shows input_ids only)
        assert all(list(row) == ["input_ids"] for row in ds), ds.column_names
        out[name] = [[int(t) for t in row["input_ids"]] for row in ds]
        print(f"  tokenize/{name}: {len(out[name])} rows, last of {len(out[name][-1])}")
    (OUT / "tokenize.json").write_text(json.dumps(out))


if __name__ == "__main__":
    which = sys.argv[1:] or ["encoder", "decode", "coder", "dense", "muon", "trajectory", "checkpoint", "tokenize"]
    table = dict(encoder=gen_encoder, decode=gen_decode, coder=gen_sparse_coder, dense=gen_dense,
                 muon=gen_muon, trajectory=gen_trajectory, checkpoint=gen_checkpoint, tokenize=gen_tokenize)
    for w in which:
        print(f"[{w}]")
        table[w]()
