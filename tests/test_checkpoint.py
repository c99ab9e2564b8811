"""Checkpoint interop: tests/golden/ckpt_* were written by the
reference's own SparseCoder.save_to_disk (sparse_coder.py:167-179, tools/make_golden.py
gen_checkpoint); the build must load them, write the same layout back, and compute the
reference's outputs from them. The reverse direction (build writes, reference loads) is
asserted inside the generator, the only place where the reference is importable."""
import json

import pytest
import torch
from safetensors import safe_open

from conftest import GOLDEN, load_golden

TAGS = ["sae", "transcoder_skip"]


def stored_tensors(path):
    with safe_open(str(path / "sae.safetensors"), framework="pt", device="cpu") as f:
        return {k: f.get_tensor(k) for k in f.keys()}


@pytest.mark.parametrize("tag", TAGS)
def test_load_reference_checkpoint(tag):
    from sparsify_amd import SparseCoder
    path = GOLDEN / f"ckpt_{tag}"
    sae = SparseCoder.load_from_disk(path)
    with open(path / "cfg.json") as f:
        cfg = json.load(f)
    assert sae.d_in == cfg["d_in"] and sae.num_latents == cfg["num_latents"]
    assert sae.cfg.k == cfg["k"] and sae.cfg.transcode == cfg["transcode"]
    assert sae.cfg.skip_connection == cfg["skip_connection"] and sae.cfg.activation == cfg["activation"]
    stored = stored_tensors(path)
    state = sae.state_dict()
    assert set(state) == set(stored)                      # W_dec, b_dec, encoder.*[, W_skip]
    for k, v in stored.items():
        assert state[k].dtype == v.dtype and torch.equal(state[k], v), k


@pytest.mark.parametrize("tag", TAGS)
def test_save_writes_the_reference_layout(tag, tmp_path):
    from sparsify_amd import SparseCoder
    path = GOLDEN / f"ckpt_{tag}"
    SparseCoder.load_from_disk(path).save_to_disk(tmp_path / "out")
    with open(path / "cfg.json") as fa, open(tmp_path / "out" / "cfg.json") as fb:
        assert json.load(fa) == json.load(fb)             # same keys, same values, d_in included
    a, b = stored_tensors(path), stored_tensors(tmp_path / "out")
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_load_many_local_and_encoder_only():
    from sparsify_amd import SparseCoder
    many = SparseCoder.load_many(str(GOLDEN), local=True, layers=[f"ckpt_{t}" for t in TAGS])
    assert list(many) == sorted(f"ckpt_{t}" for t in TAGS)
    listed = SparseCoder.load_many(str(GOLDEN), local=True)
    assert set(listed) == {f"ckpt_{t}" for t in TAGS}     # every directory, natural order
    enc_only = SparseCoder.load_from_disk(GOLDEN / "ckpt_sae", decoder=False)   # strict=decoder
    assert enc_only.W_dec is None
    assert torch.equal(enc_only.encoder.weight, many["ckpt_sae"].encoder.weight)
    with pytest.raises(NotImplementedError):
        SparseCoder.load_many("some/repo", local=False)   # hub download: no network, out of scope


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_reference_checkpoint_forward(device, tag):
    """Loaded on the GPU, the checkpoint reproduces the outputs the reference computed from it
    (fp32 forward, same tolerance as test_hip_api.test_sparse_coder_fp32_vs_reference)."""
    from sparsify_amd import SparseCoder
    sae = SparseCoder.load_from_disk(GOLDEN / f"ckpt_{tag}", device="cuda")
    g = load_golden(f"ckpt_{tag}_io")
    fwd = type(sae).forward.__wrapped__
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=False):
        out = fwd(sae, g["x"].cuda(), g["y"].cuda() if "y" in g else None)
    order = out.latent_indices.argsort(-1)
    idx = out.latent_indices.gather(-1, order).cpu()
    acts = out.latent_acts.gather(-1, order).float().cpu()
    assert torch.equal(idx, g["latent_indices"])
    tol = dict(rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(acts, g["latent_acts"], **tol)
    torch.testing.assert_close(out.sae_out.cpu(), g["sae_out"], **tol)
    torch.testing.assert_close(float(out.fvu.detach()), g["fvu"], rtol=2e-4, atol=1e-6)
