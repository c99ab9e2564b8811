"""SparseCoder with the reference's surface (sparsify/sparse_coder.py:19-299): same
parameters, same method names, same ForwardOutput -- executed by the HIP kernels of
libsparsify_hip (encoder GEMM + top-k, sparse decode, fused FVU/AuxK reductions,
segmented weight gradients)."""
from __future__ import annotations

import json
from fnmatch import fnmatch
from pathlib import Path
from typing import NamedTuple

import torch
from torch import Tensor, nn

from . import dead_decode, ops, parallel
from .config import SparseCoderConfig
from .fused_encoder import EncoderOutput, FusedEncoder, bf16_shadow, block_maxima_of, low_precision_input
from .utils import SparseDecode, decoder_impl


class ForwardOutput(NamedTuple):
    sae_out: Tensor

    latent_acts: Tensor
    """Activations of the top-k latents."""

    latent_indices: Tensor
    """Indices of the top-k features."""

    fvu: Tensor
    """Fraction of variance unexplained."""

    auxk_loss: Tensor
    """AuxK loss, if applicable."""

    multi_topk_fvu: Tensor
    """Multi-TopK FVU, if applicable."""


class ResidualLoss(torch.autograd.Function):
    """loss = sum((target - pred)^2) / tv, plus the residual itself (non-differentiable).

    Fuses sparse_coder.py:225 + :253-254 (and :248-249, :260): one pass produces e and the
    sum of squares; the backward is one scaled copy, dL/dpred = (-2 g / tv) e."""

    @staticmethod
    def forward(ctx, pred: Tensor, target: Tensor, tv: Tensor):
        p2 = pred.reshape(-1, pred.shape[-1]).contiguous()
        t2 = target.reshape(-1, target.shape[-1]).contiguous()
        e, sq = ops.residual_sumsq(t2, p2, want_residual=True)
        loss = ops.scale_by_ratio(sq, 1.0, None, tv)
        ctx.save_for_backward(e, tv)
        ctx.pred_shape = pred.shape
        e_out = e.reshape(pred.shape)
        ctx.mark_non_differentiable(e_out)
        ctx.set_materialize_grads(False)
        return loss.reshape(()), e_out

    @staticmethod
    def backward(ctx, g_loss, g_e):
        e, tv = ctx.saved_tensors
        if g_loss is None:
            return None, None, None
        g = g_loss.reshape(1).contiguous().float()
        return ops.scale_by_ratio(e, -2.0, g, tv).reshape(ctx.pred_shape), None, None


class _AddSkip(torch.autograd.Function):
    """sae_out + skip with sae_out fp32 and skip bf16 (autocast) or fp32, one pass; both
    gradients are the incoming one (the product's backward casts it as autocast would)."""

    @staticmethod
    def forward(ctx, sae_out: Tensor, skip: Tensor):
        ctx.set_materialize_grads(False)
        return ops.add(sae_out.contiguous(), skip.contiguous())

    @staticmethod
    def backward(ctx, g):
        return g, g


class SkipProduct(torch.autograd.Function):
    """x @ W_skip.mT (sparse_coder.py:221-222) on the package's own GEMM kernels.

    Under bf16 autocast the reference's matmul takes bf16 operands and returns a bf16 tensor, and
    its backward runs in bf16 too (the fp32 gradient of sae_out is cast, W_skip.grad = g^T x is
    produced in bf16 and cast to the fp32 parameter): sphip_gemm_bf16_axpy with those roundings.
    Without autocast everything is the exact-fp32 MFMA product (sphip_encoder_preacts, relu off)."""

    @staticmethod
    def forward(ctx, x: Tensor, W: Tensor, low: bool):
        parallel.wait_for(W)
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        ctx.low, ctx.x_shape = low, x.shape
        if low:
            xl = low_precision_input(x2, None)
            wl = ops.cast_bf16(W.detach().contiguous())
            out = ops.gemm_bf16_axpy(xl, wl)
            ctx.save_for_backward(xl, wl)
        else:
            xf = x2 if x2.dtype == torch.float32 else ops.center_cast(x2, None, True, False)[0]
            out = ops.encoder_preacts(xf, W.detach().contiguous(), None, relu=False)
            ctx.save_for_backward(xf, W)
        ctx.set_materialize_grads(False)
        return out.reshape(*x.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        xs, ws = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1]).contiguous()
        gx = gW = None
        if ctx.low:
            gl = g2 if g2.dtype == torch.bfloat16 else ops.center_cast(g2 if g2.dtype == torch.float32 else g2.float(),
                                                                       None, False, True)[1]
            if ctx.needs_input_grad[1]:
                # g^T x contracts over the tokens: both operands transposed, the token count
                # padded with zero columns to the kernel's multiple of 8
                gW = ops.gemm_bf16_axpy(ops.transpose(gl, pad_to=8), ops.transpose(xs, pad_to=8)).float()
            if ctx.needs_input_grad[0]:
                gx = ops.gemm_bf16_axpy(gl, ops.transpose(ws)).reshape(ctx.x_shape)
        else:
            gf = g2 if g2.dtype == torch.float32 else g2.float()
            if ctx.needs_input_grad[1]:
                gW = ops.encoder_preacts(ops.transpose(gf, pad_to=4), ops.transpose(xs, pad_to=4), None, relu=False)
            if ctx.needs_input_grad[0]:
                gx = ops.encoder_preacts(gf, ops.transpose(ws.detach().contiguous()), None, relu=False).reshape(ctx.x_shape)
        return gx, gW, None


class SparseCoder(nn.Module):
    lazy_pre_acts: bool | None = None
    """(not in the reference) True: forwards that do not read pre_acts do not store them
    (fused_encoder.LazyPreActs); None: the module default fused_encoder.LAZY_PRE_ACTS."""

    def __init__(
        self,
        d_in: int,
        cfg: SparseCoderConfig,
        device: str | torch.device = "cpu",
        dtype: torch.dtype | None = None,
        *,
        decoder: bool = True,
    ):
        super().__init__()
        self.cfg = cfg
        self.d_in = d_in
        self.num_latents = cfg.num_latents or d_in * cfg.expansion_factor

        self.encoder = nn.Linear(d_in, self.num_latents, device=device, dtype=dtype)
        self.encoder.bias.data.zero_()

        if decoder:
            if cfg.transcode:
                # transcoder: zeros (sparse_coder.py:58-59)
                self.W_dec = nn.Parameter(torch.zeros_like(self.encoder.weight.data))
            else:
                # autoencoder: copy of the encoder, rows normalised (sparse_coder.py:62-65).
                # Initialisation runs once and may happen on the host before .to(device),
                # so it is written with plain tensor ops; the per-step normalisation below
                # (set_decoder_norm_to_unit_norm) is the HIP kernel.
                w = self.encoder.weight.data.clone()
                if cfg.normalize_decoder:
                    eps = torch.finfo(w.dtype).eps
                    w /= torch.norm(w, dim=1, keepdim=True) + eps
                self.W_dec = nn.Parameter(w)
        else:
            self.W_dec = None

        self.b_dec = nn.Parameter(torch.zeros(d_in, dtype=dtype, device=device))
        self.W_skip = (
            nn.Parameter(torch.zeros(d_in, d_in, device=device, dtype=dtype))
            if cfg.skip_connection
            else None
        )

    # ------------------------------------------------------------------ (de)serialisation
    # Checkpoint layout of the reference: {path}/cfg.json + {path}/sae.safetensors with keys
    # W_dec, b_dec, encoder.weight, encoder.bias[, W_skip] (sparse_coder.py:134-179).
    @staticmethod
    def load_many(name: str, local: bool = True, layers: list[str] | None = None,
                  device: str | torch.device = "cpu", *, decoder: bool = True,
                  pattern: str | None = None) -> dict[str, "SparseCoder"]:
        if not local:
            raise NotImplementedError("hub downloads are out of scope (no network); pass local=True")
        repo_path = Path(name)
        pat = pattern + "/*" if pattern is not None else None
        if layers is not None:
            return {layer: SparseCoder.load_from_disk(repo_path / layer, device=device, decoder=decoder)
                    for layer in sorted(layers, key=_natural_key)}
        dirs = [f for f in repo_path.iterdir() if f.is_dir() and (pat is None or fnmatch(f.name, pat))]
        return {f.name: SparseCoder.load_from_disk(f, device=device, decoder=decoder)
                for f in sorted(dirs, key=lambda f: _natural_key(f.name))}

    @staticmethod
    def load_from_disk(path: Path | str, device: str | torch.device = "cpu", *, decoder: bool = True) -> "SparseCoder":
        from safetensors import safe_open
        from safetensors.torch import load_model

        path = Path(path)
        with open(path / "cfg.json", "r") as f:
            cfg_dict = json.load(f)
            d_in = cfg_dict.pop("d_in")
            cfg = SparseCoderConfig.from_dict(cfg_dict, drop_extra_fields=True)
        st_path = str(path / "sae.safetensors")
        with safe_open(st_path, framework="pt", device="cpu") as f:
            first_key = next(iter(f.keys()))
            reference_dtype = f.get_tensor(first_key).dtype
        sae = SparseCoder(d_in, cfg, device=device, decoder=decoder, dtype=reference_dtype)
        load_model(model=sae, filename=st_path, device=str(device), strict=decoder)
        return sae

    def save_to_disk(self, path: Path | str):
        from safetensors.torch import save_model

        path = Path(path)
        path.mkdir(parents=True, exist_ok=True)
        parallel.wait_for_updates()               # overlapped optimizer updates (sign_sgd.py)
        save_model(self, str(path / "sae.safetensors"))
        with open(path / "cfg.json", "w") as f:
            json.dump({**self.cfg.to_dict(), "d_in": self.d_in}, f)

    @property
    def device(self):
        return self.encoder.weight.device

    @property
    def dtype(self):
        return self.encoder.weight.dtype

    # ------------------------------------------------------------------ hot path
    def encode(self, x: Tensor, lazy_pre_acts: bool | None = None) -> EncoderOutput:
        """Encode the input and select the top-k latents (sparse_coder.py:189-196).
        The `x - b_dec` of the reference is fused into the encoder op. lazy_pre_acts (not in the
        reference): fused_encoder.FusedEncoder.forward."""
        b_dec = None if self.cfg.transcode else self.b_dec
        if lazy_pre_acts is None:
            lazy_pre_acts = self.lazy_pre_acts
        return EncoderOutput(*FusedEncoder.apply(
            x, self.encoder.weight, self.encoder.bias, self.cfg.k, self.cfg.activation, b_dec, lazy_pre_acts))

    def decode(self, top_acts: Tensor, top_indices: Tensor) -> Tensor:
        """sparse_coder.py:198-202; the bias add is fused into the gather kernel."""
        assert self.W_dec is not None, "Decoder weight was not initialized."
        idx = top_indices.reshape(-1, top_indices.shape[-1]).contiguous()
        # (bf16 activations under autocast go to the kernels as they are: the cast to fp32 is exact and
        # happens in registers, and 2-byte coefficients make the decode's slot lists 4-byte entries)
        keep_bf16 = top_acts.dtype == torch.bfloat16 and self.W_dec.dtype == torch.float32
        acts = (top_acts if keep_bf16 else top_acts.to(self.dtype)).reshape(-1, top_acts.shape[-1])
        y = SparseDecode.apply(idx, self.W_dec, acts, self.b_dec)
        dead_decode.note_decode(self.W_dec, y.grad_fn)      # whose backward takes the AuxK rows along (dead_decode.py)
        return y.reshape(*top_indices.shape[:-1], self.d_in)

    # Same autocast wrapper as the reference (sparse_coder.py:204-209): the encoder linear
    # (and the skip matmul) run in bf16, everything else in fp32.
    @torch.autocast("cuda", dtype=torch.bfloat16, enabled=torch.cuda.is_available())
    def forward(self, x: Tensor, y: Tensor | None = None, *, dead_mask: Tensor | None = None,
                num_dead: int | None = None) -> ForwardOutput:
        """num_dead (not in the reference): int(dead_mask.sum()) if the caller already holds it;
        otherwise it is read here, which makes the host wait for the device mid-forward."""
        # AuxK and multi-top-k read pre_acts (sparse_coder.py:240, 257): they get the stored matrix
        reads_pre = self.cfg.multi_topk or dead_mask is not None
        top_acts, top_indices, pre_acts = self.encode(x, lazy_pre_acts=False if reads_pre else None)

        if y is None:
            y = x

        sae_out = decoded = self.decode(top_acts, top_indices)
        if self.W_skip is not None:
            from .fused_encoder import _autocast_bf16
            skip = SkipProduct.apply(x, self.W_skip, _autocast_bf16())
            sae_out = _AddSkip.apply(sae_out, skip)

        y2 = y.reshape(-1, y.shape[-1]).contiguous()
        total_variance = ops.total_variance(y2)                     # sparse_coder.py:228
        fvu, e = ResidualLoss.apply(sae_out, y, total_variance)      # :225, :253-254

        if dead_mask is not None and num_dead is None:
            num_dead = int(dead_mask.sum())
        if dead_mask is not None and num_dead > 0:
            k_aux = y.shape[-1] // 2                                  # :233
            scale = min(num_dead / k_aux, 1.0)
            k_aux = min(k_aux, num_dead)
            pre2 = pre_acts.reshape(-1, pre_acts.shape[-1])
            # the gradient w.r.t. auxk_acts would flow into pre_acts, which the encoder
            # backward drops (fused_encoder.py:53,87) -- so it is never computed here.
            if dead_decode.eligible(pre2, k_aux, self.W_dec, num_dead):
                # few dead latents: selection on their recomputed, compact pre-activations and
                # the decode + W_dec gradient as dense products (dead_decode.py)
                slot_of, dead_list = ops.dead_compact(dead_mask, num_dead)
                x_low = low_precision_input(x.reshape(-1, x.shape[-1]).contiguous(),
                                            None if self.cfg.transcode else self.b_dec)
                auxk_acts, slots = dead_decode.auxk_select_compact(
                    x_low, bf16_shadow(self.encoder.weight), bf16_shadow(self.encoder.bias), dead_list, num_dead, k_aux)
                e_hat = dead_decode.DeadLatentDecode.apply(slots, auxk_acts, self.W_dec, self.b_dec, dead_list, None,
                                                           num_dead).reshape(e.shape)
            else:
                auxk_acts, auxk_indices = ops.topk(pre2, k_aux, keep_mask=dead_mask)      # :240-243
                e_hat = self.decode(auxk_acts, auxk_indices).reshape(e.shape)
            auxk_sq, _ = ResidualLoss.apply(e_hat, e, total_variance)                  # :248-249
            auxk_loss = scale * auxk_sq
        else:
            auxk_loss = sae_out.new_zeros(())      # a fill kernel: new_tensor(0.0) is a blocking host copy

        if self.cfg.multi_topk:
            pre2 = pre_acts.reshape(-1, pre_acts.shape[-1])
            # (the block maxima the GEMM left for this matrix spare the selection its first sweep)
            acts4, idx4 = ops.topk(pre2, 4 * self.cfg.k, tile_max=block_maxima_of(pre_acts))   # :257
            top_acts4 = acts4.reshape(*x.shape[:-1], -1)
            top_indices4 = idx4.reshape(*x.shape[:-1], -1)
            if self.cfg.activation == "topk" and acts4.dtype == top_acts.dtype and self.W_dec is not None:
                # the 4k set holds the top-k set (one selection rule, one matrix), whose rows `decoded` has summed:
                # the forward gathers the other 3k rows only and adds. The backward is the whole 4k list's -- its
                # W_dec gradient weighs every row with this branch's own residual, and nothing of it may reach the
                # encoder through top_acts (the reference drops the gradient of pre_acts, fused_encoder.py:53)
                rest = ops.exclude_selected(acts4, idx4, top_acts.reshape(-1, top_acts.shape[-1]).contiguous(),
                                            top_indices.reshape(-1, top_indices.shape[-1]).contiguous())
                keep_bf16 = acts4.dtype == torch.bfloat16 and self.W_dec.dtype == torch.float32
                sae_out = SparseDecode.apply(idx4, self.W_dec, acts4 if keep_bf16 else acts4.to(self.dtype), self.b_dec,
                                             (rest, decoded)).reshape(decoded.shape)
            else:
                sae_out = self.decode(top_acts4, top_indices4)
            top_acts, top_indices = top_acts4, top_indices4
            multi_topk_fvu, _ = ResidualLoss.apply(sae_out, y, total_variance)         # :260
        else:
            multi_topk_fvu = sae_out.new_zeros(())

        return ForwardOutput(sae_out, top_acts, top_indices, fvu, auxk_loss, multi_topk_fvu)

    @torch.no_grad()
    def set_decoder_norm_to_unit_norm(self):
        """sparse_coder.py:273-279, one fused pass over W_dec."""
        assert self.W_dec is not None, "Decoder weight was not initialized."
        parallel.wait_for(self.W_dec)
        ops.unit_norm_rows_(self.W_dec.data)
        torch.autograd.graph.increment_version(self.W_dec)

    @torch.no_grad()
    def remove_gradient_parallel_to_decoder_directions(self):
        """sparse_coder.py:281-295, one fused pass over (W_dec.grad, W_dec)."""
        assert self.W_dec is not None, "Decoder weight was not initialized."
        assert self.W_dec.grad is not None
        parallel.wait_for(self.W_dec)
        ops.remove_parallel_grad_(self.W_dec.grad, self.W_dec.data)


def _natural_key(s: str):
    import re
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s)]


# Allow for alternate naming conventions
Sae = SparseCoder
