"""Latent-dim sharded sparse coder (BASELINE.json configs[4]: a 262144-latent transcoder with
W_enc / W_dec split over 8 GPUs). New functionality -- the reference has no such mode; the
contract is "identical to the unsharded SparseCoder up to ties" (SURVEY.md section 8e).

Rank r owns latents [r*M/W, (r+1)*M/W) of encoder.weight, encoder.bias and W_dec; b_dec (and
the token batch) are replicated. Per forward:
  1. local pre-activations and local top-k (the same HIP kernels, on M/W latents);
  2. all-gather of the k local (value, global index) candidates per token  [RCCL];
  3. merge: top-k of the W*k candidates -> the global selection, identical on every rank;
  4. partial decode of the selected latents this rank owns (other slots are inactive: index
     -1, see sphip_gather_rows_weighted), then an all-reduce(SUM) of the partial
     reconstructions [RCCL];
  5. loss on the full reconstruction (replicated, cheap).
Backward needs no collective at all: dL/d sae_out is replicated, every rank reduces the
gradient rows of its own latents, and the gradient of a selected activation is needed only
by its owner. The replicated b_dec is the one shared parameter; its gradient (D floats) is
summed over ranks once per step by reduce_shared_grads().

The other terms of SparseCoder.forward take the same route (sharded_select below): AuxK
(sparse_coder.py:230-251) is a second selection, k_aux of the dead latents -- every shard offers
its k_aux best dead latents (its slice of the replicated dead mask), the merge keeps the k_aux
best of all, the partial reconstructions of the residual are summed; multi-top-k
(sparse_coder.py:256-262) is the same with 4k and no mask. Neither sends a gradient into the
encoder (the reference drops grad_preacts, fused_encoder.py:53,87). The skip connection
(sparse_coder.py:219-222) is replicated: every rank holds W_skip and computes x W_skip^T and its
gradient in full, so the replicas stay identical without a collective.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
from torch import Tensor, nn

from . import ops, parallel
from .config import SparseCoderConfig
from .fused_encoder import _autocast_bf16, bf16_shadow
from .sparse_coder import ForwardOutput, ResidualLoss, SkipProduct, _AddSkip
from .utils import SparseDecode


def merge_candidates(cand_v: Tensor, cand_i: Tensor, k: int, lo: int, hi: int):
    """Global top-k of the gathered candidates (N, W*kk) -> (values, global indices, local view).

    The candidates arrive shard-major, and inside a shard in the local top-k's order (ties in
    ascending index order), so the top-k kernel's tie rule -- lowest position first -- is the
    unsharded rule, lowest global index first: the merged SET equals sphip_topk over the whole
    row. The local view holds `index - lo` for the latents this rank owns and -1 (inactive slot,
    see sphip_gather_rows_weighted) elsewhere."""
    mv, mpos = ops.topk(cand_v.contiguous(), k)
    gidx = parallel.candidate_indices(cand_i, mpos)
    mine = (gidx >= lo) & (gidx < hi)
    local_idx = torch.where(mine, gidx - lo, torch.full_like(gidx, -1))
    return mv, gidx, local_idx


def sharded_select(pre: Tensor, k_sel: int, lo: int, hi: int, keep_local: Tensor | None = None):
    """Global top-k_sel over the shards' pre-activations, optionally among the latents a mask keeps:
    local top-k (at most k_sel candidates per shard), candidate all-gather, merge. A shard with
    fewer kept latents than it offers fills up with masked-out entries (value -inf), which no merge
    prefers to a kept latent of another shard."""
    kk = min(k_sel, pre.shape[1])
    lv, li = ops.topk(pre, kk, keep_mask=keep_local)
    cand_v, cand_i = parallel.gather_topk_candidates(lv, li, lo)
    return merge_candidates(cand_v, cand_i, k_sel, lo, hi)


class _ShardedEncode(torch.autograd.Function):
    """Local encoder GEMM + local top-k + candidate exchange + merge. Returns the merged
    (values, global indices) and this rank's local view of the selection."""

    @staticmethod
    def forward(ctx, x, weight, bias, k: int, lo: int, hi: int, b_dec, want_pre: bool = False):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        low = _autocast_bf16()
        if b_dec is not None:
            rows, x_low = ops.center_cast(x2, b_dec.detach().contiguous(), want_f32=True, want_bf16=low)
        else:
            rows = x2
            x_low = (x2 if x2.dtype == torch.bfloat16 else ops.center_cast(x2, None, False, True)[1]) if low else None
        if low:
            pre, tile_max = ops.encoder_preacts(x_low, bf16_shadow(weight), bf16_shadow(bias), want_tile_max=True)
        else:
            xf = rows if rows.dtype == torch.float32 else ops.center_cast(rows, None, True, False)[0]
            pre, tile_max = ops.encoder_preacts(xf, weight.detach().contiguous(), bias.detach().contiguous()), None
        kk = min(k, pre.shape[1])
        lv, li = ops.topk(pre, kk, tile_max=tile_max)
        cand_v, cand_i = parallel.gather_topk_candidates(lv, li, lo)          # (N, W*kk)
        mv, gidx, local_idx = merge_candidates(cand_v, cand_i, k, lo, hi)
        ops.forget_segments(local_idx)
        ctx.save_for_backward(rows, weight, local_idx, b_dec)
        ctx.k = k
        ctx.set_materialize_grads(False)
        if want_pre:
            # the shard's pre-activations, for the AuxK / multi-top-k selections (no gradient:
            # fused_encoder.py:53,87)
            ctx.mark_non_differentiable(gidx, local_idx, pre)
            return mv, gidx, local_idx, pre
        ctx.mark_non_differentiable(gidx, local_idx)
        return mv, gidx, local_idx, None

    @staticmethod
    def backward(ctx, grad_values, _g1, _g2, _g3):
        rows, weight, local_idx, b_dec = ctx.saved_tensors
        if grad_values is None:
            return (None,) * 8
        gv = grad_values.reshape(-1, ctx.k).contiguous()
        seg = ops.cached_segments(local_idx, weight.shape[0])
        gW, gb, _ = ops.segmented_weight_grad(seg, gv, rows, want_bias=True)
        g_bdec = None
        if b_dec is not None and ctx.needs_input_grad[6]:
            # this shard's part of -sum_n grad_x[n,:]; summed over shards by reduce_shared_grads()
            g_bdec = ops.vec_mat(gb, weight.detach(), -1.0)
        return None, gW, gb, None, None, None, g_bdec, None


class _AllReduceSum(torch.autograd.Function):
    """y = sum over ranks of the partial reconstructions; dL/dpartial = dL/dy on every rank."""

    @staticmethod
    def forward(ctx, partial):
        out = partial.clone()
        parallel.all_reduce_(out, "sum")
        return out

    @staticmethod
    def backward(ctx, g):
        return g


class _ScaleGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, scale: float):
        ctx.scale = scale
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.scale, None


class ShardedSparseCoder(nn.Module):
    def __init__(self, d_in: int, cfg: SparseCoderConfig, device="cuda", dtype=torch.float32):
        super().__init__()
        self.cfg, self.d_in = cfg, d_in
        self.num_latents = cfg.num_latents or d_in * cfg.expansion_factor
        self.lo, self.hi = parallel.shard_range(self.num_latents)
        m = self.hi - self.lo
        # same init as SparseCoder.__init__ (sparse_coder.py:53-74), drawn for the full
        # dictionary so that the shards are slices of the unsharded initialisation
        full = nn.Linear(d_in, self.num_latents, device=device, dtype=dtype)
        self.encoder = nn.Linear(d_in, m, device=device, dtype=dtype)
        with torch.no_grad():
            self.encoder.weight.copy_(full.weight[self.lo:self.hi])
            self.encoder.bias.zero_()
        if cfg.transcode:
            self.W_dec = nn.Parameter(torch.zeros_like(self.encoder.weight.data))
        else:
            w = self.encoder.weight.data.clone()
            if cfg.normalize_decoder:
                w /= torch.norm(w, dim=1, keepdim=True) + torch.finfo(w.dtype).eps
            self.W_dec = nn.Parameter(w)
        self.b_dec = nn.Parameter(torch.zeros(d_in, dtype=dtype, device=device))
        # replicated on every rank (sparse_coder.py:76-80)
        self.W_skip = (nn.Parameter(torch.zeros(d_in, d_in, device=device, dtype=dtype))
                       if cfg.skip_connection else None)
        del full

    @property
    def dtype(self):
        return self.encoder.weight.dtype

    @property
    def device(self):
        return self.encoder.weight.device

    def save_to_disk(self, path):
        """Collective: the shards are gathered and rank 0 writes the reference's unsharded
        checkpoint (cfg.json + sae.safetensors, sparse_coder.py:167-179), which
        SparseCoder.load_from_disk -- and the reference -- load as an ordinary coder."""
        import json
        from pathlib import Path

        from safetensors.torch import save_file
        tensors = {
            "W_dec": parallel.all_gather_rows(self.W_dec.data),
            "b_dec": self.b_dec.data,
            "encoder.bias": parallel.all_gather_rows(self.encoder.bias.data),
            "encoder.weight": parallel.all_gather_rows(self.encoder.weight.data),
        }
        if self.W_skip is not None:
            tensors["W_skip"] = self.W_skip.data
        if parallel.rank() == 0:
            path = Path(path)
            path.mkdir(parents=True, exist_ok=True)
            save_file({k: v.detach().contiguous().cpu() for k, v in tensors.items()}, str(path / "sae.safetensors"))
            with open(path / "cfg.json", "w") as f:
                json.dump({**self.cfg.to_dict(), "d_in": self.d_in}, f)

    def _decode_all(self, local_idx: Tensor, acts: Tensor) -> Tensor:
        """Full reconstruction from a merged selection: this rank's partial sum over the slots it
        owns, summed over ranks, plus b_dec. The decode-side gradient of the replicated b_dec is
        identical on every rank: 1/W of it is kept here so that reduce_shared_grads() (one SUM
        over ranks, which also adds up the shard-local encoder-side terms) yields the unsharded
        gradient."""
        partial = SparseDecode.apply(local_idx, self.W_dec, acts.to(self.dtype), None)
        return _AllReduceSum.apply(partial) + _ScaleGrad.apply(self.b_dec, 1.0 / parallel.world_size())

    @torch.autocast("cuda", dtype=torch.bfloat16, enabled=torch.cuda.is_available())
    def forward(self, x: Tensor, y: Tensor | None = None, *, dead_mask: Tensor | None = None,
                num_dead: int | None = None) -> ForwardOutput:
        """dead_mask: the (num_latents,) mask of the whole dictionary, identical on every rank (the
        Trainer's counters see the merged selection); num_dead as in SparseCoder.forward."""
        if self.cfg.activation != "topk":
            raise NotImplementedError("the latent-sharded coder selects with top-k (groupmax groups would straddle shards)")
        b_dec = None if self.cfg.transcode else self.b_dec
        want_pre = dead_mask is not None or self.cfg.multi_topk
        top_acts, top_indices, local_idx, pre = _ShardedEncode.apply(
            x, self.encoder.weight, self.encoder.bias, self.cfg.k, self.lo, self.hi, b_dec, want_pre)
        if y is None:
            y = x
        sae_out = self._decode_all(local_idx, top_acts)
        if self.W_skip is not None:
            sae_out = _AddSkip.apply(sae_out, SkipProduct.apply(x.reshape(-1, x.shape[-1]), self.W_skip, _autocast_bf16()))
        y2 = y.reshape(-1, y.shape[-1]).contiguous()
        tv = ops.total_variance(y2)
        fvu, e = ResidualLoss.apply(sae_out, y2, tv)
        zero = sae_out.new_zeros(())               # a fill kernel: new_tensor(0.0) is a blocking host copy

        auxk_loss = zero
        if dead_mask is not None and num_dead is None:
            num_dead = int(dead_mask.sum())
        if dead_mask is not None and num_dead > 0:                                  # sparse_coder.py:230-251
            if dead_mask.shape[0] != self.num_latents:
                raise ValueError(f"dead_mask must cover the whole dictionary ({self.num_latents} latents)")
            k_aux = y.shape[-1] // 2
            scale = min(num_dead / k_aux, 1.0)
            k_aux = min(k_aux, num_dead)
            keep = dead_mask[self.lo:self.hi].contiguous()
            aux_acts, _, aux_local = sharded_select(pre, k_aux, self.lo, self.hi, keep)
            e_hat = self._decode_all(aux_local, aux_acts)
            auxk_sq, _ = ResidualLoss.apply(e_hat, e, tv)
            auxk_loss = scale * auxk_sq

        multi_topk_fvu = zero
        if self.cfg.multi_topk:                                                       # sparse_coder.py:256-262
            top_acts, top_indices, local4 = sharded_select(pre, 4 * self.cfg.k, self.lo, self.hi)
            sae_out = self._decode_all(local4, top_acts)
            multi_topk_fvu, _ = ResidualLoss.apply(sae_out, y2, tv)

        lead = x.shape[:-1]
        return ForwardOutput(sae_out.reshape(*lead, -1), top_acts.reshape(*lead, -1), top_indices.reshape(*lead, -1),
                             fvu, auxk_loss, multi_topk_fvu)

    @torch.no_grad()
    def reduce_shared_grads(self):
        """Call once after backward: the only gradient that crosses shards is b_dec's (D floats)."""
        if self.b_dec.grad is not None:
            parallel.all_reduce_(self.b_dec.grad, "sum")

    @torch.no_grad()
    def set_decoder_norm_to_unit_norm(self):
        ops.unit_norm_rows_(self.W_dec.data)
        torch.autograd.graph.increment_version(self.W_dec)

    @torch.no_grad()
    def remove_gradient_parallel_to_decoder_directions(self):
        ops.remove_parallel_grad_(self.W_dec.grad, self.W_dec.data)
