"""Sparse decoder dispatch with the reference's seam (sparsify/utils.py:81-103):
``decoder_impl(top_indices, top_acts, W_dec_T)``. There is exactly one implementation,
the HIP gather / segmented-reduction pair; nothing here falls back to torch."""
from __future__ import annotations

import torch
from torch import Tensor

from . import ops, parallel


class SparseDecode(torch.autograd.Function):
    """Counterpart of xFormersEmbeddingBag (sparsify/xformers.py:188-209)."""

    @staticmethod
    def forward(ctx, indices: Tensor, weight: Tensor, per_sample_weights: Tensor, bias: Tensor | None = None,
                summed: tuple[Tensor, Tensor] | None = None):
        """summed = (rest, base), multi-top-k only (sparse_coder.py:256-262): `base` (N, D) already holds the bias and
        the rows of those slots that `rest` (= indices with these slots set to -1, ops.exclude_selected) leaves out,
        so only the others are gathered. The backward is that of the whole list either way."""
        ops.forget_segments(indices)
        parallel.wait_for(weight)                  # an overlapped optimizer update (sign_sgd.py)
        w = weight.detach()
        ctx.later_node = None
        if summed is not None:
            from . import dead_decode
            # the decode noted last over this weight (SparseCoder.decode) runs its backward after this one
            ctx.later_node = dead_decode._latest_decode_node(weight)
        ctx.empty = indices.shape[0] == 0
        if ctx.empty:                              # no rows: nothing to launch
            ops._req(indices, "indices", torch.int64, 2)
            ctx.save_for_backward(indices, weight, per_sample_weights)
            ctx.has_bias = bias is not None
            ctx.set_materialize_grads(False)
            return torch.empty(0, w.shape[1], dtype=torch.float32, device=w.device)
        if summed is not None:
            rest, base = summed
            out = ops.add(ops.gather_rows_weighted(rest, per_sample_weights.detach().contiguous(), w, None,
                                                   table_owner=weight), base.detach().reshape(-1, w.shape[1]).contiguous())
        else:
            out = ops.gather_rows_weighted(indices, per_sample_weights.detach().contiguous(), w,
                                           None if bias is None else bias.detach(), table_owner=weight)
        ctx.save_for_backward(indices, weight, per_sample_weights)
        ctx.has_bias = bias is not None
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, gradient):
        indices, weight, psw = ctx.saved_tensors
        if gradient is None:
            return None, None, None, None, None
        if ctx.empty:
            need = ctx.needs_input_grad
            return (None, torch.zeros_like(weight) if need[1] else None, torch.zeros_like(psw) if need[2] else None,
                    torch.zeros(weight.shape[1], dtype=torch.float32, device=weight.device)
                    if ctx.has_bias and need[3] else None, None)
        g = gradient.contiguous()
        if g.dtype != torch.float32:
            g = g.float()
        need_w, need_a = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        gW = gA = gB = None
        if need_w or need_a:
            # in token blocks where the batch needs them (the reference's default step is 65536 tokens); on a later
            # micro-step of a gradient-accumulating Trainer the sums are added to weight.grad inside the kernel
            from . import dead_decode
            sink = ops.grad_sink(weight) if need_w else None
            # a dense gradient an earlier node of this pass parked for the same weight (the multi-top-k decode)
            parked = dead_decode.claim_dense(weight) if need_w else None
            target = sink if sink is not None else parked
            gW, _, gA, added = ops.weight_grad_maybe_into(target, indices, weight.shape[0], psw.detach().contiguous(), g,
                                                          weight.detach(), want_dcoeff=need_a)
            if gA is not None and gA.dtype != psw.dtype:
                gA = gA.to(psw.dtype)
            if need_w:
                if added and target is parked:
                    gW = parked
                elif parked is not None:                        # (the kernel form could not add, or a sink took this one)
                    gW = ops.add(gW, parked) if gW is not None else parked
                dead_decode.claim_pending_rows(weight, sink if gW is None else gW)     # AuxK rows of this pass (dead_decode.py)
                if gW is not None and dead_decode.later_decode_will_run(weight, ctx.later_node):
                    dead_decode.park_dense(weight, gW)
                    gW = None
        if ctx.has_bias and ctx.needs_input_grad[3]:
            gB = ops.colsum(g)
        return None, gW, gA, gB, None


def hip_decode(top_indices: Tensor, top_acts: Tensor, W_dec: Tensor) -> Tensor:
    """Called as decoder_impl(top_indices, top_acts, self.W_dec.mT) (sparse_coder.py:201);
    like both reference implementations it transposes back, so the table is (M, D)."""
    table = W_dec.mT
    if not table.is_contiguous():
        table = table.contiguous()
    idx = top_indices.reshape(-1, top_indices.shape[-1])
    acts = top_acts.reshape(-1, top_acts.shape[-1])
    out = SparseDecode.apply(idx.contiguous(), table, acts)
    return out.reshape(*top_indices.shape[:-1], table.shape[1])


decoder_impl = hip_decode
