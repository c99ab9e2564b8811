"""AuxK decode over few dead latents as dense matrix-core products (sparse_coder.py:230-251).

``self.decode(auxk_acts, auxk_indices)`` sums k_aux = d_in / 2 decoder rows per token, all of
them rows of dead latents. As a row gather that is 10x the bytes of the main decode. When the
dead latents are few the (tokens x dead latents) matrix A of selected activations is dense
enough (k_aux / n_dead) that ``A @ W_dec[dead] + b_dec`` and ``A.T @ grad`` on the matrix cores
are an order of magnitude cheaper. Exactness: see csrc/dead_dense.hip (A exact in bf16, the
fp32 factor split into three bf16 terms, fp32 accumulation: only the summation order differs
from the reference's fp32 embedding_bag)."""
from __future__ import annotations

import os
import weakref

import torch
from torch import Tensor

from . import ops, parallel

# dense when n_dead <= DENSE_RATIO * k_aux: the gather moves k_aux * D * 4 bytes per token at
# ~10 TB/s, the three bf16 products cost 6 * n_dead * D FLOP per token at ~1.3 PFLOP/s (x2 with
# the gradient on both sides), which crosses near n_dead = 80 k_aux; the preparation passes and
# the (N, n_dead) temporaries scale with n_dead too, so the switch is set well below that.
DENSE_RATIO = 24
MIN_ENTRIES = 1 << 22


def eligible(pre_acts: Tensor, k_aux: int, W_dec: Tensor, num_dead: int) -> bool:
    """pre_acts: the (N, M) pre-activations the AuxK selection runs on (bf16 under autocast)."""
    force = os.environ.get("SPHIP_AUXK")
    if (force == "gather" or pre_acts.dtype != torch.bfloat16 or W_dec is None or W_dec.dtype != torch.float32
            or W_dec.shape[1] % 8):
        return False
    if force == "dense":
        return True
    return num_dead <= DENSE_RATIO * k_aux and pre_acts.shape[0] * k_aux >= MIN_ENTRIES


# Gradient rows of the dead latents waiting for the dense W_dec gradient of the same backward pass.
# DeadLatentDecode.backward runs before the main decode's backward (it was recorded later); instead
# of returning an (M, D) tensor that is zero outside the dead rows -- a 2 GiB fill plus an (M, D)
# addition by the autograd engine at the headline shape, 1.6 ms -- it parks its (n_dead, D) rows
# here, keyed by the parameter's storage address (validated through a weak reference), and
# utils.SparseDecode.backward adds them into the dense gradient it returns (claim_pending_rows).
# Rows are parked only when that backward is certain to come: the forward notes the main decode's
# autograd node (note_decode), and DeadLatentDecode.backward asks the engine whether it will execute
# that node in this very pass. Otherwise (auxk_loss back-propagated on its own, torch.autograd.grad
# over a sub-graph) it returns the dense zero-filled gradient like any other op, so W_dec.grad is only
# ever written by the engine's own accumulation, with every hook that hangs on it (gradient reducers,
# the sign exchange's early start). The callback below is a safety net for a pass that died half-way.
_decode_nodes: dict[int, tuple] = {}


def note_decode(weight: Tensor, node) -> None:
    """The autograd node of the latest row-gather decode over `weight` (SparseCoder.decode)."""
    if node is not None:
        _decode_nodes[weight.data_ptr()] = (weakref.ref(weight), weakref.ref(node))


def _latest_decode_node(weight: Tensor):
    hit = _decode_nodes.get(weight.data_ptr())
    if hit is None or hit[0]() is not weight:
        return None
    return hit[1]()


_pending_rows: dict[int, list] = {}


def claim_pending_rows(weight: Tensor, gW: Tensor) -> None:
    """Add the parked dead-latent gradient rows of `weight` into its dense gradient gW, in place."""
    for ref, G, dead_list, num_dead in _pending_rows.pop(weight.data_ptr(), []):
        w = ref()
        if w is not None and w.data_ptr() == weight.data_ptr() and w.shape == gW.shape:
            ops.scatter_add_rows_(gW, G, dead_list, num_dead)
        elif w is not None:                                     # another tensor at a recycled address
            _pending_rows.setdefault(w.data_ptr(), []).append((ref, G, dead_list, num_dead))


def _settle_pending_rows() -> None:
    for key in list(_pending_rows):
        for ref, G, dead_list, num_dead in _pending_rows.pop(key, []):
            w = ref()
            if w is None:
                continue
            if w.grad is None:
                w.grad = ops.scatter_rows(G, dead_list, num_dead, w.shape[0])
            else:
                ops.scatter_add_rows_(w.grad, G, dead_list, num_dead)


# The same for a whole dense gradient: the multi-top-k decode (sparse_coder.py:256-262) runs its backward before the
# main decode's and would hand the engine a second (M, D) tensor to add (6 GiB of traffic at the headline shape).
# It parks its gradient instead, under the same condition (a decode node over the same weight, noted in the forward,
# is going to execute in this pass), and that node's weight-gradient kernels add their sums into it
# (sphip_segmented_weight_grad_acc) and return it as theirs.
_pending_dense: dict[int, list] = {}


def park_dense(weight: Tensor, gW: Tensor) -> None:
    _pending_dense.setdefault(weight.data_ptr(), []).append((weakref.ref(weight), gW))
    torch.autograd.Variable._execution_engine.queue_callback(_settle_pending_dense)


def claim_dense(weight: Tensor) -> Tensor | None:
    """The dense gradient(s) parked for `weight` in this pass, as one tensor."""
    total = None
    for ref, G in _pending_dense.pop(weight.data_ptr(), []):
        w = ref()
        if w is not None and w.data_ptr() == weight.data_ptr() and w.shape == G.shape:
            total = G if total is None else ops.add(total, G)
        elif w is not None:                                     # another tensor at a recycled address
            _pending_dense.setdefault(w.data_ptr(), []).append((ref, G))
    return total


def _settle_pending_dense() -> None:
    for key in list(_pending_dense):
        for ref, G in _pending_dense.pop(key, []):
            w = ref()
            if w is None:
                continue
            w.grad = G if w.grad is None else ops.add(w.grad, G)


def later_decode_will_run(weight: Tensor, node) -> bool:
    return (node is not None and weight.is_leaf and isinstance(weight, torch.nn.Parameter)
            and torch._C._will_engine_execute_node(node))


class DeadLatentDecode(torch.autograd.Function):
    """decode(auxk_acts, auxk_indices) for indices that all lie on dead latents; gradients for
    W_dec and b_dec only (the gradient w.r.t. the activations would flow into pre_acts, which
    the encoder backward drops, fused_encoder.py:53,87). `dead_list` / `slot_of` come from
    ops.dead_compact; slot_of None means the indices already are slots (auxk_select_compact)."""

    @staticmethod
    def forward(ctx, indices: Tensor, acts: Tensor, W_dec: Tensor, b_dec: Tensor, dead_list: Tensor,
                slot_of: Tensor | None, num_dead: int):
        parallel.wait_for(W_dec)
        # rows can only stay parked across passes when a backward pass died half-way: drop them
        _pending_rows.pop(W_dec.data_ptr(), None)
        N = indices.shape[0]
        M, D = W_dec.shape
        Kd = -(-num_dead // 64) * 64
        A, At = ops.dead_scatter_acts(indices, acts.contiguous(), slot_of, Kd, num_valid=num_dead)
        w_parts = ops.split_bf16_transposed(W_dec.detach(), dead_list, num_dead, Kd)      # 3 x (D, Kd)
        out = ops.broadcast_rows(b_dec.detach(), N)
        for p in range(3):
            ops.gemm_bf16_f32acc(A, w_parts[p], out, accumulate=True)
        ctx.save_for_backward(At, dead_list)
        ctx.weight_ref = weakref.ref(W_dec) if isinstance(W_dec, torch.nn.Parameter) else None
        ctx.main_node = _latest_decode_node(W_dec)      # (both nodes belong to the same forward's graph)
        ctx.dims = (N, M, D, num_dead)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None, None, None, None
        At, dead_list = ctx.saved_tensors
        N, M, D, num_dead = ctx.dims
        g = g.reshape(N, D).contiguous().float()
        gW = gb = None
        if ctx.needs_input_grad[2]:
            g_parts = ops.split_bf16_transposed(g, None, N, At.shape[1])                  # 3 x (D, ceil64(N))
            G = torch.empty(At.shape[0], D, dtype=torch.float32, device=g.device)
            for p in range(3):
                ops.gemm_bf16_f32acc(At, g_parts[p], G, accumulate=p > 0)
            w = ctx.weight_ref() if ctx.weight_ref is not None else None
            claimed = (w is not None and w.is_leaf and ctx.main_node is not None
                       and torch._C._will_engine_execute_node(ctx.main_node))
            if claimed:
                _pending_rows.setdefault(w.data_ptr(), []).append((ctx.weight_ref, G, dead_list, num_dead))
                torch.autograd.Variable._execution_engine.queue_callback(_settle_pending_rows)
            else:
                gW = ops.scatter_rows(G, dead_list, num_dead, M)
        if ctx.needs_input_grad[3]:
            gb = ops.colsum(g, 1.0)
        return None, None, gW, gb, None, None, None


def auxk_select_compact(x_low: Tensor, W_low: Tensor, b_low: Tensor | None, dead_list: Tensor, num_dead: int, k_aux: int,
                        M: int | None = None):
    """torch.where(dead_mask, pre_acts, -inf).topk(k_aux) (sparse_coder.py:240-243) without the
    sweep over all M columns: the dead latents' pre-activations are recomputed as a compact
    (N, n_dead) matrix from their gathered encoder rows -- the same GEMM kernel, hence the same
    values -- and the top-k runs on that. Returns (acts, slots): slots index dead_list."""
    rows_out = -(-num_dead // ops.TOPK_TILE) * ops.TOPK_TILE      # padding columns: zero rows, relu -> 0,
    Wd = ops.gather_rows_bf16(W_low, dead_list, num_dead, rows_out)   # placed last, never needed (k_aux <= n_dead)
    bd = None if b_low is None else ops.gather_rows_bf16(b_low, dead_list, num_dead, rows_out)
    # the same GEMM kernel as the full-width product used (its two tile shapes round a few
    # outputs differently): the block entry point always runs the 256x256 one
    N, D = x_low.shape
    if ops.encoder_preacts_uses_tile256(N, D, W_low.shape[0] if M is None else M):
        pre_d = torch.empty(N, rows_out, dtype=torch.bfloat16, device=x_low.device)
        tmax = torch.empty(-(-N // 64), rows_out // ops.TOPK_TILE, 64, dtype=torch.bfloat16, device=x_low.device)
        ops.encoder_preacts_block(x_low, Wd, bd, pre_d, tmax, 0, rows_out)
    else:
        pre_d, tmax = ops.encoder_preacts(x_low, Wd, bd, want_tile_max=True)
    return ops.topk(pre_d, k_aux, tile_max=tmax)
