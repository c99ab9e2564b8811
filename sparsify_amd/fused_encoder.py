"""Fused encoder op with the reference's surface (sparsify/fused_encoder.py:7-103):
``fused_encoder(input, weight, bias, k, activation) -> EncoderOutput`` and the
``FusedEncoder`` autograd.Function, running on the HIP kernels of libsparsify_hip.

Forward : pre = relu(x W^T + b) (bf16 MFMA under autocast, exact-fp32 MFMA otherwise),
          then row top-k / groupmax.
Backward: only the k selected columns -- segmented (latent-major) weight/bias gradient and,
          if asked for, the gathered grad-input. ``grad_preacts`` is ignored exactly as in
          the reference (fused_encoder.py:53,87).
"""
from __future__ import annotations

import os
from typing import Literal, NamedTuple

import torch
from torch import Tensor

from . import ops, parallel


class EncoderOutput(NamedTuple):
    top_acts: Tensor
    """Activations of the top-k latents."""

    top_indices: Tensor
    """Indices of the top-k features."""

    pre_acts: Tensor
    """Activations before the top-k selection."""


import weakref

# Opt-in (SPHIP_LAZY_PREACTS=1, TrainConfig.lazy_pre_acts, or set this attribute): on eligible
# shapes the forward does not store the (N, M) pre-activations; EncoderOutput.pre_acts is then a
# LazyPreActs that recomputes them on first use. Measured at the headline shape
# (benchmarks/lazy_bench.py): 12.6 ms against 12.25 ms for GEMM + top-k -- the GEMM is bound by
# the matrix pipe, not by its 4 GiB of stores, so what this buys is memory (4 GiB per 16384
# tokens, 3 GiB of HBM writes and 1 GB of reads per step), not time. Hence off by default.
LAZY_PRE_ACTS = os.environ.get("SPHIP_LAZY_PREACTS", "0") == "1"


class LazyPreActs:
    """Stand-in for EncoderOutput.pre_acts (fused_encoder.py:50) when the forward did not store
    the matrix: shape / dtype / device answer at once; any other use -- a torch function, an
    attribute, indexing -- first recomputes relu(x W^T + b) with the ordinary kernel, from the
    operands of the forward. Reading it after the encoder weights changed raises."""

    def __init__(self, x_low: Tensor, weight: Tensor, bias: Tensor | None, lead: tuple):
        self._x, self._weight, self._bias, self._lead = x_low, weight, bias, tuple(lead)
        self._version = (weight._version, None if bias is None else bias._version)
        self._value: Tensor | None = None

    @property
    def shape(self):
        return torch.Size((*self._lead, self._weight.shape[0]))

    @property
    def dtype(self):
        return torch.bfloat16

    @property
    def device(self):
        return self._x.device

    @property
    def is_materialized(self) -> bool:
        return self._value is not None

    def materialize(self) -> Tensor:
        if self._value is None:
            now = (self._weight._version, None if self._bias is None else self._bias._version)
            if now != self._version:
                raise RuntimeError("pre_acts of a lazy EncoderOutput read after the encoder parameters changed; "
                                   "read it before the optimizer step, or run with lazy pre_acts off")
            pre = ops.encoder_preacts(self._x, bf16_shadow(self._weight),
                                      None if self._bias is None else bf16_shadow(self._bias))
            self._value = pre.reshape(*self._lead, pre.shape[-1])
            self._x = None
        return self._value

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    def __getitem__(self, item):
        return self.materialize()[item]

    def __len__(self):
        return self.shape[0]

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def unwrap(a):
            if isinstance(a, LazyPreActs):
                return a.materialize()
            if isinstance(a, (list, tuple)):
                return type(a)(unwrap(b) for b in a)
            return a
        return func(*unwrap(args), **{k_: unwrap(v) for k_, v in (kwargs or {}).items()})


# address -> (weak reference to the owning tensor, (version, shape), bf16 buffer). The weak
# reference pins an entry to the tensor it was made for: the caching allocator hands a freed
# block to the next same-sized request, so a later parameter can live at the same address with
# the same shape and version counter (every coder loaded from disk sits at version 2). An entry
# whose owner has died is dropped, together with its (M, D) bf16 buffer.
_shadows: dict[int, tuple] = {}


def _shadow_entry(t: Tensor):
    hit = _shadows.get(t.data_ptr())
    if hit is None:
        return None
    owner = hit[0]()
    if owner is None or owner.data_ptr() != t.data_ptr() or owner.shape != t.shape or owner._version != t._version:
        # the owner is gone, or `t` is another tensor at a recycled address (an alias of the owner
        # -- a view, .detach() -- shares its version counter and passes)
        if owner is None:
            _shadows.pop(t.data_ptr(), None)
        return None
    return hit


def _store_shadow(t: Tensor, shadow: Tensor) -> None:
    key = t.data_ptr()

    def _drop(ref, key=key):
        hit = _shadows.get(key)
        if hit is not None and hit[0] is ref:
            del _shadows[key]

    _shadows[key] = (weakref.ref(t, _drop), (t._version, tuple(t.shape)), shadow)


def bf16_shadow(t: Tensor) -> Tensor:
    """bf16 copy of an fp32 parameter, re-cast only when the parameter changed.

    Autocast re-casts encoder.weight (2 GiB read at the headline shape) on every call
    (sparse_coder.py:205-209 + fused_encoder.py:29); the values are identical, we just keep
    them until the optimizer bumps the parameter's version. Pass the parameter itself (not a
    temporary alias): the entry lives as long as the tensor object it was made for."""
    if t.dtype == torch.bfloat16:
        return t
    hit = _shadow_entry(t)
    if hit is not None and hit[1] == (t._version, tuple(t.shape)):
        return hit[2]
    buf = hit[2] if hit is not None and hit[2].shape == t.shape else None
    sh = ops.cast_bf16(t.detach().contiguous(), buf)
    _store_shadow(t, sh)
    return sh


def register_bf16_shadow(t: Tensor, shadow: Tensor) -> None:
    """Tell the cache that `shadow` already equals bf16(t) (written by a fused update)."""
    _store_shadow(t, shadow)


def peek_bf16_shadow(t: Tensor) -> Tensor | None:
    """The buffer holding (a possibly stale) bf16(t), for an update kernel to refresh in place."""
    hit = _shadows.get(t.data_ptr())
    if hit is None:
        return None
    owner = hit[0]()
    if owner is None or owner.data_ptr() != t.data_ptr() or owner.shape != t.shape:
        return None
    return hit[2]


def shadow_is_current(t: Tensor) -> bool:
    hit = _shadow_entry(t)
    return hit is not None and hit[1] == (t._version, tuple(t.shape))


# Optimizer-in-backward for the encoder matrix (sign_sgd.SignSGD.fuse_encoder_update,
# schedule_free.ScheduleFreeWrapper.fuse_encoder_update): the optimizer registers a Parameter with
# a callable returning its learning rate (or, schedule-free, lr + z + beta + the step's c). For a registered
# weight the backward applies p -= lr * sign(grad) inside the gradient kernel and returns no
# gradient. Keyed by storage address, validated through a weak reference to the Parameter, so
# an entry can never outlive its parameter and match another tensor at the same address.
_inplace_signsgd: dict[int, tuple] = {}


def register_inplace_signsgd(param: Tensor, lr_of) -> None:
    _inplace_signsgd[param.data_ptr()] = (weakref.ref(param), lr_of)


def unregister_inplace_signsgd(param: Tensor) -> None:
    _inplace_signsgd.pop(param.data_ptr(), None)


def _inplace_lr(weight: Tensor):
    hit = _inplace_signsgd.get(weight.data_ptr())
    if hit is None:
        return None
    owner = hit[0]()
    if owner is None or owner.data_ptr() != weight.data_ptr() or owner.shape != weight.shape:
        _inplace_signsgd.pop(weight.data_ptr(), None)
        return None
    return hit[1]


# The block maxima the GEMM epilogue left for a stored pre_acts matrix (ops.encoder_preacts(..., want_tile_max=True)),
# for a later top-k over the same matrix (multi-top-k, sparse_coder.py:257): address -> (weak reference to the
# matrix, maxima). The entry dies with the matrix.
_block_maxima: dict[int, tuple] = {}


def note_block_maxima(pre: Tensor, tile_max: Tensor) -> None:
    import weakref
    key = pre.data_ptr()

    def _drop(ref, key=key):
        cur = _block_maxima.get(key)
        if cur is not None and cur[0] is ref:
            del _block_maxima[key]

    _block_maxima[key] = (weakref.ref(pre, _drop), tile_max)


def block_maxima_of(pre: Tensor) -> Tensor | None:
    """The maxima noted for `pre` (or for a tensor `pre` is a same-shaped alias of), if it is still that matrix."""
    hit = _block_maxima.get(pre.data_ptr())
    if hit is None:
        return None
    owner = hit[0]()
    if (owner is None or owner.data_ptr() != pre.data_ptr() or owner.numel() != pre.numel()
            or owner.shape[-1] != pre.shape[-1] or owner._version != pre._version or owner._version != 0):
        return None
    return hit[1]


def _autocast_bf16() -> bool:
    return torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16


def low_precision_input(x2: Tensor, b_dec: Tensor | None) -> Tensor:
    """The bf16 operand FusedEncoder.forward feeds the GEMM: bf16(x - b_dec), or bf16(x)."""
    if b_dec is None and x2.dtype == torch.bfloat16:
        return x2
    return ops.center_cast(x2, None if b_dec is None else b_dec.detach().contiguous(), want_f32=False, want_bf16=True)[1]


# the in-backward update of encoder.weight runs in the L2-sliced kernel: up to this mean segment length (entries per
# latent and step)
FUSE_MEAN_SEGMENT = int(os.environ.get("SPHIP_FUSE_MEAN", "64"))


class FusedEncoder(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, weight, bias, k: int, activation: Literal["groupmax", "topk"], b_dec=None,
                lazy_pre_acts: bool | None = None):
        """
        input:  (N, D)      weight: (M, D)      bias: (M,)
        k:      number of latents to keep per row
        b_dec:  optional (D,) decoder bias to subtract from the input first
                (SparseCoder.encode, sparse_coder.py:191-192), fused here.
        lazy_pre_acts: None = the module default LAZY_PRE_ACTS; False = always store pre_acts
                (a caller that will read them: AuxK, multi-top-k).
        """
        if activation not in ("topk", "groupmax"):
            raise ValueError(f"Unknown activation: {activation}")
        lead = input.shape[:-1]
        x2 = input.reshape(-1, input.shape[-1]).contiguous()
        low = _autocast_bf16() or (weight.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16)
        ctx.empty = x2.shape[0] == 0
        if ctx.empty:
            ops._req(x2, "input")
            # no rows: nothing to launch (torch's linear / topk return empty tensors here too)
            odt = torch.bfloat16 if low else torch.float32
            ctx.save_for_backward(x2, weight, bias, None, b_dec, None)
            ctx.k, ctx.lead, ctx.in_shape = k, lead, input.shape
            values = torch.empty(*lead, k, dtype=odt, device=x2.device)
            indices = torch.empty(*lead, k, dtype=torch.int64, device=x2.device)
            pre = torch.empty(*lead, weight.shape[0], dtype=odt, device=x2.device)
            ctx.mark_non_differentiable(indices, pre)
            ctx.set_materialize_grads(False)
            return values, indices, pre

        shift = None
        if b_dec is not None and low and x2.dtype == torch.bfloat16:
            # bf16 activations: the fp32 (x - b_dec) is never materialised; the backward reads
            # the bf16 rows (half the gather bytes) and applies the shift analytically
            x_low = ops.center_cast(x2, b_dec.detach().contiguous(), want_f32=False, want_bf16=True)[1]
            rows, shift = x2, b_dec.detach().clone()
        elif b_dec is not None:
            rows, x_low = ops.center_cast(x2, b_dec.detach().contiguous(), want_f32=True, want_bf16=low)
        else:
            rows = x2
            x_low = None
            if low:
                x_low = x2 if x2.dtype == torch.bfloat16 else ops.center_cast(x2, None, False, True)[1]
        tile_max = None
        # an overlapped data-parallel update of the weight (sign_sgd.SignSGD(overlap=True)) may
        # still be arriving, row block by row block: the GEMM then follows it block by block
        blocks = parallel.pending_blocks(weight)
        if blocks and not (low and len(blocks) > 1 and shadow_is_current(weight) and x_low.shape[1] % 64 == 0):
            torch.cuda.current_stream(weight.device).wait_event(blocks[-1][2])
            blocks = None
        lazy = LAZY_PRE_ACTS if lazy_pre_acts is None else lazy_pre_acts
        lazy = (lazy and low and not blocks and activation == "topk"
                and ops.lazy_eligible(x_low.shape[0], x_low.shape[1], weight.shape[0], k))
        if lazy:
            values, indices, _ = ops.encode_topk_lazy(
                x_low, bf16_shadow(weight), None if bias is None else bf16_shadow(bias), k)
            pre = LazyPreActs(x_low, weight, bias, lead)
        elif low and blocks:
            N, M = x_low.shape[0], weight.shape[0]
            pre = torch.empty(N, M, dtype=torch.bfloat16, device=x_low.device)
            if activation == "topk":
                tile_max = torch.empty(-(-N // 64), -(-M // ops.TOPK_TILE), 64, dtype=torch.bfloat16, device=x_low.device)
            w_low, b_low = bf16_shadow(weight), None if bias is None else bf16_shadow(bias)
            cur = torch.cuda.current_stream(weight.device)
            for lo, hi, ev in blocks:
                cur.wait_event(ev)
                ops.encoder_preacts_block(x_low, w_low, b_low, pre, tile_max, lo, hi)
        elif low:
            # the bf16 GEMM epilogue also leaves the maximum of every 128-latent block of pre;
            # top-k then reads only the ~k blocks per row that can hold a selected latent
            pre = ops.encoder_preacts(x_low, bf16_shadow(weight), None if bias is None else bf16_shadow(bias),
                                      want_tile_max=(activation == "topk"))
            if activation == "topk":
                pre, tile_max = pre
        else:
            xf = rows if rows.dtype == torch.float32 else ops.center_cast(rows, None, True, False)[0]
            pre = ops.encoder_preacts(xf, weight.detach().contiguous(),
                                      None if bias is None else bias.detach().contiguous())

        if lazy:
            pass
        elif activation == "topk":
            values, indices = ops.topk(pre, k, tile_max=tile_max)
        else:
            values, indices = ops.groupmax(pre, k)
        ops.forget_segments(indices)

        ctx.save_for_backward(rows, weight, bias, indices, b_dec, shift)
        ctx.k = k
        ctx.lead = lead
        ctx.in_shape = input.shape
        ctx.set_materialize_grads(False)        # no (N, M) zero tensor for the ignored grad_preacts
        if lazy:
            ctx.mark_non_differentiable(indices)
            return values.reshape(*lead, k), indices.reshape(*lead, k), pre
        pre = pre.reshape(*lead, pre.shape[-1])
        ctx.mark_non_differentiable(indices, pre)
        if tile_max is not None:
            note_block_maxima(pre, tile_max)
        return values.reshape(*lead, k), indices.reshape(*lead, k), pre

    @staticmethod
    def backward(ctx, grad_values, grad_indices, grad_preacts):
        rows, weight, bias, indices, b_dec, shift = ctx.saved_tensors
        grad_input = grad_weight = grad_bias = grad_bdec = None
        if grad_values is None:
            return None, None, None, None, None, None, None
        if ctx.empty:
            need = ctx.needs_input_grad
            return (torch.zeros(ctx.in_shape, dtype=torch.float32, device=weight.device) if need[0] else None,
                    torch.zeros_like(weight) if need[1] else None,
                    torch.zeros_like(bias) if bias is not None and need[2] else None, None, None,
                    torch.zeros_like(b_dec) if b_dec is not None and need[5] else None, None)
        M = weight.shape[0]
        gv = grad_values.reshape(-1, ctx.k).contiguous()
        if weight.dtype != torch.float32:
            raise TypeError("FusedEncoder.backward needs fp32 master weights (trainer.py:85-87)")
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        need_bdec = b_dec is not None and ctx.needs_input_grad[5]

        lr_of = _inplace_lr(weight) if need_w else None
        fuse = (lr_of is not None and weight.is_contiguous() and rows.shape[1] % 8 == 0
                and gv.numel() < FUSE_MEAN_SEGMENT * M)
        if fuse:
            seg = ops.cached_segments(indices, M)
            gb_full = ops.segment_coeff_sums(seg, gv)
            grad_bias = gb_full if bias is not None else None
            # b_dec's encoder-side gradient -gb^T W (W before its update) rides in the fused kernel
            fold_bdec = need_bdec and not need_x
        elif need_w or need_b or need_bdec:
            sink = ops.grad_sink(weight) if need_w else None        # gradient accumulation: see ops.register_grad_sinks
            grad_weight, grad_bias, _, _ = ops.weight_grad_maybe_into(sink, indices, M, gv, rows, want_bias=True,
                                                                      row_shift=shift)
            if bias is None:
                gb_full, grad_bias = grad_bias, None
            else:
                gb_full = grad_bias
        if need_x:
            gx = ops.gather_rows_weighted(indices, gv, weight.detach())
            grad_input = gx.reshape(ctx.in_shape)
            if need_bdec:
                grad_bdec = ops.colsum(gx, -1.0)
        elif need_bdec and not fuse:
            # sum_n grad_x[n,:] = gb^T W : one pass over W instead of an (N,k)-row gather
            grad_bdec = ops.vec_mat(gb_full, weight.detach(), -1.0)
        if fuse:
            # every reader of the old weight in this backward has been enqueued: update in place
            shadow = peek_bf16_shadow(weight)
            how = lr_of()                    # lr, or {"lr", "z", "beta", "c"} for the schedule-free wrapper
            sums = gb_full if fold_bdec else None
            if isinstance(how, dict):
                res = ops.segmented_weight_grad_signsgd_(seg, gv, rows, weight.data, how["lr"], shadow, row_shift=shift,
                                                         schedule_free=(how["z"], how["beta"], how["c"]), coeff_sums=sums)
            else:
                res = ops.segmented_weight_grad_signsgd_(seg, gv, rows, weight.data, how, shadow, row_shift=shift,
                                                         coeff_sums=sums)
            if fold_bdec:
                grad_bdec = ops.colsum(res[1], -1.0)
            torch.autograd.graph.increment_version(weight)
            if shadow is not None:
                register_bf16_shadow(weight, shadow)
        return grad_input, grad_weight, grad_bias, None, None, grad_bdec, None


def fused_encoder(input, weight, bias, k: int, activation: Literal["groupmax", "topk"],
                  lazy_pre_acts: bool | None = None) -> EncoderOutput:
    """nn.Linear + ReLU + `activation`, with the sparse backward (fused_encoder.py:90-103).
    lazy_pre_acts (not in the reference): see FusedEncoder.forward."""
    return EncoderOutput(*FusedEncoder.apply(input, weight, bias, k, activation, None, lazy_pre_acts))
