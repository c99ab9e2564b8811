"""SignSGD with the reference's optimizer surface (sparsify/sign_sgd.py:5-23); the update
p -= lr * sign(grad) is one HIP pass per parameter, which can also refresh the bf16 shadow
the encoder GEMM reads (so the next step needs no separate cast)."""
from __future__ import annotations

import os

import torch
from torch.optim import Optimizer

from . import ops, parallel
from .fused_encoder import peek_bf16_shadow, register_bf16_shadow


class SignSGD(Optimizer):
    """Steepest descent in the L-infty norm. From <https://arxiv.org/abs/1802.04434>"""

    # row blocks of an overlapped exchange of the encoder matrix: the next forward's GEMM runs
    # block by block behind the arriving updates (sphip_encoder_preacts_block)
    OVERLAP_BLOCKS = 8

    def __init__(self, params, lr: float = 1e-3, exchange_signs: bool = False, overlap: bool = False):
        """exchange_signs (not in the reference): under torch.distributed the optimizer itself
        averages the gradients of the large matrices across ranks, as a reduce-scatter plus an
        all-gather of 1-byte signs (parallel.SignExchange); the caller then all-reduces only
        the parameters for which `reduces(p)` is False.
        overlap: run those exchanges and the updates they feed on a second stream, encoder
        matrices first, so that step() returns at once and the next forward overlaps the
        traffic. Readers inside this package wait for the rows they need; anything else that
        reads the parameters must call parallel.wait_for_updates() first."""
        if lr <= 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        super().__init__(params, {"lr": lr})
        self.exchange_signs = exchange_signs
        self.overlap = overlap and exchange_signs
        # ids of "renorm_rows" parameters whose rows were left unit-norm by the last step();
        # the trainer consumes an id instead of normalising that decoder again
        self.renormalised: set[int] = set()
        self._early_hooks: list = []
        self._early_done: dict[int, tuple] = {}
        self.early_exchanges = 0                   # how often a hook started an exchange (tests)

    # fraction of a decoder matrix's rows whose exchange starts as soon as its gradient exists
    EARLY_FRACTION = 0.5

    def enable_early_exchange(self, params) -> None:
        """Overlapped data-parallel step only: start the exchange (and update) of the first
        EARLY_FRACTION of the rows of `params` (decoder matrices) from a post-accumulate-grad
        hook, i.e. inside the backward as soon as their gradient is complete, under the rest of
        the backward. step() then exchanges the remaining rows after the encoder matrices. The
        wire is busy earlier, the update is the same. Needs one backward per step()."""
        for group in self.param_groups:
            for p in group["params"]:
                if any(p is q for q in params) and p.is_cuda and hasattr(p, "register_post_accumulate_grad_hook"):
                    self._early_hooks.append(p.register_post_accumulate_grad_hook(
                        lambda t, g=group: self._early_hook(t, g)))

    def _early_hook(self, p, group) -> None:
        if not (self.overlap and self.reduces(p)) or p.grad is None or not p.grad.is_contiguous():
            return
        w = parallel.world_size()
        rows = int(p.shape[0] * self.EARLY_FRACTION) // w * w
        if rows <= 0 or rows >= p.shape[0] or id(p) in self._early_done:
            return
        with torch.no_grad():
            events = self._exchange_step(p, p.grad, group["lr"], group.get("project_rows", False),
                                         group.get("renorm_rows", False), overlap=True, rows=(0, rows), finish=False)
        self._early_done[id(p)] = (rows, events)
        self.early_exchanges += 1

    def fuse_encoder_update(self, weights) -> None:
        """Optimizer-in-backward for encoder matrices (not in the reference): FusedEncoder's
        backward applies this optimizer's step, p -= lr * sign(grad), inside the weight-gradient
        kernel and leaves p.grad None, so step() skips them. Bit for bit the separate step; the
        (M, D) gradient is never written. Only valid when nothing else consumes that gradient:
        one rank, one backward per optimizer step, no clipping or logging of it."""
        from .fused_encoder import register_inplace_signsgd
        for group in self.param_groups:
            for p in group["params"]:
                if any(p is w for w in weights):
                    if group.get("project_rows", False):
                        raise ValueError("decoder matrices need the projected gradient: not fusable")
                    # (the group is looked up per call: load_state_dict replaces the group dictionaries)
                    register_inplace_signsgd(p, lambda p=p: next(
                        gr for gr in self.param_groups if any(q is p for q in gr["params"]))["lr"])

    def unfuse_encoder_update(self) -> None:
        from .fused_encoder import unregister_inplace_signsgd
        for group in self.param_groups:
            for p in group["params"]:
                unregister_inplace_signsgd(p)

    def reduces(self, p) -> bool:
        return self.exchange_signs and parallel.SignExchange.eligible(p)

    @torch.no_grad()
    def step(self, closure: None = None) -> None:
        assert closure is None, "Closure is not supported."
        deferred = []
        for group in self.param_groups:
            lr = group["lr"]
            project = group.get("project_rows", False)
            renorm = group.get("renorm_rows", False)
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if self.reduces(p):
                    if self.overlap and p.is_cuda:
                        deferred.append((p, g, lr, project, renorm))
                    else:
                        self._exchange_step(p, g, lr, project, renorm)
                    continue
                if project:
                    # decoder matrix: remove_gradient_parallel_to_decoder_directions
                    # (sparse_coder.py:281-295) fused with the sign step, one pass over (W, G);
                    # "renorm_rows" also applies the set_decoder_norm_to_unit_norm that opens
                    # the next train step (trainer.py:408) while the rows are in registers
                    # (the slice-major copy the L2-resident decode reads is rewritten in the same pass)
                    sm = ops.peek_slice_major(p) if os.environ.get("SPHIP_SM_FOLD", "1") != "0" else None
                    ops.decoder_project_signsgd_(p.data, g, lr, renorm, slice_major=sm)
                    torch.autograd.graph.increment_version(p)
                    if sm is not None:
                        ops.register_slice_major(p, sm)
                    if renorm:
                        self.renormalised.add(id(p))
                    continue
                shadow = peek_bf16_shadow(p)
                sm = (ops.peek_slice_major(p) if shadow is None and p.ndim == 2 and p.is_contiguous()
                      and os.environ.get("SPHIP_SM_FOLD", "1") != "0" else None)
                if sm is not None:
                    # a decoder matrix that is neither projected nor normalised (a transcoder's) and that the
                    # L2-resident decode reads through its slice-major copy: rewritten in the same pass
                    ops.signsgd_step_rows_(p.data, g, lr, sm)
                    torch.autograd.graph.increment_version(p)
                    ops.register_slice_major(p, sm)
                    continue
                ops.signsgd_step_(p.data, g, lr, shadow)
                torch.autograd.graph.increment_version(p)
                if shadow is not None:
                    register_bf16_shadow(p, shadow)

        # overlapped exchanges, in the order the next forward needs the results: encoder
        # matrices (read by the GEMM that opens it) before decoder matrices (read after top-k)
        for item in sorted(deferred, key=lambda it: it[3]):
            early = self._early_done.pop(id(item[0]), None)
            if early is None:
                self._exchange_step(*item, overlap=True)
            else:                                  # the first rows went out during the backward
                self._exchange_step(*item, overlap=True, rows=(early[0], item[0].shape[0]), prior=early[1])

    def _exchange_step(self, p, g, lr, project: bool, renorm: bool, overlap: bool = False, rows=None, prior=None,
                       finish: bool = True):
        """One large matrix under data parallelism: see parallel.SignExchange."""
        w = p.data
        shadow = None if project else peek_bf16_shadow(p)
        # the slice-major copy the L2-resident decode reads is rewritten row block by row block by
        # the kernel that applies the signs (fused form: with the re-normalisation)
        sm = ops.peek_slice_major(p) if project and renorm and p.shape[1] % 32 == 0 and p.shape[1] <= 8192 else None
        if project:
            def apply_fn(signs, lo, hi):
                ops.decoder_apply_signs_(w[lo:hi], signs, lr, renorm, slice_major=sm, row0=lo)
        else:
            def apply_fn(signs, lo, hi):
                ops.apply_signs_(w[lo:hi], signs, lr, None if shadow is None else shadow[lo:hi])
        events = self.run_exchange(p, g, project, apply_fn, overlap, rows=rows, prior=prior, finish=finish)
        if finish:
            if project and renorm:
                self.renormalised.add(id(p))
            if shadow is not None:
                register_bf16_shadow(p, shadow)
            if sm is not None:
                ops.register_slice_major(p, sm)
        return events

    def run_exchange(self, p, g, project: bool, apply_fn, overlap: bool, rows=None, prior=None, finish: bool = True):
        """Reduce-scatter the gradient, take the signs (decoder matrices: of the gradient
        projected on the complement of the parameter's rows), all-gather them and hand them to
        `apply_fn(signs, row_lo, row_hi)`; with `overlap` on the exchange stream, recording one
        event per finished row block (parallel.set_pending). Also used by the schedule-free
        wrapper, whose apply_fn is its own step. `rows` = (lo, hi) restricts the call to a row
        range (early exchange); `prior` are the events of rows exchanged before; with finish =
        False the events are returned instead of published (more rows follow)."""
        w = p.data
        r0, r1 = (0, p.shape[0]) if rows is None else rows
        if project:
            def sign_fn(shard, lo, hi):
                return ops.decoder_project_sign_i8(w[r0 + lo:r0 + hi], shard)
        else:
            def sign_fn(shard, lo, hi):
                return ops.sign_i8(shard)

        def apply_rows(signs, lo, hi):
            apply_fn(signs, r0 + lo, r0 + hi)

        g_rows = g if rows is None else g[r0:r1]
        done: list = list(prior or [])
        if overlap:
            cur = torch.cuda.current_stream(p.device)
            side = parallel.exchange_stream(p.device)
            if not prior:
                parallel.wait_for(p)               # an earlier update of p that nobody read
            side.wait_stream(cur)                  # the gradient (and p's last readers) are done
            g.record_stream(side)                  # zero_grad() must not recycle it under the exchange
            blocks = 1 if project else parallel.SignExchange.row_blocks(r1 - r0, self.OVERLAP_BLOCKS, 256)

            def on_block(lo, hi):
                ev = torch.cuda.Event()
                ev.record(side)
                done.append((r0 + lo, r0 + hi, ev))

            with torch.cuda.stream(side):
                parallel.SignExchange.step(g_rows, sign_fn, apply_rows, blocks=blocks, on_block=on_block)
            if finish:
                parallel.set_pending(p, done)
        else:
            parallel.SignExchange.step(g_rows, sign_fn, apply_rows)
        if finish:
            torch.autograd.graph.increment_version(p)
        return done
