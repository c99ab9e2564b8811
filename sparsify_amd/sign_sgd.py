"""SignSGD with the reference's optimizer surface (sparsify/sign_sgd.py:5-23); the update
p -= lr * sign(grad) is one HIP pass per parameter, which can also refresh the bf16 shadow
the encoder GEMM reads (so the next step needs no separate cast)."""
from __future__ import annotations

import torch
from torch.optim import Optimizer

from . import ops, parallel
from .fused_encoder import peek_bf16_shadow, register_bf16_shadow


class SignSGD(Optimizer):
    """Steepest descent in the L-infty norm. From <https://arxiv.org/abs/1802.04434>"""

    def __init__(self, params, lr: float = 1e-3, exchange_signs: bool = False):
        """exchange_signs (not in the reference): under torch.distributed the optimizer itself
        averages the gradients of the large matrices across ranks, as a reduce-scatter plus an
        all-gather of 1-byte signs (parallel.SignExchange); the caller then all-reduces only
        the parameters for which `reduces(p)` is False."""
        if lr <= 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        super().__init__(params, {"lr": lr})
        self.exchange_signs = exchange_signs
        # ids of "renorm_rows" parameters whose rows were left unit-norm by the last step();
        # the trainer consumes an id instead of normalising that decoder again
        self.renormalised: set[int] = set()

    def reduces(self, p) -> bool:
        return self.exchange_signs and parallel.SignExchange.eligible(p)

    @torch.no_grad()
    def step(self, closure: None = None) -> None:
        assert closure is None, "Closure is not supported."
        for group in self.param_groups:
            lr = group["lr"]
            project = group.get("project_rows", False)
            renorm = group.get("renorm_rows", False)
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if self.reduces(p):
                    self._exchange_step(p, g, lr, project, renorm)
                    continue
                if project:
                    # decoder matrix: remove_gradient_parallel_to_decoder_directions
                    # (sparse_coder.py:281-295) fused with the sign step, one pass over (W, G);
                    # "renorm_rows" also applies the set_decoder_norm_to_unit_norm that opens
                    # the next train step (trainer.py:408) while the rows are in registers
                    ops.decoder_project_signsgd_(p.data, g, lr, renorm)
                    torch.autograd.graph.increment_version(p)
                    if renorm:
                        self.renormalised.add(id(p))
                    continue
                shadow = peek_bf16_shadow(p)
                ops.signsgd_step_(p.data, g, lr, shadow)
                torch.autograd.graph.increment_version(p)
                if shadow is not None:
                    register_bf16_shadow(p, shadow)

    def _exchange_step(self, p, g, lr, project: bool, renorm: bool):
        """One large matrix under data parallelism: see parallel.SignExchange."""
        w = p.data
        if project:
            def sign_fn(shard, lo, hi):
                return ops.decoder_project_sign_i8(w[lo:hi], shard)

            def apply_fn(signs):
                ops.decoder_apply_signs_(w, signs, lr, renorm)
        else:
            shadow = peek_bf16_shadow(p)

            def sign_fn(shard, lo, hi):
                return ops.sign_i8(shard)

            def apply_fn(signs):
                ops.apply_signs_(w, signs, lr, shadow)
        parallel.SignExchange.step(g, sign_fn, apply_fn)
        torch.autograd.graph.increment_version(p)
        if project and renorm:
            self.renormalised.add(id(p))
        if not project and shadow is not None:
            register_bf16_shadow(p, shadow)
