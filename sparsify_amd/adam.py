"""Adam with torch.optim.Adam's surface and state layout (the "adam" optimizer of the
reference's trainer, sparsify/trainer.py:92-122, when bitsandbytes is absent): betas, eps, no
weight decay, no amsgrad. The update of one parameter is one HIP pass over (p, grad, exp_avg,
exp_avg_sq) which can also refresh the bf16 shadow the encoder GEMM reads, so the next step
needs no separate cast. The state keys ("step", "exp_avg", "exp_avg_sq") are torch's, so a
state_dict moves between the two."""
from __future__ import annotations

import torch
from torch.optim import Optimizer

from . import ops
from .fused_encoder import peek_bf16_shadow, register_bf16_shadow


class Adam(Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas: tuple[float, float] = (0.9, 0.999), eps: float = 1e-8):
        if lr < 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        if eps < 0.0:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 0: {betas[0]}")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 1: {betas[1]}")
        # (the remaining keys are torch.optim.Adam's, at the only values this step implements, so
        # that a state_dict loads into either class)
        super().__init__(params, {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": 0, "amsgrad": False,
                                  "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                                  "fused": None, "decoupled_weight_decay": False})

    @torch.no_grad()
    def step(self, closure: None = None) -> None:
        assert closure is None, "Closure is not supported."
        for group in self.param_groups:
            lr, (beta1, beta2), eps = group["lr"], group["betas"], group["eps"]
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
                raise NotImplementedError("Adam: weight_decay, amsgrad and maximize are not implemented")
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = torch.tensor(0.0)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                state["step"] += 1
                shadow = peek_bf16_shadow(p)
                ops.adam_step_(p.data, g, state["exp_avg"], state["exp_avg_sq"], lr, beta1, beta2, eps,
                               int(state["step"]), shadow)
                torch.autograd.graph.increment_version(p)
                if shadow is not None:
                    register_bf16_shadow(p, shadow)
