"""CPU statement of schedule-free SignSGD -- the reference's default optimizer "signum",
ScheduleFreeWrapper(SignSGD(...), momentum=0.95) (sparsify/trainer.py:146-162) -- written from the
published algorithm, not from sparsify_amd/schedule_free.py.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED: the wrapper comes from the third-party package `schedulefree` (listed without a
version in the reference's pyproject.toml, not vendored, absent from this image, untouched by any
reference test), so nothing importable pins these numbers. What is restated here is Defazio et al.,
"The Road Less Scheduled" (2024), Algorithm 1 in its wrapper form, with the base step of
sparsify/sign_sgd.py:23:

    y_t     = (1 - beta) z_t + beta x_t            gradients are taken at y   (train mode)
    z_{t+1} = z_t - lr * sign(grad f(y_t))         the base optimizer's iterate
    x_{t+1} = (1 - c_{t+1}) x_t + c_{t+1} z_{t+1}  running average, c_{t+1} = w_t / sum_{i<=t} w_i,
                                                   w_t = lr_max_t ** 2  (weight_lr_power = 2, r = 0)

The state carried between steps is (y, z) -- the parameter tensor holds y, as in the wrapper, and
x = (y - (1 - beta) z) / beta is implied. That matters for decoder matrices: the trainer's
row normalisation (set_decoder_norm_to_unit_norm, trainer.py:406-408) acts on the PARAMETER, i.e.
on y, between two optimizer steps and leaves z alone, so the implied x moves with it.

Order of operations for a decoder matrix in one train step (trainer.py:406-408, 513-518):
    1. y <- y / (||y||_row + eps)                       (start of the hook; the fused HIP kernel does
                                                         this at the END of the previous step)
    2. g = dL/dy; g <- g - <g, y>_row y                 projection on the complement of y's rows
    3. the schedule-free step above with sign(g)
Everything is evaluated in float64 and rounded once at the end of a step, so the comparison with
the fp32 kernels is against the mathematics, not against another fp32 operation order.
"""
from __future__ import annotations

import torch
from torch import Tensor

EPS32 = float(torch.finfo(torch.float32).eps)


class ScheduleFreeSignSGD:
    """State (y, z) of one parameter and the scalar weights of its group."""

    def __init__(self, y0: Tensor, lr: float, beta: float = 0.95, weight_lr_power: float = 2.0):
        self.y = y0.double().clone()
        self.z = y0.double().clone()          # the wrapper starts z at the parameter
        self.lr, self.beta, self.power = lr, beta, weight_lr_power
        self.lr_max = 0.0
        self.weight_sum = 0.0

    def next_c(self) -> float:
        self.lr_max = max(self.lr_max, self.lr)
        w = self.lr_max ** self.power
        self.weight_sum += w
        return w / self.weight_sum if self.weight_sum > 0 else 0.0

    def x(self) -> Tensor:
        return (self.y - (1.0 - self.beta) * self.z) / self.beta

    def step(self, grad: Tensor, *, project_rows: bool = False, renorm_rows: bool = False) -> float:
        g = grad.double()
        if project_rows:
            g = g - (g * self.y).sum(1, keepdim=True) * self.y
        c = self.next_c()
        x = self.x()
        z = self.z - self.lr * torch.sign(g)
        x = (1.0 - c) * x + c * z
        y = (1.0 - self.beta) * z + self.beta * x
        if renorm_rows:
            # fp32 arithmetic of sparse_coder.py:273-279 on the fp32 parameter
            y32 = y.float()
            y = (y32 / (torch.norm(y32, dim=1, keepdim=True) + EPS32)).double()
        self.y, self.z = y, z
        return c

    def eval_weights(self) -> Tensor:
        """What .eval() leaves in the parameter: the averaged iterate x."""
        return self.x()
