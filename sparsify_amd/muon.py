"""Muon with the reference's surface (sparsify/muon.py:11-181). The update math is the
reference's; the Newton-Schulz iterations are plain dense bf16 GEMMs and go to the vendor
GEMM library through torch.matmul (library GEMMs are not part of the hand-written hot
path). Under DDP the parameters are dealt round-robin to ranks, each rank orthogonalises
its own, and the bf16 updates are all-gathered asynchronously (RCCL) exactly as
muon.py:143-181 does."""
from __future__ import annotations

import torch
import torch.distributed as dist
from torch import Tensor


def quintic_newtonschulz(G: Tensor, steps: int) -> Tensor:
    """Quintic Newton-Schulz orthogonalisation, coefficients (3.4445, -4.7750, 2.0315),
    bf16, rows <= cols orientation, Frobenius normalisation + 1e-7 (muon.py:11-40)."""
    assert G.ndim >= 2
    a, b, c = (3.4445, -4.7750, 2.0315)
    X = G.bfloat16()
    tall = G.size(-2) > G.size(-1)
    if tall:
        X = X.mT
    X = X / (X.norm(dim=(-2, -1), keepdim=True) + 1e-7)
    for _ in range(steps):
        A = X @ X.mT
        B = b * A + c * A @ A
        X = a * X + B @ X
    return X.mT if tall else X


class Muon(torch.optim.Optimizer):
    """MomentUm Orthogonalized by Newton-schulz."""

    def __init__(self, params, lr: float = 1e-3, momentum: float = 0.95, nesterov: bool = True,
                 weight_decay: float = 0.1, ns_steps: int = 5, ddp: bool = True):
        defaults = dict(lr=lr, momentum=momentum, nesterov=nesterov, ns_steps=ns_steps, weight_decay=weight_decay)
        params = list(params)
        use_ddp = dist.is_initialized() and ddp and dist.get_world_size() > 1
        self.rank = dist.get_rank() if use_ddp else 0
        self.world_size = dist.get_world_size() if use_ddp else 1
        if use_ddp:
            devices = {p.device for p in params}
            assert len(devices) == 1, "Muon does not support model parallelism."
            device = next(iter(devices))
            groups = []
            for size in sorted({p.numel() for p in params}):
                buf = torch.empty(self.world_size, size, dtype=torch.bfloat16, device=device)
                groups.append(dict(params=[p for p in params if p.numel() == size], update_buffer=buf,
                                   update_buffer_views=[buf[i] for i in range(self.world_size)]))
            super().__init__(groups, defaults)
        else:
            super().__init__(params, defaults)

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            params: list[Tensor] = group["params"]
            if group["weight_decay"] > 0.0:
                for p in params:
                    p.mul_(1 - group["lr"] * group["weight_decay"])
            update_buffer = group.get("update_buffer")
            views = group.get("update_buffer_views", [])
            beta = group["momentum"]
            handle, params_world = None, None

            def apply_prev():
                handle.wait()
                for p_world, g_world in zip(params_world, views):
                    scale = 0.2 * max(p_world.shape) ** 0.5          # arXiv:2502.16982 heuristic
                    p_world.add_(g_world.view_as(p_world), alpha=-group["lr"] * scale)

            for i in range(0, len(params), self.world_size):
                if i + self.rank < len(params):
                    p = params[i + self.rank]
                    state = self.state[p]
                    g = p.grad
                    assert g is not None
                    if beta > 0.0:
                        if "exp_avg" not in state:
                            state["exp_avg"] = torch.zeros_like(g)
                        buf = state["exp_avg"].lerp_(g, 1 - beta)
                        g = g.lerp_(buf, beta) if group["nesterov"] else buf
                    if g.ndim == 4:
                        g = g.view(len(g), -1)
                    g = quintic_newtonschulz(g, steps=group["ns_steps"])
                else:
                    g = views[self.rank]
                if self.world_size > 1:
                    if i > 0:
                        apply_prev()
                    handle = dist.all_gather_into_tensor(update_buffer, g.flatten(), async_op=True)
                    params_world = params[i: i + self.world_size]
                else:
                    scale = 0.2 * max(params[i].shape) ** 0.5
                    params[i].add_(g, alpha=-group["lr"] * scale)
            if self.world_size > 1:
                apply_prev()
