"""Schedule-free wrapper around a base optimizer (used for the reference's default
``optimizer="signum"``: ScheduleFreeWrapper(SignSGD, momentum=0.95), trainer.py:146-162).

PARITY UNPINNED: the reference takes this class from the third-party package
``schedulefree`` (unpinned in its pyproject.toml, not vendored, absent from this image and
untouched by any reference test). This file restates the published schedule-free wrapper
algorithm (Defazio et al., "The Road Less Scheduled", 2024):

    z : the base optimizer's iterate        x : the running average (evaluation point)
    y = (1 - beta) z + beta x               : where gradients are taken (train mode)

    step:  y -> x (undo the interpolation), base optimizer steps on z,
           x <- (1 - c) x + c z   with  c = w / sum(w),  w = lr_max ** weight_lr_power,
           parameter <- y = beta x + (1 - beta) z
    eval(): parameter <- x          train(): parameter <- y

With the SignSGD base of this package on HIP tensors the whole step is ONE kernel per parameter
(sphip_schedulefree_signsgd_step, for decoder groups with the projected gradient and the
re-normalisation fused in: sphip_decoder_schedulefree_signsgd_step) instead of the ~10
full-tensor passes of the generic path below; both paths implement the same arithmetic.
"""
from __future__ import annotations

import torch


class ScheduleFreeWrapper:
    def __init__(self, base: torch.optim.Optimizer, momentum: float = 0.9, weight_lr_power: float = 2.0,
                 r: float = 0.0):
        self.base = base
        self.momentum = momentum
        self.weight_lr_power = weight_lr_power
        self.r = r
        self.train_mode = False
        for group in self.param_groups:
            group.setdefault("k", 0)
            group.setdefault("weight_sum", 0.0)
            group.setdefault("lr_max", 0.0)

    @property
    def param_groups(self):
        return self.base.param_groups

    @property
    def state(self):
        return self.base.state

    def zero_grad(self, set_to_none: bool = True):
        self.base.zero_grad(set_to_none)

    def state_dict(self):
        return {"base": self.base.state_dict(), "train_mode": self.train_mode}

    def load_state_dict(self, sd):
        self.base.load_state_dict(sd["base"])
        self.train_mode = sd["train_mode"]

    @torch.no_grad()
    def eval(self):
        if self.train_mode:
            for group in self.param_groups:
                for p in group["params"]:
                    st = self.state[p]
                    if "z" in st:
                        p.lerp_(st["z"], 1 - 1 / self.momentum)      # y -> x
            self.train_mode = False

    @torch.no_grad()
    def train(self):
        if not self.train_mode:
            for group in self.param_groups:
                for p in group["params"]:
                    st = self.state[p]
                    if "z" in st:
                        p.lerp_(st["z"], 1 - self.momentum)          # x -> y
            self.train_mode = True

    @property
    def renormalised(self):
        return getattr(self.base, "renormalised", None)

    def reduces(self, p) -> bool:
        """True for parameters whose gradients this optimizer averages across ranks itself
        (the base SignSGD's sign exchange); the trainer all-reduces the others."""
        f = getattr(self.base, "reduces", None)
        return bool(f and f(p))

    def _peek_weights(self, group) -> float:
        """The c the NEXT step() of this group will use, without advancing its state."""
        lr = group["lr"] * group.get("d", 1.0)
        lr_max = max(lr, group["lr_max"])
        weight = ((group["k"] + 1) ** self.r) * (lr_max ** self.weight_lr_power)
        total = group["weight_sum"] + weight
        return weight / total if total > 0 else 0.0

    def fuse_encoder_update(self, weights) -> None:
        """Optimizer-in-backward for encoder matrices, as SignSGD.fuse_encoder_update: FusedEncoder's
        backward applies THIS wrapper's step to (p = y, z) inside the weight-gradient kernel and
        leaves p.grad None, so step() only advances the group's weights for them. Same validity
        conditions (one rank, one backward per optimizer step, nothing else reads that gradient)."""
        from .fused_encoder import register_inplace_signsgd
        from .sign_sgd import SignSGD
        if not isinstance(self.base, SignSGD):
            raise TypeError("fuse_encoder_update needs the SignSGD base optimizer")
        for group in self.param_groups:
            for p in group["params"]:
                if any(p is w for w in weights):
                    if group.get("project_rows", False):
                        raise ValueError("decoder matrices need the projected gradient: not fusable")

                    def how(p=p):
                        if not self.train_mode:
                            raise RuntimeError("call .train() on the schedule-free optimizer before stepping")
                        # (looked up per call: load_state_dict replaces the group dictionaries)
                        g = next(gr for gr in self.param_groups if any(q is p for q in gr["params"]))
                        st = self.state[p]
                        if "z" not in st:
                            st["z"] = p.detach().clone(memory_format=torch.preserve_format)
                        return {"lr": g["lr"], "z": st["z"], "beta": self.momentum, "c": self._peek_weights(g)}

                    register_inplace_signsgd(p, how)

    def _weights(self, group) -> float:
        lr = group["lr"] * group.get("d", 1.0)
        group["lr_max"] = max(lr, group["lr_max"])
        weight = ((group["k"] + 1) ** self.r) * (group["lr_max"] ** self.weight_lr_power)
        group["weight_sum"] += weight
        return weight / group["weight_sum"] if group["weight_sum"] > 0 else 0.0

    def _fusable(self) -> bool:
        from .sign_sgd import SignSGD
        if not isinstance(self.base, SignSGD):
            return False
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    return False
                if group.get("project_rows", False) and (p.ndim != 2 or p.shape[1] % 4 or p.shape[1] > 8192):
                    return False
        return True

    @torch.no_grad()
    def _fused_step(self):
        from . import ops
        from .fused_encoder import peek_bf16_shadow, register_bf16_shadow
        deferred = []
        for group in self.param_groups:
            c = self._weights(group)
            project, renorm = group.get("project_rows", False), group.get("renorm_rows", False)
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if "z" not in st:
                    st["z"] = p.detach().clone(memory_format=torch.preserve_format)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if self.reduces(p):
                    # data parallel: the base optimizer's sign exchange feeds this step
                    item = (p, st["z"], g, group["lr"], c, project, renorm)
                    if self.base.overlap:
                        deferred.append(item)
                    else:
                        self._exchange_step(*item, overlap=False)
                    continue
                sm = None
                if project:
                    # (the slice-major copy of y the L2-resident decode reads is rewritten in the same pass)
                    sm = ops.peek_slice_major(p) if p.shape[1] % 32 == 0 else None
                    ops.decoder_schedulefree_signsgd_step_(p.data, st["z"], g, group["lr"], self.momentum, c, renorm,
                                                           slice_major=sm)
                    if renorm:
                        self.base.renormalised.add(id(p))
                else:
                    shadow = peek_bf16_shadow(p)
                    if (shadow is None and p.ndim == 2 and p.is_contiguous() and p.shape[1] % 32 == 0
                            and p.shape[1] <= 8192):
                        sm = ops.peek_slice_major(p)      # a transcoder's decoder, read by the L2-resident decode
                    if sm is not None:
                        ops.schedulefree_signsgd_step_rows_(p.data, st["z"], g, group["lr"], self.momentum, c, sm)
                    else:
                        ops.schedulefree_signsgd_step_(p.data, st["z"], g, group["lr"], self.momentum, c, shadow)
                    if shadow is not None:
                        register_bf16_shadow(p, shadow)
                torch.autograd.graph.increment_version(p)
                if sm is not None:
                    ops.register_slice_major(p, sm)
            group["k"] += 1
        for item in sorted(deferred, key=lambda it: it[5]):       # encoder matrices first (sign_sgd.py)
            self._exchange_step(*item, overlap=True)

    def _exchange_step(self, p, z, g, lr, c, project: bool, renorm: bool, overlap: bool):
        from . import ops
        from .fused_encoder import peek_bf16_shadow, register_bf16_shadow
        w, beta = p.data, self.momentum
        shadow = None if project else peek_bf16_shadow(p)
        sm = ops.peek_slice_major(p) if project and p.shape[1] % 32 == 0 else None        # as SignSGD._exchange_step
        if project:
            def apply_fn(signs, lo, hi):
                ops.decoder_schedulefree_apply_signs_(w[lo:hi], z[lo:hi], signs, lr, beta, c, renorm,
                                                      slice_major=sm, row0=lo)
        else:
            def apply_fn(signs, lo, hi):
                ops.schedulefree_apply_signs_(w[lo:hi], z[lo:hi], signs, lr, beta, c,
                                              None if shadow is None else shadow[lo:hi])
        self.base.run_exchange(p, g, project, apply_fn, overlap)
        if project and renorm:
            self.base.renormalised.add(id(p))
        if shadow is not None:
            register_bf16_shadow(p, shadow)
        if sm is not None:
            ops.register_slice_major(p, sm)

    @torch.no_grad()
    def step(self, closure=None):
        if not self.train_mode:
            raise RuntimeError("call .train() on the schedule-free optimizer before stepping")
        if self._fusable():
            return self._fused_step()
        # generic path; decoder groups: project the gradient w.r.t. the parameter (= y) first and
        # keep the base optimizer from projecting again w.r.t. z
        flagged = [g for g in self.param_groups if g.get("project_rows", False)]
        if flagged:
            from . import ops
            for group in flagged:
                for p in group["params"]:
                    if p.grad is not None:
                        ops.remove_parallel_grad_(p.grad, p.data)
                group["project_rows"] = False
        overlap, self.base.overlap = getattr(self.base, "overlap", False), False    # torch ops below read p at once
        try:
            self._generic_step()
        finally:
            for group in flagged:
                group["project_rows"] = True
            if hasattr(self.base, "overlap"):
                self.base.overlap = overlap
        for group in flagged:
            if group.get("renorm_rows", False):
                from . import ops
                for p in group["params"]:
                    if p.grad is not None:
                        ops.unit_norm_rows_(p.data)
                        self.base.renormalised.add(id(p))

    @torch.no_grad()
    def _generic_step(self):
        swapped = []
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if "z" not in st:
                    st["z"] = p.detach().clone(memory_format=torch.preserve_format)
                z = st["z"]
                p.lerp_(z, 1 - 1 / self.momentum)        # y -> x
                # let the base optimizer act on z: exchange the contents of p and z
                tmp = p.detach().clone()
                p.copy_(z)
                z.copy_(tmp)                             # z now holds x
                swapped.append((group, p))
        self.base.step()
        for group in self.param_groups:
            c = self._weights(group)
            for p in group["params"]:
                if p.grad is None:
                    continue
                x = self.state[p]["z"]                   # holds x since the exchange
                z_new = p.detach().clone()
                p.copy_(x)                               # p = x
                self.state[p]["z"].copy_(z_new)          # z = stepped z
                p.lerp_(self.state[p]["z"], c)           # x <- (1-c) x + c z
                p.lerp_(self.state[p]["z"], 1 - self.momentum)    # -> y
            group["k"] += 1
