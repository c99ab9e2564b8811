"""Muon on the HIP path: momentum, Nesterov look-ahead, quintic Newton-Schulz orthogonalisation
and the scaled update of sparsify/muon.py:11-40, 115-181, with the optimizer surface the
reference's trainer uses (``Muon(params, lr=..., ddp=...)``, trainer.py:123-145).

How the step is laid out here (none of this structure is the reference's):

* One pass over the gradient (sphip_muon_momentum) updates the momentum buffer, forms the
  Nesterov update, casts it to bf16 and takes its Frobenius sum of squares; a second small pass
  normalises.
* The iterate is kept TALL, T = (long side, short side), whatever the parameter's orientation
  (the reference orthogonalises with rows <= columns, X = T^T). One Newton-Schulz step is then

      A = T^T T          contraction over the long side -> the kernel reads T^T (one transposition)
      B = b A + c A A    short x short (A is bitwise symmetric, so A A = A A^T)
      T = a T + T B      (B symmetric)

  three launches of the bf16 MFMA GEMM whose epilogue applies torch's bf16 roundings
  (sphip_gemm_bf16_axpy); element for element these are the reference's X X^T, b A + c A A and
  a X + B X.
* Data parallel (``ddp=True`` under torch.distributed): the reference deals whole parameters to
  ranks round-robin and all-gathers their bf16 updates (muon.py:143-181), which leaves all but two
  ranks idle when a coder has two big matrices. Here the rows of a tall iterate are split over
  the ranks instead: every rank keeps 1/W of the momentum buffer (gathered into the reference's
  whole-parameter ``exp_avg`` on save, cut back on load), runs the products on its rows,
  the short x short Gram matrix is summed over ranks in fp32 (one all-reduce of S x S per
  Newton-Schulz step, RCCL), and the bf16 update rows are all-gathered once at the end -- the same
  bytes on the wire as the reference's all-gather, 1/W of the FLOPs per rank for every matrix.
  Matrices that are not tall (W_skip) are orthogonalised redundantly on every rank.

The gradients must already be averaged over ranks when step() runs (the Trainer's GradReducer
does that; the reference relies on DDP's all-reduce in the same way).
"""
from __future__ import annotations

import math

import torch
from torch import Tensor

from . import ops, parallel
from .fused_encoder import peek_bf16_shadow, register_bf16_shadow

NS_A, NS_B, NS_C = 3.4445, -4.7750, 2.0315          # quintic coefficients, muon.py:22


def _gram_tall(T: Tensor, Tt: Tensor, shared: bool) -> Tensor:
    """bf16 A = T^T T (short x short). shared: T holds this rank's rows only -- the partial Gram
    matrices are summed over ranks in fp32 before the one rounding to bf16."""
    ops.transpose_bf16(T, Tt)
    if not shared:
        return ops.gemm_bf16_axpy(Tt, Tt)
    S = Tt.shape[0]
    part = torch.empty(S, S, dtype=torch.float32, device=T.device)
    ops.gemm_bf16_f32acc(Tt, Tt, part, accumulate=False)
    parallel.all_reduce_(part, "sum")
    return ops.cast_bf16(part)


def newton_schulz_tall(T: Tensor, steps: int, shared: bool = False) -> Tensor:
    """Orthogonalises the normalised bf16 iterate T (L, S), L >= S (or this rank's rows of it when
    `shared`); returns the result (a fresh tensor or T's partner buffer)."""
    L, S = T.shape
    Tt = torch.empty(S, L, dtype=torch.bfloat16, device=T.device)
    nxt = torch.empty_like(T)
    for _ in range(steps):
        A = _gram_tall(T, Tt, shared)
        B = ops.gemm_bf16_axpy(A, A, C=A, alpha=NS_C, beta=NS_B)
        ops.gemm_bf16_axpy(T, B, C=T, alpha=1.0, beta=NS_A, out=nxt)
        T, nxt = nxt, T
    return T


def quintic_newtonschulz(G: Tensor, steps: int) -> Tensor:
    """muon.py:11-40 for one matrix on the HIP kernels: bf16 orthogonalisation of G (any float
    dtype, 2-D), returned in G's orientation."""
    assert G.ndim == 2, "quintic_newtonschulz: one matrix at a time"
    u, norm = ops.muon_momentum(G.detach().float().contiguous(), None, 0.0, False)
    ops.muon_normalise_(u, norm)
    R, C = u.shape
    if R > C:
        return newton_schulz_tall(u, steps)
    return ops.transpose_bf16(newton_schulz_tall(ops.transpose_bf16(u), steps))


class Muon(torch.optim.Optimizer):
    """MomentUm Orthogonalized by Newton-schulz (Keller Jordan et al.), for >= 2-D parameters."""

    def __init__(self, params, lr: float = 1e-3, momentum: float = 0.95, nesterov: bool = True,
                 weight_decay: float = 0.1, ns_steps: int = 5, ddp: bool = True):
        super().__init__(list(params), dict(lr=lr, momentum=momentum, nesterov=nesterov, ns_steps=ns_steps,
                                            weight_decay=weight_decay))
        self.row_sharded = bool(ddp) and parallel.world_size() > 1

    def _rows_of(self, p: Tensor):
        """(lo, hi, shared): the rows of p's gradient this rank orthogonalises."""
        R = p.shape[0]
        C = p.numel() // R
        w = parallel.world_size()
        if self.row_sharded and R > C and R % w == 0 and (R // w) % 64 == 0 and C % 4 == 0:
            per = R // w
            return parallel.rank() * per, (parallel.rank() + 1) * per, True
        return 0, R, False

    # ---- optimizer state in the reference's format: state["exp_avg"] with the parameter's whole shape
    # (muon.py:154-157; rank 0 writes optimizer_{i}.pt, trainer.py:663-668). Between steps a
    # row-sharded rank holds its rows only, so state_dict() is a collective there (every rank calls
    # it, every rank gets the whole buffer) and load_state_dict() keeps this rank's rows.
    collective_state_dict = True

    def _indexed_params(self):
        i = 0
        for group in self.param_groups:
            for p in group["params"]:
                yield i, p
                i += 1

    @torch.no_grad()
    def state_dict(self):
        sd = super().state_dict()
        for i, p in self._indexed_params():
            st = sd["state"].get(i)
            if not st or "exp_avg" not in st:
                continue
            buf = st["exp_avg"]
            lo, hi, shared = self._rows_of(p)
            if shared:
                buf = parallel.all_gather_rows(buf.contiguous())
            sd["state"][i] = dict(st, exp_avg=buf.reshape(p.shape))
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for _, p in self._indexed_params():
            st = self.state.get(p)
            if not st:
                continue
            if "momentum_buffer" in st and "exp_avg" not in st:     # this package's key before round 3
                st["exp_avg"] = st.pop("momentum_buffer")
            if "exp_avg" not in st:
                continue
            R = p.shape[0]
            C = p.numel() // R
            lo, hi, _ = self._rows_of(p)
            buf = st["exp_avg"]
            if buf.numel() == p.numel():
                st["exp_avg"] = buf.reshape(R, C)[lo:hi].to(torch.float32).contiguous().clone()
            elif tuple(buf.shape) == (hi - lo, C):
                st["exp_avg"] = buf.to(torch.float32).contiguous()
            else:
                raise ValueError(f"Muon: momentum state of shape {tuple(buf.shape)} fits neither the parameter "
                                 f"{tuple(p.shape)} nor this rank's rows ({hi - lo}, {C}): a checkpoint written by one "
                                 "rank of a row-sharded run before the state was gathered on save")

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None, "Closure is not supported."
        for group in self.param_groups:
            lr, beta, wd = group["lr"], group["momentum"], group["weight_decay"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise TypeError("Muon: fp32 parameters on the GPU only (no CPU fallback)")
                parallel.wait_for(p)
                R = p.shape[0]
                C = p.numel() // R                                  # muon.py:162-163: >2-D -> (len, -1)
                g = p.grad.reshape(R, C)
                g = g if g.is_contiguous() else g.contiguous()
                lo, hi, shared = self._rows_of(p)
                st = self.state[p]
                if beta > 0 and "exp_avg" not in st:                 # the reference's key, muon.py:154-157
                    st["exp_avg"] = torch.zeros(hi - lo, C, dtype=torch.float32, device=p.device)
                u, norm = ops.muon_momentum(g[lo:hi], st.get("exp_avg"), beta, group["nesterov"])
                if shared:
                    parallel.all_reduce_(norm[:1], "sum")           # Frobenius norm over all rows
                    ops.muon_denominator_(norm)
                ops.muon_normalise_(u, norm)
                if R > C:
                    o = newton_schulz_tall(u, group["ns_steps"], shared)
                    if shared:
                        o = parallel.all_gather_rows(o)
                else:
                    o = ops.transpose_bf16(newton_schulz_tall(ops.transpose_bf16(u), group["ns_steps"]))
                shadow = peek_bf16_shadow(p)
                ops.muon_apply_(p.data, o, 1.0 - lr * wd if wd > 0 else 1.0, lr * 0.2 * math.sqrt(max(R, C)), shadow)
                torch.autograd.graph.increment_version(p)
                if shadow is not None:
                    register_bf16_shadow(p, shadow)
