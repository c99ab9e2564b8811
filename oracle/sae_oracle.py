"""CPU restatement of the reference SAE / transcoder train step (torch, CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py). The backward pass is written
out by hand from the math instead of going through autograd, so it is an
independent statement of what the reference's autograd graph computes; it is
pinned against the real reference by tests/test_oracle_golden.py.

Notation: N tokens, D = d_in, M latents, k active latents.
``bf16=True`` reproduces the dtype flow of the reference's bf16 autocast
(sparsify/sparse_coder.py:205-209): only the encoder linear runs in bf16, its
output (pre-activations, top-k values) is bf16, everything downstream is fp32,
and the gradient that flows back into the top-k values is rounded to bf16 by
the backward of ``top_acts.to(self.dtype)`` (sparse_coder.py:201).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F
from torch import Tensor


@dataclass
class OracleConfig:
    k: int
    activation: str = "topk"          # "topk" | "groupmax"  (fused_encoder.py:32-45)
    transcode: bool = False           # sparse_coder.py:191
    multi_topk: bool = False          # sparse_coder.py:256
    normalize_decoder: bool = True


# --------------------------------------------------------------------------- encoder
def pre_activations(xin: Tensor, W_enc: Tensor, b_enc: Tensor, bf16: bool) -> Tensor:
    """relu(xin @ W_enc^T + b_enc)  -- fused_encoder.py:29.

    Under autocast every operand of the linear is cast to bf16 and the result is bf16.
    """
    if bf16:
        z = F.linear(xin.bfloat16(), W_enc.bfloat16(), b_enc.bfloat16())
    else:
        z = F.linear(xin, W_enc, b_enc)
    return z.clamp_min_(0)


def select_latents(pre: Tensor, k: int, activation: str) -> tuple[Tensor, Tensor]:
    """Top-k (unsorted) or group-max selection -- fused_encoder.py:32-45."""
    if activation == "topk":
        return torch.topk(pre, k, dim=-1, sorted=False)
    if activation == "groupmax":
        M = pre.shape[-1]
        width = M // k
        vals, local = pre.reshape(*pre.shape[:-1], k, width).max(dim=-1)
        return vals, local + torch.arange(0, M, width, device=pre.device)
    raise ValueError(f"Unknown activation: {activation}")


def encode(x: Tensor, P: dict[str, Tensor], cfg: OracleConfig, bf16: bool = False):
    """SparseCoder.encode -- sparse_coder.py:189-196. Returns (acts, idx, pre, xin)."""
    xin = x if cfg.transcode else x - P["b_dec"]
    pre = pre_activations(xin, P["W_enc"], P["b_enc"], bf16)
    acts, idx = select_latents(pre, cfg.k, cfg.activation)
    return acts, idx, pre, xin


# --------------------------------------------------------------------------- decoder
def gather_sum(idx: Tensor, coeff: Tensor, table: Tensor) -> Tensor:
    """out[n,:] = sum_s coeff[n,s] * table[idx[n,s],:]  -- utils.py:82-85 (embedding_bag, mode=sum)."""
    out = torch.zeros(idx.shape[0], table.shape[1], dtype=table.dtype)
    for s in range(idx.shape[1]):
        out.addcmul_(coeff[:, s : s + 1].to(table.dtype), table[idx[:, s]])
    return out


def decode(idx: Tensor, acts: Tensor, W_dec: Tensor, b_dec: Tensor) -> Tensor:
    """SparseCoder.decode -- sparse_coder.py:198-202."""
    return gather_sum(idx, acts.to(W_dec.dtype), W_dec) + b_dec


def scatter_rows(idx: Tensor, coeff: Tensor, rows: Tensor, M: int, chunk: int = 4096) -> Tensor:
    """G[j,:] = sum_{(n,s): idx[n,s]=j} coeff[n,s] * rows[n,:]   (dense (M,D) result).

    This is the weight-gradient of both the decoder (xformers.py:110-136, coeff = top
    activations, rows = dL/d sae_out) and the encoder (fused_encoder.py:67-77, coeff =
    dL/d top_acts, rows = encoder input). Accumulated token-major in chunks so the
    (N*k, D) temporary of the reference never exists in full.
    """
    N, k = idx.shape
    G = torch.zeros(M, rows.shape[1], dtype=torch.float32)
    for a in range(0, N, chunk):
        b = min(N, a + chunk)
        contrib = coeff[a:b].unsqueeze(2) * rows[a:b].unsqueeze(1)      # promotes like the reference
        G.index_add_(0, idx[a:b].reshape(-1), contrib.reshape(-1, rows.shape[1]).float())
    return G


def scatter_scalar(idx: Tensor, coeff: Tensor, M: int) -> Tensor:
    """g[j] = sum_{idx[n,s]=j} coeff[n,s]  -- fused_encoder.py:80-84."""
    g = torch.zeros(M, dtype=torch.float32)
    g.index_add_(0, idx.reshape(-1), coeff.reshape(-1).float())
    return g


# --------------------------------------------------------------------------- loss pieces
def total_variance(y: Tensor) -> Tensor:
    """sum((y - mean_0(y))^2) over batch and features -- sparse_coder.py:228.

    For bf16 activations the reference runs under CUDA autocast: mean and the subtraction
    stay in bf16, pow and sum are on autocast's fp32 list."""
    return (y - y.mean(0)).float().pow(2).sum()


def auxk_select(pre: Tensor, dead_mask: Tensor, d_in: int):
    """AuxK dead-latent selection -- sparse_coder.py:231-243.

    Returns (scale, acts, idx) or None when there is no dead latent.
    """
    num_dead = int(dead_mask.sum())
    if num_dead <= 0:
        return None
    k_aux = d_in // 2
    scale = min(num_dead / k_aux, 1.0)
    k_aux = min(k_aux, num_dead)
    masked = torch.where(dead_mask[None], pre, -torch.inf)
    acts, idx = masked.topk(k_aux, sorted=False)
    return scale, acts, idx


# --------------------------------------------------------------------------- full step
def forward_backward(
    P: dict[str, Tensor],
    x: Tensor,
    y: Tensor | None,
    cfg: OracleConfig,
    *,
    dead_mask: Tensor | None = None,
    auxk_alpha: float = 0.0,
    loss_scale: float = 1.0,
    bf16: bool = False,
    need_grads: bool = True,
    forced: dict | None = None,
):
    """SparseCoder.forward (sparse_coder.py:210-271) followed by the Trainer's
    ``(fvu + alpha*auxk + multi_topk/8).div(acc_steps).backward()`` (trainer.py:444-447).

    ``loss_scale`` is 1/acc_steps. Returns (out, grads): ``out`` mirrors ForwardOutput
    plus ``pre_acts``; ``grads`` maps parameter name -> dense fp32 gradient.

    ``forced`` (tests only) replaces a selection by one made elsewhere: keys "top", "aux",
    "multi" -> (acts, idx). bf16 pre-activations tie exactly at the k-th boundary on a
    sizeable fraction of rows and torch.topk's pick among ties is unspecified, so the GPU
    path is checked (a) for choosing an equally valid set and (b) against this oracle
    evaluated on the very set it chose.
    """
    forced = forced or {}
    W_enc, b_enc, W_dec, b_dec = P["W_enc"], P["b_enc"], P["W_dec"], P["b_dec"]
    W_skip = P.get("W_skip")
    M, D = W_enc.shape
    acts, idx, pre, xin = encode(x, P, cfg, bf16)
    if "top" in forced:
        acts, idx = forced["top"]
    if y is None:
        y = x

    sae_out = decode(idx, acts, W_dec, b_dec)
    if W_skip is not None:
        if bf16:
            sae_out += x.bfloat16() @ W_skip.bfloat16().mT
        else:
            sae_out += x.to(W_dec.dtype) @ W_skip.mT

    e = y - sae_out
    tv = total_variance(y)

    aux = auxk_select(pre, dead_mask, y.shape[-1]) if dead_mask is not None else None
    if aux is not None:
        a_scale, a_acts, a_idx = aux
        if "aux" in forced:
            a_acts, a_idx = forced["aux"]
        e_hat = decode(a_idx, a_acts, W_dec, b_dec)
        auxk_loss = a_scale * (e_hat - e).pow(2).sum() / tv
    else:
        auxk_loss = sae_out.new_tensor(0.0)

    fvu = e.pow(2).sum() / tv

    out_acts, out_idx, out_sae = acts, idx, sae_out
    if cfg.multi_topk:
        m_acts, m_idx = forced["multi"] if "multi" in forced else pre.topk(4 * cfg.k, sorted=False)
        m_out = decode(m_idx, m_acts, W_dec, b_dec)
        multi_topk_fvu = (m_out - y).pow(2).sum() / tv
        out_acts, out_idx, out_sae = m_acts, m_idx, m_out
    else:
        multi_topk_fvu = sae_out.new_tensor(0.0)

    out = dict(sae_out=out_sae, latent_acts=out_acts, latent_indices=out_idx, fvu=fvu,
               auxk_loss=auxk_loss, multi_topk_fvu=multi_topk_fvu, pre_acts=pre,
               top_acts=acts, top_indices=idx)
    if not need_grads:
        return out, {}

    # ---------------- backward, by hand -------------------------------------------
    # d fvu / d sae_out = -2 e / TV.  TV depends on the data only.
    g = (-2.0 * loss_scale / tv) * e                                   # (N, D) fp32
    gW_dec = scatter_rows(idx, acts.to(torch.float32), g, M)
    gb_dec = g.sum(0)
    # dL/d top_acts[n,s] = <g[n,:], W_dec[idx[n,s],:]>  (xformers.py:126-131)
    d_acts = torch.einsum("nd,nsd->ns", g, W_dec[idx]) if idx.numel() * D < 2**28 else \
        torch.stack([(g * W_dec[idx[:, s]]).sum(-1) for s in range(idx.shape[1])], 1)
    # backward of `top_acts.to(fp32)`: the gradient is cast to the dtype of top_acts.
    gv = d_acts.to(acts.dtype)
    gW_skip = None
    if W_skip is not None:
        # sae_out += x @ W_skip^T ; under autocast the matmul runs in bf16 and so does
        # its backward (grad cast to bf16, result cast back to the fp32 parameter).
        if bf16:
            gW_skip = (g.bfloat16().mT @ x.bfloat16()).float()
        else:
            gW_skip = g.mT @ x.to(g.dtype)

    if aux is not None:
        # auxk = a_scale * sum((e_hat - e.detach())^2) / TV  (sparse_coder.py:247-249)
        g_aux = (2.0 * a_scale * auxk_alpha * loss_scale / tv) * (e_hat - e)
        gW_dec += scatter_rows(a_idx, a_acts.to(torch.float32), g_aux, M)
        gb_dec = gb_dec + g_aux.sum(0)
        # the gradient w.r.t. a_acts flows into pre_acts and is dropped by the encoder
        # backward (fused_encoder.py:53,87), so nothing else to do.
    if cfg.multi_topk:
        g_m = (2.0 * loss_scale / (8.0 * tv)) * (m_out - y)
        gW_dec += scatter_rows(m_idx, m_acts.to(torch.float32), g_m, M)
        gb_dec = gb_dec + g_m.sum(0)

    # encoder backward over the k selected columns only (fused_encoder.py:52-87)
    gW_enc = scatter_rows(idx, gv, xin, M)
    gb_enc = scatter_scalar(idx, gv, M)
    if not cfg.transcode:
        # xin = x - b_dec  =>  b_dec also receives -sum_n grad_xin[n,:]
        g_xin = gather_sum(idx, gv.to(W_enc.dtype), W_enc)
        gb_dec = gb_dec - g_xin.sum(0)

    grads = {"W_enc": gW_enc, "b_enc": gb_enc, "W_dec": gW_dec, "b_dec": gb_dec}
    if gW_skip is not None:
        grads["W_skip"] = gW_skip
    out["d_top_acts"] = gv
    return out, grads


def backward_from_output_grad(P: dict[str, Tensor], xin: Tensor, idx: Tensor, acts: Tensor, g: Tensor,
                              cfg: OracleConfig):
    """Parameter gradients (and the gradient w.r.t. the encoder input) of any scalar loss, given
    g = dL/d sae_out for sae_out = decode(acts, idx): the decoder backward (xformers.py:88-136), the
    cast of dL/d top_acts to the dtype of top_acts (backward of sparse_coder.py:201) and the
    encoder backward over the selected columns (fused_encoder.py:52-87). This is what the
    end-to-end losses (trainer.py:425-430, 486-500) send through a coder: there g comes out of the
    language model instead of the FVU. `xin` is the encoder input (x - b_dec for autoencoders)."""
    W_enc, W_dec = P["W_enc"], P["W_dec"]
    M = W_enc.shape[0]
    g = g.float()
    gW_dec = scatter_rows(idx, acts.to(torch.float32), g, M)
    gb_dec = g.sum(0)
    d_acts = torch.einsum("nd,nsd->ns", g, W_dec[idx])
    gv = d_acts.to(acts.dtype)
    gW_enc = scatter_rows(idx, gv, xin, M)
    gb_enc = scatter_scalar(idx, gv, M)
    g_xin = gather_sum(idx, gv.to(W_enc.dtype), W_enc)
    if not cfg.transcode:
        gb_dec = gb_dec - g_xin.sum(0)
    return {"W_enc": gW_enc, "b_enc": gb_enc, "W_dec": gW_dec, "b_dec": gb_dec}, g_xin


def fused_encoder_fwd_bwd(x, W, b, k, activation, grad_values):
    """FusedEncoder forward + backward for an arbitrary upstream grad on the values
    (fused_encoder.py:20-87); mirrors tests/test_encode.py:43-47 when grad_values == 1."""
    pre = pre_activations(x, W, b, False)
    vals, idx = select_latents(pre, k, activation)
    gx = gather_sum(idx, grad_values.to(W.dtype), W)
    gW = scatter_rows(idx, grad_values, x, W.shape[0])
    gb = scatter_scalar(idx, grad_values, W.shape[0])
    return vals, idx, pre, gx, gW, gb


def decode_backward(idx, acts, W, g):
    """Backward of the sparse decoder (xformers.py:88-136): (W.grad, acts.grad)."""
    gW = scatter_rows(idx, acts.float(), g, W.shape[0])
    d_acts = torch.einsum("nd,nsd->ns", g, W[idx])
    return gW, d_acts


# --------------------------------------------------------------------------- dense updates
def unit_norm_rows(W: Tensor) -> Tensor:
    """W / (||W||_row + eps)  -- sparse_coder.py:273-279."""
    eps = torch.finfo(W.dtype).eps
    return W / (torch.norm(W, dim=1, keepdim=True) + eps)


def remove_parallel_component(G: Tensor, W: Tensor) -> Tensor:
    """G - <G,W>_row * W  -- sparse_coder.py:281-295."""
    par = (G * W).sum(1, keepdim=True)
    return G - par * W


def signsgd_update(p: Tensor, g: Tensor, lr: float) -> Tensor:
    """p - lr*sign(g)  -- sign_sgd.py:23."""
    return p - lr * torch.sign(g)


def newton_schulz5(G: Tensor, steps: int = 5) -> Tensor:
    """Quintic Newton-Schulz orthogonalisation in bf16 -- muon.py:11-40."""
    a, b, c = 3.4445, -4.7750, 2.0315
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


def muon_update(p: Tensor, g: Tensor, buf: Tensor, lr: float, momentum: float = 0.95,
                nesterov: bool = True, weight_decay: float = 0.1, ns_steps: int = 5):
    """One single-process Muon step for one matrix -- muon.py:115-178.

    Returns (new_p, new_buf). ``g`` is modified like the reference does (lerp_ in place
    on a copy here).
    """
    p = p * (1 - lr * weight_decay)
    buf = torch.lerp(buf, g, 1 - momentum)
    upd = torch.lerp(g, buf, momentum) if nesterov else buf
    o = newton_schulz5(upd, ns_steps)
    scale = 0.2 * max(p.shape) ** 0.5
    return p - lr * scale * o.to(p.dtype), buf


def signum_lr(num_latents: int) -> float:
    """trainer.py:152"""
    return 5e-3 / (num_latents / (2**14)) ** 0.5


def adam_lr(num_latents: int) -> float:
    """trainer.py:109"""
    return 2e-4 / (num_latents / (2**14)) ** 0.5


# --------------------------------------------------------------------------- init + loop
def init_params(d_in: int, num_latents: int, *, transcode: bool = False, skip: bool = False,
                seed: int = 0, normalize_decoder: bool = True) -> dict[str, Tensor]:
    """SparseCoder.__init__ -- sparse_coder.py:39-74 (nn.Linear default init, bias 0,
    W_dec = W_enc row-normalised for SAEs / zeros for transcoders, b_dec = 0)."""
    torch.manual_seed(seed)
    lin = torch.nn.Linear(d_in, num_latents)
    W_enc = lin.weight.data.clone()
    P = {"W_enc": W_enc, "b_enc": torch.zeros(num_latents)}
    if transcode:
        P["W_dec"] = torch.zeros_like(W_enc)
    else:
        P["W_dec"] = unit_norm_rows(W_enc.clone()) if normalize_decoder else W_enc.clone()
    P["b_dec"] = torch.zeros(d_in)
    if skip:
        P["W_skip"] = torch.zeros(d_in, d_in)
    return P


def train_step_signsgd(P, x, y, cfg: OracleConfig, lr: float, *, step: int, bf16: bool = False,
                       dead_mask=None, auxk_alpha: float = 0.0):
    """One Trainer micro-step + step tail with plain SignSGD, grad_acc_steps = 1
    (trainer.py:392-447 and 510-522). Mutates P in place, returns the ForwardOutput dict."""
    if step == 0:
        # bias init -- trainer.py:394-404
        if cfg.transcode:
            P["b_enc"] = -(x.float().mean(0)) @ P["W_enc"].T
        P["b_dec"] = (y if y is not None else x).mean(0).float()
    if cfg.normalize_decoder and not cfg.transcode:
        P["W_dec"] = unit_norm_rows(P["W_dec"])
    out, G = forward_backward(P, x, y, cfg, dead_mask=dead_mask, auxk_alpha=auxk_alpha, bf16=bf16)
    if cfg.normalize_decoder and not cfg.transcode:
        G["W_dec"] = remove_parallel_component(G["W_dec"], P["W_dec"])
    for name in list(P):
        P[name] = signsgd_update(P[name], G[name], lr)
    return out
