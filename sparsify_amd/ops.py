"""Thin tensor-level wrappers over the C ABI (include/sparsify_hip.h).

PyTorch is used only as plumbing here: device memory (torch.empty), the current HIP
stream and dtype bookkeeping. Every wrapper validates devices/dtypes/contiguity on the
host and then passes raw pointers; nothing in this module computes with torch ops.
"""
from __future__ import annotations

import os

import torch
from torch import Tensor

from . import _lib
from ._lib import BF16, F32, check


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _dt(t: Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype}; the HIP path handles float32 and bfloat16")


def _stream(t: Tensor) -> int:
    """The current stream of t's device as a hipStream_t. (A dozen calls per step: the raw handle costs a tenth of
    building a torch.cuda.Stream object for it, which shows on launch-bound shapes.)"""
    if _raw_stream is not None and t.device.index is not None:
        return _raw_stream(t.device.index)
    return torch.cuda.current_stream(t.device).cuda_stream


def _req(t: Tensor, name: str, dtype=None, ndim: int | None = None) -> Tensor:
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on {t.device}: sparsify_amd runs on the MI355X HIP path only (no CPU fallback)"
        )
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if ndim is not None and t.ndim != ndim:
        raise ValueError(f"{name}: expected {ndim} dims, got shape {tuple(t.shape)}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def _ptr(t: Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


class KernelTimer:
    """Optional per-entry-point HIP-event timing on the launching stream (bench.py's roofline
    numbers). Disabled by default: no events, no overhead."""

    def __init__(self):
        self.enabled = False
        self.events: dict[str, list] = {}

    def start(self, name: str, t: Tensor):
        if not self.enabled:
            return None
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record(torch.cuda.current_stream(t.device))
        self.events.setdefault(name, []).append(ev)
        return ev

    def stop(self, ev, t: Tensor):
        if ev is not None:
            ev[1].record(torch.cuda.current_stream(t.device))

    def summary(self) -> dict[str, dict]:
        """name -> {calls, mean_ms, total_ms}; call after torch.cuda.synchronize()."""
        out = {}
        for name, evs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[name] = {"calls": len(ms), "mean_ms": sum(ms) / len(ms), "total_ms": sum(ms)}
        return out

    def reset(self):
        self.events = {}


timer = KernelTimer()

_workspaces: dict[tuple, Tensor] = {}


def _workspace(device: torch.device, nbytes: int, tag: str) -> Tensor:
    """Grow-only per-(device, stream, tag) scratch buffer (uint8, 256-B aligned by the allocator)."""
    stream = (_raw_stream(device.index) if _raw_stream is not None and device.index is not None
              else torch.cuda.current_stream(device).cuda_stream)
    key = (device.index, stream, tag)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        # whole 2 MiB units: with an odd size the caching allocator placed the 290 MB weight-gradient
        # workspace such that the kernels streaming through it ran 8 % slower (measured, same code)
        nbytes = -(-max(nbytes, 1 << 16) // (2 << 20)) * (2 << 20) if nbytes > (1 << 20) else max(nbytes, 1 << 16)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


# ------------------------------------------------------------------ encoder forward
def center_cast(x: Tensor, b_dec: Tensor | None, want_f32: bool, want_bf16: bool):
    """(x - b_dec) as fp32 and/or bf16. reference: sparse_coder.py:191-192 (+ autocast cast)."""
    _req(x, "x", ndim=2)
    N, D = x.shape
    if b_dec is not None:
        _req(b_dec, "b_dec", torch.float32, 1)
    of = torch.empty(N, D, dtype=torch.float32, device=x.device) if want_f32 else None
    ob = torch.empty(N, D, dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    _ev = timer.start("sphip_center_cast", x)
    check(_lib.load().sphip_center_cast(x.data_ptr(), _dt(x), _ptr(b_dec), _ptr(of), _ptr(ob), N, D, _stream(x)),
          "sphip_center_cast")
    timer.stop(_ev, x)
    return of, ob


def cast_bf16(src: Tensor, out: Tensor | None = None) -> Tensor:
    _req(src, "src", torch.float32)
    if out is None:
        out = torch.empty_like(src, dtype=torch.bfloat16)
    check(_lib.load().sphip_cast_f32_to_bf16(src.data_ptr(), out.data_ptr(), src.numel(), _stream(src)),
          "sphip_cast_f32_to_bf16")
    return out


TOPK_TILE = 128      # SPHIP_TOPK_TILE: latents per block maximum


def encoder_preacts(x: Tensor, W: Tensor, bias: Tensor | None, relu: bool = True, want_tile_max: bool = False):
    """relu(x @ W.T + bias); reference: fused_encoder.py:29. With want_tile_max also returns the
    block maxima that topk() uses to skip most of each row: a (ceil(N/64), ceil(M/128), 64)
    tensor, [n // 64, t, n % 64] = max of pre[n, 128 t : 128 (t + 1)] (rows past N undefined)."""
    _req(x, "x", ndim=2)
    _req(W, "weight", x.dtype, 2)
    if bias is not None:
        _req(bias, "bias", x.dtype, 1)
    N, D = x.shape
    M = W.shape[0]
    if W.shape[1] != D or (bias is not None and bias.shape[0] != M):
        raise ValueError(f"shape mismatch: x {tuple(x.shape)}, weight {tuple(W.shape)}")
    pre = torch.empty(N, M, dtype=x.dtype, device=x.device)
    tmax = torch.empty(-(-N // 64), -(-M // TOPK_TILE), 64, dtype=x.dtype, device=x.device) if want_tile_max else None
    _ev = timer.start("sphip_encoder_preacts", x)
    check(_lib.load().sphip_encoder_preacts(x.data_ptr(), W.data_ptr(), _ptr(bias), pre.data_ptr(), _ptr(tmax), _dt(x),
                                           N, D, M, int(relu), _stream(x)), "sphip_encoder_preacts")
    timer.stop(_ev, x)
    return (pre, tmax) if want_tile_max else pre


def encoder_preacts_uses_tile256(N: int, D: int, M: int) -> bool:
    """Whether encoder_preacts computes a bf16 (N, D) x (M, D) product with the 256x256 kernel."""
    return bool(_lib.load().sphip_encoder_preacts_uses_tile256(BF16, N, D, M))


def encoder_preacts_block(x: Tensor, W: Tensor, bias: Tensor | None, pre: Tensor, tile_max: Tensor | None,
                          lo: int, hi: int, relu: bool = True) -> None:
    """Fills pre[:, lo:hi] (and the matching blocks of tile_max) from rows lo:hi of W: one latent
    block of encoder_preacts, written into the caller's full-width buffers."""
    _req(x, "x", torch.bfloat16, 2)
    _req(W, "weight", torch.bfloat16, 2)
    _req(pre, "pre_acts", torch.bfloat16, 2)
    N, D = x.shape
    M = W.shape[0]
    if W.shape[1] != D or pre.shape != (N, M) or (bias is not None and bias.shape != (M,)):
        raise ValueError(f"shape mismatch: x {tuple(x.shape)}, weight {tuple(W.shape)}, pre {tuple(pre.shape)}")
    if not (0 <= lo < hi <= M) or lo % TOPK_TILE or (hi - lo) % TOPK_TILE:
        raise ValueError(f"latent block [{lo}, {hi}) must be whole {TOPK_TILE}-latent tiles inside [0, {M})")
    T = -(-M // TOPK_TILE)
    if tile_max is not None:
        _req(tile_max, "tile_max", torch.bfloat16, 3)
        if tuple(tile_max.shape) != (-(-N // 64), T, 64):
            raise ValueError(f"tile_max must be ({-(-N // 64)}, {T}, 64), got {tuple(tile_max.shape)}")
    es = 2
    _ev = timer.start("sphip_encoder_preacts_block", x)
    check(_lib.load().sphip_encoder_preacts_block(
        x.data_ptr(), W.data_ptr() + lo * D * es, None if bias is None else _req(bias, "bias", torch.bfloat16, 1).data_ptr() + lo * es,
        pre.data_ptr() + lo * es, None if tile_max is None else tile_max.data_ptr() + (lo // TOPK_TILE) * 64 * es,
        BF16, N, D, hi - lo, M, T, int(relu), _stream(x)), "sphip_encoder_preacts_block")
    timer.stop(_ev, x)


def topk(pre: Tensor, k: int, keep_mask: Tensor | None = None, tile_max: Tensor | None = None):
    """torch.topk(pre, k, sorted=False) (optionally restricted to keep_mask); fused_encoder.py:33.
    tile_max: the block maxima encoder_preacts(..., want_tile_max=True) returned for this `pre`."""
    _req(pre, "pre_acts", ndim=2)
    N, M = pre.shape
    if tile_max is not None:
        _req(tile_max, "tile_max", pre.dtype, 3)
        if tuple(tile_max.shape) != (-(-N // 64), -(-M // TOPK_TILE), 64):
            raise ValueError(f"tile_max must be ({-(-N // 64)}, {-(-M // TOPK_TILE)}, 64), got {tuple(tile_max.shape)}")
    if keep_mask is not None:
        _req(keep_mask, "keep_mask", ndim=1)
        if keep_mask.dtype == torch.bool:
            keep_mask = keep_mask.view(torch.uint8)
        if keep_mask.dtype != torch.uint8 or keep_mask.shape[0] != M:
            raise ValueError("keep_mask must be a (M,) bool/uint8 tensor")
    vals = torch.empty(N, k, dtype=pre.dtype, device=pre.device)
    idx = torch.empty(N, k, dtype=torch.int64, device=pre.device)
    _ev = timer.start("sphip_topk", pre)
    check(_lib.load().sphip_topk(pre.data_ptr(), _dt(pre), _ptr(keep_mask), _ptr(tile_max), N, M, k, vals.data_ptr(),
                                idx.data_ptr(), _stream(pre)), "sphip_topk")
    timer.stop(_ev, pre)
    return vals, idx


def exclude_selected(vals_wide: Tensor, idx_wide: Tensor, vals_sel: Tensor, idx_sel: Tensor) -> Tensor:
    """idx_wide (N, kw) with the members of the narrower selection of the same rows (vals_sel, idx_sel: N x k, both
    from topk over one matrix) replaced by -1, the inactive-slot mark: the multi-top-k branch
    (sparse_coder.py:256-262) decodes only the rows the main decode has not summed."""
    _req(vals_wide, "vals_wide", ndim=2)
    _req(idx_wide, "idx_wide", torch.int64, 2)
    _req(vals_sel, "vals_sel", vals_wide.dtype, 2)
    _req(idx_sel, "idx_sel", torch.int64, 2)
    N, kw = idx_wide.shape
    k = idx_sel.shape[1]
    if vals_wide.shape != idx_wide.shape or vals_sel.shape != idx_sel.shape or idx_sel.shape[0] != N or k > kw:
        raise ValueError(f"exclude_selected: shapes {tuple(vals_wide.shape)} {tuple(idx_wide.shape)} "
                         f"{tuple(vals_sel.shape)} {tuple(idx_sel.shape)} do not match")
    out = torch.empty_like(idx_wide)
    if N == 0:
        return out
    check(_lib.load().sphip_exclude_selected(vals_wide.data_ptr(), idx_wide.data_ptr(), kw, vals_sel.data_ptr(),
                                            idx_sel.data_ptr(), k, _dt(vals_wide), N, out.data_ptr(), _stream(idx_wide)),
          "sphip_exclude_selected")
    return out


LAZY_CAP = 4096          # chunks (8 latents each) a token's candidate list can hold (csrc/lazy.hip)
LAZY_MAX_K = 256


def lazy_split(M: int) -> int:
    """Latents computed the ordinary way by encode_topk_lazy (the first M_a of M): about M / 4,
    whole 256-latent tiles. Measured at the headline shape (benchmarks/lazy_bench.py): M/8 13.2 ms,
    3M/16 13.9 (fewer 128-latent blocks than k: that top-k sweeps its rows), M/4 12.6, 3M/8 12.6."""
    return max(256, -(-(M // 4) // 256) * 256)


def lazy_eligible(N: int, D: int, M: int, k: int) -> bool:
    """Shapes encode_topk_lazy takes (and pays off on): bf16, the 256x256 kernel on both parts."""
    Ma = lazy_split(M)
    return (k <= LAZY_MAX_K and D % 64 == 0 and M % 8 == 0 and M - Ma >= 256 and Ma >= 4 * k
            and encoder_preacts_uses_tile256(N, D, Ma))


LAZY_REDO_ROWS = 2048          # tokens per launch of the predicated fallback of encode_topk_lazy


def encode_topk_lazy(x: Tensor, W: Tensor, bias: Tensor | None, k: int, fallback: bool = True):
    """top-k of relu(x @ W.T + bias) without storing the (N, M) matrix (include/sparsify_hip.h,
    "top-k without the (N, M) matrix"; fused_encoder.py:29-33 when pre_acts is not read).
    Returns (vals, idx, flag): the same set as topk(encoder_preacts(...)), value-descending;
    flag is a device int32, non-zero when a candidate list overflowed -- with fallback=True the
    ordinary path has then been run behind it on the device (no host round trip) and the outputs
    are right either way."""
    _req(x, "x", torch.bfloat16, 2)
    _req(W, "weight", torch.bfloat16, 2)
    if bias is not None:
        _req(bias, "bias", torch.bfloat16, 1)
    N, D = x.shape
    M = W.shape[0]
    if W.shape[1] != D or (bias is not None and bias.shape[0] != M):
        raise ValueError(f"shape mismatch: x {tuple(x.shape)}, weight {tuple(W.shape)}")
    Ma = lazy_split(M)
    if not (k <= LAZY_MAX_K and D % 64 == 0 and M % 8 == 0 and Ma >= k and M - Ma >= 8):
        raise ValueError(f"encode_topk_lazy: unsupported shape D={D} M={M} k={k}")
    lib = _lib.load()
    st = _stream(x)
    # 1. the first Ma latents, the ordinary way
    pre_a, tmax_a = encoder_preacts(x, W[:Ma], None if bias is None else bias[:Ma], want_tile_max=True)
    vals_a, idx_a = topk(pre_a, k, tile_max=tmax_a)
    # 2. per-token threshold
    tau = torch.empty(N, dtype=torch.bfloat16, device=x.device)
    check(lib.sphip_row_min(vals_a.data_ptr(), N, k, tau.data_ptr(), st), "sphip_row_min")
    # 3. the other latents: candidate chunks only
    ws = _workspace(x.device, lib.sphip_lazy_workspace_bytes(N, LAZY_CAP), "lazy")
    _ev = timer.start("sphip_encoder_preacts_emit", x)
    check(lib.sphip_encoder_preacts_emit(x.data_ptr(), W.data_ptr() + Ma * D * 2, None if bias is None else bias.data_ptr() + Ma * 2,
                                         N, D, M - Ma, Ma, tau.data_ptr(), LAZY_CAP, 1, ws.data_ptr(), ws.numel(), st),
          "sphip_encoder_preacts_emit")
    timer.stop(_ev, x)
    # 4. selection
    vals = torch.empty(N, k, dtype=torch.bfloat16, device=x.device)
    idx = torch.empty(N, k, dtype=torch.int64, device=x.device)
    _ev = timer.start("sphip_topk_candidates", x)
    check(lib.sphip_topk_candidates(vals_a.data_ptr(), idx_a.data_ptr(), tau.data_ptr(), N, k, LAZY_CAP, ws.data_ptr(),
                                    ws.numel(), vals.data_ptr(), idx.data_ptr(), st), "sphip_topk_candidates")
    timer.stop(_ev, x)
    flag = ws[:4].view(torch.int32)
    if fallback:
        # 5. predicated: the launches return at once unless the flag is up. The redo runs LAZY_REDO_ROWS
        # tokens at a time through one small workspace (512 MiB at M = 131072): reserving the whole
        # (N, M) matrix for a path that almost never runs would hold more memory than the ordinary
        # forward, whose pre-activations go back to the allocator after every step
        R = min(N, LAZY_REDO_ROWS)
        full = _workspace(x.device, R * M * 2, "lazy_full").view(torch.bfloat16)
        T = -(-M // TOPK_TILE)
        tm = _workspace(x.device, -(-R // 64) * T * 64 * 2, "lazy_full_tmax")
        for r0 in range(0, N, R):
            n = min(R, N - r0)
            check(lib.sphip_encoder_preacts_if(flag.data_ptr(), x[r0:r0 + n].data_ptr(), W.data_ptr(), _ptr(bias),
                                               full.data_ptr(), tm.data_ptr(), n, D, M, 1, st), "sphip_encoder_preacts_if")
            check(lib.sphip_topk_if(flag.data_ptr(), full.data_ptr(), BF16, None, tm.data_ptr(), n, M, k,
                                    vals[r0:r0 + n].data_ptr(), idx[r0:r0 + n].data_ptr(), st), "sphip_topk_if")
    return vals, idx, flag


def groupmax(pre: Tensor, k: int):
    _req(pre, "pre_acts", ndim=2)
    N, M = pre.shape
    vals = torch.empty(N, k, dtype=pre.dtype, device=pre.device)
    idx = torch.empty(N, k, dtype=torch.int64, device=pre.device)
    check(_lib.load().sphip_groupmax(pre.data_ptr(), _dt(pre), N, M, k, vals.data_ptr(), idx.data_ptr(), _stream(pre)),
          "sphip_groupmax")
    return vals, idx


# ------------------------------------------------------------------ gathers / segments
_slice_major: dict[int, tuple] = {}


def slice_major_copy(table: Tensor) -> Tensor:
    """[D/32][M][32] copy of a (M, D) fp32 table (sphip_slice_major_f32), refreshed only when the
    table's version changed -- the optimizer steps bump it, like the bf16 shadow of encoder.weight.
    An entry is pinned to the tensor it was made for by a weak reference (a later parameter at a
    recycled address with the same shape and version must not hit it) and dies with that tensor."""
    import weakref
    key = table.data_ptr()
    hit = _slice_major.get(key)
    if hit is not None:
        owner = hit[0]()
        if owner is None or owner.data_ptr() != key or owner.shape != table.shape or owner._version != table._version:
            hit = None
    state = (table._version, tuple(table.shape))
    if hit is not None and hit[1] == state:
        return hit[2]
    M, D = table.shape
    buf = hit[2] if hit is not None and hit[2].numel() == table.numel() else torch.empty(D // 32, M, 32, dtype=torch.float32,
                                                                                        device=table.device)
    check(_lib.load().sphip_slice_major_f32(table.data_ptr(), buf.data_ptr(), M, D, _stream(table)), "sphip_slice_major_f32")

    def _drop(ref, key=key):
        cur = _slice_major.get(key)
        if cur is not None and cur[0] is ref:
            del _slice_major[key]

    _slice_major[key] = (weakref.ref(table, _drop), state, buf)
    return buf


def peek_slice_major(table: Tensor) -> Tensor | None:
    """The (possibly stale) slice-major buffer of `table`, for an update kernel to rewrite."""
    hit = _slice_major.get(table.data_ptr())
    if hit is None:
        return None
    owner = hit[0]()
    if owner is None or owner.data_ptr() != table.data_ptr() or owner.shape != table.shape:
        return None
    return hit[2]


def register_slice_major(table: Tensor, buf: Tensor) -> None:
    """Tell the cache that `buf` holds the slice-major copy of `table` as it is now (a fused
    optimizer step wrote it)."""
    import weakref
    key = table.data_ptr()

    def _drop(ref, key=key):
        cur = _slice_major.get(key)
        if cur is not None and cur[0] is ref:
            del _slice_major[key]

    _slice_major[key] = (weakref.ref(table, _drop), (table._version, tuple(table.shape)), buf)


def slice_major_copy_of(table: Tensor, owner: Tensor | None) -> Tensor | None:
    if owner is None or owner.data_ptr() != table.data_ptr() or owner.shape != table.shape:
        return None
    return slice_major_copy(owner)


def gather_rows_weighted(idx: Tensor, coeff: Tensor, table: Tensor, bias: Tensor | None = None,
                         table_owner: Tensor | None = None) -> Tensor:
    """out[n] = sum_s coeff[n,s] * table[idx[n,s]] (+ bias); utils.py:81-85 / fused_encoder.py:58-64."""
    _req(idx, "indices", torch.int64, 2)
    _req(coeff, "coeff", ndim=2)
    _req(table, "table", torch.float32, 2)
    if bias is not None:
        _req(bias, "bias", torch.float32, 1)
    N, k = idx.shape
    M, D = table.shape
    if coeff.shape != idx.shape:
        raise ValueError(f"coeff {tuple(coeff.shape)} vs indices {tuple(idx.shape)}")
    out = torch.empty(N, D, dtype=torch.float32, device=table.device)
    lib = _lib.load()
    # tables far beyond the caches: the L2-resident form (gather_l2.hip); SPHIP_GATHER=full|sliced keeps
    # the token-major kernels (A/B runs, bitwise comparisons between them)
    l2_bytes = lib.sphip_gather_l2_workspace_bytes(N, k, M, D) if os.environ.get("SPHIP_GATHER", "l2") == "l2" else 0
    _ev = timer.start("sphip_gather_rows_weighted", table)
    if l2_bytes:
        ws = _workspace(table.device, l2_bytes, "gather_l2")
        # table_owner: the tensor whose version says when `table` (a detached view of it) changed;
        # without it the row-major table is read directly (no copy to keep current)
        sm = slice_major_copy_of(table, table_owner) if os.environ.get("SPHIP_GATHER_SM", "1") != "0" else None
        check(lib.sphip_gather_rows_weighted_l2(idx.data_ptr(), coeff.data_ptr(), _dt(coeff), table.data_ptr(), _ptr(sm),
                                                _ptr(bias), out.data_ptr(), N, k, M, D, ws.data_ptr(), ws.numel(),
                                                _stream(table)), "sphip_gather_rows_weighted_l2")
    else:
        check(lib.sphip_gather_rows_weighted(idx.data_ptr(), coeff.data_ptr(), _dt(coeff), table.data_ptr(),
                                             _ptr(bias), out.data_ptr(), N, k, M, D, _stream(table)),
              "sphip_gather_rows_weighted")
    timer.stop(_ev, table)
    return out


class LatentSegments:
    """perm / seg_begin of one (N,k) index tensor over M latents (xformers.py:139-163)."""

    __slots__ = ("perm", "seg_begin", "N", "k", "M")

    def __init__(self, perm, seg_begin, N, k, M):
        self.perm, self.seg_begin, self.N, self.k, self.M = perm, seg_begin, N, k, M


def build_latent_segments(idx: Tensor, M: int) -> LatentSegments:
    _req(idx, "indices", torch.int64, 2)
    N, k = idx.shape
    lib = _lib.load()
    nbytes = lib.sphip_segments_workspace_bytes(N, k, M)
    ws = _workspace(idx.device, nbytes, "segments")
    perm = torch.empty(N * k, dtype=torch.int32, device=idx.device)       # holds uint32 entry ids
    seg = torch.empty(M + 1, dtype=torch.int32, device=idx.device)
    _ev = timer.start("sphip_build_latent_segments", idx)
    check(lib.sphip_build_latent_segments(idx.data_ptr(), N, k, M, perm.data_ptr(), seg.data_ptr(), ws.data_ptr(),
                                          ws.numel(), _stream(idx)), "sphip_build_latent_segments")
    timer.stop(_ev, idx)
    return LatentSegments(perm, seg, N, k, M)


_segment_cache: dict[int, tuple] = {}


_block_segment_cache: dict[int, tuple] = {}     # the same per token block (weight_grad_in_token_blocks)


def forget_segments(idx: Tensor) -> None:
    """Called by every forward that takes `idx`: drops a stale entry left by an earlier
    tensor that lived at the same address."""
    _segment_cache.pop(idx.data_ptr(), None)
    _block_segment_cache.pop(idx.data_ptr(), None)


def cached_segments(idx: Tensor, M: int) -> LatentSegments:
    """Decoder and encoder backward share one index tensor; sort it once per step.

    Entries are created in backward only and dropped by the next forward that sees the same
    address (forget_segments), and the tensor is kept alive by the autograd graph in
    between, so a hit always belongs to this very tensor."""
    key = idx.data_ptr()
    hit = _segment_cache.get(key)
    if hit is not None and hit[0] == (idx._version, tuple(idx.shape), M):
        return hit[1]
    seg = build_latent_segments(idx, M)
    if len(_segment_cache) > 16:
        _segment_cache.clear()
    _segment_cache[key] = ((idx._version, tuple(idx.shape), M), seg)
    return seg


def segmented_weight_grad(seg: LatentSegments, coeff: Tensor, rows: Tensor, W: Tensor | None = None,
                          want_bias: bool = False, want_dcoeff: bool = False, row_shift: Tensor | None = None,
                          into: tuple | None = None, dcoeff_out: Tensor | None = None):
    """(gW, gbias, dcoeff); xformers.py:88-136 and fused_encoder.py:67-84.
    into = (gW, gbias or None): add this call's sums to those tensors instead of returning new ones (a batch taken in
    token blocks, weight_grad_in_token_blocks); raises HipLibraryError (status SPHIP_EUNSUPPORTED) where the kernel form the shape
    takes cannot. dcoeff_out: where to write dcoeff (a block's rows of the whole batch's tensor)."""
    _req(coeff, "coeff", ndim=2)
    _req(rows, "rows", ndim=2)
    N, k, M = seg.N, seg.k, seg.M
    D = rows.shape[1]
    if tuple(coeff.shape) != (N, k) or rows.shape[0] != N:
        raise ValueError("segmented_weight_grad: shape mismatch")
    if W is not None:
        _req(W, "W", torch.float32, 2)
        if tuple(W.shape) != (M, D):
            raise ValueError("segmented_weight_grad: W shape mismatch")
    if want_dcoeff and W is None:
        raise ValueError("dcoeff needs W")
    if row_shift is not None:
        _req(row_shift, "row_shift", torch.float32, 1)
        if row_shift.shape[0] != D:
            raise ValueError("segmented_weight_grad: row_shift shape mismatch")
    if into is not None:
        gW, gb = into
        _req(gW, "gW", torch.float32, 2)
        if tuple(gW.shape) != (M, D) or (want_bias and (gb is None or tuple(gb.shape) != (M,))):
            raise ValueError("segmented_weight_grad: `into` does not match (M, D) / (M,)")
        gb = _req(gb, "gbias", torch.float32, 1) if want_bias else None
    else:
        gW = torch.empty(M, D, dtype=torch.float32, device=rows.device)
        gb = torch.empty(M, dtype=torch.float32, device=rows.device) if want_bias else None
    dc = None
    if want_dcoeff:
        dc = torch.empty(N, k, dtype=torch.float32, device=rows.device) if dcoeff_out is None else dcoeff_out
        _req(dc, "dcoeff", torch.float32, 2)
        if tuple(dc.shape) != (N, k):
            raise ValueError("segmented_weight_grad: dcoeff_out shape mismatch")
    lib = _lib.load()
    ws = _workspace(rows.device, lib.sphip_weight_grad_workspace_bytes(N, k, D), "wgrad")
    _ev = timer.start("sphip_segmented_weight_grad", rows)
    try:
        check(lib.sphip_segmented_weight_grad_acc(seg.perm.data_ptr(), seg.seg_begin.data_ptr(), coeff.data_ptr(),
                                                  _dt(coeff), rows.data_ptr(), _dt(rows), _ptr(W) if want_dcoeff else None,
                                                  gW.data_ptr(), _ptr(gb), _ptr(dc), _ptr(row_shift), N, k, M, D,
                                                  ws.data_ptr(), ws.numel(), int(into is not None), _stream(rows)),
              "sphip_segmented_weight_grad_acc")
    finally:
        timer.stop(_ev, rows)
    return gW, gb, dc


# a block's N x 128 B slices of the rows have to fit an XCD's 4 MiB L2 next to the streams that pass through
TOKEN_BLOCK_LIMIT = 24576
TOKEN_BLOCK = 16384


def token_blocks(N: int) -> list[tuple[int, int]]:
    """[(lo, hi)]: N tokens cut into nearly equal blocks of at most TOKEN_BLOCK (multiples of 64 but the last)."""
    nb = -(-N // TOKEN_BLOCK)
    step = -(-(-(-N // nb)) // 64) * 64
    return [(lo, min(lo + step, N)) for lo in range(0, N, step)]


def weight_grad_in_token_blocks(indices: Tensor, M: int, coeff: Tensor, rows: Tensor, W: Tensor | None = None,
                                want_bias: bool = False, want_dcoeff: bool = False, row_shift: Tensor | None = None,
                                into: tuple | None = None):
    """segmented_weight_grad over the whole batch, (gW, gbias, dcoeff), for batches whose row slices do not fit the
    L2-sliced kernels at once (N > TOKEN_BLOCK_LIMIT: the reference's default step is 32 x 2048 = 65536 tokens):
    blocks of <= TOKEN_BLOCK tokens, each with its own segments, the later ones adding their sums to the first one's
    (block after block: a fixed order). Where the shape takes a kernel form that cannot add, the whole batch goes
    through one call as before. into = (gW, gbias or None): the first block adds to these as well."""
    N = indices.shape[0]
    blocks = token_blocks(N)
    # asked for every block before the first one adds anything (blocks differ in size by a few tokens)
    if not all(weight_grad_accumulates(hi - lo, coeff.shape[1], M, rows.shape[1], rows.dtype, want_dcoeff)
               for lo, hi in (blocks if into is not None else blocks[1:])):
        return segmented_weight_grad(cached_segments(indices, M), coeff, rows, W, want_bias=want_bias,
                                     want_dcoeff=want_dcoeff, row_shift=row_shift, into=into)
    # decoder and encoder backward share the index tensor: the blocks' segments are sorted once per step, under
    # the same rules as cached_segments (made in backward, dropped by the next forward that sees the address)
    stamp = (indices._version, tuple(indices.shape), M, tuple(blocks))
    hit = _block_segment_cache.get(indices.data_ptr())
    if hit is None or hit[0] != stamp:
        if len(_block_segment_cache) > 4:
            _block_segment_cache.clear()
        hit = (stamp, [build_latent_segments(indices[lo:hi], M) for lo, hi in blocks])
        _block_segment_cache[indices.data_ptr()] = hit
    gW = gb = None
    dc = torch.empty(N, coeff.shape[1], dtype=torch.float32, device=rows.device) if want_dcoeff else None
    for b, (lo, hi) in enumerate(blocks):
        gW, gb, _ = segmented_weight_grad(hit[1][b], coeff[lo:hi], rows[lo:hi], W, want_bias=want_bias,
                                          want_dcoeff=want_dcoeff, row_shift=row_shift,
                                          dcoeff_out=None if dc is None else dc[lo:hi],
                                          into=into if b == 0 else (gW, gb))
    return gW, gb, dc


# Gradient accumulation over micro-steps (trainer.py:444-447: loss / acc_steps, backward, one optimizer step per
# grad_acc_steps): autograd adds a freshly written (M, D) gradient to `.grad` in a pass of its own, 6 GiB per matrix
# and micro-step at the headline widths. For parameters registered here (the Trainer does, on one rank) the backward of
# a later micro-step adds its sums to `.grad` inside the weight-gradient kernel instead (sphip_segmented_weight_grad_acc:
# the same fp32 addition, so the same bits) and returns no gradient for them; tensor hooks on such a parameter do not
# see these micro-steps' contributions.
_grad_sinks: dict[int, object] = {}
GRAD_SINK_HITS = 0


def register_grad_sinks(params) -> None:
    import weakref
    for p in params:
        _grad_sinks[p.data_ptr()] = weakref.ref(p)


def clear_grad_sinks() -> None:
    _grad_sinks.clear()


def grad_sink(param: Tensor) -> Tensor | None:
    """param.grad if `param` is registered and already holds a dense fp32 gradient to add to."""
    ref = _grad_sinks.get(param.data_ptr())
    p = ref() if ref is not None else None
    if p is None or p is not param:
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or g.shape != p.shape or not g.is_contiguous() or g.device != p.device:
        return None
    return g


def weight_grad_maybe_into(sink: Tensor | None, indices: Tensor, M: int, coeff: Tensor, rows: Tensor,
                           W: Tensor | None = None, want_bias: bool = False, want_dcoeff: bool = False,
                           row_shift: Tensor | None = None):
    """(gW or None, gbias, dcoeff, added): the weight gradient of a backward pass, in token blocks where the batch
    needs them; with a sink (grad_sink) the sums are added to it in the kernel and gW is None (added = True) unless
    the shape takes a kernel form that cannot add (weight_grad_accumulates) -- then this is the ordinary call."""
    global GRAD_SINK_HITS
    N, k = coeff.shape
    blocked = N > TOKEN_BLOCK_LIMIT
    if sink is not None and all(weight_grad_accumulates(hi - lo, k, M, rows.shape[1], rows.dtype, want_dcoeff)
                                for lo, hi in (token_blocks(N) if blocked else [(0, N)])):
        # the bias sums are added to zeros: the same bits as written directly (the accumulate flag covers both outputs)
        gb = torch.zeros(M, dtype=torch.float32, device=rows.device) if want_bias else None
        if blocked:
            _, _, dc = weight_grad_in_token_blocks(indices, M, coeff, rows, W, want_bias=want_bias,
                                                   want_dcoeff=want_dcoeff, row_shift=row_shift, into=(sink, gb))
        else:
            _, _, dc = segmented_weight_grad(cached_segments(indices, M), coeff, rows, W, want_bias=want_bias,
                                             want_dcoeff=want_dcoeff, row_shift=row_shift, into=(sink, gb))
        GRAD_SINK_HITS += 1
        return None, gb, dc, True
    if blocked:
        return (*weight_grad_in_token_blocks(indices, M, coeff, rows, W, want_bias=want_bias, want_dcoeff=want_dcoeff,
                                             row_shift=row_shift), False)
    return (*segmented_weight_grad(cached_segments(indices, M), coeff, rows, W, want_bias=want_bias,
                                   want_dcoeff=want_dcoeff, row_shift=row_shift), False)


def weight_grad_accumulates(N: int, k: int, M: int, D: int, rows_dtype, want_dcoeff: bool) -> bool:
    """Whether segmented_weight_grad(..., into=...) is available for a call of this shape."""
    code = F32 if rows_dtype == torch.float32 else BF16 if rows_dtype == torch.bfloat16 else -1
    return bool(_lib.load().sphip_weight_grad_accumulates(N, k, M, D, code, int(want_dcoeff)))


def segment_coeff_sums(seg: LatentSegments, coeff: Tensor) -> Tensor:
    """(M,) sums of each latent's coefficients: the encoder bias gradient (fused_encoder.py:80-84)."""
    _req(coeff, "coeff", ndim=2)
    if tuple(coeff.shape) != (seg.N, seg.k):
        raise ValueError("segment_coeff_sums: shape mismatch")
    out = torch.empty(seg.M, dtype=torch.float32, device=coeff.device)
    _ev = timer.start("sphip_segment_coeff_sums", coeff)
    check(_lib.load().sphip_segment_coeff_sums(seg.perm.data_ptr(), seg.seg_begin.data_ptr(), coeff.data_ptr(), _dt(coeff),
                                              seg.M, out.data_ptr(), _stream(coeff)), "sphip_segment_coeff_sums")
    timer.stop(_ev, coeff)
    return out


def segmented_weight_grad_signsgd_(seg: LatentSegments, coeff: Tensor, rows: Tensor, param: Tensor, lr: float,
                                   bf16_shadow: Tensor | None = None, row_shift: Tensor | None = None,
                                   schedule_free: tuple | None = None, coeff_sums: Tensor | None = None):
    """param -= lr * sign(weight gradient) without materialising the gradient (fused_encoder.py:67-77
    + sign_sgd.py:23); the same arithmetic as segmented_weight_grad followed by signsgd_step_.
    schedule_free = (z, beta, c): the schedule-free step of (param = y, z) instead
    (schedulefree_signsgd_step_). coeff_sums (segment_coeff_sums(seg, coeff)): also returns the
    (ceil(M / 32), D) partial sums of coeff_sums^T param_old (their column sum is vec_mat(coeff_sums,
    param_old)), read off the parameter pieces the update touches anyway; returns (param, partials)."""
    _req(coeff, "coeff", ndim=2)
    _req(rows, "rows", ndim=2)
    _req(param, "param", torch.float32, 2)
    N, k, M = seg.N, seg.k, seg.M
    D = rows.shape[1]
    if tuple(coeff.shape) != (N, k) or rows.shape[0] != N or tuple(param.shape) != (M, D):
        raise ValueError("segmented_weight_grad_signsgd: shape mismatch")
    if row_shift is not None:
        _req(row_shift, "row_shift", torch.float32, 1)
        if row_shift.shape[0] != D:
            raise ValueError("segmented_weight_grad_signsgd: row_shift shape mismatch")
    if bf16_shadow is not None:
        _req(bf16_shadow, "bf16_shadow", torch.bfloat16, 2)
        if bf16_shadow.shape != param.shape:
            raise ValueError("segmented_weight_grad_signsgd: shadow shape mismatch")
    partials = None
    if coeff_sums is not None:
        _req(coeff_sums, "coeff_sums", torch.float32, 1)
        if coeff_sums.shape[0] != M:
            raise ValueError("segmented_weight_grad_signsgd: coeff_sums shape mismatch")
        partials = torch.empty(-(-M // 32), D, dtype=torch.float32, device=param.device)
    lib = _lib.load()
    ws = _workspace(rows.device, lib.sphip_weight_grad_workspace_bytes(N, k, D), "wgrad")
    _ev = timer.start("sphip_segmented_weight_grad_signsgd", rows)
    if schedule_free is None:
        check(lib.sphip_segmented_weight_grad_signsgd(seg.perm.data_ptr(), seg.seg_begin.data_ptr(), coeff.data_ptr(),
                                                      _dt(coeff), rows.data_ptr(), _dt(rows), _ptr(row_shift), N, k, M, D,
                                                      param.data_ptr(), float(lr), _ptr(bf16_shadow), _ptr(coeff_sums),
                                                      _ptr(partials), ws.data_ptr(),
                                                      ws.numel(), _stream(rows)), "sphip_segmented_weight_grad_signsgd")
    else:
        z, beta, c = schedule_free
        _req(z, "z", torch.float32, 2)
        if z.shape != param.shape:
            raise ValueError("segmented_weight_grad_signsgd: z shape mismatch")
        check(lib.sphip_segmented_weight_grad_schedulefree(
            seg.perm.data_ptr(), seg.seg_begin.data_ptr(), coeff.data_ptr(), _dt(coeff), rows.data_ptr(), _dt(rows),
            _ptr(row_shift), N, k, M, D, param.data_ptr(), z.data_ptr(), float(lr), float(beta), float(c),
            _ptr(bf16_shadow), _ptr(coeff_sums), _ptr(partials), ws.data_ptr(), ws.numel(), _stream(rows)),
            "sphip_segmented_weight_grad_schedulefree")
    timer.stop(_ev, rows)
    return param if coeff_sums is None else (param, partials)


# ------------------------------------------------------------------ loss pieces
def _red_ws(t: Tensor, N: int, D: int) -> Tensor:
    return _workspace(t.device, _lib.load().sphip_reduce_workspace_bytes(N, D), "reduce")


def total_variance(y: Tensor) -> Tensor:
    _req(y, "y", ndim=2)
    N, D = y.shape
    tv = torch.empty(1, dtype=torch.float32, device=y.device)
    ws = _red_ws(y, N, D)
    _ev = timer.start("sphip_total_variance", y)
    check(_lib.load().sphip_total_variance(y.data_ptr(), _dt(y), N, D, tv.data_ptr(), ws.data_ptr(), ws.numel(),
                                          _stream(y)), "sphip_total_variance")
    timer.stop(_ev, y)
    return tv


def residual_sumsq(a: Tensor, b: Tensor, want_residual: bool = True):
    """e = a - b (fp32), sq = sum(e^2)."""
    _req(a, "a", ndim=2)
    _req(b, "b", torch.float32, 2)
    if a.shape != b.shape:
        raise ValueError("residual_sumsq: shape mismatch")
    N, D = a.shape
    e = torch.empty(N, D, dtype=torch.float32, device=a.device) if want_residual else None
    sq = torch.empty(1, dtype=torch.float32, device=a.device)
    ws = _red_ws(a, N, D)
    _ev = timer.start("sphip_residual_sumsq", a)
    check(_lib.load().sphip_residual_sumsq(a.data_ptr(), _dt(a), b.data_ptr(), _ptr(e), sq.data_ptr(), N, D,
                                          ws.data_ptr(), ws.numel(), _stream(a)), "sphip_residual_sumsq")
    timer.stop(_ev, a)
    return e, sq


def scale_by_ratio(src: Tensor, alpha: float, num: Tensor | None, den: Tensor | None, out: Tensor | None = None):
    _req(src, "in", torch.float32)
    if out is None:
        out = torch.empty_like(src)
    _ev = timer.start("sphip_scale_by_ratio", src)
    check(_lib.load().sphip_scale_by_ratio(src.data_ptr(), out.data_ptr(), src.numel(), float(alpha), _ptr(num),
                                          _ptr(den), _stream(src)), "sphip_scale_by_ratio")
    timer.stop(_ev, src)
    return out


def colsum(a: Tensor, scale: float = 1.0) -> Tensor:
    _req(a, "a", ndim=2)
    N, D = a.shape
    out = torch.empty(D, dtype=torch.float32, device=a.device)
    ws = _red_ws(a, N, D)
    _ev = timer.start("sphip_colsum", a)
    check(_lib.load().sphip_colsum(a.data_ptr(), _dt(a), N, D, float(scale), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                  _stream(a)), "sphip_colsum")
    timer.stop(_ev, a)
    return out


def vec_mat(v: Tensor, W: Tensor, scale: float = 1.0) -> Tensor:
    _req(v, "v", torch.float32, 1)
    _req(W, "W", torch.float32, 2)
    M, D = W.shape
    out = torch.empty(D, dtype=torch.float32, device=W.device)
    ws = _red_ws(W, M, D)
    _ev = timer.start("sphip_vec_mat", W)
    check(_lib.load().sphip_vec_mat(v.data_ptr(), W.data_ptr(), M, D, float(scale), out.data_ptr(), ws.data_ptr(),
                                   ws.numel(), _stream(W)), "sphip_vec_mat")
    timer.stop(_ev, W)
    return out


# ------------------------------------------------------------------ dense updates
def unit_norm_rows_(W: Tensor) -> Tensor:
    _req(W, "W_dec", torch.float32, 2)
    _ev = timer.start("sphip_unit_norm_rows", W)
    check(_lib.load().sphip_unit_norm_rows(W.data_ptr(), W.shape[0], W.shape[1], _stream(W)), "sphip_unit_norm_rows")
    timer.stop(_ev, W)
    return W


def remove_parallel_grad_(G: Tensor, W: Tensor) -> Tensor:
    _req(G, "W_dec.grad", torch.float32, 2)
    _req(W, "W_dec", torch.float32, 2)
    check(_lib.load().sphip_remove_parallel_grad(G.data_ptr(), W.data_ptr(), W.shape[0], W.shape[1], _stream(W)),
          "sphip_remove_parallel_grad")
    return G


def signsgd_step_(p: Tensor, g: Tensor, lr: float, bf16_shadow: Tensor | None = None) -> Tensor:
    _req(p, "param", torch.float32)
    _req(g, "grad", torch.float32)
    if g.shape != p.shape:
        raise ValueError("signsgd: grad shape mismatch")
    if bf16_shadow is not None:
        _req(bf16_shadow, "bf16_shadow", torch.bfloat16)
    _ev = timer.start("sphip_signsgd_step", p)
    check(_lib.load().sphip_signsgd_step(p.data_ptr(), g.data_ptr(), p.numel(), float(lr), _ptr(bf16_shadow),
                                        _stream(p)), "sphip_signsgd_step")
    timer.stop(_ev, p)
    return p


def signsgd_step_rows_(W: Tensor, G: Tensor, lr: float, slice_major: Tensor | None = None) -> Tensor:
    """signsgd_step_ for an (M, D) matrix, rewriting its slice-major copy (the buffer the L2-resident decode
    gathers from) in the same pass: a decoder matrix that is neither projected nor normalised (transcoders)."""
    _req(W, "W", torch.float32, 2)
    _req(G, "grad", torch.float32, 2)
    if G.shape != W.shape:
        raise ValueError("signsgd_step_rows: grad shape mismatch")
    if slice_major is not None:
        _req(slice_major, "slice_major", torch.float32)
        if slice_major.numel() != W.numel() or W.shape[1] % 32:
            raise ValueError("signsgd_step_rows: slice-major buffer does not match W")
    _ev = timer.start("sphip_signsgd_step_rows", W)
    check(_lib.load().sphip_signsgd_step_rows(W.data_ptr(), G.data_ptr(), W.shape[0], W.shape[1], float(lr),
                                             _ptr(slice_major), _stream(W)), "sphip_signsgd_step_rows")
    timer.stop(_ev, W)
    return W


def adam_step_(p: Tensor, g: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, lr: float, beta1: float, beta2: float,
               eps: float, step: int, bf16_shadow: Tensor | None = None) -> Tensor:
    """torch.optim.Adam's update of one parameter (no weight decay, no amsgrad), one pass; step counts from 1."""
    for t, name in ((p, "param"), (g, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _req(t, name, torch.float32)
        if t.shape != p.shape:
            raise ValueError(f"adam_step: {name} shape mismatch")
    if bf16_shadow is not None:
        _req(bf16_shadow, "bf16_shadow", torch.bfloat16)
    _ev = timer.start("sphip_adam_step", p)
    check(_lib.load().sphip_adam_step(p.data_ptr(), g.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), p.numel(),
                                     float(lr), float(beta1), float(beta2), float(eps), int(step), _ptr(bf16_shadow),
                                     _stream(p)), "sphip_adam_step")
    timer.stop(_ev, p)
    return p


def schedulefree_signsgd_step_(p: Tensor, z: Tensor, g: Tensor, lr: float, beta: float, c: float,
                                bf16_shadow: Tensor | None = None) -> Tensor:
    """One pass of ScheduleFreeWrapper(SignSGD).step() over (p = y, z, g); see sparsify_hip.h."""
    _req(p, "param", torch.float32)
    _req(z, "z", torch.float32)
    _req(g, "grad", torch.float32)
    if z.shape != p.shape or g.shape != p.shape:
        raise ValueError("schedulefree_signsgd_step: shape mismatch")
    if bf16_shadow is not None:
        _req(bf16_shadow, "bf16_shadow", torch.bfloat16)
    check(_lib.load().sphip_schedulefree_signsgd_step(p.data_ptr(), z.data_ptr(), g.data_ptr(), p.numel(), float(lr),
                                                     float(beta), float(c), _ptr(bf16_shadow), _stream(p)),
          "sphip_schedulefree_signsgd_step")
    return p


def decoder_schedulefree_signsgd_step_(W: Tensor, Z: Tensor, G: Tensor, lr: float, beta: float, c: float,
                                        renorm: bool = False, slice_major: Tensor | None = None) -> Tensor:
    """The same for decoder rows: projected gradient, step, optional re-normalisation of y.
    slice_major: the [D/32][M][32] buffer to rewrite with the finished rows in the same pass."""
    _req(W, "W_dec", torch.float32, 2)
    _req(Z, "z", torch.float32, 2)
    _req(G, "W_dec.grad", torch.float32, 2)
    if Z.shape != W.shape or G.shape != W.shape:
        raise ValueError("decoder_schedulefree_signsgd_step: shape mismatch")
    if slice_major is not None:
        _req(slice_major, "slice_major", torch.float32)
        if slice_major.numel() != W.numel() or W.shape[1] % 32:
            raise ValueError("decoder_schedulefree_signsgd_step: slice_major must hold M * D floats, D % 32 == 0")
    _ev = timer.start("sphip_decoder_schedulefree_signsgd_step", W)
    check(_lib.load().sphip_decoder_schedulefree_signsgd_step(W.data_ptr(), Z.data_ptr(), G.data_ptr(), W.shape[0],
                                                             W.shape[1], float(lr), float(beta), float(c), int(renorm),
                                                             _ptr(slice_major), _stream(W)),
          "sphip_decoder_schedulefree_signsgd_step")
    timer.stop(_ev, W)
    return W


def schedulefree_signsgd_step_rows_(W: Tensor, Z: Tensor, G: Tensor, lr: float, beta: float, c: float,
                                    slice_major: Tensor | None = None) -> Tensor:
    """schedulefree_signsgd_step_ for an (M, D) matrix (no projection, no normalisation: a transcoder's decoder)
    that rewrites its slice-major copy in the same pass. D % 4 == 0, D <= 8192."""
    _req(W, "W", torch.float32, 2)
    _req(Z, "z", torch.float32, 2)
    _req(G, "grad", torch.float32, 2)
    if Z.shape != W.shape or G.shape != W.shape:
        raise ValueError("schedulefree_signsgd_step_rows: shape mismatch")
    if slice_major is not None:
        _req(slice_major, "slice_major", torch.float32)
        if slice_major.numel() != W.numel() or W.shape[1] % 32:
            raise ValueError("schedulefree_signsgd_step_rows: slice_major must hold M * D floats, D % 32 == 0")
    _ev = timer.start("sphip_schedulefree_signsgd_step_rows", W)
    check(_lib.load().sphip_schedulefree_signsgd_step_rows(W.data_ptr(), Z.data_ptr(), G.data_ptr(), W.shape[0], W.shape[1],
                                                          float(lr), float(beta), float(c), _ptr(slice_major), _stream(W)),
          "sphip_schedulefree_signsgd_step_rows")
    timer.stop(_ev, W)
    return W


# ---- sign exchange (data-parallel SignSGD, parallel.SignExchange)
def sign_i8(g: Tensor, out: Tensor | None = None) -> Tensor:
    """int8 sign(g) (-1 / 0 / +1)."""
    _req(g, "grad", torch.float32)
    if out is None:
        out = torch.empty(g.shape, dtype=torch.int8, device=g.device)
    _req(out, "signs", torch.int8)
    check(_lib.load().sphip_sign_i8(g.data_ptr(), g.numel(), out.data_ptr(), _stream(g)), "sphip_sign_i8")
    return out


def decoder_project_sign_i8(W_rows: Tensor, G_rows: Tensor, out: Tensor | None = None) -> Tensor:
    """int8 sign(G - <G,W> W) per row, for a block of decoder rows and its gradient."""
    _req(W_rows, "W_dec rows", torch.float32, 2)
    _req(G_rows, "W_dec.grad rows", torch.float32, 2)
    if W_rows.shape != G_rows.shape:
        raise ValueError("decoder_project_sign_i8: shape mismatch")
    if out is None:
        out = torch.empty(G_rows.shape, dtype=torch.int8, device=G_rows.device)
    _req(out, "signs", torch.int8)
    check(_lib.load().sphip_decoder_project_sign_i8(W_rows.data_ptr(), G_rows.data_ptr(), W_rows.shape[0], W_rows.shape[1],
                                                   out.data_ptr(), _stream(W_rows)), "sphip_decoder_project_sign_i8")
    return out


def apply_signs_(p: Tensor, signs: Tensor, lr: float, bf16_shadow: Tensor | None = None) -> Tensor:
    """p -= lr * signs (int8), optionally refreshing the bf16 shadow of p."""
    _req(p, "param", torch.float32)
    _req(signs, "signs", torch.int8)
    if signs.numel() != p.numel():
        raise ValueError("apply_signs: size mismatch")
    if bf16_shadow is not None:
        _req(bf16_shadow, "bf16_shadow", torch.bfloat16)
    check(_lib.load().sphip_apply_signs(p.data_ptr(), signs.data_ptr(), p.numel(), float(lr), _ptr(bf16_shadow),
                                       _stream(p)), "sphip_apply_signs")
    return p


def _sm_args(slice_major: Tensor | None, W: Tensor, row0: int, name: str):
    """(pointer, rows, first row) of the slice-major copy a row block W belongs to."""
    if slice_major is None:
        return None, 0, 0
    _req(slice_major, "slice_major", torch.float32)
    D = W.shape[1]
    if D % 32 or slice_major.numel() % D or row0 < 0 or row0 + W.shape[0] > slice_major.numel() // D:
        raise ValueError(f"{name}: slice_major must hold whole rows of D = {D} floats (D % 32 == 0) covering "
                         f"rows [{row0}, {row0 + W.shape[0]})")
    return slice_major.data_ptr(), slice_major.numel() // D, row0


def decoder_apply_signs_(W: Tensor, signs: Tensor, lr: float, renorm: bool = False,
                         slice_major: Tensor | None = None, row0: int = 0) -> Tensor:
    """W -= lr * signs per decoder row; renorm=True also re-normalises the updated rows.
    slice_major / row0: the [D/32][rows][32] copy of the whole matrix and W's first row in it; the
    finished rows are written there in the same pass (renorm only)."""
    _req(W, "W_dec", torch.float32, 2)
    _req(signs, "signs", torch.int8)
    if signs.numel() != W.numel():
        raise ValueError("decoder_apply_signs: size mismatch")
    sm, rows, r0 = _sm_args(slice_major, W, row0, "decoder_apply_signs")
    check(_lib.load().sphip_decoder_apply_signs(W.data_ptr(), signs.data_ptr(), W.shape[0], W.shape[1], float(lr),
                                               int(renorm), sm, rows, r0, _stream(W)), "sphip_decoder_apply_signs")
    return W


def schedulefree_apply_signs_(p: Tensor, z: Tensor, signs: Tensor, lr: float, beta: float, c: float,
                              bf16_shadow: Tensor | None = None) -> Tensor:
    """schedulefree_signsgd_step_ with int8 signs in place of sign(g)."""
    _req(p, "param", torch.float32)
    _req(z, "z", torch.float32)
    _req(signs, "signs", torch.int8)
    if signs.numel() != p.numel() or z.numel() != p.numel():
        raise ValueError("schedulefree_apply_signs: size mismatch")
    if bf16_shadow is not None:
        _req(bf16_shadow, "bf16_shadow", torch.bfloat16)
    check(_lib.load().sphip_schedulefree_apply_signs(p.data_ptr(), z.data_ptr(), signs.data_ptr(), p.numel(), float(lr),
                                                    float(beta), float(c), _ptr(bf16_shadow), _stream(p)),
          "sphip_schedulefree_apply_signs")
    return p


def decoder_schedulefree_apply_signs_(W: Tensor, Z: Tensor, signs: Tensor, lr: float, beta: float, c: float,
                                      renorm: bool = False, slice_major: Tensor | None = None, row0: int = 0) -> Tensor:
    _req(W, "W_dec", torch.float32, 2)
    _req(Z, "z", torch.float32, 2)
    _req(signs, "signs", torch.int8)
    if signs.numel() != W.numel() or Z.shape != W.shape:
        raise ValueError("decoder_schedulefree_apply_signs: size mismatch")
    sm, rows, r0 = _sm_args(slice_major, W, row0, "decoder_schedulefree_apply_signs")
    check(_lib.load().sphip_decoder_schedulefree_apply_signs(W.data_ptr(), Z.data_ptr(), signs.data_ptr(), W.shape[0],
                                                            W.shape[1], float(lr), float(beta), float(c), int(renorm),
                                                            sm, rows, r0, _stream(W)), "sphip_decoder_schedulefree_apply_signs")
    return W


def decoder_project_signsgd_(W: Tensor, G: Tensor, lr: float, renorm: bool = False,
                             slice_major: Tensor | None = None) -> Tensor:
    """W -= lr * sign(G - <G,W> W) per row; renorm=True also re-normalises the updated rows;
    slice_major: the [D/32][M][32] buffer to rewrite with the finished rows in the same pass."""
    _req(W, "W_dec", torch.float32, 2)
    _req(G, "W_dec.grad", torch.float32, 2)
    if slice_major is not None:
        _req(slice_major, "slice_major", torch.float32)
        if slice_major.numel() != W.numel() or W.shape[1] % 32:
            raise ValueError("decoder_project_signsgd: slice-major buffer does not match W")
    _ev = timer.start("sphip_decoder_project_signsgd", W)
    check(_lib.load().sphip_decoder_project_signsgd(W.data_ptr(), G.data_ptr(), W.shape[0], W.shape[1], float(lr),
                                                   int(renorm), _ptr(slice_major), _stream(W)), "sphip_decoder_project_signsgd")
    timer.stop(_ev, W)
    return W


# ------------------------------------------------------------------ AuxK as dense products
def dead_compact(dead_mask: Tensor, num_dead: int):
    """dead_mask (M,) bool/uint8 -> slot_of (M,) int32 (slot or -1), dead_list (num_dead,) int64."""
    _req(dead_mask, "dead_mask", ndim=1)
    mask = dead_mask.view(torch.uint8) if dead_mask.dtype == torch.bool else _req(dead_mask, "dead_mask", torch.uint8)
    M = mask.shape[0]
    slot_of = torch.empty(M, dtype=torch.int32, device=mask.device)
    dead_list = torch.empty(num_dead, dtype=torch.int64, device=mask.device)
    check(_lib.load().sphip_dead_compact(mask.data_ptr(), M, slot_of.data_ptr(), dead_list.data_ptr(), num_dead,
                                        _stream(mask)), "sphip_dead_compact")
    return slot_of, dead_list


def gather_rows_bf16(src: Tensor, row_list: Tensor, R: int, rows_out: int) -> Tensor:
    """(rows_out, D) bf16: rows row_list[:R] of src, zero rows after (src may be 1-D: D = 1)."""
    _req(src, "src", torch.bfloat16)
    _req(row_list, "row_list", torch.int64, 1)
    if row_list.shape[0] < R or src.ndim not in (1, 2):
        raise ValueError("gather_rows_bf16: bad arguments")
    D = 1 if src.ndim == 1 else src.shape[1]
    dst = torch.empty((rows_out,) if src.ndim == 1 else (rows_out, D), dtype=torch.bfloat16, device=src.device)
    check(_lib.load().sphip_gather_rows_bf16(src.data_ptr(), row_list.data_ptr(), R, rows_out, D, dst.data_ptr(),
                                            _stream(src)), "sphip_gather_rows_bf16")
    return dst


def dead_scatter_acts(idx: Tensor, acts: Tensor, slot_of: Tensor | None, Kd: int, num_valid: int | None = None):
    """Selected activations as dense bf16 matrices A (N, Kd) and A^T (Kd, ceil64(N)).
    slot_of None: the indices are slots of the compact dead-latent numbering already; slots
    >= num_valid (the padding columns of the compact selection) are dropped."""
    _req(idx, "indices", torch.int64, 2)
    _req(acts, "acts", torch.bfloat16, 2)
    if slot_of is not None:
        _req(slot_of, "slot_of", torch.int32, 1)
    if acts.shape != idx.shape:
        raise ValueError("dead_scatter_acts: shape mismatch")
    N, k = idx.shape
    ldT = -(-N // 64) * 64
    A = torch.empty(N, Kd, dtype=torch.bfloat16, device=idx.device)
    At = torch.empty(Kd, ldT, dtype=torch.bfloat16, device=idx.device)
    check(_lib.load().sphip_dead_scatter_acts(idx.data_ptr(), acts.data_ptr(), _ptr(slot_of), N, k,
                                             min(Kd, Kd if num_valid is None else num_valid) if slot_of is None else slot_of.shape[0],
                                             Kd, Kd, ldT, A.data_ptr(), At.data_ptr(), _stream(idx)),
          "sphip_dead_scatter_acts")
    return A, At


def split_bf16_transposed(src: Tensor, row_list: Tensor | None, R: int, ldr: int):
    """Three bf16 (C, ldr) matrices whose sum is fp32 src[rows].T exactly (columns R.. are zero)."""
    _req(src, "src", torch.float32, 2)
    if row_list is not None:
        _req(row_list, "row_list", torch.int64, 1)
        if row_list.shape[0] < R:
            raise ValueError("split_bf16_transposed: row_list shorter than R")
    elif src.shape[0] < R:
        raise ValueError("split_bf16_transposed: src has fewer than R rows")
    C = src.shape[1]
    parts = torch.empty(3, C, ldr, dtype=torch.bfloat16, device=src.device)
    check(_lib.load().sphip_split_bf16_transposed(src.data_ptr(), _ptr(row_list), R, C, ldr, parts[0].data_ptr(),
                                                 parts[1].data_ptr(), parts[2].data_ptr(), _stream(src)),
          "sphip_split_bf16_transposed")
    return parts


def gemm_bf16_f32acc(x: Tensor, W: Tensor, out: Tensor, accumulate: bool) -> Tensor:
    """out (N, M) fp32 (+)= x (N, K) @ W (M, K).T with bf16 operands and fp32 accumulation."""
    _req(x, "x", torch.bfloat16, 2)
    _req(W, "W", torch.bfloat16, 2)
    _req(out, "out", torch.float32, 2)
    N, K = x.shape
    M = W.shape[0]
    if W.shape[1] != K or out.shape != (N, M):
        raise ValueError(f"gemm_bf16_f32acc: x {tuple(x.shape)}, W {tuple(W.shape)}, out {tuple(out.shape)}")
    _ev = timer.start("sphip_gemm_bf16_f32acc", x)
    check(_lib.load().sphip_gemm_bf16_f32acc(x.data_ptr(), W.data_ptr(), out.data_ptr(), N, K, M, M, int(accumulate),
                                            _stream(x)), "sphip_gemm_bf16_f32acc")
    timer.stop(_ev, x)
    return out


def scatter_rows(src: Tensor, row_list: Tensor, R: int, M: int) -> Tensor:
    """(M, D) zeros with row row_list[r] = src[r] for r < R."""
    _req(src, "src", torch.float32, 2)
    _req(row_list, "row_list", torch.int64, 1)
    if src.shape[0] < R or row_list.shape[0] < R:
        raise ValueError("scatter_rows: fewer than R rows")
    D = src.shape[1]
    dst = torch.empty(M, D, dtype=torch.float32, device=src.device)
    check(_lib.load().sphip_scatter_rows(src.data_ptr(), row_list.data_ptr(), R, D, M, dst.data_ptr(), _stream(src)),
          "sphip_scatter_rows")
    return dst


def scatter_add_rows_(dst: Tensor, src: Tensor, row_list: Tensor, R: int) -> Tensor:
    """dst[row_list[r]] += src[r] for r < R (distinct rows), in place."""
    _req(dst, "dst", torch.float32, 2)
    _req(src, "src", torch.float32, 2)
    _req(row_list, "row_list", torch.int64, 1)
    if src.shape[0] < R or row_list.shape[0] < R or src.shape[1] != dst.shape[1]:
        raise ValueError("scatter_add_rows: shape mismatch")
    check(_lib.load().sphip_scatter_add_rows(src.data_ptr(), row_list.data_ptr(), R, dst.shape[1], dst.shape[0],
                                            dst.data_ptr(), _stream(src)), "sphip_scatter_add_rows")
    return dst


def broadcast_rows(bias: Tensor, N: int) -> Tensor:
    _req(bias, "bias", torch.float32, 1)
    out = torch.empty(N, bias.shape[0], dtype=torch.float32, device=bias.device)
    check(_lib.load().sphip_broadcast_rows(bias.data_ptr(), N, bias.shape[0], out.data_ptr(), _stream(bias)),
          "sphip_broadcast_rows")
    return out


# ------------------------------------------------------------------ Muon (muon.hip + the GEMM axpy epilogue)
def gemm_bf16_axpy(x: Tensor, W: Tensor, C: Tensor | None = None, alpha: float = 1.0, beta: float = 0.0,
                   out: Tensor | None = None) -> Tensor:
    """bf16 out (N, M) = rnd(rnd(alpha * rnd(x @ W.T)) + rnd(beta * C)): torch's roundings for
    `beta * C + alpha * (x @ W.mT)` on bf16 tensors (muon.py:36-38)."""
    _req(x, "x", torch.bfloat16, 2)
    _req(W, "W", torch.bfloat16, 2)
    N, K = x.shape
    M = W.shape[0]
    if W.shape[1] != K:
        raise ValueError(f"gemm_bf16_axpy: x {tuple(x.shape)} vs W {tuple(W.shape)}")
    if C is not None:
        _req(C, "C", torch.bfloat16, 2)
        if tuple(C.shape) != (N, M):
            raise ValueError(f"gemm_bf16_axpy: C {tuple(C.shape)} is not ({N}, {M})")
    if out is None:
        out = torch.empty(N, M, dtype=torch.bfloat16, device=x.device)
    _req(out, "out", torch.bfloat16, 2)
    if tuple(out.shape) != (N, M) or out.data_ptr() in (x.data_ptr(), W.data_ptr()):
        raise ValueError("gemm_bf16_axpy: out must be a separate (N, M) tensor")
    _ev = timer.start("sphip_gemm_bf16_axpy", x)
    check(_lib.load().sphip_gemm_bf16_axpy(x.data_ptr(), W.data_ptr(), _ptr(C), out.data_ptr(), N, K, M, M, M, float(alpha),
                                          float(beta), _stream(x)), "sphip_gemm_bf16_axpy")
    timer.stop(_ev, x)
    return out


def muon_momentum(g: Tensor, buf: Tensor | None, beta: float, nesterov: bool):
    """(bf16 update u, norm = [sum(u^2), denominator]); buf is updated in place (muon.py:157-160)."""
    _req(g, "grad", torch.float32)
    if buf is not None:
        _req(buf, "momentum buffer", torch.float32)
        if buf.shape != g.shape:
            raise ValueError("muon_momentum: buffer shape mismatch")
    elif beta > 0:
        raise ValueError("muon_momentum: momentum needs a buffer")
    u = torch.empty(g.shape, dtype=torch.bfloat16, device=g.device)
    norm = torch.empty(2, dtype=torch.float32, device=g.device)
    lib = _lib.load()
    ws = _workspace(g.device, lib.sphip_muon_workspace_bytes(), "muon")
    check(lib.sphip_muon_momentum(g.data_ptr(), _ptr(buf), g.numel(), float(beta), int(nesterov), u.data_ptr(),
                                  norm.data_ptr(), ws.data_ptr(), ws.numel(), _stream(g)), "sphip_muon_momentum")
    return u, norm


def muon_denominator_(norm: Tensor) -> Tensor:
    _req(norm, "norm", torch.float32, 1)
    check(_lib.load().sphip_muon_denominator(norm.data_ptr(), _stream(norm)), "sphip_muon_denominator")
    return norm


def muon_normalise_(x: Tensor, norm: Tensor) -> Tensor:
    _req(x, "x", torch.bfloat16)
    _req(norm, "norm", torch.float32, 1)
    check(_lib.load().sphip_muon_normalise(x.data_ptr(), x.numel(), norm.data_ptr(), _stream(x)), "sphip_muon_normalise")
    return x


def transpose(src: Tensor, out: Tensor | None = None, pad_to: int = 1) -> Tensor:
    """(R, C) -> (C, ld) with ld = R rounded up to a multiple of pad_to, padding columns zero."""
    _req(src, "src", ndim=2)
    R, C = src.shape
    ld = -(-R // pad_to) * pad_to
    if out is None:
        out = torch.empty(C, ld, dtype=src.dtype, device=src.device)
    _req(out, "out", src.dtype, 2)
    if tuple(out.shape) != (C, ld):
        raise ValueError(f"transpose: out must be ({C}, {ld})")
    check(_lib.load().sphip_transpose(src.data_ptr(), out.data_ptr(), R, C, ld, _dt(src), _stream(src)), "sphip_transpose")
    return out


def transpose_bf16(src: Tensor, out: Tensor | None = None) -> Tensor:
    _req(src, "src", torch.bfloat16, 2)
    return transpose(src, out)


def add(a: Tensor, b: Tensor) -> Tensor:
    """fp32 a + b (b bf16 or fp32, same number of elements) as a new fp32 tensor of a's shape."""
    _req(a, "a", torch.float32)
    _req(b, "b")
    if b.numel() != a.numel():
        raise ValueError("add: size mismatch")
    out = torch.empty_like(a)
    check(_lib.load().sphip_add(a.data_ptr(), b.data_ptr(), _dt(b), out.data_ptr(), a.numel(), _stream(a)), "sphip_add")
    return out


def muon_apply_(p: Tensor, o: Tensor, decay: float, step: float, bf16_shadow: Tensor | None = None) -> Tensor:
    """p = p * decay - step * o (muon.py:121-122, 178-180), o bf16 in p's layout."""
    _req(p, "param", torch.float32)
    _req(o, "update", torch.bfloat16)
    if o.numel() != p.numel():
        raise ValueError("muon_apply: size mismatch")
    if bf16_shadow is not None:
        _req(bf16_shadow, "bf16_shadow", torch.bfloat16)
    check(_lib.load().sphip_muon_apply(p.data_ptr(), o.data_ptr(), p.numel(), float(decay), float(step), _ptr(bf16_shadow),
                                      _stream(p)), "sphip_muon_apply")
    return p


# ------------------------------------------------------------------ dead-latent bookkeeping
def mark_fired_(did_fire: Tensor, idx: Tensor) -> Tensor:
    _req(idx, "indices", torch.int64)
    _req(did_fire, "did_fire", ndim=1)
    buf = did_fire.view(torch.uint8) if did_fire.dtype == torch.bool else did_fire
    _ev = timer.start("sphip_mark_fired", idx)
    check(_lib.load().sphip_mark_fired(idx.data_ptr(), idx.numel(), buf.shape[0], buf.data_ptr(), _stream(idx)),
          "sphip_mark_fired")
    timer.stop(_ev, idx)
    return did_fire


def dead_counter_step_(counts: Tensor, did_fire: Tensor, n_tokens: int) -> None:
    _req(counts, "num_tokens_since_fired", torch.int64, 1)
    buf = did_fire.view(torch.uint8) if did_fire.dtype == torch.bool else did_fire
    _ev = timer.start("sphip_dead_counter_step", counts)
    check(_lib.load().sphip_dead_counter_step(counts.data_ptr(), buf.data_ptr(), counts.shape[0], int(n_tokens),
                                             _stream(counts)), "sphip_dead_counter_step")
    timer.stop(_ev, counts)


def dead_mask(counts: Tensor, threshold: int):
    """(mask bool (M,), n_dead int32 device scalar); trainer.py:415 + sparse_coder.py:231."""
    _req(counts, "num_tokens_since_fired", torch.int64, 1)
    M = counts.shape[0]
    mask = torch.empty(M, dtype=torch.bool, device=counts.device)
    n_dead = torch.empty(1, dtype=torch.int32, device=counts.device)
    check(_lib.load().sphip_dead_mask(counts.data_ptr(), M, int(threshold), mask.view(torch.uint8).data_ptr(),
                                     n_dead.data_ptr(), _stream(counts)), "sphip_dead_mask")
    return mask, n_dead


# ------------------------------------------------------------------ timing of the remaining entry points
def _timed(label: str, fn):
    """Wrap an op that has no timer calls of its own (auxiliary passes: AuxK preparation, the
    data-parallel sign exchange, Muon's element-wise steps) so that KernelTimer sees it."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if not timer.enabled:
            return fn(*args, **kwargs)
        t = next(a for a in args if isinstance(a, Tensor))
        ev = timer.start(label, t)
        try:
            return fn(*args, **kwargs)
        finally:
            timer.stop(ev, t)
    return wrapper


for _name in ("cast_bf16", "groupmax", "slice_major_copy", "remove_parallel_grad_", "schedulefree_signsgd_step_", "sign_i8",
              "decoder_project_sign_i8", "apply_signs_", "decoder_apply_signs_", "schedulefree_apply_signs_",
              "decoder_schedulefree_apply_signs_", "dead_compact", "gather_rows_bf16", "dead_scatter_acts",
              "split_bf16_transposed", "scatter_rows", "scatter_add_rows_", "broadcast_rows", "muon_momentum",
              "muon_denominator_", "muon_normalise_", "transpose", "add", "muon_apply_", "dead_mask"):
    globals()[_name] = _timed("sphip_" + _name.rstrip("_"), globals()[_name])
del _name
