"""Multi-GPU pieces of the train step, one process per GPU over torch.distributed
(backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).

Reference call sites: sparsify/trainer.py:457-468 (DDP gradient all-reduce), :614-628
(maybe_all_reduce), :630-646 (distribute_modules plan), :358-380 (activation all-gathers).
Latent-dim sharding (BASELINE.json configs[4]) does not exist in the reference; its
contract here is "identical to the unsharded SparseCoder" (SURVEY.md section 8e).

Everything in this file is device-agnostic collective plumbing on tensors that already
exist; the compute stays in the HIP kernels."""
from __future__ import annotations

import torch
import torch.distributed as dist
from torch import Tensor


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_initialized() else 0


def _host_staged(x: Tensor) -> bool:
    """gloo moves device tensors through its own pinned staging buffers and streams; that path
    stalls intermittently on ROCm when ranks share one GPU (single-GPU rehearsals of the
    multi-rank code). Device tensors under gloo are therefore staged through host memory here;
    under RCCL ("nccl") they never leave the device."""
    return x.is_cuda and dist.get_backend() != "nccl"


def _all_reduce_raw(x: Tensor, op) -> None:
    if _host_staged(x):
        h = x.cpu()
        dist.all_reduce(h, op=op)
        x.copy_(h)
    else:
        dist.all_reduce(x, op=op)


def all_reduce_(x: Tensor, op: str = "mean") -> Tensor:
    """In-place all-reduce with the reference's op names (trainer.py:614-628)."""
    if world_size() == 1:
        return x
    if op == "sum":
        _all_reduce_raw(x, dist.ReduceOp.SUM)
    elif op == "mean":
        _all_reduce_raw(x, dist.ReduceOp.SUM)
        x /= world_size()
    elif op == "max":
        buf = x.view(torch.uint8) if x.dtype == torch.bool else x      # max is boolean "any"
        _all_reduce_raw(buf, dist.ReduceOp.MAX)
    else:
        raise ValueError(f"Unknown reduction op '{op}'")
    return x


def _all_gather_raw(out: Tensor, src: Tensor) -> None:
    if _host_staged(src):
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h, src.cpu())
        out.copy_(h)
    else:
        dist.all_gather_into_tensor(out, src)


def module_plan(hookpoints: list[str], world: int) -> list[list[str]]:
    """Contiguous blocks of hookpoints per rank (trainer.py:637-644)."""
    per_rank, rem = divmod(len(hookpoints), world)
    assert rem == 0, "Number of modules must be divisible by world size"
    return [hookpoints[s: s + per_rank] for s in range(0, len(hookpoints), per_rank)]


def all_gather_rows(x: Tensor) -> Tensor:
    """(B, ...) on every rank -> (W*B, ...) (trainer.py:358-377)."""
    if world_size() == 1:
        return x
    x = x.contiguous()
    as_bool = x.dtype == torch.bool
    src = x.view(torch.uint8) if as_bool else x
    out = src.new_empty(src.shape[0] * world_size(), *src.shape[1:])
    _all_gather_raw(out, src)
    return out.view(torch.bool) if as_bool else out


def gather_rows(x: Tensor, dst: int) -> Tensor | None:
    """(B, ...) on every rank -> (W*B, ...) in rank order on rank `dst` only, None elsewhere.

    The `distribute_modules` hook (trainer.py:358-380) all-gathers every hookpoint's activations
    to every rank and all but the owner throw them away; only the owner needs them ("scattered by
    layer", BASELINE.json north_star; the reference's own comment at trainer.py:70 points at an
    all-to-all). A gather to the owner moves 1/W of the all-gather's bytes over xGMI: per forward a
    rank receives W x its own hookpoints' activations instead of W x all hookpoints'."""
    w = world_size()
    if w == 1:
        return x
    x = x.contiguous()
    as_bool = x.dtype == torch.bool
    src = x.view(torch.uint8) if as_bool else x
    me = rank()
    if _host_staged(src):
        h = src.cpu()
        parts = [torch.empty_like(h) for _ in range(w)] if me == dst else None
        dist.gather(h, parts, dst=dst)
        if me != dst:
            return None
        out = torch.cat(parts).to(src.device)
    else:
        out = src.new_empty(src.shape[0] * w, *src.shape[1:]) if me == dst else None
        parts = list(out.chunk(w)) if me == dst else None
        dist.gather(src, parts, dst=dst)
        if me != dst:
            return None
    return out.view(torch.bool) if as_bool else out


def module_owner(plan: list[list[str]], hookpoint: str) -> int:
    """Rank that trains `hookpoint` under a module_plan()."""
    for r, names in enumerate(plan):
        if hookpoint in names:
            return r
    raise KeyError(hookpoint)


class GradReducer:
    """Averages dense gradients across ranks once per optimizer step (the reference's DDP
    wrapper reduces on every micro-step backward, trainer.py:457-468, because it never uses
    no_sync()).

    Large gradients (>= `bucket_bytes`: the (M, D) matrices under Adam / Muon) are all-reduced one
    by one, asynchronously, from a post-accumulate-grad hook -- i.e. inside the backward, as soon
    as the gradient is complete, on the communication stream and under the rest of the backward --
    whenever the reducer is `armed` (the Trainer arms it for the last micro-step of an optimizer
    step). Everything else (biases, b_dec) is packed into one flat bucket and reduced by a single
    call in ``finish()``, which also waits for the asynchronous ones; it runs right before the
    optimizer step. Without arming (plain use, CPU tests) ``finish()`` does all of it itself.
    With RCCL every call is stream-ordered (no host sync)."""

    def __init__(self, params, bucket_bytes: int = 32 << 20):
        self.params = [p for p in params]
        self.bucket_bytes = bucket_bytes
        self.armed = False
        self._inflight: list = []           # (param, work) of asynchronous all-reduces
        self._done: set[int] = set()
        self._hooks = []
        if world_size() > 1:
            for p in self.params:
                if self._large(p) and hasattr(p, "register_post_accumulate_grad_hook"):
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _large(self, p) -> bool:
        return p.numel() * p.element_size() >= self.bucket_bytes

    def _on_grad(self, p) -> None:
        if not self.armed or p.grad is None or id(p) in self._done or _host_staged(p.grad):
            return
        g = p.grad if p.grad.is_contiguous() else None
        if g is None:
            return
        avg = dist.get_backend() == "nccl"
        work = dist.all_reduce(g, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, async_op=True)
        self._inflight.append((p, work, avg))
        self._done.add(id(p))

    def finish(self):
        w = world_size()
        if w == 1:
            return
        avg = dist.get_backend() == "nccl"
        for p, work, was_avg in self._inflight:
            work.wait()                              # stream-ordered under RCCL
            if not was_avg:
                p.grad /= w
        small = []
        for p in self.params:
            if p.grad is None or id(p) in self._done:
                continue
            if self._large(p):
                if avg:
                    dist.all_reduce(p.grad, op=dist.ReduceOp.AVG)
                else:
                    _all_reduce_raw(p.grad, dist.ReduceOp.SUM)
                    p.grad /= w
            else:
                small.append(p)
        if small:
            # one flat bucket for the small ones; a dtype of its own per group
            by_dtype: dict = {}
            for p in small:
                by_dtype.setdefault(p.grad.dtype, []).append(p)
            for group in by_dtype.values():
                flat = torch.cat([p.grad.reshape(-1) for p in group])
                if avg:
                    dist.all_reduce(flat, op=dist.ReduceOp.AVG)
                else:
                    _all_reduce_raw(flat, dist.ReduceOp.SUM)
                    flat /= w
                o = 0
                for p in group:
                    n = p.grad.numel()
                    p.grad.copy_(flat[o:o + n].view_as(p.grad))
                    o += n
        self._inflight, self._done, self.armed = [], set(), False


# ------------------------------------------------------------------ overlapped updates
# A parameter update that is still running on the exchange stream is recorded here as
# (row_lo, row_hi, event) per finished row block; every reader of the parameter in this package
# (FusedEncoder.forward, the decode op, the SparseCoder row operations, checkpointing) first makes
# its own stream wait for those events. Code outside the package that reads parameters directly
# right after Optimizer.step() must call wait_for_updates() first; the overlap is therefore
# opt-in (SignSGD(overlap=True), TrainConfig.overlap_exchange for the Trainer).
_side_streams: dict[int, "torch.cuda.Stream"] = {}
_pending: dict[int, list] = {}


def exchange_stream(device: torch.device) -> "torch.cuda.Stream":
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _side_streams.get(idx)
    if st is None:
        st = _side_streams[idx] = torch.cuda.Stream(device=device)
    return st


def set_pending(t: Tensor, blocks: list) -> None:
    _pending[t.data_ptr()] = (t.device, blocks)


def pending_blocks(t: Tensor) -> list | None:
    """[(row_lo, row_hi, event)] of an update of `t` that may still be running, else None.
    The caller takes over: it must wait for every event before reading those rows."""
    hit = _pending.pop(t.data_ptr(), None)
    return None if hit is None else hit[1]


def wait_for(t: Tensor) -> None:
    """Make the current stream wait until `t` holds its last update."""
    hit = _pending.pop(t.data_ptr(), None)
    if hit and hit[1]:
        cur = torch.cuda.current_stream(t.device)
        cur.wait_event(hit[1][-1][2])              # blocks finish in order on one stream


def wait_for_updates() -> None:
    """Make the current stream of every device that has an overlapped update in flight wait for
    it (an event belongs to the device of the parameter it was recorded for)."""
    for device, blocks in _pending.values():
        if blocks:
            torch.cuda.current_stream(device).wait_event(blocks[-1][2])
    _pending.clear()


class SignExchange:
    """Gradient exchange of data-parallel SignSGD for the large matrices: the update only needs
    sign(mean gradient), so instead of all-reducing 4 B per element both ways
    (reduce-scatter + all-gather inside the all-reduce), the fp32 gradients are
    reduce-scattered by row blocks, each rank takes the sign of its block (for the decoder:
    of the projected gradient, a row-local operation), and 1-byte signs are all-gathered:
    (4 + 1) / (4 + 4) of the bytes over xGMI, and the result is exactly the update of the
    averaged gradient (trainer.py:457-468 + sign_sgd.py:23).

    `sign_fn(grad_rows, row_lo, row_hi) -> int8 signs` and `apply_fn(signs, row_lo, row_hi)` are
    the per-parameter kernels (sign_sgd.SignSGD passes the HIP ones). With `blocks` > 1 the rows
    are exchanged and updated block by block (block b = rows [b R/blocks, (b+1) R/blocks), of
    which rank r reduces the r-th W-th), so that a reader can start on the first rows while the
    later ones are still on the wire; every row gets the same update either way."""

    MIN_NUMEL = 1 << 20
    FORCE = False          # tests / single-GPU rehearsals: exchange even with one rank

    @staticmethod
    def eligible(p: Tensor) -> bool:
        w = world_size()
        if w == 1 and not (SignExchange.FORCE and dist.is_initialized()):
            return False
        return p.ndim == 2 and p.shape[0] % w == 0 and p.numel() >= SignExchange.MIN_NUMEL

    @staticmethod
    def row_blocks(rows: int, want: int, multiple: int = 1) -> int:
        """Largest block count <= want such that every block is a whole number of
        `multiple`-row units per rank."""
        w = world_size()
        b = max(1, want)
        while b > 1 and (rows % b or (rows // b) % (w * multiple)):
            b -= 1
        return b

    @staticmethod
    def step(grad: Tensor, sign_fn, apply_fn, blocks: int = 1, on_block=None) -> None:
        w, r = world_size(), rank()
        total = grad.shape[0]
        if total % (blocks * w):
            raise ValueError(f"{total} rows do not split into {blocks} blocks of {w} equal shards")
        per_block = total // blocks
        rows = per_block // w
        grad = grad.contiguous()
        backend = dist.get_backend()
        for b in range(blocks):
            lo = b * per_block
            gb = grad[lo:lo + per_block]
            shard = torch.empty(rows, *grad.shape[1:], dtype=grad.dtype, device=grad.device)
            if backend == "nccl":
                dist.reduce_scatter_tensor(shard, gb, op=dist.ReduceOp.AVG)
            elif _host_staged(grad):
                h = torch.empty(shard.shape, dtype=shard.dtype)
                dist.reduce_scatter_tensor(h, gb.cpu(), op=dist.ReduceOp.SUM)
                shard.copy_(h)
                shard /= w
            else:
                dist.reduce_scatter_tensor(shard, gb, op=dist.ReduceOp.SUM)
                shard /= w
            signs = sign_fn(shard, lo + r * rows, lo + (r + 1) * rows)
            full = torch.empty(per_block, *grad.shape[1:], dtype=torch.int8, device=grad.device)
            _all_gather_raw(full, signs.contiguous())
            apply_fn(full, lo, lo + per_block)
            if on_block is not None:
                on_block(lo, lo + per_block)


def broadcast_parameters(params, src: int = 0):
    """What DDP does at wrap time (trainer.py:463): make replicas identical."""
    if world_size() > 1:
        for p in params:
            if _host_staged(p.data):
                h = p.data.cpu()
                dist.broadcast(h, src)
                p.data.copy_(h)
            else:
                dist.broadcast(p.data, src)


# ------------------------------------------------------------------ latent-dim sharding
def shard_range(num_latents: int, r: int | None = None, w: int | None = None) -> tuple[int, int]:
    """Rank r owns latents [lo, hi) of W_enc / b_enc / W_dec."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    assert num_latents % w == 0, "num_latents must be divisible by the number of shards"
    per = num_latents // w
    return r * per, (r + 1) * per


def pack_candidates(vals: Tensor, gidx: Tensor) -> Tensor:
    """(value, global index) pairs as one int64 record each: the value's bits (bf16 or fp32) in the
    upper half, the index (< 2^32) in the lower. 8 bytes per candidate on the wire, one message."""
    if vals.dtype == torch.bfloat16:
        bits = (vals.contiguous().view(torch.int16).to(torch.int64) & 0xFFFF) << 48
    elif vals.dtype == torch.float32:
        bits = (vals.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF) << 32
    else:
        raise TypeError(f"pack_candidates: bf16 or fp32 values, got {vals.dtype}")
    return bits | gidx.to(torch.int64)


def unpack_candidates(packed: Tensor, dtype: torch.dtype) -> tuple[Tensor, Tensor]:
    """Inverse of pack_candidates: (values of `dtype`, int64 indices), both shaped like `packed`."""
    if dtype == torch.bfloat16:
        vals = (packed >> 48).to(torch.int16).view(torch.bfloat16)
    else:
        vals = (packed >> 32).to(torch.int32).view(torch.float32)
    return vals, packed & 0xFFFFFFFF


def candidate_indices(cand_i: Tensor, pos: Tensor) -> Tensor:
    """Global indices of the merged selection: `pos` (N, k) are positions in a token's shard-major
    candidate list of W*kk entries; `cand_i` is (N, W*kk), or the (N, W, kk) view of the gathered
    rank-major records, which is indexed in place (no transposed copy of the indices)."""
    if cand_i.ndim == 2:
        return cand_i.gather(1, pos)
    kk = cand_i.shape[2]
    rows = torch.arange(cand_i.shape[0], device=cand_i.device)[:, None]
    return cand_i[rows, pos // kk, pos % kk]


def gather_topk_candidates(local_vals: Tensor, local_idx: Tensor, lo: int) -> tuple[Tensor, Tensor]:
    """All-gather every shard's k local candidates as packed (value, global index) records:
    (N, k) per rank -> values (N, W*k), shard-major per token, and indices as the (N, W, k) view of the
    gathered records (see candidate_indices). One collective, W*N*k*8 bytes on the wire (SURVEY.md 8e:
    6-8 bytes per candidate; it was two collectives and 10-12 bytes with int64 indices)."""
    gidx = local_idx + lo
    if world_size() == 1:
        return local_vals, gidx
    w = world_size()
    N, k = local_vals.shape
    mine = pack_candidates(local_vals, gidx)
    got = mine.new_empty(w * N, k)
    _all_gather_raw(got, mine)
    vals, idxs = unpack_candidates(got.view(w, N, k), local_vals.dtype)
    return vals.permute(1, 0, 2).reshape(N, w * k), idxs.permute(1, 0, 2)
