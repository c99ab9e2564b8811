"""Random shapes through the AuxK dense path (dead-mask compaction, A / A^T, exact three-term split,
bf16 products with fp32 accumulation, gradient rows scattered back) against the row-gather decode on the
same selection: reconstruction, W_dec.grad and b_dec.grad to fp32 summation-order accuracy (2e-5 of the
largest element). run(seconds, max_cases) is also what tests/test_hip_fuzz.py
calls for a bounded, seeded slice. Usage: python benchmarks/auxk_fuzz.py [seconds]"""
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import ops  # noqa: E402
from sparsify_amd.dead_decode import DeadLatentDecode  # noqa: E402
from sparsify_amd.utils import SparseDecode  # noqa: E402


def run(budget: float = 60.0, max_cases: int | None = None) -> int:
    """Random cases (seeded: the sequence is the same on every run) until `budget` seconds or `max_cases`
    have passed; raises AssertionError naming the first mismatching case; returns the number of cases."""
    rng = random.Random(5)
    dev = torch.device("cuda", 0)
    t0, n = time.time(), 0
    while time.time() - t0 < budget and (max_cases is None or n < max_cases):
        D = 8 * rng.randint(1, 64)
        M = rng.choice([rng.randint(8, 300), 64 * rng.randint(1, 64)])
        n_dead = rng.randint(1, M)
        k = rng.randint(1, min(n_dead, 200))
        N = rng.choice([rng.randint(1, 70), rng.randint(70, 500)])
        g = torch.Generator(device=dev).manual_seed(n)
        dead = torch.zeros(M, dtype=torch.bool, device=dev)
        dead[torch.randperm(M, device=dev, generator=g)[:n_dead]] = True
        pre = torch.randn(N, M, device=dev, generator=g).relu().bfloat16()
        acts, idx = ops.topk(pre, k, keep_mask=dead)
        slot_of, dead_list = ops.dead_compact(dead, n_dead)
        W = torch.randn(M, D, device=dev, generator=g)
        b = torch.randn(D, device=dev, generator=g)
        gy = torch.randn(N, D, device=dev, generator=g)
        res = {}
        for name in ("gather", "dense"):
            Wd, bd = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
            if name == "gather":
                out = SparseDecode.apply(idx, Wd, acts.float(), bd)
            else:
                out = DeadLatentDecode.apply(idx, acts, Wd, bd, dead_list, slot_of, n_dead)
            (out * gy).sum().backward()
            res[name] = (out.detach(), Wd.grad, bd.grad)
        for a, r, what in zip(res["dense"], res["gather"], ("decode", "W_dec.grad", "b_dec.grad")):
            err = float((a - r).abs().max()) / max(1e-30, float(r.abs().max()))
            if not err <= 2e-5:
                raise AssertionError(f"MISMATCH case={n} {what}: rel err {err:.2e}  N={N} D={D} M={M} n_dead={n_dead} k={k}")
        if bool((res["dense"][1][~dead] != 0).any()):
            raise AssertionError(f"MISMATCH case={n}: gradient rows outside the dead latents")
        n += 1
    print(f"{n} random cases in {time.time() - t0:.0f} s: dense AuxK path = row-gather path within 2e-5", flush=True)
    return n


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
