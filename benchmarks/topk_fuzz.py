"""Random shapes through sphip_topk (all its paths: four-wave block-maxima, 1024-thread sweeps, exact
bisection, register-resident exact) against torch.topk: the selected values must be the k largest and
the indices must point at them; and a wider selection of the same row, through sphip_exclude_selected,
must mark exactly the members of the narrower one. run(seconds, max_cases) is also what tests/test_hip_fuzz.py
calls for a bounded, seeded slice. Usage: python benchmarks/topk_fuzz.py [seconds]"""
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_amd import ops  # noqa: E402


def run(budget: float = 60.0, max_cases: int | None = None) -> int:
    """Random cases (seeded: the sequence is the same on every run) until `budget` seconds or `max_cases`
    have passed; raises AssertionError naming the first mismatching case; returns the number of cases."""
    rng = random.Random(0)
    dev = torch.device("cuda", 0)
    t0, n = time.time(), 0
    while time.time() - t0 < budget and (max_cases is None or n < max_cases):
        dtype = rng.choice([torch.float32, torch.bfloat16])
        M = rng.choice([rng.randint(1, 300), rng.randint(300, 20000), 8 * rng.randint(1, 4096), 128 * rng.randint(1, 1024)])
        k = rng.choice([1, rng.randint(1, min(M, 64)), rng.randint(1, min(M, 4096)), M])
        N = rng.randint(1, 70)
        kind = rng.randint(0, 3)
        g = torch.Generator().manual_seed(n)
        if kind == 0:
            pre = torch.randn(N, M, generator=g)
        elif kind == 1:
            pre = torch.randn(N, M, generator=g).clamp_min(0)
        elif kind == 2:
            pre = torch.randint(0, 5, (N, M), generator=g).float()
        else:
            pre = torch.randn(N, M, generator=g).clamp_min(0) * (torch.rand(N, M, generator=g) < 0.02)
        pre = pre.to(dtype)
        d = pre.to(dev)
        tm = None
        if rng.random() < 0.5 and M % 8 == 0:
            T = -(-M // 128)
            pad = torch.full((-(-N // 64) * 64, T * 128), float("-inf"))
            pad[:N, :M] = pre.float()
            tm = pad.reshape(-1, 64, T, 128).amax(-1).permute(0, 2, 1).contiguous().to(dtype).to(dev)
        vals, idx = ops.topk(d, k, tile_max=tm)
        kw = min(M, rng.choice([2, 4]) * k)
        wider_ok = True
        if kw > k:
            # the wider selection of the same row holds the narrower one: exclude_selected marks exactly its members
            vw, iw = ops.topk(d, kw, tile_max=tm if rng.random() < 0.5 else None)
            marked = ops.exclude_selected(vw, iw, vals, idx).cpu() == -1
            wider_ok = bool((marked.sum(1) == k).all()) and torch.equal(
                iw.cpu()[marked].reshape(N, k).sort(-1).values, idx.cpu().sort(-1).values)
        vals, idx = vals.cpu(), idx.cpu()
        ref = pre.float().topk(k, dim=-1).values
        ok = (torch.equal(vals.float().sort(-1, descending=True).values, ref) and torch.equal(pre.float().gather(1, idx), vals.float())
              and int(idx.min()) >= 0 and int(idx.max()) < M and bool((idx.sort(-1).values.diff(dim=-1) > 0).all() if k > 1 else True))
        if not (ok and wider_ok):
            raise AssertionError(f"MISMATCH dtype={dtype} N={N} M={M} k={k} kind={kind} tile_max={tm is not None} case={n}")
        n += 1
    print(f"{n} random cases in {time.time() - t0:.0f} s: all selections exact", flush=True)
    return n


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
