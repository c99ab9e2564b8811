"""Random shapes through the decode gather and the segmented weight gradients, every kernel form
(SPHIP_GATHER / SPHIP_WGRAD = full | sliced | wide | l2, forms a shape does not take fall back), against a
dense float64 torch computation: inactive slots (-1), repeated latents, empty and very long segments,
widths that are not powers of two. run(seconds, max_cases) is also what tests/test_hip_fuzz.py
calls for a bounded, seeded slice. Usage: python benchmarks/sparse_fuzz.py [seconds]"""
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
    rng = random.Random(1)
    dev = torch.device("cuda", 0)
    t0, n = time.time(), 0
    while time.time() - t0 < budget and (max_cases is None or n < max_cases):
        D = rng.choice([4 * rng.randint(1, 64), 32 * rng.randint(1, 40), 64 * rng.randint(1, 16), 256, 768, 1024])
        M = rng.choice([rng.randint(1, 64), rng.randint(64, 3000), 32 * rng.randint(1, 200)])
        k = rng.randint(1, min(M, 48))
        N = rng.choice([rng.randint(1, 64), rng.randint(64, 700)])
        g = torch.Generator().manual_seed(n)
        skew = rng.random() < 0.3
        if skew:                                         # a few latents take most of the slots
            hot = torch.randint(0, M, (max(1, M // 50),), generator=g)
            idx = hot[torch.randint(0, hot.numel(), (N, k), generator=g)]
            mix = torch.rand(N, k, generator=g) < 0.3
            idx = torch.where(mix, torch.randint(0, M, (N, k), generator=g), idx)
        else:
            idx = torch.stack([torch.randperm(M, generator=g)[:k] for _ in range(N)])
        if rng.random() < 0.3:
            idx = torch.where(torch.rand(N, k, generator=g) < 0.1, torch.tensor(-1), idx)
        cdt = rng.choice([torch.float32, torch.bfloat16])
        rdt = rng.choice([torch.float32, torch.bfloat16])
        coeff = torch.randn(N, k, generator=g).to(cdt)
        rows = torch.randn(N, D, generator=g).to(rdt)
        W = torch.randn(M, D, generator=g)
        bias = torch.randn(D, generator=g)
        live = (idx >= 0)
        safe = idx.clamp_min(0)
        c64 = coeff.double() * live
        want_out = torch.einsum("nk,nkd->nd", c64, W.double()[safe]) + bias.double()
        want_gW = torch.zeros(M, D, dtype=torch.float64)
        want_gW.index_add_(0, safe.reshape(-1), (c64.unsqueeze(-1) * rows.double().unsqueeze(1)).reshape(-1, D))
        want_gb = torch.zeros(M, dtype=torch.float64).index_add_(0, safe.reshape(-1), c64.reshape(-1))
        want_dc = torch.einsum("nd,nkd->nk", rows.double(), W.double()[safe]) * live
        di, dc, dr, dW, db = idx.to(dev), coeff.to(dev), rows.to(dev), W.to(dev), bias.to(dev)
        seg = ops.build_latent_segments(di, M)
        for mode in ("", "full", "sliced", "wide", "l2"):
            for var in ("SPHIP_GATHER", "SPHIP_WGRAD"):
                if mode:
                    os.environ[var] = mode
                else:
                    os.environ.pop(var, None)
            out = ops.gather_rows_weighted(di, dc, dW, db)
            gW, gb, dcf = ops.segmented_weight_grad(seg, dc, dr, dW, want_bias=True, want_dcoeff=True)
            scale = max(1.0, float(want_out.abs().max()))
            errs = {"out": float((out.cpu().double() - want_out).abs().max()) / scale,
                    "gW": float((gW.cpu().double() - want_gW).abs().max()) / max(1.0, float(want_gW.abs().max())),
                    "gb": float((gb.cpu().double() - want_gb).abs().max()) / max(1.0, float(want_gb.abs().max())),
                    "dc": float(((dcf.cpu().double() - want_dc) * live).abs().max()) / max(1.0, float(want_dc.abs().max()))}
            if max(errs.values()) > 2e-5:
                raise AssertionError(f"MISMATCH case={n} mode={mode!r} N={N} D={D} M={M} k={k} skew={skew} coeff={cdt} rows={rdt}: {errs}")
        n += 1
    print(f"{n} random cases x 5 kernel forms in {time.time() - t0:.0f} s: all within 2e-5 of float64", flush=True)
    os.environ.pop("SPHIP_GATHER", None)
    os.environ.pop("SPHIP_WGRAD", None)
    return n


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
