"""Random shapes through the fused decoder steps (sphip_decoder_project_signsgd with and without
re-normalisation and the slice-major copy, sphip_decoder_schedulefree_signsgd_step) against the separate
passes they replace (projected gradient, step, re-normalisation, copy): bit for bit for SignSGD; the
schedule-free step against its element-wise form on the projected gradient. run(seconds, max_cases) is also what tests/test_hip_fuzz.py
calls for a bounded, seeded slice. Usage: python
benchmarks/decoder_fuzz.py [seconds]"""
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
    rng = random.Random(4)
    dev = torch.device("cuda", 0)
    t0, n = time.time(), 0
    while time.time() - t0 < budget and (max_cases is None or n < max_cases):
        D = rng.choice([4 * rng.randint(1, 300), 32 * rng.randint(1, 256), 768, 4096, 5120, 8192])
        M = rng.choice([rng.randint(1, 200), 64 * rng.randint(1, 40), rng.randint(200, 3000)])
        g = torch.Generator(device=dev).manual_seed(n)
        W0 = torch.randn(M, D, device=dev, generator=g)
        W0 /= W0.norm(dim=1, keepdim=True)
        G = torch.randn(M, D, device=dev, generator=g)
        lr = 3e-3
        renorm = rng.random() < 0.7
        use_sm = D % 32 == 0 and rng.random() < 0.7
        # separate passes
        Gp = G.clone()
        ops.remove_parallel_grad_(Gp, W0)
        Wr = W0.clone()
        ops.signsgd_step_(Wr, Gp, lr, None)
        if renorm:
            ops.unit_norm_rows_(Wr)
        # fused
        W = W0.clone()
        buf = torch.full((D // 32, M, 32), float("nan"), device=dev) if use_sm else None
        ops.decoder_project_signsgd_(W, G, lr, renorm, slice_major=buf)
        ok = torch.equal(W, Wr)
        if ok and use_sm:
            ok = torch.equal(buf, W.view(M, D // 32, 32).permute(1, 0, 2).contiguous())
        if not ok:
            raise AssertionError(f"MISMATCH (signsgd) case={n} M={M} D={D} renorm={renorm} slice_major={use_sm}")
        # schedule-free: the fused decoder step = element-wise schedule-free step on the projected gradient (+ renorm of y)
        Z0 = W0 + 0.01 * torch.randn(M, D, device=dev, generator=g)
        beta, c = 0.95, 0.25
        Wy, Wz = W0.clone(), Z0.clone()
        buf = torch.full((D // 32, M, 32), float("nan"), device=dev) if use_sm else None
        ops.decoder_schedulefree_signsgd_step_(Wy, Wz, G, lr, beta, c, renorm, slice_major=buf)
        Ry, Rz = W0.clone(), Z0.clone()
        ops.schedulefree_signsgd_step_(Ry, Rz, Gp, lr, beta, c, None)
        if renorm:
            ops.unit_norm_rows_(Ry)
        ok = torch.equal(Wy, Ry) and torch.equal(Wz, Rz)
        if ok and use_sm:
            ok = torch.equal(buf, Wy.view(M, D // 32, 32).permute(1, 0, 2).contiguous())
        if not ok:
            raise AssertionError(f"MISMATCH (schedule-free) case={n} M={M} D={D} renorm={renorm} slice_major={use_sm}: "
                  f"y diff {float((Wy - Ry).abs().max()):.3e} z diff {float((Wz - Rz).abs().max()):.3e}")
        n += 1
    print(f"{n} random cases in {time.time() - t0:.0f} s: fused decoder steps = separate passes, bit for bit", flush=True)
    return n


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
