"""Random shapes through the fused encoder-gradient steps (sphip_segmented_weight_grad_signsgd and
..._schedulefree, with the b_dec partial sums) against the separate path (weight gradient, then the optimizer
kernel, then a cast for the shadow): parameters, schedule-free state and bf16 shadow bit for bit, the partial
sums to fp32 rounding. run(seconds, max_cases) is also what tests/test_hip_fuzz.py
calls for a bounded, seeded slice. Usage: python benchmarks/fused_fuzz.py [seconds]"""
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
    rng = random.Random(3)
    dev = torch.device("cuda", 0)
    t0, n = time.time(), 0
    while time.time() - t0 < budget and (max_cases is None or n < max_cases):
        D = rng.choice([8 * rng.randint(1, 40), 64 * rng.randint(1, 24), 768, 1024])
        M = rng.choice([rng.randint(1, 100), rng.randint(100, 3000), 32 * rng.randint(1, 150)])
        k = rng.randint(1, min(M, 40))
        N = rng.choice([rng.randint(1, 64), rng.randint(64, 600)])
        g = torch.Generator(device=dev).manual_seed(n)
        if rng.random() < 0.3:
            hot = torch.randint(0, M, (max(1, M // 40),), device=dev, generator=g)
            idx = hot[torch.randint(0, hot.numel(), (N, k), device=dev, generator=g)]
        else:
            idx = torch.rand(N, M, device=dev, generator=g).topk(k, dim=-1).indices
        rdt = rng.choice([torch.bfloat16, torch.bfloat16, torch.float32])
        gv = torch.randn(N, k, device=dev, generator=g).to(rng.choice([torch.bfloat16, torch.float32]))
        rows = torch.randn(N, D, device=dev, generator=g).to(rdt)
        shift = torch.randn(D, device=dev, generator=g) if rng.random() < 0.6 else None
        W0 = torch.randn(M, D, device=dev, generator=g)
        cs = torch.randn(M, device=dev, generator=g)
        lr = 3e-3
        seg = ops.build_latent_segments(idx, M)
        gW, _, _ = ops.segmented_weight_grad(seg, gv, rows, None, row_shift=shift)
        want_part = ops.vec_mat(cs, W0, 1.0) if hasattr(ops, "vec_mat") else None
        for sfree in (False, True):
            W = W0.clone()
            shadow = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
            if sfree:
                Z0 = W0 + 0.01 * torch.randn(M, D, device=dev, generator=g)
                Z, beta, c = Z0.clone(), 0.95, 0.25
                _, part = ops.segmented_weight_grad_signsgd_(seg, gv, rows, W, lr, shadow, row_shift=shift,
                                                             schedule_free=(Z, beta, c), coeff_sums=cs)
                Wr, Zr = W0.clone(), Z0.clone()
                ops.schedulefree_signsgd_step_(Wr, Zr, gW, lr, beta, c, None)
                ok = torch.equal(W, Wr) and torch.equal(Z, Zr)
            else:
                _, part = ops.segmented_weight_grad_signsgd_(seg, gv, rows, W, lr, shadow, row_shift=shift, coeff_sums=cs)
                Wr = W0.clone()
                ops.signsgd_step_(Wr, gW, lr, None)
                ok = torch.equal(W, Wr)
            ok = ok and torch.equal(shadow.view(torch.int16), Wr.bfloat16().view(torch.int16))
            got = part.double().sum(0)
            ref = cs.double() @ W0.double()
            perr = float((got - ref).abs().max()) / max(1.0, float(ref.abs().max()))
            if not ok or perr > 1e-5:
                raise AssertionError(f"MISMATCH case={n} sfree={sfree} N={N} D={D} M={M} k={k} rows={rdt} shift={shift is not None}: equal={ok} partial err {perr:.2e}")
        n += 1
    print(f"{n} random cases x 2 optimizers in {time.time() - t0:.0f} s: fused = separate, bit for bit", flush=True)
    return n


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
