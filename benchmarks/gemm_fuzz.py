"""Random shapes through sphip_encoder_preacts (both tile kernels, bf16 and fp32, with / without bias and
ReLU, with the block maxima) on small-integer operands, for which every product and sum is exact: the
result must equal the integer computation bit for bit. run(seconds, max_cases) is also what tests/test_hip_fuzz.py
calls for a bounded, seeded slice. Usage: python benchmarks/gemm_fuzz.py [seconds]"""
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
    rng = random.Random(2)
    dev = torch.device("cuda", 0)
    t0, n = time.time(), 0
    while time.time() - t0 < budget and (max_cases is None or n < max_cases):
        dtype = rng.choice([torch.bfloat16, torch.bfloat16, torch.float32])
        N = rng.choice([rng.randint(1, 300), 64 * rng.randint(1, 12), 256 * rng.randint(1, 5)])
        K = rng.choice([8 * rng.randint(1, 25), 8 * rng.randint(1, 100), 64 * rng.randint(1, 20)])     # the ABI asks for D % 8 == 0
        M = rng.choice([rng.randint(1, 500), 128 * rng.randint(1, 12), 256 * rng.randint(1, 8)])
        g = torch.Generator().manual_seed(n)
        dens = min(1.0, 40.0 / K)                                   # |sums| stay below 256: exact in bf16 too
        x = (torch.randint(-1, 2, (N, K), generator=g) * (torch.rand(N, K, generator=g) < dens)).float()
        W = torch.randint(-2, 3, (M, K), generator=g).float()
        b = torch.randint(-3, 4, (M,), generator=g).float() if rng.random() < 0.7 else None
        relu = rng.random() < 0.7
        want = x.double() @ W.double().T + (0 if b is None else b.double())
        if relu:
            want = want.clamp_min(0)
        assert float(want.abs().max()) <= 256
        for tile in ("", "128", "256"):
            if tile:
                os.environ["SPHIP_GEMM"] = tile
            else:
                os.environ.pop("SPHIP_GEMM", None)
            res = ops.encoder_preacts(x.to(dtype).to(dev), W.to(dtype).to(dev), None if b is None else b.to(dtype).to(dev),
                                      relu=relu, want_tile_max=(dtype == torch.bfloat16 and relu))
            pre, tm = res if isinstance(res, tuple) else (res, None)
            ok = torch.equal(pre.cpu().double(), want)
            if ok and tm is not None:
                T = -(-M // 128)
                pad = torch.full((-(-N // 64) * 64, T * 128), float("-inf"), dtype=torch.float64)
                pad[:N, :M] = want
                wtm = pad.reshape(-1, 64, T, 128).amax(-1).permute(0, 2, 1)
                rows = torch.arange(N)
                ok = torch.equal(tm.cpu().double()[rows // 64, :, rows % 64], wtm[rows // 64, :, rows % 64])
            if not ok:
                raise AssertionError(f"MISMATCH case={n} tile={tile!r} dtype={dtype} N={N} K={K} M={M} bias={b is not None} relu={relu}")
        n += 1
    print(f"{n} random cases x 3 tile choices in {time.time() - t0:.0f} s: all bit-exact", flush=True)
    os.environ.pop("SPHIP_GEMM", None)
    return n


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
