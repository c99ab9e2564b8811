"""Random training-size shapes with heavy-tailed latent usage through the weight-gradient entry points, so that the
chunked long-segment path (wgrad.hip: latents with more than 128 entries, 128-entry chunks, combine kernel) does a
large share of the work: sphip_segmented_weight_grad with and without the dot part, fp32 / bf16 rows, fp32 / bf16
coefficients (8- and 4-byte list entries), inactive slots, row shift -- against a float64 dense computation (2e-5
of the largest element) -- and sphip_segmented_weight_grad_signsgd / _schedulefree against gradient + separate step
(bit for bit), twice (identical bits); the accumulate form against gW + gW (bit for bit) and the
token-block driver against float64. run(seconds, max_cases) is also what tests/test_hip_fuzz.py calls for a
bounded, seeded slice. Usage: python benchmarks/long_fuzz.py [seconds]"""
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
    rng = random.Random(6)
    dev = torch.device("cuda", 0)
    t0, n, told = time.time(), 0, time.time()
    while time.time() - t0 < budget and (max_cases is None or n < max_cases):
        rdt = rng.choice([torch.float32, torch.bfloat16])
        D = rng.choice([512, 768, 1024, 2048])                       # multiples of 256: the long-segment path's widths
        rows_min = (8 << 20) // (D * (4 if rdt == torch.float32 else 2))      # N * D * bytes >= 8 MiB: training-size dispatch
        N = rows_min + rng.choice([0, 1, 37, 640, rng.randint(0, 3000)])
        M = rng.choice([rng.randint(40, 400), 32 * rng.randint(8, 128), rng.randint(400, 6000), 32 * rng.randint(100, 500)])
        k = rng.randint(2, min(M // 2, 24))
        g = torch.Generator(device=dev).manual_seed(n)
        # power-law usage: latent ranks ~ 1 / r, dealt to rows at random; a token's k latents are distinct
        w = 1.0 / torch.arange(1, M + 1, device=dev, dtype=torch.float32) ** rng.choice([0.8, 1.0, 1.3])
        perm = torch.randperm(M, device=dev, generator=g)
        scores = torch.rand(N, M, device=dev, generator=g).log_() / w            # Gumbel-style weighted sampling without replacement
        idx = perm[scores.topk(k, dim=-1).indices]
        if rng.random() < 0.3:
            idx = torch.where(torch.rand(N, k, device=dev, generator=g) < 0.05, torch.full_like(idx, -1), idx)
        cdt = rng.choice([torch.float32, torch.bfloat16])
        coeff = torch.randn(N, k, device=dev, generator=g).to(cdt)
        rows = torch.randn(N, D, device=dev, generator=g).to(rdt)
        W = torch.randn(M, D, device=dev, generator=g)
        shift = torch.randn(D, device=dev, generator=g) if rng.random() < 0.5 else None
        live = idx >= 0
        safe = idx.clamp_min(0)
        c64 = coeff.double() * live
        r64 = rows.double() - (shift.double() if shift is not None else 0.0)
        want_gW = torch.zeros(M, D, dtype=torch.float64, device=dev)
        for s in range(k):                                           # (N, D) at a time: no (N k, D) temporary
            want_gW.index_add_(0, safe[:, s], c64[:, s:s + 1] * r64)
        want_gb = torch.zeros(M, dtype=torch.float64, device=dev).index_add_(0, safe.reshape(-1), c64.reshape(-1))
        seg = ops.build_latent_segments(idx, M)
        lens = seg.seg_begin[1:] - seg.seg_begin[:-1]
        what = f"case={n} N={N} D={D} M={M} k={k} rows={rdt} coeff={cdt} shift={shift is not None} longest={int(lens.max())} long={int((lens > 128).sum())}"
        scale = max(1.0, float(want_gW.abs().max()))
        gW, gb, _ = ops.segmented_weight_grad(seg, coeff, rows, None, want_bias=True, row_shift=shift)
        err = float((gW.double() - want_gW).abs().max()) / scale
        errb = float((gb.double() - want_gb).abs().max()) / max(1.0, float(want_gb.abs().max()))
        if err > 2e-5 or errb > 2e-5:
            raise AssertionError(f"MISMATCH (no dot) {what}: gW {err:.2e} gb {errb:.2e}")
        if rdt == torch.float32 and shift is None:
            want_dc = torch.stack([(rows.double() * W.double()[safe[:, s]]).sum(-1) for s in range(k)], 1) * live
            gW2, _, dc = ops.segmented_weight_grad(seg, coeff, rows, W, want_dcoeff=True)
            err = float((gW2.double() - want_gW).abs().max()) / scale
            errd = float(((dc.double() - want_dc) * live).abs().max()) / max(1.0, float(want_dc.abs().max()))
            if err > 2e-5 or errd > 2e-5:
                raise AssertionError(f"MISMATCH (dot) {what}: gW {err:.2e} dcoeff {errd:.2e}")
            gW3, _, dc3 = ops.segmented_weight_grad(seg, coeff, rows, W, want_dcoeff=True)
            if not (torch.equal(gW2, gW3) and torch.equal(dc, dc3)):
                raise AssertionError(f"MISMATCH (dot, second run differs) {what}")
        # the accumulate form (sphip_segmented_weight_grad_acc): a second call over the same batch adds its sums to
        # what is there -- bit for bit gW + gW, whatever the kernel form; the token-block driver over the batch cut at
        # a random point against the float64 sums
        dot = rdt == torch.float32 and shift is None and rng.random() < 0.5
        if ops.weight_grad_accumulates(N, k, M, D, rdt, dot):
            first = ops.segmented_weight_grad(seg, coeff, rows, W if dot else None, want_bias=True, want_dcoeff=dot, row_shift=shift)
            acc_W, acc_b = first[0].clone(), first[1].clone()
            ops.segmented_weight_grad(seg, coeff, rows, W if dot else None, want_bias=True, want_dcoeff=dot, row_shift=shift,
                                      into=(acc_W, acc_b))
            if not (torch.equal(acc_W, first[0] + first[0]) and torch.equal(acc_b, first[1] + first[1])):
                raise AssertionError(f"MISMATCH (accumulate != gW + gW, dot={dot}) {what}")
        old_block = ops.TOKEN_BLOCK
        ops.TOKEN_BLOCK = max(64, N // rng.randint(2, 4) + rng.randint(0, 63))
        try:
            tW, tb, _ = ops.weight_grad_in_token_blocks(idx, M, coeff, rows, W if dot else None, want_bias=True,
                                                        want_dcoeff=dot, row_shift=shift)
        finally:
            ops.TOKEN_BLOCK = old_block
        err = float((tW.double() - want_gW).abs().max()) / scale
        errb = float((tb.double() - want_gb).abs().max()) / max(1.0, float(want_gb.abs().max()))
        if err > 2e-5 or errb > 2e-5:
            raise AssertionError(f"MISMATCH (token blocks, dot={dot}) {what}: gW {err:.2e} gb {errb:.2e}")
        # (the fused step is what FusedEncoder.backward takes below a mean segment of 64 entries, fused_encoder.py;
        # beyond a mean of 256 the unfused gradient goes to the wave-per-latent kernel, another summation order)
        if N * k < 64 * M:
            lr = 3e-3
            W0 = W
            sep = W0.clone()
            sh_sep = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
            ops.signsgd_step_(sep, gW, lr, sh_sep)
            for _ in range(2):
                fused = W0.clone()
                sh = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
                ops.segmented_weight_grad_signsgd_(seg, coeff, rows, fused, lr, sh, row_shift=shift, coeff_sums=gb)
                if not (torch.equal(fused, sep) and torch.equal(sh.view(torch.int16), sh_sep.view(torch.int16))):
                    raise AssertionError(f"MISMATCH (fused SignSGD step != gradient + step) {what}")
            Z0 = W0 + 0.01 * torch.randn(M, D, device=dev, generator=g)
            Wr, Zr = W0.clone(), Z0.clone()
            ops.schedulefree_signsgd_step_(Wr, Zr, gW, lr, 0.95, 0.25, None)
            Wf, Zf = W0.clone(), Z0.clone()
            ops.segmented_weight_grad_signsgd_(seg, coeff, rows, Wf, lr, None, row_shift=shift, schedule_free=(Zf, 0.95, 0.25))
            if not (torch.equal(Wf, Wr) and torch.equal(Zf, Zr)):
                raise AssertionError(f"MISMATCH (fused schedule-free step != gradient + step) {what}")
        n += 1
        if time.time() - told > 30:
            told = time.time()
            print(f"... {n} cases", flush=True)
    print(f"{n} random training-size cases in {time.time() - t0:.0f} s: long-segment path and token blocks within 2e-5 of float64, fused = separate, accumulate = sum", flush=True)
    return n


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
