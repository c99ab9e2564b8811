#!/usr/bin/env python
"""Headline benchmark: SAE train tokens/s at d_in=4096, 131072 latents, k=192 (BASELINE.json
configs[2], the configuration the metric is quoted on), bf16 autocast over fp32 parameters,
synthetic activations, plain SignSGD, one process per GPU (DDP over tokens via RCCL for N>1).

A step is one full Trainer micro-step + step tail (sparsify/trainer.py:340-447, 510-538):
decoder re-normalisation, encode (bf16 MFMA GEMM + top-k), sparse decode, FVU, backward
(segmented weight gradients), gradient all-reduce (N>1), projected gradient + SignSGD,
dead-latent counters. Inputs are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

WORKLOADS = {
    # BASELINE.json configs[2]: Llama-3-8B layer shape
    "llama3-8b-layer d_in=4096 M=131072 k=192": dict(d_in=4096, num_latents=131072, k=192),
    # BASELINE.json configs[0] (CPU-runnable reference case), for quick checks
    "gpt2-small d_in=768 M=24576 k=32": dict(d_in=768, num_latents=24576, k=32),
    # BASELINE.json configs[4]: MLP transcoder, 262144 latents, for --mode shard (latents split over the ranks)
    "llama3-8b-transcoder d_in=4096 M=262144 k=192": dict(d_in=4096, num_latents=262144, k=192, transcode=True),
}
MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: ~2.5 PF dense bf16
# Ceilings of the memory levels the byte-bound kernels are served from (MI355X_MICROARCH.md, "Indexed
# rows: gather into LDS" and "HBM"): random rows out of the XCDs' L2s 16.8-18.8 TB/s measured chip-wide
# (the upper end is used, so a fraction never flatters), out of the Infinity Cache 7.4-8.6 TB/s, HBM
# 8.0 TB/s spec (6.3 TB/s is what a streaming copy achieves).
PEAK_GBS = {"L2": 18800.0, "MALL": 8600.0, "HBM": 8000.0}


def cpu_baseline(w: dict, tokens: int, budget_s: float = 25.0) -> dict:
    """The CPU oracle (a restatement of the reference's torch path, pinned to it by golden
    vectors) timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import sae_oracle as O

    torch.manual_seed(0)
    D, M, k = w["d_in"], w["num_latents"], w["k"]
    P = O.init_params(D, M, seed=0)
    cfg = O.OracleConfig(k=k)
    lr = O.signum_lr(M)
    g = torch.Generator().manual_seed(0)
    offset = torch.randn(D, generator=g)
    times = []
    t_begin = time.perf_counter()
    step = 0
    while True:
        gs = torch.Generator().manual_seed(1000 + step)
        x = torch.randn(tokens, D, generator=gs) + offset
        t0 = time.perf_counter()
        O.train_step_signsgd(P, x, None, cfg, lr, step=step, bf16=False)
        dt = time.perf_counter() - t0
        if step > 0:
            times.append(dt)
        step += 1
        if (time.perf_counter() - t_begin > budget_s and len(times) >= 2) or len(times) >= 10:
            break
    times.sort()
    med = times[len(times) // 2]
    return {"value": tokens / med, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(times)} fp32 steps of {tokens} tokens, same shapes/init/optimizer; "
                      f"the reference's (N*k,D) fp32 temporary forbids the GPU batch size on CPU"}


def newest_profile(variant: str = ""):
    """The newest committed counter summary of this command (profiles/rNN_vNN_pmc_summary.json; with
    `variant`, rNN_vNN_<variant>_pmc_summary.json: profiles of A/B switches are not the default run's), made on the GPU box by
    benchmarks/profile.sh from this command: kernel-trace stats + FETCH_SIZE, WRITE_SIZE, TCC hit/miss
    and SQ/GRBM passes in runs of their own, FETCH_SIZE doubled for gfx950). Counters cannot be read
    inside a timed run, so the bench line quotes the profile's numbers next to its own live timings."""
    import glob
    import re
    want = re.compile(r"^r\d+_v\d+_" + (variant + "_" if variant else "") + r"pmc_summary\.json$")
    files = sorted((f for f in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_summary.json"))
                    if want.match(os.path.basename(f))),
                   key=lambda f: [int(t) for t in re.findall(r"\d+", os.path.basename(f))])      # r01_v9 < r01_v10
    if not files:
        return None, {}
    return os.path.basename(files[-1]), json.load(open(files[-1]))


def profiled(prof: dict, prefixes) -> dict | None:
    """Per-step totals over the profile's kernels whose names start with one of `prefixes`:
    duration, HBM-side bytes (FETCH_SIZE x2 + WRITE_SIZE), launches per step."""
    steps = prof.get("_steps", 6)            # profile.sh: 2 warm-up + 4 timed steps, all traced
    ms = by = n = 0.0
    for name, d in prof.items():
        if isinstance(d, dict) and any(name.startswith(p) for p in prefixes) and "avg_us" in d:
            calls = d.get("calls", 0) / steps
            ms += d["avg_us"] * 1e-3 * calls
            by += (d.get("fetch_bytes_per_launch", 0.0) + d.get("write_bytes_per_launch", 0.0)) * calls
            n += calls
    return {"ms": ms, "counter_bytes": by, "launches": n} if n else None


def kernel_reports(N: int, D: int, M: int, k: int):
    """The byte-bound entry points of the step: (timer name, what, level that serves the bulk of
    its bytes, algorithmic bytes per step (SURVEY.md 8d figures x the units one launch processes),
    the kernels it launches as named in the rocprof summaries)."""
    G = 1 << 30
    return [
        ("sphip_gather_rows_weighted", "decode: sum of k W_dec rows per token + b_dec (L2-resident gather from the "
         "slice-major copy of W_dec the decoder step leaves; includes bucketing)", "L2", N * (k * D * 4 + D * 4),
         ["sphip::gather_l2_kernel", "sphip::bucket", "sphip::slice_major"]),
        ("sphip_segmented_weight_grad", "W_dec.grad (dense, all M rows) + d top_acts: N k rows of dL/dy gathered, "
         "W_dec read, gradient written", "L2", N * k * D * 4 + 2 * (M * D * 4),
         ["sphip::seg_wgrad_sliced_kernel<float", "sphip::seg_dcoeff", "sphip::slice_major_kernel", "sphip::seg_pack",
          "sphip::seg_long", "sphip::seg_wgrad_wide"]),
        ("sphip_segmented_weight_grad_signsgd", "encoder.weight gradient fused with its SignSGD step: N k bf16 rows "
         "gathered, parameter read + written, bf16 shadow written", "L2", N * k * D * 2 + M * D * (4 + 4 + 2),
         ["sphip::seg_wgrad_sliced_kernel<unsigned short"]),
        ("sphip_topk", "row top-k from the block maxima: the maxima + one 256-B block per selected latent",
         "HBM", N * (M // 128) * 2 + N * k * 256, ["sphip::topk_kernel"]),
        ("sphip_decoder_project_signsgd", "decoder step tail: projected gradient + sign step + re-normalisation, "
         "W_dec and its gradient read, W_dec and its slice-major copy written", "HBM", 4 * M * D * 4, ["sphip::decoder_row_resident_kernel"]),
        ("sphip_vec_mat", "b_dec's encoder-side gradient gb^T W_enc: one pass over encoder.weight", "HBM", M * D * 4,
         ["sphip::col_partial_kernel<float"]),
    ]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tokens", type=int, default=16384, help="tokens per GPU per step")
    ap.add_argument("--workload", default="llama3-8b-layer d_in=4096 M=131072 k=192", choices=list(WORKLOADS))
    ap.add_argument("--cpu-tokens", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-auxk", action="store_true", help="skip the extra runs of the headline workload (AuxK, multi-top-k, signum, skewed usage)")
    ap.add_argument("--fp32", action="store_true", help="fp32 activations / no autocast (not the headline)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--force-exchange", action="store_true",
                    help="rehearsal with --gpus 1: run the data-parallel gradient exchange over a one-rank group")
    ap.add_argument("--no-overlap", action="store_true", help="gradient exchange on the compute stream (A/B)")
    ap.add_argument("--lazy-pre-acts", action="store_true",
                    help="TrainConfig.lazy_pre_acts: top-k from candidate lists, the (N, M) matrix is not stored (A/B)")
    ap.add_argument("--mode", default="ddp", choices=["ddp", "shard"],
                    help="ddp: replicas, tokens split over the ranks (weak scaling, the headline); shard: the latents of "
                         "one dictionary split over the ranks, every rank sees the same tokens (strong scaling)")
    ap.add_argument("--optimizer", default="signsgd", choices=["signsgd", "signum", "adam", "muon"],
                    help="signsgd (pinned to the reference's sign_sgd.py) is the headline; signum = the reference's "
                         "default, schedule-free SignSGD (third-party package: parity unpinned)")
    args = ap.parse_args()

    if os.environ.get("SPHIP_HANG_DUMP"):      # debugging aid: dump all stacks if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["SPHIP_HANG_DUMP"]), exit=True)
    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` on its own: one child process per GPU, started before anything here
        # touches the GPU (the reference's entry point takes its rank from LOCAL_RANK the same way,
        # sparsify/__main__.py:144-155); under torch.distributed.run the environment is already there
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)
    elif args.force_exchange:
        dist.init_process_group(args.backend, init_method="tcp://127.0.0.1:29533", rank=0, world_size=1,
                                **({"device_id": device} if args.backend == "nccl" else {}))
        from sparsify_amd import parallel as _par
        _par.SignExchange.FORCE = True

    from sparsify_amd import SparseCoderConfig, TrainConfig, ops
    from sparsify_amd.trainer import SyntheticActivations, Trainer

    w = WORKLOADS[args.workload]
    N, D, M, k = args.tokens, w["d_in"], w["num_latents"], w["k"]
    shard = args.mode == "shard"
    cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k, transcode=bool(w.get("transcode"))), shard_latents=shard,
                      optimizer=args.optimizer, hookpoints=["layers.10"],
                      save_every=10**9, wandb_log_frequency=10**9, save_dir="", log_to_wandb=False,
                      overlap_exchange=not args.no_overlap, lazy_pre_acts=args.lazy_pre_acts)
    total = args.warmup + args.steps
    src = SyntheticActivations(["layers.10"], D, N, total, device=device,
                               dtype=torch.float32 if args.fp32 else torch.bfloat16, rank=0 if shard else rank,
                               transcode=bool(w.get("transcode")))
    trainer = Trainer(cfg, src, device=device)
    sae = trainer.saes["layers.10"]
    if args.fp32:
        fwd = type(sae).forward.__wrapped__
        type(sae).forward = lambda self, *a, **kw: _fp32_forward(fwd, self, *a, **kw)
    batches = [src.batch(s) for s in range(total)]        # resident in HBM before timing
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for s in range(args.warmup):
        trainer.train_micro_batch(batches[s])
    barrier()
    ops.timer.enabled = True
    ops.timer.reset()
    t0 = time.perf_counter()
    for s in range(args.warmup, total):
        trainer.train_micro_batch(batches[s])
    barrier()
    elapsed = time.perf_counter() - t0
    ops.timer.enabled = False
    kern = ops.timer.summary()
    fvu = float(trainer._metrics["fvu/layers.10"].item()) * cfg.wandb_log_frequency / max(total, 1)

    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        from sparsify_amd import parallel
        parallel.all_reduce_(t, "max")
    elapsed = float(t.item())

    headline = args.workload.startswith("llama3-8b-layer") and N == 16384 and not args.fp32 and not shard
    variants = {}
    if headline and world == 1 and not args.no_auxk and args.optimizer == "signsgd":
        del trainer, sae, batches
        torch.cuda.empty_cache()
        for kind in ("auxk", "multitopk", "signum", "skewed"):
            variants[kind] = variant_run(kind, N, D, M, k, device)
        variants["batch65536"] = variant_run("batch65536", 65536, D, M, k, device, steps=3, warm=2)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        # one launch per step, or one per latent block when the forward follows an overlapped
        # gradient exchange (N > 1): the roofline figure is per step either way
        gemm_ms = sum(kern[n]["total_ms"] for n in ("sphip_encoder_preacts", "sphip_encoder_preacts_block",
                                                    "sphip_encoder_preacts_emit")
                      if n in kern) / args.steps
        flops = 2.0 * N * D * M / (world if shard else 1)        # algorithmic FLOP per step and rank
        tflops = flops / (gemm_ms * 1e-3) / 1e12
        prof_name, prof = newest_profile("lazy" if args.lazy_pre_acts else "") if headline else (None, {})
        roof = {"bound": "mfma", "achieved": tflops, "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tflops / MFMA_BF16_DENSE_PEAK_TFLOPS, "traffic": None,
                "kernel": "encoder_gemm256_kernel (sphip_encoder_preacts)", "ms": gemm_ms,
                "how": "achieved = 2*N*D*M FLOP / the kernel's mean duration in THIS run (HIP events on the launching "
                       "stream); `profiled` = the same from the committed rocprofv3 run of this command, on the box "
                       "and at the clock named there (boxes of the pool differ by several per cent)"}
        g = prof.get(next((n for n in prof if n.startswith("sphip::encoder_gemm256_kernel<true")), ""), None)
        if g:
            pt = flops / (g["avg_us"] * 1e-6) / 1e12
            roof["traffic"] = g.get("fetch_bytes_per_launch", 0.0) + g.get("write_bytes_per_launch", 0.0)
            roof["profiled"] = {
                "file": f"profiles/{prof_name}", "ms": g["avg_us"] * 1e-3, "achieved": pt,
                "frac": pt / MFMA_BF16_DENSE_PEAK_TFLOPS, "effective_clock_GHz": g.get("effective_clock_GHz"),
                # busy cycles are summed over 256 CUs x 4 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
                "mfma_busy_frac_of_cycles": (g["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (g["GRBM_GUI_ACTIVE"] / 8)
                                             if g.get("SQ_VALU_MFMA_BUSY_CYCLES") and g.get("GRBM_GUI_ACTIVE") else None),
                "l2_hit_rate": g.get("l2_hit_rate"),
                "traffic_note": "traffic = HBM-side bytes per launch, FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes",
            }
        byte_kernels = None
        if not shard:
            # per-rank partial work in shard mode: the one-GPU byte counts do not apply
            byte_kernels = {}
            for name, what, level, alg_bytes, kernel_names in kernel_reports(N, D, M, k):
                if name not in kern:
                    continue
                t_ms = kern[name]["total_ms"] / args.steps
                ach = alg_bytes / (t_ms * 1e-3) / 1e9
                entry = {"what": what, "bound": level, "algorithmic_bytes": alg_bytes, "ms": t_ms, "achieved_GBs": ach,
                         "peak_GBs": PEAK_GBS[level], "frac": ach / PEAK_GBS[level]}
                pk = profiled(prof, kernel_names) if prof else None
                if pk:
                    entry["profiled"] = {"ms": pk["ms"], "counter_bytes": pk["counter_bytes"], "launches": pk["launches"]}
                byte_kernels[name] = entry
        out = {
            "metric": "SAE train tokens/sec at d=4096, M=131072, k=192 (1-8 GPUs); MFMA util %",
            "value": N * (1 if shard else world) * args.steps / elapsed,
            "unit": "tokens/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "strong" if shard else "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.fp32 else "bf16",
            "data": "synthetic",
            "config": {"workload": args.workload, "tokens_per_gpu_per_step": N, "optimizer": args.optimizer,
                       "params": "fp32", "lazy_pre_acts": bool(args.lazy_pre_acts), "parallelism": f"latent-shard{world}" if shard else f"dp{world}", "mean_fvu": fvu},
            "roofline": roof,
            "kernels_ms_per_step": {n: v["total_ms"] / args.steps for n, v in kern.items()},
            "byte_bound_kernels": byte_kernels,
            "byte_bound_note": "achieved = algorithmic bytes (SURVEY.md 8d) / live HIP-event time, against the measured "
                               "ceiling of the level that serves the bulk of them (MI355X_MICROARCH.md: random rows out "
                               "of L2 16.8-18.8 TB/s, Infinity Cache 7.4-8.6 TB/s, HBM 8.0 TB/s spec / 6.3 achievable); "
                               "profiled.counter_bytes = HBM-side FETCH_SIZE x2 + WRITE_SIZE of the same kernels in "
                               f"profiles/{prof_name}",
            "extra": {"auxk_ms_per_step": variants.get("auxk", {}).get("ms_per_step"),
                      "auxk_note": "second run (SURVEY.md 8d): auxk_alpha = 1/32, 10 % of the latents dead, k_aux = 2048",
                      "multitopk_ms_per_step": variants.get("multitopk", {}).get("ms_per_step"),
                      "signum_ms_per_step": variants.get("signum", {}).get("ms_per_step"),
                      "signum_note": "the reference's default optimizer (schedule-free SignSGD, third-party package: parity "
                                     "unpinned); the headline runs plain SignSGD, which is pinned to sign_sgd.py",
                      "skewed_ms_per_step": variants.get("skewed", {}).get("ms_per_step"),
                      "skewed_usage": variants.get("skewed", {}).get("usage"),
                      "batch65536_ms_per_step": variants.get("batch65536", {}).get("ms_per_step"),
                      "batch65536_tokens_per_s": (65536e3 / variants["batch65536"]["ms_per_step"]
                                                  if "batch65536" in variants else None),
                      "batch65536_note": "the same step on 65536 tokens (the reference's default batch: 32 sequences of "
                                         "2048 tokens); the weight gradients take the batch in token blocks",
                      "skewed_note": "heavy-tailed latent usage (power-law atoms, coder planted with the dictionary) "
                                     "instead of the flat usage of Gaussian rows; same step otherwise"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, args.cpu_tokens)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def variant_run(kind: str, N, D, M, k, device, steps: int = 4, warm: int = 2) -> dict:
    """The headline step in one of its other forms; returns {"ms_per_step": ..., ...}.
      auxk      the AuxK term (sparse_coder.py:230-251): auxk_alpha = 1/32, about 10 % of the latents held dead,
                k_aux = d_in / 2 (the second run SURVEY.md 8d asks for)
      multitopk the multi-top-k term (sparse_coder.py:256-262): a second decode over the 4k best latents
      signum    the reference's default optimizer: schedule-free SignSGD (trainer.py:146-162; parity unpinned)
      batch65536 the plain step on N = 65536 tokens, the reference's default batch (config.py: batch_size 32 x ctx 2048)
      skewed    heavy-tailed latent usage instead of Gaussian rows' flat one: SyntheticActivations(skew="zipf"),
                the coder planted with the dictionary (trainer.py here); reports the usage histogram too"""
    from sparsify_amd import SparseCoderConfig, TrainConfig
    from sparsify_amd.trainer import SyntheticActivations, Trainer
    cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k, multi_topk=(kind == "multitopk")),
                      optimizer="signum" if kind == "signum" else "signsgd", hookpoints=["h"],
                      auxk_alpha=1 / 32 if kind == "auxk" else 0.0, dead_feature_threshold=10**6, save_every=10**9,
                      wandb_log_frequency=10**9, save_dir="")
    skew = dict(skew="zipf", atoms=M, draws=k) if kind == "skewed" else {}
    src = SyntheticActivations(["h"], D, N, steps + warm, device=device, **skew)
    tr = Trainer(cfg, src, device=device)
    if kind == "skewed":
        src.plant(tr.saes["h"])
    dead = torch.rand(M, device=device, generator=torch.Generator(device=device).manual_seed(0)) < 0.10
    batches = [src.batch(s) for s in range(steps + warm)]
    t0 = 0.0
    for s in range(steps + warm):
        if s == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if kind == "auxk":
            tr.num_tokens_since_fired["h"][dead] = 10**7          # these stay dead unless they fire
        tr.train_micro_batch(batches[s])
    torch.cuda.synchronize()
    out = {"ms_per_step": (time.perf_counter() - t0) / steps * 1e3}
    if kind == "skewed":
        with torch.no_grad():
            idx = tr.saes["h"].encode(batches[-1]["h"]).top_indices
        cnt = torch.bincount(idx.flatten(), minlength=M)
        f = cnt.float() / N
        out["usage"] = {"latents_firing_on_over_10pct_of_tokens": int((f > 0.1).sum()), "over_1pct": int((f > 0.01).sum()),
                        "never_in_this_batch": int((cnt == 0).sum()), "max_share_of_tokens": float(f.max()),
                        "share_of_entries_in_segments_over_128": float(cnt[cnt > 128].sum() / cnt.sum())}
    del tr, batches, src
    torch.cuda.empty_cache()
    return out


def spawn_ranks(n: int) -> int:
    """Runs this script once per rank as child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
    rendezvous on 127.0.0.1) and returns the largest exit code; rank 0's JSON line goes to this
    process's stdout. The parent never initialises the GPU."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    try:
        while any(pr.poll() is None for pr in procs):
            if any(pr.poll() not in (None, 0) for pr in procs):
                break                         # a rank that failed leaves the others waiting in a collective
            time.sleep(0.2)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
                pr.wait()
    return max(abs(pr.returncode) for pr in procs)


def _fp32_forward(fwd, self, *a, **kw):
    with torch.autocast("cuda", enabled=False):
        return fwd(self, *a, **kw)


if __name__ == "__main__":
    main()
