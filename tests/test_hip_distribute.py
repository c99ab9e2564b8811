"""BASELINE configs[3]: `distribute_modules` (sparsify/trainer.py:358-380, 630-646) end to end on
the GPU. Two ranks on ONE GPU (gloo rendezvous, as tests/test_hip_ddp.py) run
Trainer(distribute_modules=True) over four hookpoints, as SAEs and as transcoders: every rank
feeds its own batch, each hookpoint's owner trains on the batches of all ranks. Checked:

  * every owner's parameters equal, bit for bit, those of a single-rank Trainer fed the
    concatenated batch of that hookpoint (rank order, as the reference's all_gather_into_tensor
    lays them out), and so do its dead-latent counters and logged FVU;
  * that single-rank run follows the CPU oracle's train loop (oracle/sae_oracle.py, pinned to the
    reference) on the same batches, as tests/test_hip_trainer.py::test_trainer_matches_oracle_loop.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sae_oracle as O

pytestmark = pytest.mark.gpu

HOOKS = ["layers.0", "layers.1", "layers.2", "layers.3"]
D, M, K, N, STEPS, LR = 64, 1024, 8, 256, 8, 2e-3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batch(step: int, rank: int, hook_i: int, transcode: bool):
    """Rank `rank`'s activations of hookpoint hook_i at `step` (CPU, fp32): sparse codes over a fixed
    dictionary plus noise and an offset, different for every (step, rank, hookpoint)."""
    g = torch.Generator().manual_seed(77)
    dictionary = torch.randn(128, D, generator=g)
    dictionary /= dictionary.norm(dim=1, keepdim=True)
    off = torch.randn(D, generator=g)
    R = torch.randn(D, D, generator=g) / D**0.5
    g = torch.Generator().manual_seed(100003 * step + 1009 * rank + hook_i)
    codes = torch.rand(N, 128, generator=g) * (torch.rand(N, 128, generator=g) < 0.05)
    x = codes @ dictionary + 0.05 * torch.randn(N, D, generator=g) + off * (1 + hook_i)
    return (x, x @ R) if transcode else x


def _to_gpu(item):
    return tuple(t.cuda() for t in item) if isinstance(item, tuple) else item.cuda()


def _cfg(transcode: bool, distribute: bool, hooks):
    from sparsify_amd import SparseCoderConfig, TrainConfig
    return TrainConfig(sae=SparseCoderConfig(num_latents=M, k=K, transcode=transcode), optimizer="signsgd", lr=LR,
                       hookpoints=list(hooks), distribute_modules=distribute, save_dir="", save_every=10**9,
                       wandb_log_frequency=1)


class _Widths:
    def __init__(self, hooks):
        self.widths = {h: D for h in hooks}

    def __len__(self):
        return STEPS


def _host_init_saes(transcode: bool, hooks):
    """Coders initialised on the host from seed 0, i.e. with the oracle's init_params stream (the
    Trainer itself initialises on the device, like the reference)."""
    from sparsify_amd import SparseCoder, SparseCoderConfig
    out = {}
    for h in hooks:
        torch.manual_seed(0)
        out[h] = SparseCoder(D, SparseCoderConfig(num_latents=M, k=K, transcode=transcode)).cuda()
    return out


def _snapshot(tr):
    torch.cuda.synchronize()
    out = {}
    for name, sae in tr.saes.items():
        out[name] = {n: p.detach().cpu() for n, p in sae.named_parameters()}
        out[name]["counts"] = tr.num_tokens_since_fired[name].cpu()
    return out


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sparsify_amd import parallel
        from sparsify_amd.trainer import Trainer
        # at these sizes the data-parallel sign exchange would not engage by itself: make every
        # matrix eligible, so that a module owner wrongly exchanging with the other rank shows
        parallel.SignExchange.MIN_NUMEL = 1
        res = {}
        for transcode in (False, True):
            probe = Trainer(_cfg(transcode, True, HOOKS), _Widths(HOOKS), device="cuda")
            assert sorted(probe.saes) == HOOKS[2 * rank: 2 * rank + 2]       # trainer.py:637-644: contiguous blocks
            tr = Trainer(_cfg(transcode, True, HOOKS), _Widths(HOOKS), device="cuda",
                         saes=_host_init_saes(transcode, sorted(probe.saes)))
            del probe
            fvus = []
            for step in range(STEPS):
                tr.train_micro_batch({h: _to_gpu(_batch(step, rank, i, transcode)) for i, h in enumerate(HOOKS)})
                fvus.append({h: tr.last_logged[f"fvu/{h}"] for h in tr.saes})
            res[transcode] = (_snapshot(tr), fvus, tr.global_step)
        ret[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.fixture(scope="module")
def two_rank_result():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=150)
    for p in procs:
        if p.is_alive():
            p.kill()
            pytest.fail("a rank did not finish (collective mismatch?)")
        assert p.exitcode == 0
    return {0: ret[0], 1: ret[1]}


@pytest.mark.parametrize("transcode", [False, True])
def test_distribute_modules_equals_single_rank_on_gathered_batch(device, two_rank_result, transcode):
    from sparsify_amd.trainer import Trainer
    for rank in (0, 1):
        snap, fvus, gstep = two_rank_result[rank][transcode]
        assert gstep == STEPS
        mine = HOOKS[2 * rank: 2 * rank + 2]
        tr = Trainer(_cfg(transcode, False, mine), _Widths(mine), device="cuda", saes=_host_init_saes(transcode, mine))
        oracle_P = {h: O.init_params(D, M, transcode=transcode, seed=0) for h in mine}
        ocfg = O.OracleConfig(k=K, transcode=transcode)
        for h in mine:
            assert torch.equal(tr.saes[h].encoder.weight.detach().cpu(), oracle_P[h]["W_enc"])
        for step in range(STEPS):
            batch, cpu_batch = {}, {}
            for h in mine:
                i = HOOKS.index(h)
                parts = [_batch(step, r, i, transcode) for r in (0, 1)]           # rank order = gather order
                cat = tuple(torch.cat(c) for c in zip(*parts)) if transcode else torch.cat(parts)
                cpu_batch[h], batch[h] = cat, _to_gpu(cat)
            tr.train_micro_batch(batch)
            for h in mine:
                got = tr.last_logged[f"fvu/{h}"]
                assert got == fvus[step][h], (h, step)                            # the distributed run, bit for bit
                x, y = cpu_batch[h] if transcode else (cpu_batch[h], None)
                want = float(O.train_step_signsgd(oracle_P[h], x, y, ocfg, LR, step=step, bf16=True)["fvu"])
                tol = 2e-3 if step == 0 else 8e-2          # bf16 tie picks decorrelate the runs slowly
                assert abs(got - want) <= tol * want, (h, step, got, want)
        ref = _snapshot(tr)
        for h in mine:
            for name, t in ref[h].items():
                assert torch.equal(t, snap[h][name]), f"rank {rank} {h} {name}: differs from the single-rank run"
            # 2 N tokens per step went into the counters of the owner
            assert int(ref[h]["counts"].max()) <= STEPS * 2 * N and (ref[h]["counts"] % (2 * N) == 0).all()
