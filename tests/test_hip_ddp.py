"""Data-parallel Trainer with two ranks on ONE GPU (gloo rendezvous, device tensors staged
through the host): the N > 1 code of bench.py / Trainer runs end to end with the HIP kernels,
replicas stay identical, and the sign exchange (reduce-scatter + all-gather of int8 signs)
gives bit for bit the parameters of the plain gradient all-reduce. The RCCL backend itself
only runs on a multi-GPU node (the driver's scaling bench)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sparsify_amd import SparseCoderConfig, TrainConfig
        from sparsify_amd.trainer import SyntheticActivations, Trainer
        D, M, k, N, steps = 256, 4096, 16, 512, 4
        out = {}
        for exchange in (False, True):
            cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k), optimizer="signsgd", hookpoints=["h0"],
                              save_dir="", save_every=10**9, wandb_log_frequency=1, exchange_signs=exchange)
            src = SyntheticActivations(["h0"], D, N, steps, device="cuda", rank=rank)
            torch.manual_seed(0)
            tr = Trainer(cfg, src, device="cuda")
            for s in range(steps):
                tr.train_micro_batch(src.batch(s))
            sae = tr.saes["h0"]
            torch.cuda.synchronize()
            out[exchange] = {n: p.detach().cpu() for n, p in sae.named_parameters()}
            out[exchange]["fvu"] = tr.last_logged["fvu/h0"]
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def test_two_rank_trainer_sign_exchange_equals_all_reduce(device):
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=90)
    for p in procs:
        if p.is_alive():
            p.kill()
            pytest.fail("a rank did not finish (collective mismatch?)")
        assert p.exitcode == 0
    r0, r1 = ret[0], ret[1]
    for mode in (False, True):
        for name in r0[mode]:
            if name == "fvu":
                assert r0[mode]["fvu"] == pytest.approx(r1[mode]["fvu"], rel=1e-6)     # logged mean over ranks
            else:
                assert torch.equal(r0[mode][name], r1[mode][name]), f"replicas diverged: {name} (exchange={mode})"
    # two ranks: a + b is order independent, so both exchanges see the same mean gradient
    for name in r0[False]:
        if name != "fvu":
            assert torch.equal(r0[False][name], r0[True][name]), f"sign exchange != all-reduce: {name}"
    assert r0[True]["fvu"] < 1.0
