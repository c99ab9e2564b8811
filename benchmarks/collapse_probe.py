"""BASELINE configs[4] on the bench's own synthetic transcoder inputs, at a reduced shape: the HIP
Trainer next to the CPU oracle's train loop (oracle/sae_oracle.py, pinned to the reference), step
by step: FVU and the number of distinct latents that fired. Answers whether the FVU > 1 /
collapsing dictionary of the synthetic 262144-latent run is what the reference does on these inputs
(zero-initialised decoder + sign steps) or a defect of the HIP path.

    python benchmarks/collapse_probe.py [D M k N steps]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sae_oracle as O  # noqa: E402
from sparsify_amd import SparseCoderConfig, TrainConfig  # noqa: E402
from sparsify_amd.trainer import SyntheticActivations, Trainer  # noqa: E402


def run(D=256, M=16384, k=16, N=2048, steps=20, transcode=True, verbose=True):
    cfg = TrainConfig(sae=SparseCoderConfig(num_latents=M, k=k, transcode=transcode), optimizer="signsgd",
                      hookpoints=["h"], save_dir="", save_every=10**9, wandb_log_frequency=1)
    src = SyntheticActivations(["h"], D, N, steps, device="cuda", transcode=transcode)
    from sparsify_amd import SparseCoder
    torch.manual_seed(0)                 # host-side init = the oracle's init_params stream
    sae0 = SparseCoder(D, cfg.sae).cuda()
    tr = Trainer(cfg, src, device="cuda", saes={"h": sae0})
    P = O.init_params(D, M, transcode=transcode, seed=0)
    assert torch.equal(tr.saes["h"].encoder.weight.detach().cpu(), P["W_enc"])
    ocfg = O.OracleConfig(k=k, transcode=transcode)
    lr = O.signum_lr(M)
    rows = []
    for step in range(steps):
        b = src.batch(step)["h"]
        x, y = (b[0].cpu(), b[1].cpu()) if transcode else (b.cpu(), None)
        want = O.train_step_signsgd(P, x, y, ocfg, lr, step=step, bf16=True)
        tr.train_micro_batch({"h": b})
        got_fvu = tr.last_logged["fvu/h"]
        got_active = int((tr.num_tokens_since_fired["h"] == 0).sum())
        want_active = int(want["latent_indices"].unique().numel())
        rows.append((step, got_fvu, float(want["fvu"]), got_active, want_active))
        if verbose:
            print(f"step {step:3d}  fvu hip {got_fvu:9.5f} oracle {float(want['fvu']):9.5f}   "
                  f"active latents hip {got_active:6d} oracle {want_active:6d}", flush=True)
    return rows


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    run(*a)
