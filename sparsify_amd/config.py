"""Configuration dataclasses with the reference's field names and defaults
(reference: sparsify/config.py:8-131). Plain dataclasses: the CLI parsing library the
reference uses (simple_parsing) is out of scope; to_dict/from_dict keep the cfg.json
format of sparse_coder.py:143-179 readable and writable."""
from __future__ import annotations

import dataclasses
import json
from dataclasses import dataclass, field
from functools import partial
from typing import Literal


class _Serializable:
    def to_dict(self) -> dict:
        return dataclasses.asdict(self)

    @classmethod
    def from_dict(cls, d: dict, drop_extra_fields: bool = False):
        names = {f.name for f in dataclasses.fields(cls)}
        extra = set(d) - names
        if extra and not drop_extra_fields:
            raise ValueError(f"unknown fields for {cls.__name__}: {sorted(extra)}")
        kw = {k: v for k, v in d.items() if k in names}
        for f in dataclasses.fields(cls):
            if f.name in kw and dataclasses.is_dataclass(f.type if isinstance(f.type, type) else None):
                kw[f.name] = f.type.from_dict(kw[f.name], drop_extra_fields)
        return cls(**kw)

    def save_json(self, path: str) -> None:
        with open(path, "w") as f:
            json.dump(self.to_dict(), f)


@dataclass
class SparseCoderConfig(_Serializable):
    activation: Literal["groupmax", "topk"] = "topk"
    expansion_factor: int = 32
    normalize_decoder: bool = True
    num_latents: int = 0
    k: int = 32
    multi_topk: bool = False
    skip_connection: bool = False
    transcode: bool = False


SaeConfig = SparseCoderConfig
TranscoderConfig = partial(SparseCoderConfig, transcode=True)


@dataclass
class TrainConfig(_Serializable):
    sae: SparseCoderConfig = field(default_factory=SparseCoderConfig)
    batch_size: int = 32
    grad_acc_steps: int = 1
    micro_acc_steps: int = 1
    loss_fn: Literal["ce", "fvu", "kl"] = "fvu"
    optimizer: Literal["adam", "muon", "signum", "signsgd"] = "signum"
    lr: float | None = None
    lr_warmup_steps: int = 1000
    k_decay_steps: int = 0
    auxk_alpha: float = 0.0
    dead_feature_threshold: int = 10_000_000
    exclude_tokens: list[int] = field(default_factory=list)
    hookpoints: list[str] = field(default_factory=list)
    init_seeds: list[int] = field(default_factory=lambda: [0])
    layers: list[int] = field(default_factory=list)
    layer_stride: int = 1
    distribute_modules: bool = False
    exchange_signs: bool = True
    """(not in the reference) "signsgd" optimizer under DDP: exchange the large gradients as
    reduce-scatter + all-gather of 1-byte signs instead of an fp32 all-reduce (same update)."""
    overlap_exchange: bool = True
    """(not in the reference) run that exchange and the updates it feeds on a second stream, under
    the next forward: the encoder GEMM follows the arriving row blocks (same update, same
    results; parallel.wait_for_updates() before reading parameters outside the Trainer)."""
    shard_latents: bool = False
    """(not in the reference; BASELINE.json configs[4]) split every coder's latents over the ranks
    (sharded.ShardedSparseCoder): all ranks see the same tokens, the only exchanges are the top-k
    candidates and the partial reconstructions, gradients and optimizer state stay shard-local.
    The top-k FVU step with AuxK, multi-top-k and the skip connection; no end-to-end loss, no groupmax."""
    fuse_encoder_update: bool = True
    """(not in the reference) "signsgd" on one rank without gradient accumulation: apply the step of
    `encoder.weight` inside its gradient kernel (same update; the (M, D) gradient is never
    materialised, so `encoder.weight.grad` stays None)."""
    accumulate_in_kernel: bool = True
    """(not in the reference) gradient accumulation (`grad_acc_steps * micro_acc_steps > 1`) on one rank: from the
    second micro-step on the weight-gradient kernels add their sums to `encoder.weight.grad` / `W_dec.grad`
    themselves instead of autograd adding a freshly written (M, D) tensor in a pass of its own (the same fp32
    addition, the same bits; tensor hooks on these two parameters do not see those micro-steps)."""
    lazy_pre_acts: bool = False
    """(not in the reference) forward passes whose pre-activations nobody reads (no AuxK, no
    multi-top-k) do not store the (tokens, latents) matrix: top-k from candidate lists the GEMM
    epilogue emits (csrc/lazy.hip). Same selection; 4 GiB less per 16384 tokens, about 3 % more
    forward time at the headline shape."""
    save_every: int = 1000
    save_best: bool = False
    finetune: str | None = None
    log_to_wandb: bool = False
    run_name: str | None = None
    wandb_log_frequency: int = 1
    save_dir: str = "checkpoints"

    def __post_init__(self):
        """Same validation as config.py:119-131."""
        if isinstance(self.sae, dict):
            self.sae = SparseCoderConfig.from_dict(self.sae, drop_extra_fields=True)
        if self.layers and self.layer_stride != 1:
            raise ValueError("Cannot specify both `layers` and `layer_stride`.")
        if self.distribute_modules and self.loss_fn in ("ce", "kl"):
            raise ValueError(
                "Distributing modules across ranks is not compatible with the "
                "cross-entropy or KL divergence losses."
            )
        if self.shard_latents and (self.distribute_modules or self.loss_fn != "fvu" or self.sae.activation != "topk"):
            raise ValueError("`shard_latents` implements the top-k FVU step (with AuxK, multi-top-k and the skip "
                             "connection): no distribute_modules, ce / kl loss or groupmax")
        if not self.init_seeds:
            raise ValueError("Must specify at least one random seed.")
