"""Trainer: the reference's per-hookpoint train-step driver (sparsify/trainer.py:340-447, the
hook body) and step tail (:510-538) as a host loop over activation batches.

What is kept from the reference: the TrainConfig fields, step-0 bias init, per-call decoder
re-normalisation, loss = fvu + alpha*auxk + multi_topk/8 divided by acc steps, projected
gradient, optimizer/LR rules, k-decay, dead-latent counters, DDP / distribute_modules
semantics, checkpoint layout. What is not here (out of scope): running
a HuggingFace model to produce the activations -- the Trainer consumes an
``ActivationSource`` (any iterable of {hookpoint: outputs | (inputs, outputs)} batches).

MI355X-first differences, all value-preserving:
  * gradients are all-reduced (RCCL) once per optimizer step instead of DDP's all-reduce on
    every micro-step backward (trainer.py:457-468 has no no_sync());
  * logged scalars stay on the device until they are logged (the reference calls float()
    on three all-reduced scalars per hook call, trainer.py:433-441);
  * with plain SignSGD the projected gradient and the sign step of W_dec are one kernel.
"""
from __future__ import annotations

import math
import os
from collections import defaultdict
from pathlib import Path
from typing import Iterable

import torch
import torch.distributed as dist
from torch import Tensor

from . import ops, parallel
from .adam import Adam
from .config import TrainConfig
from .muon import Muon
from .schedule_free import ScheduleFreeWrapper
from .sign_sgd import SignSGD
from .sparse_coder import SparseCoder


class SyntheticActivations:
    """Seeded synthetic activation batches: x = randn(N, D) + offset(D),
    fresh batch per step (seed 1000 + step); transcoder target y = x @ R + 0.1 randn.

    skew="zipf": the heavy-tailed workload. i.i.d. Gaussian rows make every latent equally likely
    (usage Poisson around N k / M); residual-stream dictionaries are not like that: a few latents fire
    on most tokens, most fire on almost none. Here a token is a sum of `draws` atoms of a fixed
    unit-norm dictionary (`atoms` rows), drawn with replacement from a power law over the atoms
    (P(rank r) ~ r^-zipf_s), coefficients U(1, 2), plus noise and the offset; the atoms' ranks are
    dealt to rows at random, so frequent atoms are spread over the whole index range. `plant(coder)`
    puts the coder in the state training on such data converges to (encoder and decoder rows = the
    atoms), so that the *latent* usage follows the power law from the first step on: with atoms = M =
    131072, draws = k = 192 (measured): about 100 latents fire on more than 10 % of the tokens (the most
    frequent on 40 %: `b_dec` takes the mean out), 1400 on more than 1 %, a fifth never in a batch of
    16384, and a sixth of the (token, slot) entries sit in segments longer than 512."""

    def __init__(self, hookpoints: list[str], d_in: int, tokens_per_batch: int, num_batches: int, *,
                 device="cuda", dtype=torch.bfloat16, transcode: bool = False, rank: int = 0, seed: int = 0,
                 skew: str | None = None, atoms: int = 0, draws: int = 0, zipf_s: float = 1.0, noise: float = 0.02):
        self.hookpoints, self.d_in, self.n, self.num_batches = list(hookpoints), d_in, tokens_per_batch, num_batches
        self.device, self.dtype, self.transcode, self.rank = torch.device(device), dtype, transcode, rank
        g = torch.Generator(device=self.device).manual_seed(seed)
        self.offset = torch.randn(d_in, device=self.device, generator=g)
        self.R = torch.randn(d_in, d_in, device=self.device, generator=g) / d_in**0.5 if transcode else None
        self.widths = {h: d_in for h in self.hookpoints}
        if skew not in (None, "zipf"):
            raise ValueError(f"Unknown skew '{skew}'")
        self.skew, self.noise = skew, noise
        if skew == "zipf":
            if atoms <= 0 or draws <= 0:
                raise ValueError("skew='zipf' needs the number of atoms and of draws per token")
            self.atoms, self.draws = atoms, draws
            self.dictionary = torch.randn(atoms, d_in, device=self.device, generator=g)
            self.dictionary /= self.dictionary.norm(dim=1, keepdim=True)
            p = torch.arange(1, atoms + 1, device=self.device, dtype=torch.float64) ** -zipf_s
            self.cdf = (p.cumsum(0) / p.sum()).float()
            self.row_of_rank = torch.randperm(atoms, device=self.device, generator=g)

    def __len__(self):
        return self.num_batches

    def plant(self, sae) -> None:
        """Encoder and decoder rows of the first `atoms` latents <- the dictionary's atoms (zero biases):
        the dictionary a run on this source converges to."""
        a = min(self.atoms, sae.num_latents)
        with torch.no_grad():
            sae.encoder.weight[:a] = self.dictionary[:a].to(sae.encoder.weight.dtype)
            sae.encoder.bias.zero_()
            sae.W_dec[:a] = self.dictionary[:a].to(sae.W_dec.dtype)
        for p in sae.parameters():
            torch.autograd.graph.increment_version(p)

    def _zipf_rows(self, g) -> Tensor:
        u = torch.rand(self.n, self.draws, device=self.device, generator=g)
        ranks = torch.searchsorted(self.cdf, u).clamp_max(self.atoms - 1)
        coeff = 1.0 + torch.rand(self.n, self.draws, device=self.device, generator=g)
        x = torch.nn.functional.embedding_bag(self.row_of_rank[ranks], self.dictionary, per_sample_weights=coeff, mode="sum")
        return x + self.noise * torch.randn(self.n, self.d_in, device=self.device, generator=g) + self.offset

    def batch(self, step: int) -> dict:
        out = {}
        for i, h in enumerate(self.hookpoints):
            g = torch.Generator(device=self.device).manual_seed((1000 + step) * 9973 + 131 * self.rank + i)
            if self.skew == "zipf":
                x = self._zipf_rows(g)
            else:
                x = torch.randn(self.n, self.d_in, device=self.device, generator=g) + self.offset
            if self.transcode:
                y = x @ self.R + 0.1 * torch.randn(self.n, self.d_in, device=self.device, generator=g)
                out[h] = (x.to(self.dtype), y.to(self.dtype))
            else:
                out[h] = x.to(self.dtype)
        return out

    def __iter__(self):
        for s in range(self.num_batches):
            yield self.batch(s)


class Trainer:
    def __init__(self, cfg: TrainConfig, source, *, device: str | torch.device = "cuda",
                 saes: dict[str, SparseCoder] | None = None):
        """``source``: iterable (with ``widths`` and ``__len__``) of micro-batches, each a dict
        hookpoint -> outputs (N,D) for autoencoders or (inputs, outputs) for transcoders."""
        self.cfg = cfg
        self.source = source
        self.device = torch.device(device)
        if not cfg.hookpoints:
            cfg.hookpoints = list(source.widths)
        cfg.hookpoints = cfg.hookpoints[:: cfg.layer_stride]
        self.distribute_modules()

        widths = source.widths
        if cfg.distribute_modules and len({widths[h] for h in cfg.hookpoints}) > 1:
            raise ValueError("All modules must output tensors of the same shape when using "
                             f"`distribute_modules=True`, got {set(widths.values())}")

        # trainer.py:76-87: one fp32 SAE per local hookpoint (and seed)
        self.saes: dict[str, SparseCoder] = {}
        if saes is not None:
            self.saes = dict(saes)
        else:
            for hook in self.local_hookpoints():
                for seed in cfg.init_seeds:
                    torch.manual_seed(seed)
                    name = f"{hook}/seed{seed}" if len(cfg.init_seeds) > 1 else hook
                    if cfg.shard_latents:
                        from .sharded import ShardedSparseCoder
                        self.saes[name] = ShardedSparseCoder(widths[hook], cfg.sae, self.device, dtype=torch.float32)
                    else:
                        self.saes[name] = SparseCoder(widths[hook], cfg.sae, self.device, dtype=torch.float32)

        if cfg.lazy_pre_acts:
            for sae in self.saes.values():
                sae.lazy_pre_acts = True

        # the source yields one micro-batch per item (the reference counts sequences:
        # len(dataset) // batch_size, trainer.py:90)
        num_batches = max(len(source), 1) if hasattr(source, "__len__") else 1
        self._build_optimizers(num_batches)
        self._register_grad_sinks()

        self.global_step = 0
        self.num_tokens_since_fired = {
            name: torch.zeros(sae.num_latents, device=self.device, dtype=torch.long)
            for name, sae in self.saes.items()
        }
        self.did_fire = {
            name: torch.zeros(sae.num_latents, device=self.device, dtype=torch.bool)
            for name, sae in self.saes.items()
        }
        first = next(iter(self.saes.values()))
        self.initial_k = min(first.num_latents, round(first.d_in * 10))       # trainer.py:176-177
        self.final_k = cfg.sae.k
        self.best_loss = ({name: float("inf") for name in self.saes} if cfg.loss_fn == "fvu"
                          else float("inf"))                                # trainer.py:180-184
        self.num_tokens_in_step = 0
        self._metrics = defaultdict(lambda: torch.zeros((), device=self.device))
        self._pending: list = []
        self._hooks_installed = False
        self.last_logged: dict = {}

    # ------------------------------------------------------------------ optimizers (trainer.py:92-164)
    def _build_optimizers(self, num_batches: int):
        cfg = self.cfg
        self.lr_schedulers = []
        project = cfg.sae.normalize_decoder and not cfg.sae.transcode
        self._fused_projection = False

        def groups(default_lr):
            return [dict(params=list(sae.parameters()), lr=cfg.lr or default_lr(sae)) for sae in self.saes.values()]

        match cfg.optimizer:
            case "adam":
                pgs = groups(lambda s: 2e-4 / (s.num_latents / (2**14)) ** 0.5)
                adam = Adam(pgs)                   # one HIP pass per parameter, bf16 shadow included
                self.optimizers = [adam]
                self.lr_schedulers = [_linear_warmup_decay(adam, cfg.lr_warmup_steps, num_batches)]
            case "muon":
                params = [p for sae in self.saes.values() for p in sae.parameters()]
                muon_params = [p for p in params if p.ndim >= 2]
                rest = [p for p in params if p.ndim < 2]
                self.optimizers = [
                    Muon(muon_params, lr=cfg.lr or 2e-3, ddp=not (cfg.distribute_modules or cfg.shard_latents)),
                    Adam(rest, lr=cfg.lr or 2e-3),
                ]
                self.lr_schedulers = [
                    _linear_warmup_decay(self.optimizers[0], 0, num_batches),
                    _linear_warmup_decay(self.optimizers[1], cfg.lr_warmup_steps, num_batches),
                ]
            case "signum" | "signsgd":
                # SignSGD (sign_sgd.py) with the reference's learning-rate rule (trainer.py:152);
                # decoder matrices get a group of their own whose step fuses the projected
                # gradient (trainer.py:513-515) and the re-normalisation (trainer.py:406-408).
                # "signum" (the reference's default) wraps it schedule-free, "signsgd" is plain.
                pgs = []
                for sae in self.saes.values():
                    lr = cfg.lr or 5e-3 / (sae.num_latents / (2**14)) ** 0.5
                    dec = [sae.W_dec] if (project and sae.W_dec is not None) else []
                    rest = [p for p in sae.parameters() if not any(p is d for d in dec)]
                    if dec:
                        pgs.append(dict(params=dec, lr=lr, project_rows=True, renorm_rows=True))
                    pgs.append(dict(params=rest, lr=lr))
                # shard / module owners hold parameters no other rank has: nothing to exchange
                exchange = cfg.exchange_signs and not cfg.shard_latents and not cfg.distribute_modules
                if cfg.optimizer == "signum":
                    opt = ScheduleFreeWrapper(SignSGD(pgs, exchange_signs=exchange, overlap=cfg.overlap_exchange),
                                              momentum=0.95)
                    opt.train()
                    self.optimizers = [opt]
                    one_rank = not (dist.is_initialized() and dist.get_world_size() > 1)
                    if (cfg.fuse_encoder_update and (one_rank or cfg.distribute_modules) and not cfg.shard_latents
                            and cfg.grad_acc_steps * cfg.micro_acc_steps == 1 and cfg.loss_fn == "fvu"
                            and self.device.type == "cuda"):
                        opt.fuse_encoder_update([sae.encoder.weight for sae in self.saes.values()
                                                 if not opt.reduces(sae.encoder.weight)])
                else:
                    opt = SignSGD(pgs, exchange_signs=exchange, overlap=cfg.overlap_exchange)
                    self.optimizers = [opt]
                    if opt.overlap and cfg.grad_acc_steps * cfg.micro_acc_steps == 1 and cfg.loss_fn == "fvu":
                        # the decoder matrices' gradients are complete before the encoder backward
                        # runs: their exchange starts there (first half of the rows)
                        opt.enable_early_exchange([sae.W_dec for sae in self.saes.values() if sae.W_dec is not None])
                    one_rank = not (dist.is_initialized() and dist.get_world_size() > 1)
                    # (module owners exchange no gradients either: their step can be fused as well)
                    if (cfg.fuse_encoder_update and (one_rank or cfg.distribute_modules) and cfg.grad_acc_steps * cfg.micro_acc_steps == 1
                            and cfg.loss_fn == "fvu" and self.device.type == "cuda"):
                        opt.fuse_encoder_update([sae.encoder.weight for sae in self.saes.values()
                                                 if not opt.reduces(sae.encoder.weight)])
                self._fused_projection = project
            case other:
                raise ValueError(f"Unknown optimizer '{other}'")

    def _register_grad_sinks(self):
        """Gradient accumulation on one rank (or module owners, who exchange nothing): later micro-steps add into
        the dense gradients inside the weight-gradient kernels (ops.register_grad_sinks)."""
        cfg = self.cfg
        one_rank = not (dist.is_initialized() and dist.get_world_size() > 1)
        if (cfg.accumulate_in_kernel and cfg.grad_acc_steps * cfg.micro_acc_steps > 1 and not cfg.shard_latents
                and (one_rank or cfg.distribute_modules) and self.device.type == "cuda"):
            ops.register_grad_sinks([p for sae in self.saes.values() for p in (sae.encoder.weight, sae.W_dec)
                                     if p is not None])

    # ------------------------------------------------------------------ k schedule (trainer.py:229-235)
    def get_current_k(self) -> int:
        if self.global_step >= self.cfg.k_decay_steps:
            return self.final_k
        progress = self.global_step / self.cfg.k_decay_steps
        return round(self.initial_k * (1 - progress) + self.final_k * progress)

    # ------------------------------------------------------------------ distributed helpers
    @property
    def ddp(self) -> bool:
        # latent shards hold different rows of one dictionary: nothing to broadcast or average
        return (dist.is_initialized() and dist.get_world_size() > 1 and not self.cfg.distribute_modules
                and not self.cfg.shard_latents)

    def maybe_all_reduce(self, x: Tensor, op: str = "mean") -> Tensor:
        """trainer.py:614-628"""
        if not dist.is_initialized() or self.cfg.distribute_modules or self.cfg.shard_latents:
            return x        # (latent shards: every rank sees the same tokens, the values already agree)
        return parallel.all_reduce_(x, op)

    def distribute_modules(self):
        """trainer.py:630-646: contiguous blocks of hookpoints per rank."""
        if not self.cfg.distribute_modules:
            self.module_plan = []
            return
        self.module_plan = parallel.module_plan(self.cfg.hookpoints, parallel.world_size())

    def local_hookpoints(self) -> list[str]:
        if self.module_plan:
            rank = dist.get_rank() if dist.is_initialized() else 0
            return self.module_plan[rank]
        return self.cfg.hookpoints

    def _install_grad_hooks(self):
        """See parallel.GradReducer: one gradient average per optimizer step."""
        if self._hooks_installed or not self.ddp:
            return
        params = [p for sae in self.saes.values() for p in sae.parameters()]
        parallel.broadcast_parameters(params)
        # parameters whose gradients the optimizer exchanges itself (SignSGD sign exchange)
        own = [opt.reduces for opt in self.optimizers if hasattr(opt, "reduces")]
        self._reducer = parallel.GradReducer([p for p in params if not any(f(p) for f in own)])
        self._hooks_installed = True

    def _arm_grad_reduce(self):
        """Last micro-step of an optimizer step: the large gradients may leave for their all-reduce
        from inside this backward (parallel.GradReducer). One backward per hookpoint and step: with
        several coders the hooks of each fire in its own backward."""
        red = getattr(self, "_reducer", None)
        if red is not None and (self.global_step + 1) % self.cfg.grad_acc_steps == 0:
            red.armed = True

    def _finish_grad_reduce(self):
        if getattr(self, "_reducer", None) is not None:
            self._reducer.finish()

    # ------------------------------------------------------------------ the hook body (trainer.py:340-447)
    def train_on_activations(self, name: str, inputs: Tensor | None, outputs: Tensor, mask: Tensor | None = None,
                             e2e: bool = False):
        """e2e=True (loss_fn 'ce' / 'kl', trainer.py:425-430): forward and bookkeeping only; the
        caller splices out.sae_out into the model and runs the backward itself."""
        cfg = self.cfg
        hook_name = name.partition("/")[0]

        if cfg.distribute_modules and parallel.world_size() > 1:
            # trainer.py:358-380: every rank contributes its batch, only the owner trains. The
            # reference all-gathers to every rank and the non-owners drop what they received; here
            # the batches go to the owner only (same rows in the same rank order on the owner)
            owner = parallel.module_owner(self.module_plan, hook_name)
            outputs = parallel.gather_rows(outputs, owner)
            if cfg.sae.transcode:
                inputs = parallel.gather_rows(inputs, owner)
            if mask is not None:
                mask = parallel.gather_rows(mask, owner)
            if owner != dist.get_rank():
                return None

        outputs = outputs.reshape(-1, outputs.shape[-1])
        inputs = inputs.reshape(-1, inputs.shape[-1]) if cfg.sae.transcode else outputs
        if mask is not None:
            mask = mask.reshape(-1)
            outputs = outputs[mask]
            inputs = inputs[mask] if cfg.sae.transcode else outputs
        n_tokens = outputs.shape[0]

        raw = self.saes[name]
        if self.global_step == 0 and not cfg.finetune:
            # trainer.py:394-404: centre the pre-activations / initialise b_dec with the mean
            if cfg.sae.transcode:
                mean = self.maybe_all_reduce(ops.colsum(inputs.contiguous(), 1.0 / inputs.shape[0])).to(raw.dtype)
                raw.encoder.bias.data = -mean @ raw.encoder.weight.data.T
            mean = self.maybe_all_reduce(ops.colsum(outputs.contiguous(), 1.0 / outputs.shape[0]))
            raw.b_dec.data = mean.to(raw.dtype)

        if raw.cfg.normalize_decoder and not cfg.sae.transcode:
            # trainer.py:408. The fused SignSGD step already left the rows unit-norm (same
            # arithmetic, done while the updated rows were in registers): consume that once.
            done = getattr(self.optimizers[0], "renormalised", None)
            if done is not None and id(raw.W_dec) in done:
                done.discard(id(raw.W_dec))
            else:
                raw.set_decoder_norm_to_unit_norm()

        dead_mask = num_dead = None
        if cfg.auxk_alpha > 0:
            dead_mask, n_dead = ops.dead_mask(self.num_tokens_since_fired[name], cfg.dead_feature_threshold)
            # the AuxK branch needs the count on the host (sparse_coder.py:231, k_aux is a shape):
            # read it here, before the forward is enqueued, where the GEMM covers the host's catching
            # up -- inside the forward the same wait leaves the GPU idle while the backward is enqueued
            num_dead = int(n_dead)

        self._install_grad_hooks()
        out = (raw(x=inputs, y=outputs, dead_mask=dead_mask, num_dead=num_dead) if num_dead is not None
               else raw(x=inputs, y=outputs, dead_mask=dead_mask))

        ops.mark_fired_(self.did_fire[name], out.latent_indices.reshape(-1))      # trainer.py:422
        self.maybe_all_reduce(self.did_fire[name], "max")

        if e2e:
            return out, n_tokens

        denom = cfg.grad_acc_steps * cfg.micro_acc_steps * cfg.wandb_log_frequency
        self._metrics[f"fvu/{name}"] += out.fvu.detach() / denom
        if cfg.auxk_alpha > 0:
            self._metrics[f"auxk/{name}"] += out.auxk_loss.detach() / denom
        if cfg.sae.multi_topk:
            self._metrics[f"multi_topk_fvu/{name}"] += out.multi_topk_fvu.detach() / denom

        acc_steps = cfg.grad_acc_steps * cfg.micro_acc_steps
        # trainer.py:444-447: (fvu + alpha * auxk + multi_topk_fvu / 8) / acc_steps. Terms that are exact zeros and a
        # division by one are left out: the same value and the same gradients without their dozen scalar kernels
        loss = out.fvu
        if cfg.auxk_alpha > 0:
            loss = loss + cfg.auxk_alpha * out.auxk_loss
        if cfg.sae.multi_topk:
            loss = loss + out.multi_topk_fvu / 8
        if acc_steps > 1:
            loss = loss.div(acc_steps)
        self._arm_grad_reduce()
        loss.backward()
        return out, n_tokens

    # ------------------------------------------------------------------ the step tail (trainer.py:510-538)
    def step_tail(self) -> bool:
        cfg = self.cfg
        step, substep = divmod(self.global_step + 1, cfg.grad_acc_steps)
        did_step = False
        if substep == 0:
            self._finish_grad_reduce()
            if cfg.shard_latents:
                for sae in self.saes.values():
                    sae.reduce_shared_grads()          # b_dec: the one parameter the shards share
            if cfg.sae.normalize_decoder and not cfg.sae.transcode and not self._fused_projection:
                for sae in self.saes.values():
                    sae.remove_gradient_parallel_to_decoder_directions()
            for optimizer in self.optimizers:
                optimizer.step()
                optimizer.zero_grad()
            for scheduler in self.lr_schedulers:
                scheduler.step()
            k = self.get_current_k()
            for sae in self.saes.values():
                sae.cfg.k = k
            for name, counts in self.num_tokens_since_fired.items():
                ops.dead_counter_step_(counts, self.did_fire[name], self.num_tokens_in_step)
            self.num_tokens_in_step = 0
            did_step = True
            if (step + 1) % cfg.wandb_log_frequency == 0:
                self._flush_metrics(step)
            if (step + 1) % cfg.save_every == 0:
                self.save()
                if cfg.save_best:                                           # trainer.py:543-544
                    self.save_best(self._avg_losses())
        self.global_step += 1
        return did_step

    def _flush_metrics(self, step: int):
        if not self._metrics:
            return
        keys = sorted(self._metrics)
        vec = torch.stack([self._metrics[k] for k in keys])
        self.maybe_all_reduce(vec, "mean")
        vals = vec.tolist()                     # the only host sync of the logging path
        info = dict(zip(keys, vals))
        for name in self.saes:
            dead = self.num_tokens_since_fired[name] > self.cfg.dead_feature_threshold
            info[f"dead_pct/{name}"] = dead.float().mean().item()
        info["k"] = self.get_current_k()
        info["step"] = step
        self.last_logged = info
        self._metrics.clear()

    def _avg_losses(self) -> float | dict[str, float]:
        """Losses of the last logged window: per-coder FVU, or the one ce / kl scalar (the
        reference's `avg_losses`, trainer.py:501-506); a missing value never beats the best."""
        if self.cfg.loss_fn != "fvu":
            return self.last_logged.get(f"{self.cfg.loss_fn}_loss", float("inf"))
        return {name: self.last_logged.get(f"fvu/{name}", float("inf")) for name in self.saes}

    def train_micro_batch(self, batch: dict, mask: Tensor | None = None):
        """One iteration of the reference's outer loop (trainer.py:453-589) for a batch of
        ready-made activations."""
        n_step = 0
        plan = self.local_hookpoints() if not self.cfg.distribute_modules else self.cfg.hookpoints
        for hook in plan:
            item = batch[hook]
            inputs, outputs = item if isinstance(item, tuple) else (None, item)
            for name in [n for n in self._names_for(hook)] or [hook]:
                res = self.train_on_activations(name, inputs, outputs, mask)
                if res is not None:
                    n_step = max(n_step, res[1])
        if self.cfg.distribute_modules and n_step == 0:
            first = batch[self.cfg.hookpoints[0]]
            first = first[1] if isinstance(first, tuple) else first
            world = dist.get_world_size() if dist.is_initialized() else 1
            n_step = first.reshape(-1, first.shape[-1]).shape[0] * world
        self.num_tokens_in_step += n_step
        return self.step_tail()

    def _names_for(self, hook: str):
        if hook in self.saes:
            return [hook]
        return [n for n in self.saes if n.partition("/")[0] == hook]

    def fit(self):
        k = self.get_current_k()
        for sae in self.saes.values():
            sae.cfg.k = k
        for i, batch in enumerate(self.source):
            if i < self.global_step:
                continue                       # resume: skip what was already consumed
            self.train_micro_batch(batch)
        parallel.wait_for_updates()
        self.save()
        if self.cfg.save_best:                                              # trainer.py:592-594
            self.save_best(self._avg_losses())

    # ------------------------------------------------------------------ checkpoints (trainer.py:186-227, 648-724)
    def _checkpoint(self, saes: dict[str, SparseCoder], path: str, rank_zero: bool):
        parallel.wait_for_updates()
        for optimizer in self.optimizers:
            if isinstance(optimizer, ScheduleFreeWrapper):
                optimizer.eval()
        for name, sae in saes.items():
            sae.save_to_disk(f"{path}/{name}")
        if rank_zero:
            for i, scheduler in enumerate(self.lr_schedulers):
                torch.save(scheduler.state_dict(), f"{path}/lr_scheduler_{i}.pt")
            for i, optimizer in enumerate(self.optimizers):
                state = getattr(self, "_opt_states", {}).get(i)
                torch.save(state if state is not None else optimizer.state_dict(), f"{path}/optimizer_{i}.pt")
            torch.save({"global_step": self.global_step}, f"{path}/state.pt")
            self.cfg.save_json(f"{path}/config.json")
        for optimizer in self.optimizers:
            if isinstance(optimizer, ScheduleFreeWrapper):
                optimizer.train()
        rank = 0 if rank_zero else dist.get_rank()
        torch.save({"num_tokens_since_fired": self.num_tokens_since_fired, "best_loss": self.best_loss},
                   f"{path}/rank_{rank}_state.pt")

    def _collect_optimizer_states(self):
        """Optimizers whose state is split over the ranks (row-sharded Muon) gather it in state_dict():
        a collective, so every rank calls it, whoever writes the file (rank 0, trainer.py:663-668)."""
        self._opt_states = {}
        if self.cfg.save_dir and self.ddp:
            parallel.wait_for_updates()
            for i, optimizer in enumerate(self.optimizers):
                if getattr(optimizer, "collective_state_dict", False):
                    self._opt_states[i] = optimizer.state_dict()

    def save(self):
        path = f'{self.cfg.save_dir}/{self.cfg.run_name or "unnamed"}'
        rank_zero = not dist.is_initialized() or dist.get_rank() == 0
        self._collect_optimizer_states()
        if self.cfg.save_dir and (rank_zero or self.cfg.distribute_modules or self.cfg.shard_latents):
            Path(path).mkdir(parents=True, exist_ok=True)
            self._checkpoint(self.saes, path, rank_zero)
        if dist.is_initialized():
            dist.barrier()

    def save_best(self, avg_loss: float | dict[str, float]):
        """trainer.py:701-724: keep, per coder (fvu) or for the whole set (ce / kl: one scalar),
        the checkpoint with the lowest loss under {run}/best."""
        base_path = f'{self.cfg.save_dir}/{self.cfg.run_name or "unnamed"}/best'
        rank_zero = not dist.is_initialized() or dist.get_rank() == 0
        # (a sharded coder's save is a collective: every shard takes part, as in save())
        writer = bool(self.cfg.save_dir) and (rank_zero or self.cfg.distribute_modules or self.cfg.shard_latents)
        if isinstance(avg_loss, dict):
            for name in self.saes:
                if avg_loss[name] < self.best_loss[name]:
                    self.best_loss[name] = avg_loss[name]
                    self._collect_optimizer_states()          # (the averaged losses agree on all ranks)
                    if writer:
                        Path(f"{base_path}/{name}").mkdir(parents=True, exist_ok=True)
                        self._checkpoint({name: self.saes[name]}, f"{base_path}/{name}", rank_zero)
        elif avg_loss < self.best_loss:
            self.best_loss = avg_loss
            self._collect_optimizer_states()
            if writer:
                Path(base_path).mkdir(parents=True, exist_ok=True)
                self._checkpoint(self.saes, base_path, rank_zero)
        if dist.is_initialized():
            dist.barrier()

    def load_state(self, path: str):
        from glob import glob

        from safetensors.torch import load_model

        device = self.device
        parallel.wait_for_updates()
        train_state = torch.load(f"{path}/state.pt", map_location=device, weights_only=True)
        self.global_step = train_state["global_step"]
        for file in glob(f"{path}/rank_*_state.pt"):
            rank_state = torch.load(file, map_location=device, weights_only=True)
            for k in self.saes:
                if k in rank_state["num_tokens_since_fired"]:
                    self.num_tokens_since_fired[k] = rank_state["num_tokens_since_fired"][k]
                if not isinstance(rank_state["best_loss"], dict):
                    self.best_loss = rank_state["best_loss"]
                elif k in rank_state["best_loss"]:
                    self.best_loss[k] = rank_state["best_loss"][k]
        for i, scheduler in enumerate(self.lr_schedulers):
            scheduler.load_state_dict(torch.load(f"{path}/lr_scheduler_{i}.pt", map_location=device, weights_only=True))
        for i, optimizer in enumerate(self.optimizers):
            optimizer.load_state_dict(torch.load(f"{path}/optimizer_{i}.pt", map_location=device, weights_only=True))
        for name, sae in self.saes.items():
            load_model(sae, f"{path}/{name}/sae.safetensors", device=str(device))
            for p in sae.parameters():
                torch.autograd.graph.increment_version(p)
        # a checkpoint holds the averaged iterate x and train_mode = False (the save ran between
        # .eval() and .train(), trainer.py:652-654, 675-677): put the parameters back on y
        for optimizer in self.optimizers:
            if isinstance(optimizer, ScheduleFreeWrapper):
                optimizer.train()


def _linear_warmup_decay(optimizer, warmup: int, total: int):
    """transformers.get_linear_schedule_with_warmup (trainer.py:118-122, 140-145)."""

    def f(step: int) -> float:
        if step < warmup:
            return step / max(1, warmup)
        return max(0.0, (total - step) / max(1, total - warmup))

    return torch.optim.lr_scheduler.LambdaLR(optimizer, f)


# Support old name for compatibility
SaeTrainer = Trainer
