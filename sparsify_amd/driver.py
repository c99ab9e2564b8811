"""On-the-fly activation driver: trains the coders on the
activations of a frozen model while it runs, like the reference's `Trainer(cfg, dataset, model)`
(sparsify/trainer.py:28-90 hookpoint resolution and width probing, :333-447 the forward hook,
:453-508 the outer loop; sparsify/utils.py:20-61 helpers).

The language model itself is outside the hot path: it is whatever torch module the caller
brings (its forward runs through torch / rocBLAS); this file only resolves hookpoints, captures
each hooked module's (input, output) as it is produced and hands it to the HIP train step of
`Trainer.train_on_activations`, then runs the step tail. With loss_fn "fvu" every coder is
trained on its local reconstruction loss inside the hook; with "ce" / "kl" (8f rank 3,
trainer.py:318-331, 425-430, 486-500) the hook splices the reconstruction into the model, the
model's own loss is back-propagated through the frozen model into the coders, and the encoder's
grad-input path (fused_encoder.py:58-64) is exercised."""
from __future__ import annotations

from fnmatch import fnmatchcase

import torch
from torch import Tensor, nn
from torch.utils.data import DataLoader

from .config import TrainConfig
from .trainer import Trainer


def _natural_key(name: str):
    """natsort.natsorted ordering for dotted module names (layers.2 before layers.10)."""
    import re
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", name)]


def _base(model: nn.Module) -> nn.Module:
    return getattr(model, "base_model", model)


def get_layer_list(model: nn.Module) -> tuple[str, nn.ModuleList]:
    """utils.py:20-30: the ModuleList holding the model's num_hidden_layers blocks."""
    n = model.config.num_hidden_layers
    if not isinstance(n, int):
        raise TypeError(f"Expected int, got {type(n).__name__}")
    cands = [(name, mod) for name, mod in _base(model).named_modules() if isinstance(mod, nn.ModuleList) and len(mod) == n]
    assert len(cands) == 1, "Could not find the list of layers."
    return cands[0]


def _model_device(model: nn.Module) -> torch.device:
    dev = getattr(model, "device", None)
    return torch.device(dev) if dev is not None else next(model.parameters()).device


@torch.inference_mode()
def resolve_widths(model: nn.Module, module_names: list[str], dim: int = -1, dummy: dict | None = None) -> dict[str, int]:
    """utils.py:33-61: output width of each hooked module, probed with the model's dummy inputs."""
    module_to_name = {_base(model).get_submodule(name): name for name in module_names}
    shapes: dict[str, int] = {}

    def hook(module, _, output):
        if isinstance(output, tuple):
            output, *_ = output
        shapes[module_to_name[module]] = output.shape[dim]

    handles = [mod.register_forward_hook(hook) for mod in module_to_name]
    if dummy is None:
        dummy = model.dummy_inputs
    dev = _model_device(model)
    dummy = {k: (v.to(dev) if isinstance(v, Tensor) else v) for k, v in dummy.items()}
    try:
        model(**dummy)
    finally:
        for h in handles:
            h.remove()
    return shapes


def resolve_hookpoints(model: nn.Module, cfg: TrainConfig) -> list[str]:
    """trainer.py:38-59: wildcard patterns over base_model.named_modules() in natural order, or
    `layers` -> "<layer list>.<i>" (all layers when neither is given), then layer_stride."""
    if cfg.hookpoints:
        assert not cfg.layers, "Cannot specify both `hookpoints` and `layers`."
        raw = [name for name, _ in _base(model).named_modules() if any(fnmatchcase(name, pat) for pat in cfg.hookpoints)]
        points = sorted(raw, key=_natural_key)
    else:
        if not cfg.layers:
            cfg.layers = list(range(model.config.num_hidden_layers))
        layers_name, _ = get_layer_list(model)
        points = [f"{layers_name}.{i}" for i in cfg.layers]
    return points[:: cfg.layer_stride]


class _DatasetInfo:
    """What Trainer.__init__ needs from its source: widths per hookpoint and a batch count."""

    def __init__(self, widths: dict[str, int], num_batches: int):
        self.widths, self.num_batches = widths, num_batches

    def __len__(self):
        return self.num_batches


class ModelTrainer(Trainer):
    """`Trainer(cfg, dataset, model)` of the reference. `dataset`: a sized, indexable collection
    of {"input_ids": LongTensor(S)} items (HF dataset with torch format, MemmapDataset-like) or
    a (num_sequences, S) LongTensor."""

    def __init__(self, cfg: TrainConfig, dataset, model: nn.Module):
        if cfg.loss_fn not in ("fvu", "ce", "kl"):
            raise ValueError(f"Unknown loss function '{cfg.loss_fn}'")
        if cfg.loss_fn != "fvu" and (len(cfg.init_seeds) > 1 or cfg.distribute_modules):
            raise ValueError("end-to-end losses need one coder per hookpoint on every rank")
        self.model = model
        self.dataset = dataset
        points = resolve_hookpoints(model, cfg)
        if not points:
            raise ValueError("No module matches the requested hookpoints")
        cfg.hookpoints = points
        widths = resolve_widths(model, points)
        stride, cfg.layer_stride = cfg.layer_stride, 1            # already applied above
        try:
            super().__init__(cfg, _DatasetInfo(widths, len(dataset) // cfg.batch_size), device=_model_device(model))
        finally:
            cfg.layer_stride = stride
        self.exclude_tokens = torch.tensor(cfg.exclude_tokens, device=self.device, dtype=torch.long)   # trainer.py:172

    def _batches(self):
        ds = self.dataset
        if self.global_step > 0:                                  # resume, trainer.py:272-276
            n = self.global_step * self.cfg.batch_size
            ds = ds.select(range(n, len(ds))) if hasattr(ds, "select") else ds[n:]
        if isinstance(ds, Tensor):
            for i in range(0, ds.shape[0] - self.cfg.batch_size + 1, self.cfg.batch_size):
                yield ds[i: i + self.cfg.batch_size]
        else:
            for batch in DataLoader(ds, batch_size=self.cfg.batch_size, shuffle=False, drop_last=True):
                yield batch["input_ids"]

    def fit(self):
        self.model.requires_grad_(False)                          # trainer.py:242
        cfg = self.cfg
        e2e = cfg.loss_fn in ("ce", "kl")
        base = _base(self.model)
        batches = self._batches()
        first = None
        if cfg.loss_fn == "ce":
            # trainer.py:318-331: report the clean loss; end-to-end transcoders replace the modules
            first = next(batches, None)
            if first is not None:
                with torch.no_grad():
                    x0 = first.to(self.device)
                    self.clean_loss = float(self.maybe_all_reduce(self.model(x0, labels=x0).loss.detach().clone()))
            if cfg.sae.transcode:
                for point in cfg.hookpoints:
                    set_submodule(base, point, nn.Identity())
        name_to_module = {name: base.get_submodule(name) for name in cfg.hookpoints}
        module_to_name = {v: k for k, v in name_to_module.items()}
        k = self.get_current_k()
        for sae in self.saes.values():
            sae.cfg.k = k
        state = {"mask": None, "tokens": 0}
        acc_steps = cfg.grad_acc_steps * cfg.micro_acc_steps
        denom = acc_steps * cfg.wandb_log_frequency

        def hook(module: nn.Module, inputs, outputs):             # trainer.py:340-447
            aux_out = None
            if isinstance(inputs, tuple):
                inputs = inputs[0]
            if isinstance(outputs, tuple):
                outputs, *aux_out = outputs
            hookpoint = module_to_name[module]
            if e2e:
                out_shape = outputs.shape
                out, n = self.train_on_activations(hookpoint, inputs, outputs, state["mask"], e2e=True)
                state["tokens"] = max(state["tokens"], n)
                # trainer.py:425-430: the coder's output replaces the module's on the trained tokens
                if state["mask"] is None:
                    output = out.sae_out.type_as(outputs)
                else:
                    output = outputs.detach().flatten(0, 1).clone()
                    output[state["mask"].reshape(-1)] = out.sae_out.type_as(output)
                output = output.reshape(out_shape)
                return (output, *aux_out) if aux_out is not None else output
            names = self._names_for(hookpoint) or [hookpoint]     # non-owners still join the gathers
            with torch.enable_grad():
                for name in names:
                    res = self.train_on_activations(name, inputs.detach(), outputs.detach(), state["mask"])
                    if res is not None:
                        state["tokens"] = max(state["tokens"], res[1])

        def all_batches():
            if first is not None:
                yield first
            yield from batches

        for x in all_batches():
            x = x.to(self.device)
            mask = torch.isin(x, self.exclude_tokens, invert=True)    # trainer.py:455
            state["mask"], state["tokens"] = (None if bool(mask.all()) else mask), 0
            clean_probs = None
            if cfg.loss_fn == "kl":                                   # trainer.py:474-479
                with torch.no_grad():
                    clean_probs = self.model(x).logits.softmax(dim=-1)
            handles = [mod.register_forward_hook(hook) for mod in name_to_module.values()]
            try:
                if cfg.loss_fn == "fvu":
                    with torch.no_grad():
                        self.model(x)
                elif cfg.loss_fn == "ce":                             # trainer.py:487-493
                    ce = self.model(x, labels=x).loss
                    ce.div(acc_steps).backward()
                    self._metrics["ce_loss"] += ce.detach() / denom           # averaged over ranks when flushed
                else:                                                 # trainer.py:494-500
                    dirty_lps = self.model(x).logits.log_softmax(dim=-1)
                    kl = -torch.sum(clean_probs * dirty_lps, dim=-1).mean()
                    kl.div(acc_steps).backward()
                    self._metrics["kl_loss"] += kl.detach() / denom
            finally:
                for h in handles:
                    h.remove()
            n_step = state["tokens"]
            if cfg.distribute_modules and n_step == 0:
                import torch.distributed as dist
                n_step = int(mask.sum()) * (dist.get_world_size() if dist.is_initialized() else 1)
            self.num_tokens_in_step += n_step
            self.step_tail()
        from . import parallel
        parallel.wait_for_updates()
        self.save()
        if cfg.save_best:                                            # trainer.py:592-594
            self.save_best(self._avg_losses())


def set_submodule(model: nn.Module, submodule_path: str, new_submodule: nn.Module):
    """utils.py:64-78: replace a submodule by its dotted path."""
    parent_path, _, last = submodule_path.rpartition(".")
    parent = model.get_submodule(parent_path) if parent_path else model
    setattr(parent, last, new_submodule)
