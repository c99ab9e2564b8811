"""On-the-fly activation driver: trains the coders on the
activations of a frozen model while it runs, like the reference's `Trainer(cfg, dataset, model)`
(sparsify/trainer.py:28-90 hookpoint resolution and width probing, :333-447 the forward hook,
:453-508 the outer loop; sparsify/utils.py:20-61 helpers).

The language model itself is outside the hot path: it is whatever torch module the caller
brings (its forward runs through torch / rocBLAS); this file only resolves hookpoints, captures
each hooked module's (input, output) as it is produced and hands it to the HIP train step of
`Trainer.train_on_activations`, then runs the step tail. Only the local reconstruction loss
("fvu") is supported; the end-to-end "ce"/"kl" losses need the model's backward (8f rank 3)."""
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
        if cfg.loss_fn != "fvu":
            raise NotImplementedError(f"loss_fn={cfg.loss_fn!r}: only the local 'fvu' loss is on the HIP path")
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
        base = _base(self.model)
        name_to_module = {name: base.get_submodule(name) for name in self.cfg.hookpoints}
        module_to_name = {v: k for k, v in name_to_module.items()}
        k = self.get_current_k()
        for sae in self.saes.values():
            sae.cfg.k = k
        state = {"mask": None, "tokens": 0}

        def hook(module: nn.Module, inputs, outputs):             # trainer.py:340-447
            if isinstance(inputs, tuple):
                inputs = inputs[0]
            if isinstance(outputs, tuple):
                outputs = outputs[0]
            hookpoint = module_to_name[module]
            names = self._names_for(hookpoint) or [hookpoint]     # non-owners still join the gathers
            with torch.enable_grad():
                for name in names:
                    res = self.train_on_activations(name, inputs.detach(), outputs.detach(), state["mask"])
                    if res is not None:
                        state["tokens"] = max(state["tokens"], res[1])

        for x in self._batches():
            x = x.to(self.device)
            mask = torch.isin(x, self.exclude_tokens, invert=True)    # trainer.py:455
            state["mask"], state["tokens"] = (None if bool(mask.all()) else mask), 0
            handles = [mod.register_forward_hook(hook) for mod in name_to_module.values()]
            try:
                with torch.no_grad():
                    self.model(x)
            finally:
                for h in handles:
                    h.remove()
            n_step = state["tokens"]
            if self.cfg.distribute_modules and n_step == 0:
                import torch.distributed as dist
                n_step = int(mask.sum()) * (dist.get_world_size() if dist.is_initialized() else 1)
            self.num_tokens_in_step += n_step
            self.step_tail()
        self.save()
