"""Host-side configuration logic (no GPU): validation rules mirrored from the reference's
config.py:119-131 plus the ones of the options this build adds."""
import pytest

from sparsify_amd import SparseCoderConfig, TrainConfig


def test_reference_validation_rules():
    with pytest.raises(ValueError):
        TrainConfig(sae=SparseCoderConfig(), layers=[1, 2], layer_stride=2)            # config.py:121-122
    with pytest.raises(ValueError):
        TrainConfig(sae=SparseCoderConfig(), distribute_modules=True, loss_fn="ce")     # config.py:124-128
    with pytest.raises(ValueError):
        TrainConfig(sae=SparseCoderConfig(), init_seeds=[])                             # config.py:130-131
    cfg = TrainConfig(sae={"k": 16, "num_latents": 64, "transcode": True, "unknown_field": 1})
    assert isinstance(cfg.sae, SparseCoderConfig) and cfg.sae.k == 16 and cfg.sae.transcode


@pytest.mark.parametrize("kw", [dict(distribute_modules=True), dict(loss_fn="kl"),
                                dict(sae=SparseCoderConfig(activation="groupmax"))])
def test_latent_sharding_is_the_topk_fvu_step_only(kw):
    kw.setdefault("sae", SparseCoderConfig())
    with pytest.raises(ValueError):
        TrainConfig(shard_latents=True, **kw)
    TrainConfig(shard_latents=True, sae=SparseCoderConfig())
    TrainConfig(shard_latents=True, auxk_alpha=1 / 32, sae=SparseCoderConfig(multi_topk=True, skip_connection=True))


def test_defaults_of_the_added_switches():
    cfg = TrainConfig(sae=SparseCoderConfig())
    assert cfg.exchange_signs and cfg.overlap_exchange and cfg.fuse_encoder_update and not cfg.shard_latents
    d = cfg.sae.to_dict()
    assert SparseCoderConfig.from_dict({**d, "d_in": 7}, drop_extra_fields=True) == cfg.sae     # cfg.json round trip
