"""rqvae/utils/config.py:17-49 of the reference (load_config, augment_arch_defaults) without omegaconf /
easydict, which are not installable in the target image: a small attribute-dict (``Config``) provides
the pieces of the OmegaConf node interface the sampling drivers touch -- attribute get/set (including
keys that were null), ``.copy()``, ``.get``, ``**cfg`` expansion, plain python lists for list nodes."""
import copy

import yaml

from ..models.rqtransformer.configs import resolve


class Config(dict):
    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = Config(v) if isinstance(v, dict) and not isinstance(v, Config) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = Config(v) if isinstance(v, dict) and not isinstance(v, Config) else v

    def copy(self):
        return Config(copy.deepcopy(dict(self)))

    def get(self, k, default=None):
        return self[k] if k in self else default

    def to_dict(self):
        return {k: v.to_dict() if isinstance(v, Config) else copy.deepcopy(v) for k, v in self.items()}


def load_config(config_path):
    """config.py:17-22"""
    with open(config_path) as fp:
        return Config(yaml.load(fp, Loader=yaml.FullLoader))


def _to_plain(cfg):
    if hasattr(cfg, 'keys'):
        return {k: _to_plain(cfg[k]) for k in cfg.keys()}
    if isinstance(cfg, (list, tuple)) or type(cfg).__name__ == 'ListConfig':
        return [_to_plain(v) for v in cfg]
    return cfg


def _merge(base, over):
    out = copy.deepcopy(base)
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = _merge(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


# config.py:31-43 of the reference: the defaults merged UNDER a stage-1 arch config
RQVAE_ARCH_DEFAULTS = {
    'ema': None,
    'hparams': {'loss_type': 'l1', 'restart_unused_codes': False, 'use_padding_idx': False, 'masked_dropout': 0.0},
    'checkpointing': False,
}


def is_stage1_arch(arch_type):
    """config.py:25-26"""
    return 'transformer' not in arch_type


def augment_arch_defaults(arch_config):
    """config.py:29-49: OmegaConf.merge(arch_defaults, arch_config) -- the defaults first, the given config on top.
    Accepts this module's Config, a plain dict or an OmegaConf node."""
    if arch_config.type == 'rq-vae':
        return Config(_merge(RQVAE_ARCH_DEFAULTS, _to_plain(arch_config)))
    if arch_config.type == 'rq-transformer':
        return Config(resolve(arch_config))
    raise NotImplementedError
