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

    def to_dict(self):
        return {k: v.to_dict() if isinstance(v, Config) else copy.deepcopy(v) for k, v in self.items()}


def load_config(config_path):
    """config.py:17-22"""
    with open(config_path) as fp:
        return Config(yaml.load(fp, Loader=yaml.FullLoader))


def augment_arch_defaults(arch_config):
    """config.py:29-49"""
    if arch_config.type == 'rq-vae':
        out = Config({'ema': None})
        out.update(arch_config.copy())
        return out
    if arch_config.type == 'rq-transformer':
        return Config(resolve(arch_config))
    raise NotImplementedError
