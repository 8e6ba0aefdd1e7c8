"""TEST STUB of the omegaconf surface the reference drivers use (see tests/stubs/README.md)."""
import copy
import dataclasses
import sys

import yaml

MISSING = '???'


class DictConfig(dict):
    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = _wrap(v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = _wrap(v)

    def copy(self):
        return DictConfig(copy.deepcopy(_plain(self)))


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, DictConfig):
        return DictConfig(v)
    return v


def _plain(v):
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    return v


def _merge(a, b):
    out = copy.deepcopy(_plain(a))
    for k, v in _plain(b).items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = _merge(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


class OmegaConf:
    @staticmethod
    def create(d=None):
        return DictConfig(d or {})

    @staticmethod
    def structured(obj):
        return DictConfig(dataclasses.asdict(obj) if dataclasses.is_dataclass(obj) else dict(obj))

    @staticmethod
    def merge(*cfgs):
        out = {}
        for c in cfgs:
            out = _merge(out, c)
        return DictConfig(out)

    @staticmethod
    def from_cli(args=None):
        return OmegaConf.from_dotlist(sys.argv[1:] if args is None else args)

    @staticmethod
    def from_dotlist(items):
        out = {}
        for it in items:
            k, _, v = it.partition('=')
            node = out
            parts = k.split('.')
            for p in parts[:-1]:
                node = node.setdefault(p, {})
            node[parts[-1]] = yaml.safe_load(v)
        return DictConfig(out)

    @staticmethod
    def to_yaml(cfg):
        return yaml.safe_dump(_plain(cfg))

    @staticmethod
    def save(cfg, f):
        with open(f, 'w') as fp:
            fp.write(OmegaConf.to_yaml(cfg))
