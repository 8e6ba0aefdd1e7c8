"""TEST STUB (tests/stubs/README.md): import-time surface of torchvision for the reference's drivers / metrics / datasets.
Any attribute of the sub-modules resolves to a placeholder class (so `from torchvision.datasets import CocoCaptions` and
`class FIDInceptionA(torchvision.models.inception.InceptionA)` import); instantiating a placeholder raises."""
import sys
import types

import torch

__version__ = '0.0-stub'


class _Lazy(types.ModuleType):
    _base = object

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        base = self._base

        def _init(self, *a, **k):
            raise RuntimeError(f'torchvision stub: {name} is not available in the test image')
        cls = type(name, (base,), {'__init__': _init})
        setattr(self, name, cls)
        return cls


def _mod(name, base=object):
    m = _Lazy(name)
    m._base = base
    m.__path__ = []
    sys.modules[name] = m
    return m


utils = _mod('torchvision.utils')
utils.make_grid = lambda t, nrow=8, **kw: (t[0] if t.dim() == 4 else t)
utils.save_image = lambda *a, **k: None
transforms = _mod('torchvision.transforms')
transforms.functional = _mod('torchvision.transforms.functional')
datasets = _mod('torchvision.datasets', torch.utils.data.Dataset)
models = _mod('torchvision.models', torch.nn.Module)
models.inception = _mod('torchvision.models.inception', torch.nn.Module)
