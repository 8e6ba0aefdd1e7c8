"""TEST STUB (tests/stubs/README.md), picked up by Python at start-up when tests/stubs is on PYTHONPATH: the reference's
main_sampling_fid.py does `from torch.utils.tensorboard import SummaryWriter` at module level, and torch's module needs the
`tensorboard` package, which this image lacks.  A finder on sys.meta_path serves a placeholder `torch.utils.tensorboard` (its
SummaryWriter raises when constructed: the driver is run with --no-tensorboard) -- only when the real package cannot be imported;
nothing else changes."""
import importlib.abc
import importlib.machinery
import importlib.util
import sys
import types


class _TensorboardStub(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    name = 'torch.utils.tensorboard'

    def find_spec(self, fullname, path, target=None):
        if fullname != self.name or importlib.util.find_spec('tensorboard') is not None:
            return None
        return importlib.machinery.ModuleSpec(fullname, self)

    def create_module(self, spec):
        m = types.ModuleType(spec.name)

        class SummaryWriter:
            def __init__(self, *a, **k):
                raise RuntimeError('torch.utils.tensorboard stub: tensorboard is not installed in the test image')
        m.SummaryWriter = SummaryWriter
        return m

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _TensorboardStub())
