import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'rq-vae-transformer_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: CPU test that takes more than a few seconds')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture
def fake_torch_rng(monkeypatch):
    """torch.rand_like / torch.randperm drawing from a seeded numpy stream (what tests/golden/make_golden.py:gen_ema did to the
    reference), so that the dead-code restart of the EMA codebook update is replayable on any device."""
    import numpy as np
    import torch

    def install(seed):
        prng = np.random.default_rng(seed)
        monkeypatch.setattr(torch, 'randperm', lambda n, device=None: torch.from_numpy(prng.permutation(n)).to(device or 'cpu'))
        monkeypatch.setattr(torch, 'rand_like', lambda t: torch.from_numpy(prng.random(tuple(t.shape), dtype=np.float32)).to(t.device))
    return install
