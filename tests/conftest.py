import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'rq-vae-transformer_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`) is dominated by the host-emulator tests (fibers: minutes each, single-threaded): when pytest-xdist
    is installed and nobody asked for a worker count, spread it over a few processes (tests are independent; the emulator / library
    builds they trigger are serialised by file locks).  The GPU suite is never parallelised (one device).  RQ_TESTS_WORKERS=1 turns it off."""
    if os.environ.get('PYTEST_XDIST_WORKER') or hasattr(config, 'workerinput'):
        return None                                   # (this IS a worker: never nest -- its own option set says "no workers" too)
    want = os.environ.get('RQ_TESTS_WORKERS')
    markexpr = getattr(config.option, 'markexpr', '') or ''
    if 'not gpu' not in markexpr or not hasattr(config.option, 'numprocesses') or config.option.numprocesses:
        return None
    n = int(want) if want else min(4, max(1, (os.cpu_count() or 1) // 2))
    if n > 1:
        config.option.numprocesses = n
        if getattr(config.option, 'dist', 'no') == 'no':
            config.option.dist = 'load'
    return None


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
