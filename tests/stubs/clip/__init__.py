"""TEST STUB (tests/stubs/README.md): rqvae/metrics/clip_score.py imports clip at module top."""


def load(*a, **k):
    raise RuntimeError('clip stub: not available in the test image')


def tokenize(*a, **k):
    raise RuntimeError('clip stub: not available in the test image')
