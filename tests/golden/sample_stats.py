"""Inputs and count statistics of the sampling-parity cases, shared by tests/golden/make_golden.py (which runs the REFERENCE's
RQTransformer.sample on them, in the build container) and tests/test_gpu_sample_stats.py (which replays them through the HIP engine).
numpy only."""
import numpy as np

SAMPLE_STATS_N = 20000
SAMPLE_STATS_COARSE = 25          # pairwise tables: codes bucketed by code // 25 (20 x 20 cells)


def sample_stats_inputs(cfg, case, n=SAMPLE_STATS_N):
    """(partial_sample, cond or None, start_loc) of one case -- shared with tests/test_gpu_sample_stats.py, which replays them on the GPU"""
    H, W, D = cfg['block_size']
    part = np.zeros((n, H, W, D), dtype=np.int64)
    cond = (np.arange(n) % cfg['vocab_size_cond']).reshape(n, 1).astype(np.int64)
    start = (0, 0)
    if case == 'nocond':
        cond = None
    if case == 'start':
        part[:, 0] = np.random.default_rng(77).integers(0, cfg['vocab_size'], (W, D))[None]      # one fixed first row for every sample
        start = (1, 0)
    return part, cond, start


def sample_stats_counts(xs, cfg, start):
    """code marginals of the first three sampled positions x all depths, and two coarse pairwise tables"""
    H, W, D = cfg['block_size']
    V, q = cfg['vocab_size'], SAMPLE_STATS_COARSE
    flat = xs.reshape(xs.shape[0], H * W, D)
    p0 = start[0] * W + start[1]
    marg = np.stack([np.stack([np.bincount(flat[:, p0 + i, d], minlength=V) for d in range(D)]) for i in range(3)]).astype(np.int32)
    nb = (V + q - 1) // q
    def pair(a, b):
        return np.bincount((a // q) * nb + (b // q), minlength=nb * nb).reshape(nb, nb).astype(np.int32)
    return marg, pair(flat[:, p0, 0], flat[:, p0 + 1, 0]), pair(flat[:, p0, 0], flat[:, p0, 1])
