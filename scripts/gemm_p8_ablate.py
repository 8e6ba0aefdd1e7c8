#!/usr/bin/env python
"""Where a 256x256 tile's time goes: full kernel vs the same launch without the epilogue (diagnostics flag epi + 16),
interleaved rounds in one process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import torch  # noqa: E402
from gemm_p8_check import problem, timeit  # noqa: E402

M = int(os.environ.get('RQ_M', 10752))
for name, N, K, epi in (('qkv', 4608, 1536, 0), ('proj', 1536, 1536, 4), ('fc1', 6144, 1536, 1), ('fc2', 1536, 6144, 4), ('cls', 16384, 1536, 3)):
    a, ws, bias = problem(M, N, K)
    res = {'full': [], 'no-epilogue': []}
    for rnd in range(5):
        res['full'].append(timeit(a, ws, bias, epi, 256, 256, 1))
        res['no-epilogue'].append(timeit(a, ws, bias, epi + 16, 256, 256, 1))
    f, n = sorted(res['full'])[2], sorted(res['no-epilogue'])[2]
    tiles = ((M + 255) // 256) * (N // 256)
    rounds = -(-((tiles + 7) // 8) // 32)
    print(f'M={M} {name:5s}: full {f:7.1f} us, without epilogue {n:7.1f} us -> epilogue {f - n:6.1f} us = {(f - n) / rounds:5.1f} us per round '
          f'({rounds} rounds of {K // 64} K-tiles; main loop + prologue {n / rounds:5.1f} us per round)', flush=True)
