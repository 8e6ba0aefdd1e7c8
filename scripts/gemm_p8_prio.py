#!/usr/bin/env python
"""A/B: eight-phase GEMM with and without s_setprio around the MFMA bursts (interleaved rounds)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd')); sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from gemm_p8_check import problem, timeit  # noqa: E402
M = 10752
for name, N, K, epi in (('qkv', 4608, 1536, 0), ('fc1', 6144, 1536, 1), ('fc2', 1536, 6144, 4)):
    a, ws, bias = problem(M, N, K)
    r = {0: [], 256: []}
    for rnd in range(5):
        for add in (0, 256):
            r[add].append(timeit(a, ws, bias, epi + add, 256, 256, 1))
    print(f'{name}: setprio {sorted(r[0])[2]:.1f} us, no setprio {sorted(r[256])[2]:.1f} us', flush=True)
