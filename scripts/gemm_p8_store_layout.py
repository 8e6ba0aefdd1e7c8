#!/usr/bin/env python
"""Does the eight-phase GEMM's bf16 epilogue cost depend on how contiguous a tile's output rows are?  One round of ~252-256 tiles each:
N = 256 (a tile's 256 rows are one contiguous 128-KB block) against N = 4608 (512-byte row pieces 9 KB apart); epilogue cost =
time with - time without the epilogue (epi + 16)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch
from rqvae import _native
dev = 'cuda'


def t_of(a, ws, bias, epi, out, reps=20):
    for i in range(3):
        _native.dbg_gemm(a, ws[i % 4], bias, epi, 256, 256, 1, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for i in range(reps):
            _native.dbg_gemm(a, ws[i % 4], bias, epi, 256, 256, 1, out=out)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


for (M, N, K) in ((65536, 256, 1536), (3584, 4608, 1536), (64512, 256, 1536)):
    a = torch.randn((M, K), device=dev).to(torch.bfloat16)
    ws = [(torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16) for _ in range(4)]
    bias = torch.randn((N,), device=dev)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    full, skip = t_of(a, ws, bias, 0, out), t_of(a, ws, bias, 16, out)
    print(f'M={M} N={N} K={K}: tiles {((M + 255) // 256) * ((N + 255) // 256)}: with epilogue {full:.1f} us, without {skip:.1f} us -> epilogue {full - skip:.1f} us', flush=True)
