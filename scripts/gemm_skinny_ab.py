#!/usr/bin/env python
"""Small-batch decode GEMMs (M <= 64): the engine's own tile choice, timed with rotating weights.  Run once as is (skinny
kernel) and once with RQAMD_NO_SKINNY=1 (the tiled kernels it replaces)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import torch  # noqa: E402
from rqvae import _native  # noqa: E402

dev = 'cuda'
tag = 'tiled' if os.environ.get('RQAMD_NO_SKINNY') else 'skinny'
for M in (64, 32, 8):
    tot = 0.0
    for name, N, K, epi, per_layer in (('qkv', 4608, 1536, 0, 1), ('proj', 1536, 1536, 4, 1), ('fc1', 6144, 1536, 1, 1), ('fc2', 1536, 6144, 4, 1), ('cls', 16384, 1536, 3, 0)):
        a = torch.randn((M, K), device=dev).to(torch.bfloat16)
        ws = [(torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16) for _ in range(12)]      # rotate: no L2 / MALL reuse
        bias = torch.randn((N,), device=dev)
        b = None if epi == 4 else bias
        out = _native.dbg_gemm(a, ws[0], b, epi, 0, 0, 0)
        ref = a.float() @ ws[0].float().T + (0 if epi == 4 else bias)
        if epi == 1:
            ref = torch.nn.functional.gelu(ref)
        got = out.float().sum(0) if epi == 4 else out.float()
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        for i in range(6):
            _native.dbg_gemm(a, ws[i % 12], b, epi, 0, 0, 0, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 60
        e0.record()
        for i in range(reps):
            _native.dbg_gemm(a, ws[i % 12], b, epi, 0, 0, 0, out=out)
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        tot += us * per_layer
        print(f'{tag} M={M:3d} {name:5s} N={N:5d} K={K:5d}: {us:6.1f} us  {N * K * 2 / us / 1e6:5.2f} TB/s of weights  err {err:.1e}', flush=True)
    print(f'{tag} M={M:3d}: GEMMs of one layer {tot:6.1f} us (weights 56.6 MB -> {56.6 / tot:5.2f} TB/s)', flush=True)
