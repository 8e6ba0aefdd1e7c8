#!/usr/bin/env python
"""GPU timing of the 256x256 GEMM's epilogue families at M rows (RQ_LIB selects the library build): bf16, GELU, fp32 slab,
in-place residual update (4 + 2048), fp32 logits."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch  # noqa: E402
from rqvae import _native  # noqa: E402
if os.environ.get('RQ_LIB'):
    _native.LIB_PATH = os.environ['RQ_LIB']
dev = 'cuda'
torch.manual_seed(0)
M = int(os.environ.get('RQ_M', 10752))


def timeit(a, ws, bias, epi, reps=30):
    run = lambda w, out=None: _native.dbg_gemm(a, w, None if epi % 16 == 4 and epi < 2048 else bias, epi, 256, 256, 1, out=out)
    out = run(ws[0])
    for i in range(3):
        run(ws[i % 4], out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        run(ws[i % 4], out)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


tot = 0.0
for name, N, K, epi in (('qkv', 4608, 1536, 0), ('proj', 1536, 1536, 4), ('proj+x', 1536, 1536, 4 + 2048), ('fc1', 6144, 1536, 1),
                        ('fc2', 1536, 6144, 4), ('fc2+x', 1536, 6144, 4 + 2048), ('cls', 16384, 1536, 3)):
    a = torch.randn((M, K), device=dev).to(torch.bfloat16)
    ws = [(torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16) for _ in range(4)]
    bias = torch.randn((N,), device=dev)
    t = sorted(timeit(a, ws, bias, epi) for _ in range(5))
    if name in ('qkv', 'proj+x', 'fc1', 'fc2+x'):
        tot += t[2]
    print(f'M={M} {name:6s} N={N:5d} K={K:5d}: med {t[2]:7.1f} us ({2.0 * M * N * K / t[2] / 1e6:6.0f} TF) min {t[0]:7.1f}', flush=True)
print(f'layer (qkv + proj+x + fc1 + fc2+x): {tot:.1f} us')
