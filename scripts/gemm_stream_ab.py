#!/usr/bin/env python
"""Small-batch decode GEMMs (M <= 512): the weight-streaming kernel (tile codes 66x32 / 130x32, gemm_stream_kernel) against the
engine's tile choice without it (RQAMD_NO_STREAM=1 semantics: explicit bm = bn = 0 with the picker's stream rule bypassed via
the `auto_old` call), timed with rotating weights, interleaved.  RQ_MS=64,128 selects the row counts."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch  # noqa: E402
from rqvae import _native  # noqa: E402

if os.environ.get('RQ_LIB'):          # A/B a differently-built kernel library (scripts/build_variant.py; diagnostics only)
    _native.LIB_PATH = os.environ['RQ_LIB']
dev = 'cuda'
E = int(os.environ.get('RQ_E', 1536))      # 1536: the 1.4B model; 2560: the 3.8B model (BASELINE configs[3]); 1024: the 355M model
SHAPES = (('qkv', 3 * E, E, 0), ('proj', E, E, 4), ('fc1', 4 * E, E, 1), ('fc2', E, 4 * E, 4), ('cls', 16384 if E != 1024 else 2048, E, 3))


def timed(fn, reps=60):
    for i in range(6):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for M in [int(x) for x in os.environ.get('RQ_MS', '8,64,100,128,256,500').split(',')]:
    tot_old = tot_new = 0.0
    for name, N, K, epi in SHAPES:
        a = torch.randn((M, K), device=dev).to(torch.bfloat16)
        ws = [(torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16) for _ in range(12)]      # rotate: no L2 / MALL reuse
        bias = torch.randn((N,), device=dev)
        b = None if epi == 4 else bias
        ref = a.float() @ ws[0].float().T + (0 if epi == 4 else bias)
        if epi == 1:
            ref = torch.nn.functional.gelu(ref)
        os.environ['RQAMD_NO_STREAM'] = '1'
        res = []
        out_old = _native.dbg_gemm(a, ws[0], b, epi, 0, 0, 0)
        t_old = timed(lambda i: _native.dbg_gemm(a, ws[i % 12], b, epi, 0, 0, 0, out=out_old))
        best = None
        for bm, bn in (((66, 32), (66, 64)) if M <= 64 else ((130, 32),)):      # 66x64: 64-row weight tiles (round 4)
            for sk in ((1, 2, 4, 8) if epi == 4 else (1,)):
                if (K // 64) % sk:
                    continue
                out = _native.dbg_gemm(a, ws[0], b, epi, bm, bn, sk)
                got = out.float().sum(0) if epi == 4 else out.float()
                err = ((got - ref).abs().max() / ref.abs().max()).item()
                t = timed(lambda i: _native.dbg_gemm(a, ws[i % 12], b, epi, bm, bn, sk, out=out))
                res.append(f'{bm}x{bn}sk{sk}:{t:5.1f}')
                if best is None or t < best[0]:
                    best = (t, f'{bm}x{bn}/sk{sk}', err)
        tot_old += t_old if name != 'cls' else 0
        tot_new += best[0] if name != 'cls' else 0
        print(f'M={M:4d} {name:5s} N={N:5d} K={K:5d}: tiled/auto {t_old:6.1f} us | stream {best[0]:6.1f} us ({best[1]}, err {best[2]:.1e}; '
              f'{N * K * 2 / best[0] / 1e6:5.2f} TB/s of weights) [{" ".join(res)}]', flush=True)
    print(f'M={M:4d}: GEMMs of one layer (E = {E}, weights {12 * E * E * 2 / 1e6:.0f} MB): {tot_old:6.1f} -> {tot_new:6.1f} us', flush=True)
