#!/usr/bin/env python
"""GPU microbenchmark of the decode-step GEMM (diagnostics ABI entry).  Usage: python scripts/gemm_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch  # noqa: E402
from rqvae import _native  # noqa: E402
if os.environ.get('RQ_LIB'):          # A/B a differently-built kernel library (diagnostics only)
    _native.LIB_PATH = os.environ['RQ_LIB']


def bench(M, N, K, epi, bm, bn, sk, reps=40):
    dev = 'cuda'
    a = torch.randn((M, K), device=dev).to(torch.bfloat16)
    ws = [torch.randn((N, K), device=dev).to(torch.bfloat16) * 0.05 for _ in range(8)]     # rotate weights: no L2/MALL reuse
    bias = torch.randn((N,), device=dev)
    out = _native.dbg_gemm(a, ws[0], bias if epi % 16 != 4 else None, epi, bm, bn, sk)
    ref = a.float() @ ws[0].float().T + (bias if epi % 16 != 4 else 0)
    if epi % 16 == 4 and sk <= 0:
        out.zero_()
        out = _native.dbg_gemm(a, ws[0], None, epi, bm, bn, sk, out=out)
    got = out.float().sum(0) if epi % 16 == 4 else out.float()
    err = (got - ref).abs().max().item() / (ref.abs().max().item() + 1e-9)
    for i in range(5):
        _native.dbg_gemm(a, ws[i % 8], bias if epi % 16 != 4 else None, epi, bm, bn, sk, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        _native.dbg_gemm(a, ws[i % 8], bias if epi % 16 != 4 else None, epi, bm, bn, sk, out=out)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    tf = 2.0 * M * N * K / us / 1e6
    gbs = (N * K * 2 + M * K * 2) / us / 1e3
    return us, tf, gbs, err


if __name__ == '__main__':
    shapes = [('qkv', 4608, 1536, 0), ('proj', 1536, 1536, 4), ('fc1', 6144, 1536, 1), ('fc2', 1536, 6144, 4), ('cls', 16384, 1536, 3)]
    for M in ([int(x) for x in os.environ['RQ_MS'].split(',')] if os.environ.get('RQ_MS') else (64, 256, 512, 1024)):
        for name, N, K, epi in shapes:
            best = None
            rows = []
            for bm, bn in ((64, 64), (64, 128), (128, 64), (128, 128), (256, 128), (257, 128)):
                for gl in ((64, 96) if bm == 257 else (0, 64, 96) if bm >= 128 and os.environ.get('RQ_GL', '1') != '0' else (0,)):   # LDS-DMA staging: +64 / +96
                    for sk in ((1, 2, 4, 8) if epi == 4 else (1,)):
                        try:
                            us, tf, gbs, err = bench(M, N, K, epi + gl, bm, bn, sk)
                        except Exception as e:
                            print('fail', M, name, bm, bn, sk, gl, e)
                            continue
                        if err > 0.02 and epi != 1:
                            print('WRONG', M, name, bm, bn, sk, gl, err)
                        rows.append((us, f'{bm}g{gl // 32}' if gl else bm, bn, sk, tf, gbs, err))
            us, tf_, gbs_, err_ = bench(M, N, K, epi, 0, 0, 0)
            rows.sort()
            b = rows[0]
            print(f'M={M:5d} {name:5s} N={N:5d} K={K:5d} | auto {us:7.1f} us {tf_:7.1f} TF {gbs_:7.0f} GB/s err {err_:.1e} | '
                  f'best {b[0]:7.1f} us tile {b[1]}x{b[2]} sk{b[3]} {b[4]:7.1f} TF {b[5]:7.0f} GB/s | '
                  + ' '.join(f'{r[1]}x{r[2]}s{r[3]}:{r[0]:.0f}' for r in rows[:6]), flush=True)
