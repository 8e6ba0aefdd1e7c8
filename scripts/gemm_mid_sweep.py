#!/usr/bin/env python
"""Mid-batch decode-step GEMMs (M = 200 .. 512 rows: the batches of the reference's Figure 4): tile shape / wave layout / LDS-DMA ring
depth / K-split sweep of gemm_bf16_kernel on a diagnostics build (-DRQ_GEMM_SWEEP, scripts/build_variant.py), per launch INSIDE a
captured graph (what a launch costs in the sampler), rotating weights (> 400 MB per shape: nothing comes from the MALL).

  RQ_LIB=rq-vae-transformer_amd/variants/librqamd_sweep.so RQ_MS=200,500 python scripts/gemm_mid_sweep.py

Tile code = BM + wave layout (0: 2 x 2 wavefronts, 1: 4 x 1, 4: 4 x 2, 8: 4 x 4), `g<stages>`; `r` = register-staged (the shipped kernels
below 512 rows).  Ablations of the best variants: `-dma` = no operand staging (MFMAs + fragment reads + barriers only), `-mma` =
staging, waits and barriers only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch  # noqa: E402
from rqvae import _native  # noqa: E402

if os.environ.get('RQ_LIB'):
    _native.LIB_PATH = os.environ['RQ_LIB']
dev = 'cuda'
E = int(os.environ.get('RQ_E', 1536))
SHAPES = (('qkv', 3 * E, E, 0), ('proj', E, E, 4), ('fc1', 4 * E, E, 1), ('fc2', E, 4 * E, 4), ('cls', 16384, E, 3))
only = os.environ.get('RQ_SHAPES')
REPS = 48

# (code, bn, BM, stages list)
TILES = [(128, 64, 128, (3, 6)), (128, 128, 128, (3, 5)), (129, 64, 128, (3, 6)), (129, 96, 128, (3, 5)), (129, 128, 128, (3, 5)),
         (129, 160, 128, (3, 4)), (129, 192, 128, (3, 4)), (132, 64, 128, (3, 6)), (132, 128, 128, (3, 5)), (132, 192, 128, (3, 4)),
         (64, 128, 64, (3, 6)), (64, 192, 64, (3, 5)), (258, 64, 256, (3, 4)), (258, 96, 256, (3,)), (260, 64, 256, (3, 4)),
         (136, 128, 128, (2, 3, 5)), (136, 256, 128, (2, 3)), (264, 128, 256, (2, 3))]
if os.environ.get('RQ_TILES'):          # e.g. RQ_TILES=129x128,132x128,136x128: only these tile codes
    keep = set(os.environ['RQ_TILES'].split(','))
    TILES = [t for t in TILES if f'{t[0]}x{t[1]}' in keep]
if os.environ.get('RQ_STAGES'):         # e.g. RQ_STAGES=3,4: these ring depths for every tile
    TILES = [(c, bn, bm, tuple(int(x) for x in os.environ['RQ_STAGES'].split(','))) for c, bn, bm, _ in TILES]
REG = [(64, 64), (128, 64), (64, 128), (128, 128)]
if os.environ.get('RQ_TILES'):
    REG = [t for t in REG if f'{t[0]}x{t[1]}r' in os.environ['RQ_TILES'].split(',')]


def graph_time(fn):
    """us per launch of fn(i), REPS launches captured in one graph, best of 5 replays"""
    fn(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(REPS):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REPS)
    return best


def cdiv(a, b):
    return (a + b - 1) // b


for M in [int(x) for x in os.environ.get('RQ_MS', '200,500').split(',')]:
    layer_old = layer_new = 0.0
    for name, N, K, epi in SHAPES:
        if only and name not in only.split(','):
            continue
        nrot = int(os.environ['RQ_NROT']) if os.environ.get('RQ_NROT') else max(4, int(420e6 / (N * K * 2)) + 1)     # RQ_NROT=1: the same (cache-warm) weights every launch
        a = torch.randn((M, K), device=dev).to(torch.bfloat16)
        ws = [(torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16) for _ in range(nrot)]
        bias = torch.randn((N,), device=dev)
        b = None if epi == 4 else bias
        ref = a.float() @ ws[0].float().T + (0 if epi == 4 else bias)
        if epi == 1:
            ref = torch.nn.functional.gelu(ref)
        scale = ref.abs().max().item()
        results = []

        def run(tag, bm, bn, sk, ecode, accum=False):
            try:
                if accum:
                    out = torch.zeros((M, N), device=dev)
                    _native.dbg_gemm(a, ws[0], bias, ecode + 2048, bm, bn, 1, out=out)
                    err = ((out - (ref + bias)).abs().max() / scale).item()
                    t = graph_time(lambda i: _native.dbg_gemm(a, ws[i % nrot], bias, ecode + 2048, bm, bn, 1, out=out))
                else:
                    out = _native.dbg_gemm(a, ws[0], b, ecode, bm, bn, sk)
                    got = out.float().sum(0) if epi == 4 else out.float()
                    err = ((got - ref).abs().max() / scale).item()
                    t = graph_time(lambda i: _native.dbg_gemm(a, ws[i % nrot], b, ecode, bm, bn, sk, out=out))
            except (RuntimeError, ValueError) as ex:
                print(f'   {tag}: {str(ex)[:90]}', flush=True)
                return None
            if err > 2e-2:
                tag += f'!ERR{err:.1e}'
            results.append((t, tag, bm, bn, sk, ecode, accum))
            return t

        t_auto = run('auto', 0, 0, 0, epi)
        sks = (1, 2, 3, 4, 6, 8) if epi == 4 else (1,)
        for bm, bn in REG:
            for sk in sks:
                if (K // 64) % sk == 0 and K // 64 // sk >= 4:
                    run(f'{bm}x{bn}r/s{sk}', bm, bn, sk, epi)
        for code, bn, BM, stages in TILES:
            for st in stages:
                for sk in sks:
                    if (K // 64) % sk or K // 64 // sk < 4:
                        continue
                    wgs = cdiv(M, BM) * cdiv(N, bn) * sk
                    if wgs < 96 or wgs > 640:
                        continue
                    run(f'{code}x{bn}g{st}/s{sk}', code, bn, sk, epi + 32 * st)
                if epi == 4 and cdiv(M, BM) * cdiv(N, bn) >= 96:
                    run(f'{code}x{bn}g{st}/acc', code, bn, 1, epi + 32 * st, accum=True)
        results.sort()
        best = results[0]
        print(f'M={M:4d} {name:5s} N={N:5d} K={K:5d} ({nrot} weight copies): auto {t_auto:6.2f} us | best {best[0]:6.2f} us {best[1]} '
              f'({2.0 * M * N * K / best[0] / 1e6:6.1f} TF)', flush=True)
        print('     ' + '  '.join(f'{tag}:{t:.1f}' for t, tag, *_ in results[:14]), flush=True)
        # ablations of the three best LDS-DMA variants
        abl = []
        for t, tag, bm, bn, sk, ecode, accum in [r for r in results if 'g' in r[1] and not r[6]][:int(os.environ.get('RQ_NABL', 3))]:
            out = _native.dbg_gemm(a, ws[0], b, ecode, bm, bn, sk)
            t1 = graph_time(lambda i: _native.dbg_gemm(a, ws[i % nrot], b, ecode + 4096, bm, bn, sk, out=out))
            t2 = graph_time(lambda i: _native.dbg_gemm(a, ws[i % nrot], b, ecode + 8192, bm, bn, sk, out=out))
            t3 = graph_time(lambda i: _native.dbg_gemm(a, ws[i % nrot], b, ecode + 8192 + 16, bm, bn, sk, out=out))
            abl.append(f'{tag}: {t:.1f} | -dma {t1:.1f} | -mma {t2:.1f} | -mma-epilogue {t3:.1f}')
        print('     ablations: ' + ' ;  '.join(abl), flush=True)
        if name != 'cls':
            layer_old += t_auto
            layer_new += best[0]
        del ws
    print(f'M={M:4d}: GEMMs of one layer (E = {E}): auto {layer_old:6.1f} us -> best per shape {layer_new:6.1f} us', flush=True)
