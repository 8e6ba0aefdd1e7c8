#!/usr/bin/env python
"""Decode-step GEMMs at 8 .. 512 rows: the K-split kernel (csrc/gemm_ks.h; fragment-packed weights) against the engine's round-3
choice (weight-streaming kernel up to 128 rows, tiled kernels beyond), timed interleaved with rotating weights inside a captured
graph of 48 launches (what a decode step is), per launch.  RQ_MS=64,100,200,500  RQ_E=1536."""
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
SHAPES = (('qkv', 3 * E, E, 0), ('proj', E, E, 4), ('fc1', 4 * E, E, 1), ('fc2', E, 4 * E, 4))
NW = 12


def graph_time(fn, n=48, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(n):
                fn(i)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n)


for M in [int(x) for x in os.environ.get('RQ_MS', '8,64,100,128,200,256,384,500').split(',')]:
    tot_old = tot_new = 0.0
    for name, N, K, epi in SHAPES:
        a = torch.randn((M, K), device=dev).to(torch.bfloat16)
        ws = [(torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16) for _ in range(NW)]
        wps = [_native.dbg_pack_w(w) for w in ws]
        bias = torch.randn((N,), device=dev)
        b = None if epi == 4 else bias
        ref = a.float() @ ws[0].float().T + (0 if epi == 4 else bias)
        if epi == 1:
            ref = torch.nn.functional.gelu(ref)
        out_old = _native.dbg_gemm(a, ws[0], b, epi, 0, 0, 0)
        t_old = graph_time(lambda i: _native.dbg_gemm(a, ws[i % NW], b, epi, 0, 0, 0, out=out_old))
        pick = _native.dbg_pick_ks(M, N, K, epi == 4)
        res = []
        best = None
        cands = [pick] if pick else []
        if pick and pick[0] == 128:
            cands += [(128, bn, 4, sk) for bn in (32, 64, 96) for sk in ((1, 2, 4) if epi == 4 else (1,))
                      if (128, bn, 4, sk) != pick and (K // 64) % sk == 0 and (K // 64) // sk >= 16]
        elif pick and epi == 4:
            cands += [(64, 32, 8, sk) for sk in (1, 2, 4, 8) if sk != pick[3] and (K // 64) % sk == 0]
        for (bm, bn, nw, sk) in cands:
            out = _native.dbg_gemm_ks(a, wps[0], N, b, epi, bm, bn, sk)
            got = out.float().sum(0) if epi == 4 else out.float()
            err = ((got - ref).abs().max() / ref.abs().max()).item()
            t = graph_time(lambda i: _native.dbg_gemm_ks(a, wps[i % NW], N, b, epi, bm, bn, sk, out=out))
            tag = f'{bm}x{bn}/sk{sk}'
            res.append(f'{tag}:{t:5.1f}')
            if (bm, bn, nw, sk) == pick:
                best = (t, tag, err)
        tot_old += t_old
        tot_new += best[0] if best else t_old
        print(f'M={M:4d} {name:5s} N={N:5d} K={K:5d}: round 3 {t_old:6.1f} us | K-split {best[0] if best else float("nan"):6.1f} us ({best[1] if best else "-"}, '
              f'err {best[2] if best else 0:.1e}; {N * K * 2 / (best[0] if best else t_old) / 1e6:5.2f} TB/s of weights) [{" ".join(res)}]', flush=True)
    print(f'M={M:4d}: GEMMs of one layer (E = {E}): {tot_old:6.1f} -> {tot_new:6.1f} us (in-graph, per launch incl. the launch boundary)', flush=True)
