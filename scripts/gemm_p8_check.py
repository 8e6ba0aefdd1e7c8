#!/usr/bin/env python
"""GPU race screen + A/B of the 256x256 eight-phase GEMM (gemm_p8_kernel) against the tiles the engine used before.
Every launch is compared bitwise with the first launch of the same problem (a hazard shows up as run-to-run differences)
and once with a torch fp32 reference; then interleaved timing rounds (median and min per variant, same process)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch  # noqa: E402
from rqvae import _native  # noqa: E402
if os.environ.get('RQ_LIB'):          # A/B a differently-built kernel library (scripts/build_variant.py; diagnostics only)
    _native.LIB_PATH = os.environ['RQ_LIB']

dev = 'cuda'
torch.manual_seed(0)


def problem(M, N, K):
    a = torch.randn((M, K), device=dev).to(torch.bfloat16)
    ws = [(torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16) for _ in range(4)]
    bias = torch.randn((N,), device=dev)
    return a, ws, bias


def run(a, w, bias, epi, bm, bn, sk, out=None):
    return _native.dbg_gemm(a, w, None if epi % 16 == 4 else bias, epi, bm, bn, sk, out=out)


def screen(M, N, K, epi, sk, reps=12):
    a, ws, bias = problem(M, N, K)
    first = run(a, ws[0], bias, epi + 1024, 256, 256, sk).clone()
    for _ in range(reps):          # the two-phase schedule must give the four-phase schedule's bits, launch after launch
        if not torch.equal(run(a, ws[0], bias, epi + 512, 256, 256, sk), first):
            print(f'TWO-PHASE SCHEDULE DIFFERS M={M} N={N} K={K} epi={epi} sk={sk}', flush=True)
            return False
    ref = a.float() @ ws[0].float().T + (0 if epi == 4 else bias)
    got = first.float().sum(0) if epi == 4 else first.float()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    bad = 0
    for _ in range(reps):
        o = run(a, ws[0], bias, epi, 256, 256, sk)
        bad += int(not torch.equal(o, first))
    print(f'screen M={M} N={N} K={K} epi={epi} sk={sk}: rel err {err:.2e}, {bad}/{reps} launches differ from the first', flush=True)
    return err < (2e-2 if epi in (0, 1) else 3e-3) and bad == 0


def timeit(a, ws, bias, epi, bm, bn, sk, reps=30):
    out = run(a, ws[0], bias, epi, bm, bn, sk)
    for i in range(3):
        run(a, ws[i % 4], bias, epi, bm, bn, sk, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        run(a, ws[i % 4], bias, epi, bm, bn, sk, out=out)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


if __name__ == "__main__":
    ok = True
    for (M, N, K, epi, sk) in ((8192, 4608, 1536, 0, 1), (8192, 1536, 1536, 4, 1), (8192, 6144, 1536, 1, 1), (8192, 1536, 6144, 4, 2),
                               (4096, 16384, 1536, 3, 1), (500, 6144, 1536, 1, 1), (2049, 1536, 6144, 4, 4), (8192, 2560, 2560, 4, 1)):
        ok &= screen(M, N, K, epi, sk)
    print('RACE SCREEN', 'OK' if ok else 'FAILED', flush=True)
    Ms = [int(x) for x in os.environ.get('RQ_MS', '8192,4096,2048').split(',')]
    shapes = [('qkv', 4608, 1536, 0), ('proj', 1536, 1536, 4), ('fc1', 6144, 1536, 1), ('fc2', 1536, 6144, 4), ('cls', 16384, 1536, 3)]
    for M in Ms:
        for name, N, K, epi in shapes:
            a, ws, bias = problem(M, N, K)
            variants = [('auto', 0, 0, 0), ('p8', 256, 256, 1)] if os.environ.get('RQ_AUTO') else [('ph4', 256, 256, 1), ('ph2', 256, 256, 1)]   # picker vs forced 256x256; or four / two phases per K-tile (epi + 1024 / + 512)
            if epi == 4:
                variants += [('p8/sk2', 256, 256, 2), ('p8/sk4', 256, 256, 4)]
            res = {v[0]: [] for v in variants}
            for rnd in range(5):                      # interleaved rounds in one process
                for v in variants:
                    res[v[0]].append(timeit(a, ws, bias, epi + (512 if v[0] == 'ph2' else 1024 if v[0] == 'ph4' else 0), v[1], v[2], v[3]))
            fl = 2.0 * M * N * K
            line = f'M={M:5d} {name:5s} N={N:5d} K={K:5d} |'
            for v in variants:
                t = sorted(res[v[0]])
                line += f' {v[0]}: med {t[2]:7.1f} us ({fl / t[2] / 1e6:6.0f} TF) min {t[0]:7.1f} |'
            print(line, flush=True)
