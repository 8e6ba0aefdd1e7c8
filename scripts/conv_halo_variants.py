#!/usr/bin/env python
"""Interleaved timing of the halo-conv variants: 8-row per-tile (baseline), 8-row persistent, 16-row on 8 wavefronts.
(profiles/r02_conv_halo_variants.txt also has the removed 16-row / 4-wavefront form, "w4".)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch
from rqvae import _native
if os.environ.get('RQ_LIB'):          # A/B a differently-built kernel library (diagnostics only)
    _native.LIB_PATH = os.path.join(ROOT, 'rq-vae-transformer_amd', os.environ['RQ_LIB'])

dev = 'cuda'
VARIANTS = (('th8', dict(persistent=False)), ('th8p', dict(persistent=True)))


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


shapes = ((8, 256, 128, 128, 0),) if os.environ.get('RQ_QUICK') else ((8, 256, 128, 128, 0), (32, 256, 128, 128, 0), (8, 128, 128, 128, 0), (8, 128, 256, 128, 0), (8, 64, 256, 256, 0), (8, 128, 256, 256, 0),
          (8, 256, 128, 128, 1), (8, 128, 256, 256, 1))
for B, H, Cin, Cout, ups in shapes:
    Hs = H // 2 if ups else H
    x = torch.randn((B, Hs, Hs, Cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((Cout, 3, 3, Cin), device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn((Cout,), device=dev)
    resid = torch.randn((B, H, H, Cout), device=dev).to(torch.bfloat16)
    gn = torch.stack([1 + 0.2 * torch.randn((B, Cin), device=dev), 0.3 * torch.randn((B, Cin), device=dev)], -1).contiguous()
    fl = 2.0 * B * H * H * Cout * 9 * Cin / 1e6
    outs = {n: torch.empty_like(resid) for n, _ in VARIANTS}
    cases = (('plain', {}),) if ups else (('plain', {}), ('GN+resid+stats', dict(gn=gn, resid=resid, st=True)), ('GN+stats', dict(gn=gn, st=True)))
    print(f'B{B} {Cin}->{Cout}@{H}{" ups" if ups else ""}:', flush=True)
    for name, kw in cases:
        kw = dict(kw)
        st = kw.pop('st', False)
        fns = {}
        for vn, vk in VARIANTS:
            th = 8
            stats = torch.zeros((B, (H // th) * (H // 32), 32, 2), device=dev) if st else None
            fns[vn] = (lambda vk=vk, vn=vn, stats=stats: _native.dbg_conv_halo(x, w, bias, out=outs[vn], stats=stats, ups=bool(ups), **vk, **kw))
        for f in fns.values():
            f(); f()
        best = {vn: 1e30 for vn in fns}
        for _ in range(5):
            for vn, f in fns.items():
                best[vn] = min(best[vn], timed(f, 6))
        ref = outs['th8'].float()
        line = f'   {name:15s}'
        for vn in fns:
            err = (outs[vn].float() - ref).abs().max().item()
            line += f' | {vn} {best[vn]:7.1f} us {fl / best[vn]:5.0f} TF' + ('' if vn == 'th8' else f' (d {err:.0e})')
        print(line, flush=True)
