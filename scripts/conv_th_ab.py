#!/usr/bin/env python
"""A/B of the halo conv tile heights (16 = 128x64 wave tiles, 8 = round-1 default) on the decoder's layer shapes:
interleaved rounds, correctness vs torch fp32 on the first shape."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from rqvae import _native  # noqa: E402

dev = 'cuda'


def t_of(fn, reps=8):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for B, H, Cin, Cout in ((32, 256, 128, 128), (32, 128, 128, 128), (32, 128, 256, 128), (32, 64, 256, 256), (32, 128, 256, 256)):
    x = torch.randn((B, H, H, Cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((Cout, 3, 3, Cin), device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn((Cout,), device=dev)
    resid = torch.randn((B, H, H, Cout), device=dev).to(torch.bfloat16)
    gn = torch.stack([1 + 0.2 * torch.randn((B, Cin), device=dev), 0.3 * torch.randn((B, Cin), device=dev)], -1).contiguous()
    out = torch.empty_like(resid)
    fl = 2.0 * B * H * H * Cout * 9 * Cin / 1e6
    res = {}
    for rnd in range(3):
        for th in (8, 16):
            stats = torch.zeros((B, (H // th) * (H // 32), 32, 2), device=dev)
            res.setdefault(('plain', th), []).append(t_of(lambda: _native.dbg_conv_halo(x, w, bias, out=out, tile_h=th)))
            res.setdefault(('fused', th), []).append(t_of(lambda: _native.dbg_conv_halo(x, w, bias, gn=gn, resid=resid, stats=stats, out=out, tile_h=th)))
    if H == 256:
        ref = F.conv2d(x[:2].float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1)
        for th in (8, 16):
            o = _native.dbg_conv_halo(x[:2].contiguous(), w, bias, tile_h=th).float()
            print(f'  th={th}: rel err {((o - ref).abs().max() / ref.abs().max()).item():.2e}')
    line = f'B{B} {Cin}->{Cout}@{H}:'
    for k in (('plain', 8), ('plain', 16), ('fused', 8), ('fused', 16)):
        t = sorted(res[k])[1]
        line += f' {k[0]}/th{k[1]} {t:7.1f} us {fl / t:6.0f} TF |'
    print(line, flush=True)
