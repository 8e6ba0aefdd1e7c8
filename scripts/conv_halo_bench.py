#!/usr/bin/env python
"""GPU check + microbenchmark of the halo-reuse 3x3 conv against the implicit-GEMM conv and torch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch
import torch.nn.functional as F
from rqvae import _native
if os.environ.get('RQ_LIB'):          # A/B a differently-built kernel library (scripts/build_variant.py; diagnostics only)
    _native.LIB_PATH = os.environ['RQ_LIB']

dev = 'cuda'


def bench(fn, reps=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for B, H, Cin, Cout in ((2, 64, 128, 128), (8, 256, 128, 128), (8, 128, 128, 128), (8, 128, 256, 128), (8, 64, 256, 256), (8, 128, 256, 256)):
    x = torch.randn((B, H, H, Cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((Cout, 3, 3, Cin), device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn((Cout,), device=dev)
    resid = torch.randn((B, H, H, Cout), device=dev).to(torch.bfloat16)
    gn = torch.stack([1 + 0.2 * torch.randn((B, Cin), device=dev), 0.3 * torch.randn((B, Cin), device=dev)], -1).contiguous()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1)
    o1 = _native.dbg_conv_halo(x, w, bias).float()
    e1 = ((o1 - ref).abs().max() / ref.abs().max()).item()
    xn = F.silu(x.float() * gn[:, None, None, :, 0] + gn[:, None, None, :, 1]).to(torch.bfloat16).float()
    ref2 = F.conv2d(xn.permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1) + resid.float()
    o2 = _native.dbg_conv_halo(x, w, bias, gn=gn, resid=resid).float()
    e2 = ((o2 - ref2).abs().max() / ref2.abs().max()).item()
    out = torch.empty_like(resid)
    fl = 2.0 * B * H * H * Cout * 9 * Cin / 1e6
    t_g = bench(lambda: _native.dbg_conv(x, w, bias, None, 3, 1, 0, 0, 0, 0, out=out))
    t_h = bench(lambda: _native.dbg_conv_halo(x, w, bias, out=out))
    t_hg = bench(lambda: _native.dbg_conv_halo(x, w, bias, gn=gn, resid=resid, out=out))
    t_hr = bench(lambda: _native.dbg_conv_halo(x, w, bias, resid=resid, out=out))
    t_hn = bench(lambda: _native.dbg_conv_halo(x, w, bias, gn=gn, out=out))
    t_gr = bench(lambda: _native.dbg_conv(x, w, bias, resid, 3, 1, 0, 0, 0, 0, out=out))
    print(f'B{B} {Cin}->{Cout}@{H}: err plain {e1:.1e} fused {e2:.1e} | implicit-GEMM {t_g:7.1f} us {fl / t_g:6.1f} TF (+resid {t_gr:7.1f}) | '
          f'halo {t_h:7.1f} us {fl / t_h:6.1f} TF | halo+GN+SiLU+resid {t_hg:7.1f} us {fl / t_hg:6.1f} TF | halo+resid {t_hr:7.1f} | halo+GN {t_hn:7.1f}', flush=True)
