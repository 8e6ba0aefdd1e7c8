#!/usr/bin/env python
"""Decoder.conv_out (norm_out + swish fused) at the decode sub-batch of the benchmark: time per launch and HBM rate."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'rq-vae-transformer_amd'))
from rqvae import _native as nat
if os.environ.get('RQ_LIB'):          # A/B a differently-built kernel library (scripts/build_variant.py; diagnostics only)
    nat.LIB_PATH = os.environ['RQ_LIB']
dev = 'cuda'
B, H, Cin, Cout = int(os.environ.get('RQ_B', 64)), 256, 128, 3
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, H, H, Cin, device=dev, generator=g).to(torch.bfloat16)
w = 0.05 * torch.randn(Cout, 3, 3, Cin, device=dev, generator=g)
bias = torch.randn(Cout, device=dev, generator=g)
gn = torch.stack([1 + 0.2 * torch.randn(B, Cin, device=dev, generator=g), 0.3 * torch.randn(B, Cin, device=dev, generator=g)], -1).contiguous()
for tag, gnv in (('plain', None), ('fused GroupNorm+SiLU', gn)):
    out = nat.dbg_conv_out(x, w, bias, gn=gnv)
    ref_in = x.float() if gnv is None else torch.nn.functional.silu(x.float() * gnv[:, None, None, :, 0] + gnv[:, None, None, :, 1]).to(torch.bfloat16).float()
    ref = torch.nn.functional.conv2d(ref_in[:4].permute(0, 3, 1, 2), w.to(torch.bfloat16).float().permute(0, 3, 1, 2), bias, padding=1)
    err = (out[:4] - ref).abs().max().item() / ref.abs().max().item()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): nat.dbg_conv_out(x, w, bias, gn=gnv)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 4 * 1e3)
    ts.sort()
    by = B * H * H * (Cin * 2 + Cout * 4)
    print(f'conv_out {tag:22s} B={B}: {ts[3]:8.1f} us per launch  {by / ts[3] / 1e6:6.2f} TB/s of activations  rel err {err:.1e}')
