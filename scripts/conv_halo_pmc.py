#!/usr/bin/env python
"""A few launches of the halo conv (dominant decoder layer) for counter collection: scripts/gpu.sh sqpmc."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch
from rqvae import _native
dev = 'cuda'
B, H, Cin, Cout = 8, 256, 128, 128
x = torch.randn((B, H, H, Cin), device=dev).to(torch.bfloat16)
w = (torch.randn((Cout, 3, 3, Cin), device=dev) * 0.05).to(torch.bfloat16)
bias = torch.randn((Cout,), device=dev)
resid = torch.randn((B, H, H, Cout), device=dev).to(torch.bfloat16)
gn = torch.stack([1 + 0.2 * torch.randn((B, Cin), device=dev), 0.3 * torch.randn((B, Cin), device=dev)], -1).contiguous()
out = torch.empty_like(resid)
st = torch.zeros((B, (H // 8) * (H // 32), 32, 2), device=dev)
pers = os.environ.get('RQ_PERSIST', '0') == '1'
for _ in range(4):
    _native.dbg_conv_halo(x, w, bias, out=out, persistent=pers)
    _native.dbg_conv_halo(x, w, bias, gn=gn, resid=resid, stats=st, out=out, persistent=pers)
torch.cuda.synchronize()
