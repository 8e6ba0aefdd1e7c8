#!/usr/bin/env python
"""RQVAE.get_codes at 256 images, a few repetitions, for `rocprofv3 --kernel-trace --stats` (encode-side kernel shares)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch
from rqvae import presets
dev = torch.device('cuda:0')
vae, ar, cfg = presets.build('huge', device=dev, seed=0)
del ar
x = torch.randn((256, 3, 256, 256), device=dev).clamp(-1, 1)
for _ in range(4):
    vae.get_codes(x)
torch.cuda.synchronize()
