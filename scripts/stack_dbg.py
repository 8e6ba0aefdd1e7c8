#!/usr/bin/env python
"""Diagnostics: one teacher-forced pass through the stack kernel, one launch per phase, traced (RQAMD_STACK_TRACE)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
os.environ.setdefault('RQAMD_STACK', '1')
import torch
from rqvae import presets
dev = torch.device('cuda:0')
vae, ar, cfg = presets.build(os.environ.get('RQ_PRESET', 'small'), device=dev, seed=0)
B = int(os.environ.get('RQ_B', 1))
codes = torch.zeros((B, 8, 8, 4), dtype=torch.long, device=dev)
out = ar(codes, model_aux=vae, cond=torch.zeros((B, 1), dtype=torch.long, device=dev))
torch.cuda.synchronize()
print('ok', float(out.float().abs().max()))
