#!/usr/bin/env python
"""AR sampling of the 1.4B model at RQ_B images (default 2048) with each KV-cache storage format (RQAMD_KV read when an engine is created):
bf16 (default), int8k (body keys as bytes + scale), int8kv (round 6: body keys and values).  Interleaved, best of 3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch
from rqvae import presets
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
B = int(os.environ.get('RQ_B', 2048))
fmts = os.environ.get('RQ_FMTS', 'bf16,int8k,int8kv').split(',')
ars, vae = {}, None
part = torch.zeros((B, 8, 8, 4), device=dev, dtype=torch.long)
cond = torch.zeros((B, 1), device=dev, dtype=torch.long)
for f in fmts:
    os.environ['RQAMD_KV'] = f
    vae, ar, _ = presets.build('huge', device=dev, seed=0)
    ar.sample(part[:2], model_aux=vae, cond=cond[:2], top_k=1024, top_p=0.95)
    ars[f] = ar
os.environ.pop('RQAMD_KV')
res = {f: [] for f in fmts}
for f in fmts:
    ars[f].sample(part, model_aux=vae, cond=cond, top_k=1024, top_p=0.95)
for rep in range(3):
    for f in fmts:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ars[f].sample(part, model_aux=vae, cond=cond, top_k=1024, top_p=0.95); e1.record(); e1.synchronize()
        res[f].append(e0.elapsed_time(e1))
print(f'1.4B AR sampling at {B} images: ' + ' | '.join(f'{f}: {min(res[f]):.1f} ms ({B / min(res[f]) * 1e3:.0f} img/s AR only)' for f in fmts), flush=True)
