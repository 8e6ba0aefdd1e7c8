#!/usr/bin/env python
"""Small-batch sampling A/B (round 4): RQTransformer.sample of the 1.4B model (or RQ_MODEL) at the per-GPU batches SURVEY 8d names, on
engine variants that live in one process and are timed interleaved.  A variant is a name plus environment switches that are read
when an engine is created, e.g. RQ_VARIANTS='base,nostream:RQAMD_NO_STREAM=1' (switches that a launcher caches in a function-local
static on first use cannot be A/B'd this way).  RQ_BS=64,100,200,500  RQ_MODEL=huge.  Prints AR ms per batch and images/s (AR only);
the codes of two variants differ wherever rounding differs -- the parity of each is what the GPU tests check."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch  # noqa: E402
from rqvae import _native, presets  # noqa: E402

if os.environ.get('RQ_LIB'):
    _native.LIB_PATH = os.environ['RQ_LIB']
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
model = os.environ.get('RQ_MODEL', 'huge')
batches = [int(b) for b in os.environ.get('RQ_BS', '64,100,128').split(',')]
variants = os.environ.get('RQ_VARIANTS', 'base').split(',')
engines = {}
vae = None
# a variant = environment switches that the engine / launchers read when first used ("name:VAR=val+VAR2=val2"; "base" = none)
def _env_of(v):
    return dict(kv.split('=', 1) for kv in v.split(':', 1)[1].split('+')) if ':' in v else {}


ALL_KEYS = sorted({k for v in variants for k in _env_of(v)})
for v in variants:
    for k in ALL_KEYS:
        os.environ.pop(k, None)
    os.environ.update(_env_of(v))
    vae, ar, cfg = presets.build(model, device=dev, seed=0)
    part = torch.zeros((2,) + tuple(ar.block_size), device=dev, dtype=torch.long)
    ar.sample(part, model_aux=vae, cond=torch.zeros((2, ar.block_size_cond), device=dev, dtype=torch.long), top_k=1024, top_p=0.95)
    engines[v] = ar
for k in ALL_KEYS:
    os.environ.pop(k, None)
for B in batches:
    part = torch.zeros((B,) + tuple(engines[variants[0]].block_size), device=dev, dtype=torch.long)
    cond = torch.zeros((B, engines[variants[0]].block_size_cond), device=dev, dtype=torch.long)
    res = {v: [] for v in variants}
    codes = {}
    for v in variants:
        torch.cuda.manual_seed_all(5)
        codes[v] = engines[v].sample(part, model_aux=vae, cond=cond, top_k=1024, top_p=0.95)      # warm-up: graphs captured
    for rep in range(4):
        for v in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(2):
                engines[v].sample(part, model_aux=vae, cond=cond, top_k=1024, top_p=0.95)
            e1.record()
            e1.synchronize()
            res[v].append(e0.elapsed_time(e1) / 2)
    line = ' | '.join(f'{v}: {min(res[v]):7.1f} ms/batch ({B / min(res[v]) * 1e3:6.1f} img/s AR only; runs ' + ' '.join(f'{t:.0f}' for t in res[v]) + ')'
                      for v in variants)
    same = '' if len(variants) < 2 else f' | codes equal across variants: {float((codes[variants[0]] == codes[variants[1]]).float().mean()):.3f}'
    print(f'{model} B={B:4d}: {line}{same}', flush=True)
