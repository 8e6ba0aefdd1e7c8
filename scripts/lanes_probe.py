#!/usr/bin/env python
"""Probe: does running S independent sub-batches of B/S rows on S streams (S engines) beat one engine on B rows at small B?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch
from rqvae import presets

dev = torch.device('cuda:0')
SMAX = 8
models = []
vae = None
for i in range(SMAX):
    v, ar, cfg = presets.build('huge', device=dev, seed=0)
    if vae is None:
        vae = v
    models.append(ar)
    del v
streams = [torch.cuda.Stream(dev) for _ in range(SMAX)]


def run(B, S, reps=2):
    rows = [B // S + (1 if i < B % S else 0) for i in range(S)]
    ins = [(torch.zeros((r, 8, 8, 4), dtype=torch.long, device=dev), torch.zeros((r, 1), dtype=torch.long, device=dev)) for r in rows]

    def once():
        outs = []
        for i in range(S):
            with torch.cuda.stream(streams[i]):
                outs.append(models[i].sample(ins[i][0], model_aux=vae, cond=ins[i][1], top_k=1024, top_p=0.95))
        torch.cuda.synchronize(dev)
        return outs
    once()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    return (time.perf_counter() - t0) / reps


for B in (64, 128, 500, 1024, 2048):
    line = f'B={B:5d}:'
    for S in (1, 2, 4, 8):
        t = run(B, S)
        line += f'  S={S}: {t * 1e3:7.1f} ms ({B / t:7.1f} img/s AR only)'
    print(line, flush=True)
