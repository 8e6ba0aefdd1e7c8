#!/usr/bin/env python
"""Small-batch A/B: per-launch decode kernels vs the persistent stack kernel (RQAMD_STACK=1), ImageNet 1.4B shape.
Parity of the teacher-forced logits first (grid-barrier form and one-launch-per-phase form), then RQTransformer.sample times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch
from rqvae import presets

dev = torch.device('cuda:0')
preset = os.environ.get('RQ_PRESET', 'huge')


def build(**env):
    for k in ('RQAMD_STACK', 'RQAMD_STACK_STEPWISE', 'RQAMD_STACK_ROWS'):
        os.environ.pop(k, None)
    os.environ.update(env)
    vae, ar, cfg = presets.build(preset, device=dev, seed=0)
    codes = torch.zeros((1, 8, 8, 4), dtype=torch.long, device=dev)
    print('built', env, flush=True)
    if os.environ.get('RQAMD_STACK_TRACE'):
        ar(codes, model_aux=vae, cond=torch.zeros((1, 1), dtype=torch.long, device=dev))
        torch.cuda.synchronize()
    ar.sample(codes, model_aux=vae, cond=torch.zeros((1, 1), dtype=torch.long, device=dev), top_k=1024, top_p=0.95)   # engine created under this env
    torch.cuda.synchronize()
    print('first sample done', flush=True)
    return vae, ar


vae, ar0 = build()
g = torch.Generator(device=dev).manual_seed(1)
B = 6
codes = torch.randint(0, 16384, (B, 8, 8, 4), device=dev, generator=g)
cond = torch.randint(0, 1000, (B, 1), device=dev, generator=g)
l0 = ar0(codes, model_aux=vae, cond=cond).float()
if os.environ.get('RQ_STEPWISE', '1') == '1':
    _, ar2 = build(RQAMD_STACK='1', RQAMD_STACK_STEPWISE='1')
    l2 = ar2(codes, model_aux=vae, cond=cond).float()
    print(f'stack kernel, one launch per phase vs per-launch kernels: max |dlogit| {(l2 - l0).abs().max().item():.5f}  mean {(l2 - l0).abs().mean().item():.6f}  (|logits| max {l0.abs().max().item():.2f})', flush=True)
    del ar2
_, ar1 = build(RQAMD_STACK='1', RQAMD_STACK_ROWS=os.environ.get('RQ_ROWS', '256'))
l1 = ar1(codes, model_aux=vae, cond=cond).float()
print(f'stack kernel, grid barriers       vs per-launch kernels: max |dlogit| {(l1 - l0).abs().max().item():.5f}  mean {(l1 - l0).abs().mean().item():.6f}', flush=True)
l1b = ar1(codes, model_aux=vae, cond=cond).float()
print(f'stack kernel run twice: identical = {bool(torch.equal(l1, l1b))}', flush=True)


def t_sample(ar, Bs, reps=2):
    ps = torch.zeros((Bs, 8, 8, 4), dtype=torch.long, device=dev)
    cd = torch.zeros((Bs, 1), dtype=torch.long, device=dev)
    ar.sample(ps, model_aux=vae, cond=cd, top_k=1024, top_p=0.95)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = ar.sample(ps, model_aux=vae, cond=cd, top_k=1024, top_p=0.95)
    torch.cuda.synchronize()
    assert int(out.min()) >= 0 and int(out.max()) < 16384
    return (time.perf_counter() - t0) / reps


for Bs in [int(b) for b in os.environ.get('RQ_BS', '16,64,128,256').split(',')]:
    a, b = t_sample(ar0, Bs), t_sample(ar1, Bs)
    print(f'B={Bs:4d}: per-launch {a * 1e3:7.1f} ms ({Bs / a:6.1f} img/s AR only) | stack kernel {b * 1e3:7.1f} ms ({Bs / b:6.1f} img/s)  x{a / b:.2f}', flush=True)
