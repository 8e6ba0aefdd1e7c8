#!/usr/bin/env python
"""GPU microbenchmark of the sampler kernels (rqamd_sample_logits) at the 1.4B decode shape: 4096 x 16384 logits."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch  # noqa: E402
from rqvae import _native  # noqa: E402

rows, V = int(os.environ.get('RQ_ROWS', 4096)), 16384
g = torch.Generator(device='cuda').manual_seed(0)
logits = 2.5 * torch.randn((rows, V), device='cuda', generator=g)
for k, p in ((None, None), (1024, None), (None, 0.95), (1024, 0.95), (1024, 1.0), (100, 0.5)):
    for _ in range(2):
        _native.sample_logits(logits, 1.0, k, p, seed=1, offset=0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(10):
        _native.sample_logits(logits, 1.0, k, p, seed=1, offset=4 * i)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    print(f'rows {rows} V {V} top_k {k} top_p {p}: {us:8.1f} us  ({rows * V * 4 / us / 1e3:6.0f} GB/s of logits)', flush=True)
