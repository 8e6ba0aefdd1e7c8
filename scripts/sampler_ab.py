#!/usr/bin/env python
"""Sampler kernels of two library builds on the same logits and seeds: identical draws and filtered probabilities (the multi-bit threshold
searches of round 5 must not move a single decision), and the time per call.  RQ_LIB_A / RQ_LIB_B = the two libraries (B defaults to the
in-tree one); two processes are not needed: each library is its own ctypes handle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch  # noqa: E402
from rqvae import _native  # noqa: E402

libs = {'A': _native._bind(os.environ['RQ_LIB_A']), 'B': _native._bind(os.environ.get('RQ_LIB_B', _native.LIB_PATH))}


def run(which, *a, **k):
    _native._lib = libs[which]
    return _native.sample_logits(*a, **k)


g = torch.Generator(device='cuda').manual_seed(0)
for rows, V in ((10752, 16384), (500, 16384), (4096, 2048)):
    base = 2.5 * torch.randn((rows, V), device='cuda', generator=g)
    variants = {'gaussian': base, 'heavy ties': (base * 2).round() / 2, 'peaked': base * 4}
    for name, logits in variants.items():
        for T, k, p in ((1.0, 1024, 0.95), (0.8, 100, 0.5), (1.0, 50, 0.9), (1.0, 1024, 1.0), (1.0, None, 0.95)):
            if k is not None and k >= V:
                continue
            outs = {}
            for w in ('A', 'B'):
                idx, probs = run(w, logits, T, k, p, seed=3, offset=8, want_probs=True)
                outs[w] = (idx.clone(), probs.clone())
            same_idx = bool(torch.equal(outs['A'][0], outs['B'][0]))
            same_pr = bool(torch.equal(outs['A'][1], outs['B'][1]))
            t = {}
            for w in ('A', 'B', 'A', 'B'):
                for _ in range(2):
                    run(w, logits, T, k, p, seed=1, offset=0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(8):
                    run(w, logits, T, k, p, seed=1, offset=4 * i)
                e1.record()
                e1.synchronize()
                t[w] = min(t.get(w, 1e9), e0.elapsed_time(e1) * 1e3 / 8)
            print(f'rows {rows:5d} V {V:5d} {name:10s} T {T} top_k {k} top_p {p}: draws equal {same_idx}, probabilities equal {same_pr}; '
                  f'A {t["A"]:7.1f} us  B {t["B"]:7.1f} us', flush=True)
