#!/usr/bin/env python
"""Probe (round 5): does the 256 x 256 GEMM's operand intake depend on the ROW STRIDE of its operands?  Rows of A and W are K * 2 bytes
apart (3072 / 12288 bytes for the 1.4B model): if the L2's channel selection folds such strides onto a few channels, padding the
leading dimension would spread them.  Needs a variant library built from a patched copy of csrc/ (RQ_LDPAD = elements of padding; the
tree's own kernels take W as [N][K] dense): RQ_LIB=<variant> RQ_PAD=<elements> python scripts/gemm_ldpad_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch
from rqvae import _native
if os.environ.get('RQ_LIB'):
    _native.LIB_PATH = os.environ['RQ_LIB']
PAD = int(os.environ.get('RQ_PAD', 0))
M = int(os.environ.get('RQ_M', 10752))
dev = 'cuda'
lib = _native.lib()
g = torch.Generator(device=dev).manual_seed(1)
for N, K in ((4608, 1536), (6144, 1536), (1536, 6144), (1536, 1536), (7680, 2560), (2560, 10240)):
    A = torch.randn((M, K + PAD), device=dev, generator=g).to(torch.bfloat16)
    W = (0.05 * torch.randn((N, K + PAD), device=dev, generator=g)).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)

    def run():
        _native.check(lib.rqamd_dbg_gemm_bf16(_native.ptr(A, torch.bfloat16), _native.ptr(W, torch.bfloat16), M, N, K, None, 0,
                                              _native.ptr(out), 256, 256, 1, _native.stream_of(A)))
    run()
    ref = A[:512, :K].float() @ W[:, :K].float().t()
    err = float((out[:512].float() - ref).abs().max() / ref.abs().max())
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f'pad {PAD:3d}  M {M} N {N:5d} K {K:5d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF   rel err {err:.1e}', flush=True)
