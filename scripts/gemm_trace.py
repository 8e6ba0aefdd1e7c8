#!/usr/bin/env python
"""Epilogue timeline of one workgroup of the eight-phase GEMM (needs librqamd_trace.so from scripts/gemm_trace.sh)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch
from rqvae import _native
_native.LIB_PATH = os.path.join(ROOT, 'rq-vae-transformer_amd', 'librqamd_trace.so')
_native.lib()
dev = 'cuda'
fn = C.CDLL(_native.LIB_PATH).rqamd_dbg_gemm_trace
fn.argtypes = [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * 128)()
for (name, M, N, K, epi) in (('qkv', 10752, 4608, 1536, 0), ('fc1', 10752, 6144, 1536, 1)):
    a = torch.randn((M, K), device=dev).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn((N,), device=dev)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        _native.dbg_gemm(a, w, bias, epi, 256, 256, 1, out=out)
    torch.cuda.synchronize()
    assert fn(buf) == 0
    print(f'== {name}: cycles from kernel start; [main loop end] [wait at barrier 1] [pack -> LDS] [wait at barrier 2] [LDS -> global stores]')
    for wv in range(8):
        ph = [buf[wv * 16 + i] for i in range(6, 11)]
        print(f'wave {wv}: K-tile 10 phases: ' + ' '.join(str(ph[i + 1] - ph[i]) for i in range(4)) + f' (sum {ph[4] - ph[0]})')
    for wv in range(8):
        t = [buf[wv * 16 + i] for i in range(6)]
        print(f'wave {wv}: main loop {t[1] - t[0]:6d} | barrier {t[2] - t[1]:5d} | pack {t[3] - t[2]:5d} | barrier {t[4] - t[3]:5d} | store loop {t[5] - t[4]:5d} | total {t[5] - t[0]:6d}')
