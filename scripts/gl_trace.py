#!/usr/bin/env python
"""Cycle timeline of one workgroup of the LDS-DMA tiled GEMM (gemm_bf16_kernel, plain loop) at a mid-batch shape.  Needs a variant
built with  scripts/build_variant.py --out rq-vae-transformer_amd/variants/librqamd_gltrace.so --file-flags gemm.hip=-DRQ_GL_TRACE=9
Per wavefront: set-up, first DMA burst, then for the first six K-tiles [wait vmcnt | barrier | DMA issue | fragment reads + MFMAs],
the whole loop, the epilogue.  RQ_MODE: 0 whole kernel, 2 no staging (-dma), 4 no MFMA side (-mma).  RQ_NROT=1: warm weights."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch  # noqa: E402
from rqvae import _native  # noqa: E402
_native.LIB_PATH = os.environ.get('RQ_LIB', os.path.join(ROOT, 'rq-vae-transformer_amd', 'variants', 'librqamd_gltrace.so'))
_native.lib()
fn = C.CDLL(_native.LIB_PATH).rqamd_dbg_gl_trace
fn.argtypes = [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * 512)()
dev = 'cuda'
M = int(os.environ.get('RQ_M', 500))
bm, bn = [int(v) for v in os.environ.get('RQ_TILE', '136x128').split('x')]
nw = {132: 8, 136: 16, 264: 16}.get(bm, 4)
for mode in [int(v) for v in os.environ.get('RQ_MODE', '0,2,4').split(',')]:
    for (name, N, K, epi) in (('qkv', 4608, 1536, 0),):
        nrot = int(os.environ.get('RQ_NROT', 30))
        a = torch.randn((M, K), device=dev).to(torch.bfloat16)
        ws = [(torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16) for _ in range(nrot)]
        bias = torch.randn((N,), device=dev)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        ecode = epi + 32 * 3 + (4096 if mode == 2 else 8192 if mode == 4 else 0)
        for i in range(2 * nrot + 1):
            _native.dbg_gemm(a, ws[i % nrot], bias, ecode, bm, bn, 1, out=out)
        torch.cuda.synchronize()
        assert fn(buf) == 0
        print(f'== {name} M={M} tile {bm}x{bn} mode {mode} ({"cold" if nrot > 1 else "warm"} weights): cycles per wavefront')
        for wv in range(nw):
            t = [buf[wv * 32 + i] for i in range(28)]
            tiles = ' '.join('[' + ' '.join(f'{t[3 + 4 * k + j] - t[2 + 4 * k + j]:4d}' for j in range(3)) + f' | wait {t[2 + 4 * k] - (t[1 + 4 * k] if k else t[1]):5d}]' for k in range(6))
            print(f'w{wv:2d}: first burst {t[1] - t[0]:5d} | tiles 0..5 [barrier issue compute | wait before] {tiles} | loop {t[26] - t[1]:6d} | epilogue {t[27] - t[26]:6d}')
