#!/usr/bin/env python
"""Where a small-batch decode GEMM launch spends its time (round 4): per-workgroup constant-clock stamps of gemm_stream_kernel
(diagnostics build, `python scripts/build_variant.py --out rq-vae-transformer_amd/variants/librqamd_strace.so --file-flags
gemm.hip=-DRQ_STREAM_TRACE`).  For each of the four layer shapes: one launch on cold weights (a fresh matrix, never touched),
stamps of every workgroup's wavefront 0 and 3: entry, first DMA burst issued, first K-tile landed, main loop done, after the
barrier, partial tiles in LDS, stores issued, stores acknowledged.  Printed relative to the earliest entry stamp of the launch."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch  # noqa: E402
from rqvae import _native  # noqa: E402

_native.LIB_PATH = os.environ.get('RQ_LIB', os.path.join(ROOT, 'rq-vae-transformer_amd', 'variants', 'librqamd_strace.so'))
lib = _native.lib()
lib.rqamd_dbg_stream_trace.restype = C.c_int
lib.rqamd_dbg_stream_trace.argtypes = [C.c_void_p, C.c_int]
dev = 'cuda'
E = int(os.environ.get('RQ_E', 1536))
M = int(os.environ.get('RQ_M', 64))
NAMES = ['entry', 'issued', 'tile0', 'loop', 'barrier', 'in LDS', 'stored', 'acked']


def trace(name, N, K, epi, sk):
    a = torch.randn((M, K), device=dev).to(torch.bfloat16)
    ws = [(torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16) for _ in range(12)]
    bias = None if epi == 4 else torch.randn((N,), device=dev)
    bm = 66 if M <= 64 else 130
    out = _native.dbg_gemm(a, ws[0], bias, epi, bm, 32, sk)
    nwg = (N + 31) // 32 * sk
    for mode in ('cold', 'warm'):
        rows = []
        for rep in range(4):
            if mode == 'cold':
                junk = torch.randn((64 << 20,), device=dev)       # 256 MB: push the weights out of the Infinity Cache
                junk.mul_(1.0001)
                del junk
            torch.cuda.synchronize()
            lib.rqamd_dbg_stream_trace(None, 1)
            # warm: the last of 12 back-to-back launches on rotating weights (instruction / scalar caches hot, as inside the graphs)
            for i in range(1 if mode == 'cold' else 12):
                _native.dbg_gemm(a, ws[(rep + i) % 12], bias, epi, bm, 32, sk, out=out)
            torch.cuda.synchronize()
            buf = np.zeros(1024 * 2 * 24, np.uint64)
            _native.check(lib.rqamd_dbg_stream_trace(buf.ctypes.data, 0))
            t = buf.reshape(1024, 2, 24).astype(np.int64)[:nwg]
            rows.append((t - t[:, :, 0].min()) * 0.01)             # 100 MHz -> us since the first wavefront of the launch entered
        rel = np.stack(rows)                                       # (rep, wg, wave, slot)
        print(f'{name:4s} M={M} N={N} K={K} splitk={sk}: {nwg} workgroups, {mode}; us since the first wavefront entered (median over 4 launches of: '
              f'min / median / max over workgroups)')
        for w in (0, 1):
            line = []
            for sl in range(8):
                v = rel[:, :, w, sl]
                line.append(f'{NAMES[sl]} {np.median(v.min(1)):.2f}/{np.median(np.median(v, 1)):.2f}/{np.median(v.max(1)):.2f}')
            print(f'   wave {0 if w == 0 else 3}: ' + ' | '.join(line))
        # per K-tile of wavefront 0: (tile landed, MFMAs issued) relative to the workgroup's own entry, median over workgroups and launches
        own = rel[:, :, 0, :] - rel[:, :, 0, 0:1]
        its = []
        for i in range(8):
            if rel[:, :, 0, 8 + 2 * i].max() > 0:
                its.append(f'{np.median(own[:, :, 8 + 2 * i]):.2f}>{np.median(own[:, :, 9 + 2 * i]):.2f}')
        print('   wave 0 per K-tile, us since its own entry (landed>mfma done): ' + ' '.join(its))
    # per launch inside a back-to-back chain (events around 48 launches)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(12):
        _native.dbg_gemm(a, ws[i % 12], bias, epi, bm, 32, sk, out=out)
    e0.record()
    for i in range(48):
        _native.dbg_gemm(a, ws[i % 12], bias, epi, bm, 32, sk, out=out)
    e1.record()
    e1.synchronize()
    print(f'   {e0.elapsed_time(e1) * 1e3 / 48:.2f} us per launch, 48 back-to-back eager launches (with the stamps)')
    sys.stdout.flush()


for name, N, K, epi, sk in (('qkv', 3 * E, E, 0, 1), ('proj', E, E, 4, 4), ('fc1', 4 * E, E, 1, 1), ('fc2', E, 4 * E, 4, 4)):
    trace(name, N, K, epi, sk)
