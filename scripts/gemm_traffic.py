#!/usr/bin/env python
"""Launch each decode-step GEMM shape (1.4B layer shapes at M = batch rows) a few times with rotating weights;
run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` to get HBM traffic per launch (scripts/gpu_pmc2.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch  # noqa: E402
from rqvae import _native  # noqa: E402

M = int(os.environ.get('RQ_M', 2048))
shapes = [('qkv', 4608, 1536, 0), ('proj', 1536, 1536, 4), ('fc1', 6144, 1536, 1), ('fc2', 1536, 6144, 4), ('cls', 16384, 1536, 3)]
for name, N, K, epi in shapes:
    a = torch.randn((M, K), device='cuda').to(torch.bfloat16)
    ws = [torch.randn((N, K), device='cuda').to(torch.bfloat16) for _ in range(6)]
    bias = torch.randn((N,), device='cuda')
    out = None
    if epi == 4 and M >= 2048 and not os.environ.get('RQAMD_NO_FUSE_RESID'):      # (below 2048 rows the engine's tile choice splits K: slabs)
        # proj / fc2 as the engine launches them when K is not split: the fp32 residual stream updated in place by the epilogue
        epi, out = 4 + 2048, torch.randn((M, N), device='cuda')
    for i in range(6):
        out = _native.dbg_gemm(a, ws[i], None if epi == 4 else bias, epi, 0, 0, 0 if epi < 2048 else 1, out=out)
    torch.cuda.synchronize()
    print(name, M, N, K, 'algorithmic bytes', N * K * 2 + M * K * 2 + M * N * (8 if epi >= 2048 else 4 if epi >= 3 else 2))
