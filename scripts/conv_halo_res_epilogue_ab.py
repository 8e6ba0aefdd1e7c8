#!/usr/bin/env python
"""A/B (round 5) of the residual epilogue of the per-tile halo conv: RQ_LIB_A / RQ_LIB_B = two builds of the library (B: -DRQ_HALO_RES_F32=0, the
residual's round trip through the bf16 LDS tile).  Bit-identity of the outputs and the GroupNorm statistics, then time per launch, alternating."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch
from rqvae import _native
dev = 'cuda'
libs = {k: C.CDLL(os.environ[k]) for k in ('RQ_LIB_A', 'RQ_LIB_B')}
for l in libs.values():
    l.rqamd_dbg_conv_halo_bf16.restype = C.c_int
    l.rqamd_dbg_conv_halo_bf16.argtypes = _native._SIGS['rqamd_dbg_conv_halo_bf16'][1]


def run(l, x, w, bias, gn, resid, out, stats):
    B, H, W, Cin = x.shape
    rc = l.rqamd_dbg_conv_halo_bf16(_native.ptr(x, torch.bfloat16), _native.ptr(w, torch.bfloat16), _native.ptr(bias, torch.float32), _native.ptr(gn),
                                    _native.ptr(resid), B, H, W, Cin, w.shape[0], 0, _native.ptr(out), _native.ptr(stats), _native.stream_of(x))
    assert rc == 0, rc


g = torch.Generator(device=dev).manual_seed(3)
for B, H, Cin, Cout in ((8, 256, 128, 128), (8, 128, 256, 128), (8, 128, 256, 256), (8, 64, 256, 256), (16, 64, 512, 256)):
    x = torch.randn((B, H, H, Cin), device=dev, generator=g).to(torch.bfloat16)
    w = (0.05 * torch.randn((Cout, 3, 3, Cin), device=dev, generator=g)).to(torch.bfloat16)
    bias = torch.randn((Cout,), device=dev, generator=g)
    resid = torch.randn((B, H, H, Cout), device=dev, generator=g).to(torch.bfloat16)
    gn = torch.stack([1 + 0.2 * torch.randn((B, Cin), device=dev, generator=g), 0.3 * torch.randn((B, Cin), device=dev, generator=g)], -1).contiguous()
    res = {}
    for fused in (True, False):
        outs = {}
        for k, l in libs.items():
            out = torch.empty_like(resid)
            st = torch.zeros((B, (H // 8) * (H // 32), 32, 2), device=dev)
            run(l, x, w, bias, gn if fused else None, resid, out, st)
            outs[k] = (out.clone(), st.clone())
        same = torch.equal(outs['RQ_LIB_A'][0], outs['RQ_LIB_B'][0]) and torch.equal(outs['RQ_LIB_A'][1], outs['RQ_LIB_B'][1])
        t = {k: [] for k in libs}
        out = torch.empty_like(resid)
        st = torch.zeros((B, (H // 8) * (H // 32), 32, 2), device=dev)
        for rep in range(4):
            for k, l in libs.items():
                for _ in range(3):
                    run(l, x, w, bias, gn if fused else None, resid, out, st)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    run(l, x, w, bias, gn if fused else None, resid, out, st)
                e1.record(); e1.synchronize()
                t[k].append(e0.elapsed_time(e1) * 1e3 / 20)
        a, b = min(t['RQ_LIB_A']), min(t['RQ_LIB_B'])
        print(f'B{B} {Cin}->{Cout}@{H} {"GN+SiLU+resid+stats" if fused else "resid+stats         "}: identical {same} | fp32 tile {a:7.1f} us | bf16 tile {b:7.1f} us | {100 * (a / b - 1):+.1f} %', flush=True)
