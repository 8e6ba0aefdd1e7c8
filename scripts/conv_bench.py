#!/usr/bin/env python
"""GPU microbenchmark + correctness check of the implicit-GEMM conv (diagnostics ABI entry)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from rqvae import _native  # noqa: E402
if os.environ.get('RQ_LIB'):          # A/B a differently-built kernel library (diagnostics only)
    _native.LIB_PATH = os.environ['RQ_LIB']


def run(B, H, Cin, Cout, ks=3, stride=1, ups=0, resid=False, flags=0, bm=0, bn=0, reps=10, check=False):
    dev = 'cuda'
    x = torch.randn((B, H >> ups, H >> ups, Cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((Cout, ks, ks, Cin), device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn((Cout,), device=dev)
    Ho = H // 2 if stride == 2 else H
    r = torch.randn((B, Ho, Ho, Cout), device=dev).to(torch.bfloat16) if resid else None
    out = _native.dbg_conv(x, w, bias, r, ks, stride, ups, bm, bn, flags)
    err = None
    if check:
        xi = x.float().permute(0, 3, 1, 2)
        if ups:
            xi = F.interpolate(xi, scale_factor=2.0, mode='nearest')
        wi = w.float().permute(0, 3, 1, 2)
        if stride == 2:
            ref = F.conv2d(F.pad(xi, (0, 1, 0, 1)), wi, bias, stride=2)
        else:
            ref = F.conv2d(xi, wi, bias, padding=ks // 2)
        ref = ref.permute(0, 2, 3, 1)
        if resid:
            ref = ref + r.float()
        err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
    for _ in range(2):
        _native.dbg_conv(x, w, bias, r, ks, stride, ups, bm, bn, flags, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _native.dbg_conv(x, w, bias, r, ks, stride, ups, bm, bn, flags, out=out)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    tf = 2.0 * B * Ho * Ho * Cout * ks * ks * Cin / us / 1e6
    return us, tf, err


if __name__ == '__main__':
    quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
    print('correctness (small):')
    for args in ((2, 16, 64, 128, 3, 1, 0, False), (2, 16, 128, 64, 3, 1, 1, True), (2, 16, 64, 64, 3, 2, 0, False), (3, 8, 128, 192, 1, 1, 0, True)):
        us, tf, err = run(*args, check=True, reps=2)
        print('  B%d H%d %d->%d k%d s%d ups%d resid=%s: rel err %.2e' % (*args, err))
    print('decoder shapes, batch 8 (us / TFLOP/s):  full | no-epilogue | +resid')
    for H, Cin, Cout in ((256, 128, 128), (128, 128, 128), (128, 256, 128), (64, 256, 256), (32, 256, 256), (32, 512, 256), (16, 512, 512), (8, 512, 512)):
        B = 8 if H >= 64 else 32
        a = run(B, H, Cin, Cout)
        b = run(B, H, Cin, Cout, flags=1)
        c = run(B, H, Cin, Cout, resid=True)
        print(f'  {Cin:4d}->{Cout:4d} @{H:3d}^2 B{B}: {a[0]:8.1f} us {a[1]:6.1f} TF | {b[0]:8.1f} us {b[1]:6.1f} TF | {c[0]:8.1f} us {c[1]:6.1f} TF', flush=True)
        if quick:
            break
