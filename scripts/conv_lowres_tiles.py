#!/usr/bin/env python
"""The implicit-GEMM conv at the decoder's 8^2 / 16^2 levels (512 -> 512 channels), 64 and 256 images, by tile (round 5 probe: these layers
are ~2.7 % of the headline step at 0.18-0.32 of the MFMA peak; 256 images at once run the 8^2 level at 806 TFLOP/s against 281-389 at 64)."""
import os, sys
sys.path.insert(0, '/root/repo/scripts')
from conv_bench import run
for H, Cin, Cout, B in ((16, 512, 512, 64), (8, 512, 512, 64), (16, 512, 512, 256), (8, 512, 512, 256)):
    for bm, bn in ((128, 128), (256, 128), (128, 64), (64, 128)):
        try:
            a = run(B, H, Cin, Cout, bm=bm, bn=bn)
            c = run(B, H, Cin, Cout, bm=bm, bn=bn, resid=True)
            print(f'{Cin}->{Cout}@{H} x{B} tile {bm}x{bn}: {a[0]:7.1f} us {a[1]:6.1f} TF | resid {c[0]:7.1f} us {c[1]:6.1f} TF', flush=True)
        except Exception as e:
            print(bm, bn, repr(e)[:100])
