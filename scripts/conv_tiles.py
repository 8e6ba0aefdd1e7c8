#!/usr/bin/env python
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from conv_bench import run
for H, Cin, Cout, B in ((256, 128, 128, 8), (128, 256, 128, 8), (64, 256, 256, 8), (16, 512, 512, 32)):
    for bm, bn in ((128, 128), (128, 64), (64, 128), (64, 64)):
        a = run(B, H, Cin, Cout, bm=bm, bn=bn)
        b = run(B, H, Cin, Cout, bm=bm, bn=bn, flags=1)
        print(f'{Cin}->{Cout}@{H} tile {bm}x{bn}: full {a[0]:7.1f} us {a[1]:6.1f} TF | no-epi {b[0]:7.1f} us {b[1]:6.1f} TF', flush=True)
