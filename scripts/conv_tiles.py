#!/usr/bin/env python
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from conv_bench import run
print('correctness 256x128 tile:')
for args in ((2, 16, 64, 128, 3, 1, 0, False), (2, 16, 128, 128, 3, 1, 1, True), (5, 16, 64, 128, 3, 2, 0, False), (3, 8, 128, 192, 1, 1, 0, True)):
    us, tf, err = run(*args, check=True, reps=2, bm=256, bn=128)
    print('  B%d H%d %d->%d k%d s%d ups%d resid=%s: rel err %.2e' % (*args, err))
for H, Cin, Cout, B in ((256, 128, 128, 8), (128, 256, 128, 8), (64, 256, 256, 8), (16, 512, 512, 32)):
    for bm, bn in ((128, 128), (256, 128)):
        a = run(B, H, Cin, Cout, bm=bm, bn=bn)
        b = run(B, H, Cin, Cout, bm=bm, bn=bn, flags=1)
        c = run(B, H, Cin, Cout, bm=bm, bn=bn, resid=True)
        print(f'{Cin}->{Cout}@{H} tile {bm}x{bn}: full {a[0]:7.1f} us {a[1]:6.1f} TF | no-epi {b[0]:7.1f} us {b[1]:6.1f} TF | resid {c[0]:7.1f} us {c[1]:6.1f} TF', flush=True)
