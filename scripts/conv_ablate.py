#!/usr/bin/env python
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from conv_bench import run
for H, Cin, Cout, B in ((256, 128, 128, 8), (64, 256, 256, 8)):
    for name, fl in (('full', 0), ('no epilogue', 1), ('no epi, no staging (LDS reads + MFMA + barriers)', 3), ('no epi, loads but no LDS writes', 5)):
        a = run(B, H, Cin, Cout, flags=fl)
        print(f'{Cin}->{Cout}@{H}: {name:50s} {a[0]:7.1f} us {a[1]:6.1f} TF', flush=True)
