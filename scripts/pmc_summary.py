#!/usr/bin/env python
"""Sum rocprofv3 --pmc counter CSVs per kernel name: python scripts/pmc_summary.py <dir> [name filter]"""
import csv, glob, sys
from collections import defaultdict
d, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'gemm')
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if flt not in k:
            continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k].add(r['Dispatch_Id'])
for k, v in acc.items():
    n = len(cnt[k])
    print(k, 'dispatches', n)
    for c, x in sorted(v.items()):
        print(f'   {c:32s} {x / n:16.1f} per dispatch')
