#!/usr/bin/env python
"""rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE CSVs of scripts/gemm_traffic.py (separate passes, scripts/gpu.sh pmc) ->
the traffic JSON that bench.py reads for roofline.traffic (copy it to profiles/rNN_gemm_traffic_m<M>.json).
Usage: python scripts/summarize_traffic.py <M> <out.json>"""
import csv
import glob
import json
import sys

M = int(sys.argv[1])
dst = sys.argv[2]
# scripts/gemm_traffic.py launches 6 GEMMs per shape, shapes in a fixed order: group the GEMM dispatches by order
# (several shapes share a kernel instantiation and even a grid, so names cannot tell them apart)
per = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    rows = []
    for f in glob.glob(f'gpurun_out/pmc_{c}/**/*counter_collection.csv', recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == c and 'gemm_' in r['Kernel_Name']]
    disp = {}
    for r in rows:          # one row per dispatch (a counter may be reported per XCD/instance: sum them)
        d = disp.setdefault(int(r['Dispatch_Id']), {'kernel': r['Kernel_Name'].split('(')[0], 'grid': r.get('Grid_Size', '?'), 'v': 0.0})
        d['v'] += float(r['Counter_Value'])
    per[c] = [disp[k] for k in sorted(disp)]
    print(c, 'gemm dispatches:', len(per[c]))
n = min(len(per['FETCH_SIZE']), len(per['WRITE_SIZE']))
assert n % 6 == 0 and n >= 30, n
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'rq-vae-transformer_amd'))
from rqvae import _native  # noqa: E402
fused = not os.environ.get('RQAMD_NO_FUSE_RESID') and M >= 2048      # (below 2048 rows the engine's tile choice splits K: slab epilogue, as gemm_traffic.py launches it)
rb = 8 if fused else 4          # residual epilogue: fp32 stream read + written in place; slab epilogue: fp32 slabs written (counted as one: the ratio then includes the extra slabs)
shapes = [('qkv', 4608, 1536, 2, 4224), ('proj (in-place residual epilogue)' if fused else 'proj (split-K slab)', 1536, 1536, rb, 4224),
          ('fc1', 6144, 1536, 2, 4224), ('fc2 (in-place residual epilogue)' if fused else 'fc2 (split-K slab)', 1536, 6144, rb, 4224),
          ('classifier', 16384, 1536, 4, 256)]
out = {'source': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over scripts/gemm_traffic.py on MI355X; '
                 'hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024 (gfx950: FETCH_SIZE counts half of wide coalesced reads, '
                 'MI355X_MICROARCH.md HBM section; Infinity-Cache hits are included in this fabric-side counter)',
       'batch_rows': M, 'kernel_sources_sha16': _native.kernel_source_hash(), 'kernel_sources': 'csrc/gemm.h + csrc/gemm.hip', 'shapes': []}
tot = totw = 0
for i, (name, N, K, ob, w) in enumerate(shapes):
    s0 = 6 * i
    f = sum(d['v'] for d in per['FETCH_SIZE'][s0:s0 + 6]) / 6
    wr = sum(d['v'] for d in per['WRITE_SIZE'][s0:s0 + 6]) / 6
    hbm = (2 * f + wr) * 1024
    alg = N * K * 2 + M * K * 2 + M * N * ob
    out['shapes'].append({'gemm': f'{name} (M={M},N={N},K={K})', 'kernel': per['FETCH_SIZE'][s0 + 5]['kernel'], 'grid': per['FETCH_SIZE'][s0 + 5]['grid'],
                          'launches_per_batch': w, 'algorithmic_bytes': alg, 'FETCH_SIZE_KiB': f, 'WRITE_SIZE_KiB': wr,
                          'hbm_bytes_per_launch': hbm, 'ratio': hbm / alg})
    tot += w * hbm
    totw += w
out['hbm_bytes_per_launch_weighted'] = tot / totw
json.dump(out, open(dst, 'w'), indent=1)
print(f'M={M}: weighted {tot / totw / 1e6:.1f} MB per launch;', [(s['gemm'].split()[0], round(s['ratio'], 2)) for s in out['shapes']])
