#!/usr/bin/env python
"""Turn gpurun_out/gemm_traffic.json (per-kernel PMC sums from scripts/gpu_pmc2.sh) into the committed
profiles/r01_gemm_traffic_m<M>.json that bench.py reads for roofline.traffic."""
import json
import sys

M = int(sys.argv[1])
src = sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out/gemm_traffic.json'
t = json.load(open(src))
shapes = [('qkv', 4608, 1536, 2, 4224), ('proj (split-K slabs)', 1536, 1536, 4, 4224), ('fc1', 6144, 1536, 2, 4224),
          ('fc2 (split-K slabs)', 1536, 6144, 4, 4224), ('classifier', 16384, 1536, 4, 256)]
merged = False
assert len(t) == len(shapes), (len(t), 'PMC rows; expected one per GEMM shape in launch order')
out = {'source': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over scripts/gemm_traffic.py on MI355X; '
                 'hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024 (gfx950: FETCH_SIZE counts half of wide coalesced reads, '
                 'MI355X_MICROARCH.md HBM section; Infinity-Cache hits are included in this fabric-side counter)',
       'batch_rows': M, 'shapes': []}
tot = totw = 0
for (name, N, K, ob, w), k in zip(shapes, t):
    alg = N * K * 2 + M * K * 2 + M * N * ob
    out['shapes'].append({'gemm': f'{name} (M={M},N={N},K={K})', 'kernel': k['kernel'], 'launches_per_batch': w,
                          'algorithmic_bytes': alg, 'hbm_bytes_per_launch': k['hbm_bytes_per_launch_x2fetch'],
                          'ratio': k['hbm_bytes_per_launch_x2fetch'] / alg})
    tot += w * k['hbm_bytes_per_launch_x2fetch']
    totw += w
out['hbm_bytes_per_launch_weighted'] = tot / totw
if merged:
    out['note'] = 'proj and fc2 share one PMC row (same kernel instantiation and grid); its value is the mean of the two shapes'
json.dump(out, open(f'profiles/r01_gemm_traffic_m{M}.json', 'w'), indent=1)
print(f'M={M}: weighted {tot / totw / 1e6:.1f} MB per launch;', [(s['gemm'].split()[0], round(s['ratio'], 2)) for s in out['shapes']])
