#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_$c -o pmc --output-format csv -- python $R/scripts/gemm_traffic.py > $R/gpurun_out/pmc_$c.log 2>&1
  echo "$c exit $?"
done
cd $R
grep algorithmic gpurun_out/pmc_FETCH_SIZE.log
python - <<'PY'
import csv, glob, json
# scripts/gemm_traffic.py launches 6 GEMMs per shape, shapes in a fixed order: group the GEMM dispatches by order
# (several shapes share a kernel instantiation and even a grid, so names cannot tell them apart)
per = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    rows = []
    for f in glob.glob(f'gpurun_out/pmc_{c}/**/*counter_collection.csv', recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == c and 'gemm_' in r['Kernel_Name'] and 'Args' in r['Kernel_Name'] or
                 (r['Counter_Name'] == c and ('gemm_bf16_kernel' in r['Kernel_Name'] or 'gemm_rb_kernel' in r['Kernel_Name']))]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    # one row per dispatch (a counter may be reported per XCD/instance: sum them)
    disp = {}
    for r in rows:
        d = disp.setdefault(int(r['Dispatch_Id']), {'kernel': r['Kernel_Name'].split('(')[0], 'grid': r.get('Grid_Size', '?'), 'v': 0.0})
        d['v'] += float(r['Counter_Value'])
    per[c] = [disp[k] for k in sorted(disp)]
    print(c, 'gemm dispatches:', len(per[c]))
n = min(len(per['FETCH_SIZE']), len(per['WRITE_SIZE']))
assert n % 6 == 0 and n >= 30, n
out = []
for s0 in range(0, 30, 6):
    f = sum(d['v'] for d in per['FETCH_SIZE'][s0:s0 + 6]) / 6
    w = sum(d['v'] for d in per['WRITE_SIZE'][s0:s0 + 6]) / 6
    k = per['FETCH_SIZE'][s0 + 5]
    out.append({'kernel': k['kernel'], 'grid': k['grid'], 'launches': 6, 'FETCH_SIZE_KiB': f, 'WRITE_SIZE_KiB': w,
                'hbm_bytes_per_launch_x2fetch': (2 * f + w) * 1024})
    print(out[-1])
json.dump(out, open('gpurun_out/gemm_traffic.json', 'w'), indent=1)
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
