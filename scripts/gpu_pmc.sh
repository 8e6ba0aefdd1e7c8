#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters (two separate passes, MI355X_MICROARCH.md §HBM)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
B=${B:-2048}
mkdir -p gpurun_out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  RQAMD_GRAPH=0 timeout 900 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_$c -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --batch $B --no-cpu-baseline --no-profile > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections, json
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(f'gpurun_out/pmc_{c}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != c: continue
            k = r['Kernel_Name'].split('(')[0]
            acc[k] += float(r['Counter_Value']); cnt[k] += 1
    for k in acc:
        out.setdefault(k, {})[c] = acc[k]; out[k]['launches'] = cnt[k]
res = {'batch': $B, 'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), 1 sample+decode pass; counters in KiB; gfx950 FETCH_SIZE counts half of wide coalesced reads (x2 applied in hbm_bytes)', 'kernels': {}}
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get('FETCH_SIZE', 0)):
    if v.get('launches', 0) == 0: continue
    f, w, n = v.get('FETCH_SIZE', 0.0), v.get('WRITE_SIZE', 0.0), v['launches']
    res['kernels'][k] = {'launches': n, 'fetch_KiB_per_launch': f / n, 'write_KiB_per_launch': w / n, 'hbm_bytes_per_launch': (2 * f + w) * 1024 / n}
json.dump(res, open('gpurun_out/gemm_traffic.json', 'w'), indent=1)
for k, v in list(res['kernels'].items())[:12]:
    print(k[:70], v)
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
