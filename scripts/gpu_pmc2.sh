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
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(f'gpurun_out/pmc_{c}/**/*counter_collection.csv', recursive=True):
        rows = list(csv.DictReader(open(f)))
        if rows: print(c, 'columns:', list(rows[0].keys()))
        for r in rows:
            if r['Counter_Name'] != c or 'gemm_bf16' not in r['Kernel_Name']: continue
            key = (r['Kernel_Name'].split('(')[0], r.get('Grid_Size', '?'))
            acc[key][c] += float(r['Counter_Value']); cnt[key][c] += 1
out = []
for k in acc:
    n = max(cnt[k].values())
    f, w = acc[k]['FETCH_SIZE'] / max(cnt[k]['FETCH_SIZE'], 1), acc[k]['WRITE_SIZE'] / max(cnt[k]['WRITE_SIZE'], 1)
    out.append({'kernel': k[0], 'grid': k[1], 'launches': n, 'FETCH_SIZE_KiB': f, 'WRITE_SIZE_KiB': w, 'hbm_bytes_per_launch_x2fetch': (2 * f + w) * 1024})
    print(out[-1])
json.dump(out, open('gpurun_out/gemm_traffic.json', 'w'), indent=1)
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
