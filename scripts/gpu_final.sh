#!/bin/bash
# end-of-round evidence: default bench line + kernel-trace stats of the same command + PMC traffic at the bench batch
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1
tail -1 gpurun_out/bench_default.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof.log 2>&1
cd $R
python - <<'PY'
import sqlite3, glob
for f in glob.glob('gpurun_out/prof/**/*.db', recursive=True) + glob.glob('gpurun_out/prof/*.db'):
    db = sqlite3.connect(f)
    rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open('gpurun_out/kernel_stats.md', 'w') as o:
        o.write("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|\n")
        for n, c, t, a, p in rows[:26]:
            n = n.split('(')[0][:80] if not n.startswith('void at::') else 'torch: ' + n[:50].replace('|', '/')
            o.write(f"| `{n}` | {c} | {t:.0f} | {a:.2f} | {p:.2f} |\n")
    print(open('gpurun_out/kernel_stats.md').read())
    break
PY
rm -rf gpurun_out/prof
RQ_M=8192 bash scripts/gpu_pmc2.sh > /dev/null 2>&1; cp gpurun_out/gemm_traffic.json gpurun_out/gemm_traffic_m8192.json
