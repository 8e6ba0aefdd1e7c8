#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
timeout 900 python scripts/gemm_bench.py > gpurun_out/gemm_bench.log 2>&1
cat gpurun_out/gemm_bench.log
cd /tmp
PB=${PROF_BATCH:-1024}
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 1 --warmup 1 --batch $PB --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof.log 2>&1
cd $R
python - <<'PY'
import sqlite3, glob
for f in glob.glob('gpurun_out/prof/**/*.db', recursive=True) + glob.glob('gpurun_out/prof/*.db'):
    db = sqlite3.connect(f)
    rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open('gpurun_out/kernel_stats.md', 'w') as o:
        o.write("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|\n")
        for n, c, t, a, p in rows[:24]:
            n = n.split('(')[0][:80] if not n.startswith('void at::') else 'torch: ' + n[:50].replace('|', '/')
            o.write(f"| `{n}` | {c} | {t:.0f} | {a:.2f} | {p:.2f} |\n")
    print(open('gpurun_out/kernel_stats.md').read())
    break
PY
rm -rf gpurun_out/prof
