#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" > gpurun_out/summary.txt
tail -3 gpurun_out/pytest_gpu.log
for B in ${BATCHES:-512 1024}; do
timeout 600 python bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline > gpurun_out/bench_b$B.log 2>&1
echo "bench b$B exit $?" >> gpurun_out/summary.txt
tail -1 gpurun_out/bench_b$B.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','ar_ms_per_image','decode_ms_per_image')}, d['roofline'])"
done
cd /tmp
PB=${PROF_BATCH:-512}
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 1 --warmup 1 --batch $PB --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof.log 2>&1
echo "rocprof exit $?" >> $R/gpurun_out/summary.txt
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
cat gpurun_out/summary.txt
