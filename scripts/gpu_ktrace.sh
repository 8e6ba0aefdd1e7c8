#!/bin/bash
# kernel-trace stats of one bench step (used to A/B individual kernels under the real launch mix)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
cd /tmp
if [ -n "$RQ_CMD" ]; then
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o trace -- python $R/$RQ_CMD > $R/gpurun_out/rocprof.log 2>&1
else
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile "$@" > $R/gpurun_out/rocprof.log 2>&1
fi
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
    import os
    pat = os.environ.get('RQ_TRACE_KERNEL')
    if pat:
        tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
        kt = [t for t in tabs if 'kernel_dispatch' in t and 'rocpd' in t] or [t for t in tabs if 'kernel' in t]
        print('tables', tabs[:40])
        try:
            rows = list(db.execute("select name, start, end from kernels where name like ? order by start", ('%' + pat + '%',)))
            d = [(e - s0) / 1e3 for _, s0, e in rows]
            n = len(d)
            print('calls', n)
            step = max(1, n // 256)
            print(' '.join(f'{x:.0f}' for x in d[n // 2::step][:200]))
            print('first position of the timed step:', ' '.join(f'{x:.0f}' for x in d[n // 2:n // 2 + 70]))
            print('last position:', ' '.join(f'{x:.0f}' for x in d[n - 66:]))
        except Exception as ex:
            print('trace query failed', ex)
    break
PY
rm -rf gpurun_out/prof
