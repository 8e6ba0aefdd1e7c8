#!/bin/bash
# One parametrised GPU-box script (replaces the per-run scripts of round 1).  Usage, through gpurun:
#   gpurun --timeout 900 -- 'bash scripts/gpu.sh tests bench'
# Tasks (any order, run left to right; outputs under gpurun_out/<TAG>_*):
#   tests            pytest -m gpu (+ smoke)
#   bench [args]     python bench.py  (RQ_BENCH_ARGS="--steps 2 ..." for arguments)
#   trace            rocprofv3 --kernel-trace --stats of one bench step -> <TAG>_kernel_stats.md
#   pmc              FETCH_SIZE / WRITE_SIZE of the decode GEMM shapes at RQ_M rows (separate passes) -> <TAG>_gemm_traffic_m<M>.json
#   ktrace           rocprofv3 --kernel-trace --stats of "$RQ_PMC_CMD" -> <TAG>_ktrace.md
#   sqpmc            SQ counters (three passes) of "$RQ_PMC_CMD", kernels matching $RQ_PMC_FILTER -> <TAG>_sqpmc.txt
#   gemm             scripts/gemm_bench.py (RQ_MS=4096,8192 ...)
#   cmd              run "$RQ_CMD"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
TAG=${RQ_TAG:-r03}
mkdir -p gpurun_out

stats_md() {   # $1 = rocprof output dir, $2 = markdown file
python - "$1" "$2" <<'PY'
import sqlite3, glob, sys
d, out = sys.argv[1], sys.argv[2]
for f in glob.glob(d + '/**/*.db', recursive=True) + glob.glob(d + '/*.db'):
    db = sqlite3.connect(f)
    rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(out, 'w') as o:
        o.write("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|\n")
        for n, c, t, a, p in rows[:30]:
            n = n.split('(')[0][:90] if not n.startswith('void at::') else 'torch: ' + n[:50].replace('|', '/')
            o.write(f"| `{n}` | {c} | {t:.0f} | {a:.2f} | {p:.2f} |\n")
    print(open(out).read())
    break
PY
}

for task in "$@"; do
case $task in
tests)
  timeout 1500 python -m pytest tests -m gpu -x -q -s -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc $?"
  tail -5 gpurun_out/${TAG}_pytest.log; grep -E "max err|agreement|err " gpurun_out/${TAG}_pytest.log | head -40
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
bench)
  timeout 1200 python bench.py $RQ_BENCH_ARGS > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc $?"
  tail -1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench.json; cat gpurun_out/${TAG}_bench.json ;;
trace)
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --sweep "" --also "" --formats 0 $RQ_TRACE_ARGS > $R/gpurun_out/${TAG}_rocprof.log 2>&1
  cd $R; stats_md gpurun_out/prof gpurun_out/${TAG}_kernel_stats.md; rm -rf gpurun_out/prof ;;
pmc)
  M=${RQ_M:-8192}
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    RQ_M=$M timeout 600 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_$c -o pmc --output-format csv -- python $R/scripts/gemm_traffic.py > $R/gpurun_out/pmc_$c.log 2>&1
    echo "$c exit $?"
  done
  cd $R; grep algorithmic gpurun_out/pmc_FETCH_SIZE.log
  python scripts/summarize_traffic.py $M gpurun_out/${TAG}_gemm_traffic_m$M.json
  rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE ;;
ktrace)
  # kernel trace of "$RQ_PMC_CMD" -> <TAG>_ktrace.md
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_k -o trace -- $RQ_PMC_CMD > $R/gpurun_out/${TAG}_ktrace.log 2>&1
  cd $R; stats_md gpurun_out/prof_k gpurun_out/${TAG}_ktrace.md; rm -rf gpurun_out/prof_k ;;
sqpmc)
  # SQ counters of "$RQ_PMC_CMD" (kernels matching $RQ_PMC_FILTER), two passes -> <TAG>_sqpmc.txt
  cd /tmp; : > $R/gpurun_out/${TAG}_sqpmc.txt
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/sqpmc_$i -o pmc --output-format csv -- $RQ_PMC_CMD > $R/gpurun_out/sqpmc_$i.log 2>&1
    echo "pass $i exit $?"
    python $R/scripts/pmc_summary.py $R/gpurun_out/sqpmc_$i "${RQ_PMC_FILTER:-gemm}" >> $R/gpurun_out/${TAG}_sqpmc.txt
    rm -rf $R/gpurun_out/sqpmc_$i
  done
  cd $R; cat gpurun_out/${TAG}_sqpmc.txt ;;
gemm)
  timeout 900 python scripts/gemm_bench.py > gpurun_out/${TAG}_gemm_bench.txt 2>&1; cat gpurun_out/${TAG}_gemm_bench.txt ;;
cmd)
  timeout ${RQ_CMD_TIMEOUT:-900} bash -c "$RQ_CMD" > gpurun_out/${TAG}_cmd.log 2>&1; echo "cmd rc $?"; tail -${RQ_TAIL:-60} gpurun_out/${TAG}_cmd.log ;;
*) echo "unknown task $task" ;;
esac
done
