export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
RQ_GL=0 RQ_MS=4096 timeout 300 python scripts/gemm_bench.py 2>&1 | grep -E "^M=|WRONG" | cut -c1-60 > gpurun_out/ab_new.txt
RQAMD_GEMM_SCHED1=1 RQ_GL=0 RQ_MS=4096 timeout 300 python scripts/gemm_bench.py 2>&1 | grep -E "^M=|WRONG" | cut -c1-60 > gpurun_out/ab_old.txt
paste -d'\n' gpurun_out/ab_new.txt gpurun_out/ab_old.txt
RQ_M=4096 bash scripts/gpu_pmc2.sh > /dev/null 2>&1; python scripts/summarize_traffic.py 4096 gpurun_out/gemm_traffic.json
