#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python scripts/gemm_bench.py > gpurun_out/gemm_bench.log 2>&1
echo "gemm_bench exit $?" > gpurun_out/summary.txt
cat gpurun_out/gemm_bench.log
timeout 900 python -m pytest tests -m gpu -q -s --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/summary.txt
tail -15 gpurun_out/pytest_gpu.log
for B in 256 512; do
timeout 600 python bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline > gpurun_out/bench_b$B.log 2>&1
echo "bench b$B exit $?" >> gpurun_out/summary.txt
tail -1 gpurun_out/bench_b$B.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','ar_ms_per_image','decode_ms_per_image')}, d['roofline'])"
done
cat gpurun_out/summary.txt
