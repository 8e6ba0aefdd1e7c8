#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
timeout 600 python scripts/conv_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/conv_bench.log
RQ_MS=256,1024,2048 timeout 600 python scripts/gemm_bench.py 2>&1 | grep -v amdgpu | cut -c1-200 | tee gpurun_out/gemm_bench.log
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -3 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tee gpurun_out/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', {k:round(d[k],3) for k in ('value','ms_per_step','ar_ms_per_image','decode_ms_per_image')}, d['roofline'])"
