#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
timeout 600 python scripts/conv_bench.py > gpurun_out/conv_bench.log 2>&1
cat gpurun_out/conv_bench.log
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -3 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 2 --warmup 1 --batch ${B:-1024} --no-cpu-baseline > gpurun_out/bench.log 2>&1
tail -1 gpurun_out/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','ar_ms_per_image','decode_ms_per_image')}, d['roofline'])"
