#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -3 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --top-k 0 --top-p 1.0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unfiltered', {k:round(d[k],3) for k in ('value','ms_per_step','ar_ms_per_image','decode_ms_per_image')})"
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --top-k 1024 --top-p 0.95 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k1024 p0.95', {k:round(d[k],3) for k in ('value','ms_per_step','ar_ms_per_image','decode_ms_per_image')})"
