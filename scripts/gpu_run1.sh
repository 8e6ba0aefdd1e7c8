#!/bin/bash
# first GPU contact: parity tests, smoke, a short bench, a kernel-trace profile
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/device.txt
echo "== pytest" > gpurun_out/summary.txt
timeout 1200 python -m pytest tests -m gpu -q -s --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/summary.txt
tail -40 gpurun_out/pytest_gpu.log
echo "== smoke" >> gpurun_out/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/summary.txt
tail -5 gpurun_out/smoke.log
echo "== bench" >> gpurun_out/summary.txt
timeout 600 python bench.py --steps 2 --warmup 1 --batch 64 --no-cpu-baseline > gpurun_out/bench_b64.log 2>&1
echo "bench b64 exit $?" >> gpurun_out/summary.txt
tail -3 gpurun_out/bench_b64.log
timeout 600 python bench.py --steps 2 --warmup 1 --batch 256 --no-cpu-baseline > gpurun_out/bench_b256.log 2>&1
echo "bench b256 exit $?" >> gpurun_out/summary.txt
tail -3 gpurun_out/bench_b256.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_b64" -o trace -- python "$OLDPWD/bench.py" --steps 1 --warmup 1 --batch 64 --no-cpu-baseline --no-profile > "$OLDPWD/gpurun_out/rocprof_b64.log" 2>&1
echo "rocprof exit $?" >> "$OLDPWD/gpurun_out/summary.txt"
cd "$OLDPWD"
find gpurun_out/prof_b64 -name "*stats*" | head
cat gpurun_out/summary.txt
