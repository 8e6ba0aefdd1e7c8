#!/bin/bash
# round 6: evidence of the final build -- GPU tests, GEMM traffic (hash-stamped, the GEMM sources changed this round), the default bench line,
# kernel traces at 10752 / 64 / 500 images, get_codes trace, SQ counters of the quantiser
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
export RQ_TAG=r06
bash scripts/gpu.sh tests
for m in 10752 500 64; do RQ_M=$m bash scripts/gpu.sh pmc > /dev/null 2>&1; ls gpurun_out/r06_gemm_traffic_m$m.json; done
bash scripts/gpu.sh bench | cut -c1-1500
RQ_TAG=r06_b10752 bash scripts/gpu.sh trace > /dev/null 2>&1; head -24 gpurun_out/r06_b10752_kernel_stats.md
RQ_TAG=r06_b64 RQ_TRACE_ARGS="--batch 64 --steps 2" bash scripts/gpu.sh trace > /dev/null 2>&1; head -8 gpurun_out/r06_b64_kernel_stats.md
RQ_TAG=r06_b500 RQ_TRACE_ARGS="--batch 500 --steps 3" bash scripts/gpu.sh trace > /dev/null 2>&1; head -8 gpurun_out/r06_b500_kernel_stats.md
RQ_TAG=r06_encode RQ_PMC_CMD="python $R/scripts/encode_trace.py" bash scripts/gpu.sh ktrace > /dev/null 2>&1; head -12 gpurun_out/r06_encode_ktrace.md
RQ_TAG=r06_rq RQ_PMC_CMD="python $R/scripts/rq_bench.py" RQ_PMC_FILTER=rq_quantize bash scripts/gpu.sh sqpmc > /dev/null 2>&1; head -12 gpurun_out/r06_rq_sqpmc.txt
python scripts/rq_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r06_rq_bench.txt; cat gpurun_out/r06_rq_bench.txt
