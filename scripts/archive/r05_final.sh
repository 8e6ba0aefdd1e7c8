#!/bin/bash
# round 5: evidence of the final build -- GPU tests, the default bench line, kernel traces, GEMM traffic (hash-stamped), halo-conv SQ counters
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
export RQ_TAG=r05
bash scripts/gpu.sh tests
echo "(GEMM traffic files: unchanged GEMM sources since the earlier run of this script -- not repeated)"
bash scripts/gpu.sh bench
RQ_TAG=r05_b10752 bash scripts/gpu.sh trace > /dev/null 2>&1; head -24 gpurun_out/r05_b10752_kernel_stats.md
RQ_TAG=r05_b64 RQ_TRACE_ARGS="--batch 64 --steps 2" bash scripts/gpu.sh trace > /dev/null 2>&1; head -8 gpurun_out/r05_b64_kernel_stats.md
RQ_TAG=r05_b500 RQ_TRACE_ARGS="--batch 500 --steps 3" bash scripts/gpu.sh trace > /dev/null 2>&1; head -8 gpurun_out/r05_b500_kernel_stats.md
RQ_TAG=r05_conv_halo RQ_PMC_CMD="python $R/scripts/conv_halo_pmc.py" RQ_PMC_FILTER=conv3x3_halo bash scripts/gpu.sh sqpmc > /dev/null 2>&1; head -30 gpurun_out/r05_conv_halo_sqpmc.txt
