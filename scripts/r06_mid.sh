#!/bin/bash
# round 6, first evidence run: GPU tests, quantiser SQ counters, get_codes kernel trace (encode side), short default bench line
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
export RQ_TAG=r06
bash scripts/gpu.sh tests
RQ_TAG=r06_rq RQ_PMC_CMD="python $R/scripts/rq_bench.py" RQ_PMC_FILTER=rq_quantize bash scripts/gpu.sh sqpmc > /dev/null 2>&1; cat gpurun_out/r06_rq_sqpmc.txt
RQ_TAG=r06_encode RQ_PMC_CMD="python $R/scripts/encode_trace.py" bash scripts/gpu.sh ktrace > /dev/null 2>&1; head -24 gpurun_out/r06_encode_ktrace.md
RQ_BENCH_ARGS="--steps 2 --warmup 1" bash scripts/gpu.sh bench | cut -c1-3000
