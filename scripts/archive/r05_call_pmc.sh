#!/bin/bash
# round 5: (a) does the fabric traffic of the 256 x 256 GEMM decide its time?  group height of the XCD walk (RQAMD_P8_GM) vs FETCH/WRITE_SIZE
# and vs launch time at the bench batch; (b) SQ counters of the residual quantiser; (c) the default traffic file of the final GEMM sources
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out; mkdir -p $O
{
echo "== gemm_p8_kernel at M = 10752: XCD walk group height (RQAMD_P8_GM; default 6 for 42 m-tiles) vs fabric traffic and time"
for gm in 3 6 14; do
  echo "-- RQAMD_P8_GM=$gm: time per launch"
  RQAMD_P8_GM=$gm RQ_M=10752 timeout 300 python scripts/gemm_epi_ab.py 2>&1 | grep -v amdgpu.ids
  RQ_TAG=r05_gm$gm RQAMD_P8_GM=$gm RQ_M=10752 bash scripts/gpu.sh pmc 2>&1 | grep "weighted"
done
} > $O/r05_gemm_p8_traffic_vs_time.txt 2>&1
cat $O/r05_gemm_p8_traffic_vs_time.txt
RQ_TAG=r05 RQ_M=10752 bash scripts/gpu.sh pmc 2>&1 | tail -3
RQ_TAG=r05_rq RQ_PMC_CMD="python $R/scripts/rq_bench.py" RQ_PMC_FILTER=rq_quantize bash scripts/gpu.sh sqpmc 2>&1 | tail -40
