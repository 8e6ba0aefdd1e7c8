#!/bin/bash
# round 5, GPU call 1: mid-batch GEMM tile sweep (+ nt weights), stream-kernel nt A/B, encode kernel trace
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out; mkdir -p $O
V=$R/rq-vae-transformer_amd/variants
( RQ_LIB=$V/librqamd_sweep.so RQ_MS=500,200 timeout 600 python scripts/gemm_mid_sweep.py ) > $O/r05_gemm_mid_sweep.txt 2>&1
echo "sweep rc $?"; grep -E "^M=|ablations" $O/r05_gemm_mid_sweep.txt | cut -c1-400
( RQ_LIB=$V/librqamd_sweep_nt.so RQ_MS=500 RQ_SHAPES=qkv,fc1,fc2 timeout 300 python scripts/gemm_mid_sweep.py ) > $O/r05_gemm_mid_sweep_nt.txt 2>&1
echo "sweep nt rc $?"; grep -E "^M=" $O/r05_gemm_mid_sweep_nt.txt | cut -c1-300
( RQ_LIB=$V/librqamd_sweep.so RQ_E=2560 RQ_MS=500 timeout 400 python scripts/gemm_mid_sweep.py ) > $O/r05_gemm_mid_sweep_e2560.txt 2>&1
echo "sweep e2560 rc $?"; grep -E "^M=" $O/r05_gemm_mid_sweep_e2560.txt | cut -c1-300
{
  echo "== gemm_stream_kernel: weight DMAs default policy vs nt (aux = 2); activations default in both"
  for rep in 1 2; do
    echo "-- default, run $rep"; RQ_MS=64,128 timeout 200 python scripts/gemm_stream_ab.py 2>&1 | grep "M="
    echo "-- nt weights, run $rep"; RQ_LIB=$V/librqamd_streamnt.so RQ_MS=64,128 timeout 200 python scripts/gemm_stream_ab.py 2>&1 | grep "M="
  done
  echo "== RQTransformer.sample, 1.4B, AR only"
  echo "-- default"; RQ_BS=64,128 timeout 300 python scripts/small_batch_ab.py 2>&1 | grep "B="
  echo "-- nt weights"; RQ_LIB=$V/librqamd_streamnt.so RQ_BS=64,128 timeout 300 python scripts/small_batch_ab.py 2>&1 | grep "B="
  echo "-- default (again)"; RQ_BS=64,128 timeout 300 python scripts/small_batch_ab.py 2>&1 | grep "B="
  echo "-- nt weights (again)"; RQ_LIB=$V/librqamd_streamnt.so RQ_BS=64,128 timeout 300 python scripts/small_batch_ab.py 2>&1 | grep "B="
} > $O/r05_stream_nt_ab.txt 2>&1
cat $O/r05_stream_nt_ab.txt | cut -c1-260
RQ_TAG=r05_encode RQ_PMC_CMD="python $R/scripts/encode_trace.py" bash scripts/gpu.sh ktrace 2>&1 | head -30
