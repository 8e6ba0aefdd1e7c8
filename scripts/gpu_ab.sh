#!/bin/bash
# A/B kernel library variants on the conv and GEMM microbenchmarks
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for v in "" _v1 _v2; do
  echo "=== variant '$v'"
  RQ_LIB=$PWD/rq-vae-transformer_amd/librqamd$v.so timeout 300 python scripts/conv_bench.py 2>&1 | grep -E "@|err" | grep -v amdgpu
  RQ_MS=1024 RQ_LIB=$PWD/rq-vae-transformer_amd/librqamd$v.so timeout 300 python scripts/gemm_bench.py 2>&1 | grep -v amdgpu | cut -c1-150
done
