#!/bin/bash
# round 6: fragment prefetch (PF) on the 4 / 8 / 16-wavefront forms of the 128 x 128 and 128 x 64 tiles, sweep builds, 3 and 4 stages
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export RQ_TILES=128x128,129x128,132x128,136x128,128x64,132x64 RQ_STAGES=3,4 RQ_NABL=6 RQ_SHAPES=qkv,fc1
for lib in rq-vae-transformer_amd/variants/librqamd_sw_pf1.so rq-vae-transformer_amd/variants/librqamd_sw_pf0.so; do
  echo "== per launch, RQ_LIB=${lib}"
  RQ_LIB=$lib RQ_MS=500 timeout 600 python scripts/gemm_mid_sweep.py 2>&1 | grep -v amdgpu.ids
  echo "== warm weights (RQ_NROT=1), RQ_LIB=${lib}"
  RQ_NROT=1 RQ_LIB=$lib RQ_MS=500 timeout 600 python scripts/gemm_mid_sweep.py 2>&1 | grep -v amdgpu.ids
done
