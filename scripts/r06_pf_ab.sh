#!/bin/bash
# round 6: fragment reads one K-tile ahead (RQ_GEMM_PF=1, the tree) against the plain loop (variants/librqamd_pf0.so), same box:
# per launch in-graph (gemm_mid_sweep.py on the shipped tile codes) and end to end (small_batch_ab.py)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export RQ_TILES=132x64,136x128,132x192 RQ_STAGES=3 RQ_NABL=3
for lib in "" rq-vae-transformer_amd/variants/librqamd_pf0.so; do
  echo "== per launch, RQ_LIB=${lib:-<tree: PF=1>}"
  RQ_LIB=$lib RQ_MS=200,500 timeout 600 python scripts/gemm_mid_sweep.py 2>&1 | grep -v amdgpu.ids
done
for lib in "" rq-vae-transformer_amd/variants/librqamd_pf0.so "" rq-vae-transformer_amd/variants/librqamd_pf0.so; do
  echo "== sampling, RQ_LIB=${lib:-<tree: PF=1>}"
  RQ_LIB=$lib RQ_BS=200,500 timeout 600 python scripts/small_batch_ab.py 2>&1 | grep -v amdgpu.ids
done
