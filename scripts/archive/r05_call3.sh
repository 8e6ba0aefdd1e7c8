#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
export RQ_LIB=rq-vae-transformer_amd/variants/librqamd_sweep.so RQ_NABL=0
T=132x64,132x128,132x192,136x128,260x64,264x128,136x256,64x64r,128x64r,128x128r
RQ_TILES=$T RQ_MS=300,768,1024,1536 timeout 900 python scripts/gemm_mid_sweep.py > $O/r05_gemm_mid_sweep_m.txt 2>&1
RQ_TILES=$T RQ_E=1024 RQ_MS=200,500 timeout 600 python scripts/gemm_mid_sweep.py > $O/r05_gemm_mid_sweep_e1024.txt 2>&1
RQ_TILES=$T RQ_E=2560 RQ_MS=200,1024 timeout 600 python scripts/gemm_mid_sweep.py > $O/r05_gemm_mid_sweep_e2560b.txt 2>&1
grep -h -A1 "^M=" $O/r05_gemm_mid_sweep_m.txt $O/r05_gemm_mid_sweep_e1024.txt $O/r05_gemm_mid_sweep_e2560b.txt | cut -c1-420
