#!/bin/bash
# round-3 A/B call 1: halo conv variants (base / new / new without SLP packing), GEMM two-phase schedule with BH1 in the burst
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
V=rq-vae-transformer_amd/variants
for lib in base new noslp; do
  echo "== conv_halo_bench: $lib"
  if [ $lib = new ]; then unset RQ_LIB; else export RQ_LIB=$PWD/$V/librqamd_$lib.so; fi
  timeout 300 python scripts/conv_halo_bench.py 2>&1 | tail -8
done
unset RQ_LIB
echo "== gemm ph2 vs ph3"
timeout 400 python scripts/gemm_p8_ph3_ab.py 2>&1 | tail -40
