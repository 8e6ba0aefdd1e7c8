#!/bin/bash
# A/B of the per-tile halo conv's weight staging (round 5): base = HEAD before the change, new = LDS-DMA weight ring (the product build: residual
# pieces requested behind the tap's weight request), rmid0 = residual pieces at the top of the tap, pmid = patch pieces behind the weight
# request too, fd2 = new + fragment reads two k-steps ahead.
mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conv_kernels or vae" 2>&1 | tail -3
for v in ${VARIANTS:-base rmid0 new pmid fd2 base rmid0 new pmid fd2}; do
  echo "== conv_halo_bench: $v"
  if [ $v = new ]; then unset RQ_LIB; else export RQ_LIB=rq-vae-transformer_amd/variants/librqamd_$v.so; fi
  python scripts/conv_halo_bench.py 2>&1 | grep -v amdgpu.ids | sed -e 's/implicit-GEMM.*) | halo/halo/'
done
} > gpurun_out/r05_conv_wdma_ab.txt 2>&1
tail -80 gpurun_out/r05_conv_wdma_ab.txt
