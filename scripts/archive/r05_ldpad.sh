mkdir -p gpurun_out
{
for rep in 1 2; do
RQ_PAD=0 python scripts/gemm_ldpad_probe.py
for pad in 8 32 64; do RQ_PAD=$pad RQ_LIB=rq-vae-transformer_amd/variants/librqamd_ldpad$pad.so python scripts/gemm_ldpad_probe.py; done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_gemm_ldpad_probe.txt
