mkdir -p gpurun_out
{
for lib in sweep kpb2; do
echo "== library $lib (g4 / g6 = $([ $lib = kpb2 ] && echo 'two K-tiles per barrier, one / two iterations of flight' || echo 'one K-tile per barrier'))"
RQ_LIB=rq-vae-transformer_amd/variants/librqamd_$lib.so RQ_MS=${RQ_MS:-200,500} RQ_TILES=132x64,129x64,128x64,132x128,64x128 RQ_STAGES=3,4,6 RQ_NABL=3 RQ_SHAPES=${RQ_SHAPES:-qkv,proj,fc1,fc2} python scripts/gemm_mid_sweep.py 2>&1 | grep -v "amdgpu.ids\|exceed the LDS"
done
} | tee gpurun_out/r05_gemm_kpb2_sweep.txt
