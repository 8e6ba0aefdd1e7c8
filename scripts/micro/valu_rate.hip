// Microbenchmark: issue cost of VALU / transcendental instructions, alone and beside another wavefront's MFMA stream (gfx950).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/valu_rate.hip -o valu_rate && ./valu_rate
// 512 threads per workgroup: wavefronts 0-3 (one per SIMD) run stream A, wavefronts 4-7 (the second wavefront of each SIMD) stream B.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { S_NONE, S_MFMA, S_EXP, S_RCP, S_FMA, S_PKFMA, S_SILU };

template <int KIND>
__device__ __forceinline__ float stream(int n, float seed) {
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = seed + i * 0.01f;
    if (KIND == S_MFMA) {
        f32x16 acc[4];
        for (int a = 0; a < 4; ++a) for (int q = 0; q < 16; ++q) acc[a][q] = seed * (a + q);
        bf16x8 x, y;
        for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(seed + e); y[e] = (__bf16)(seed - e); }
        for (int i = 0; i < n; i += 4)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
        float s = 0.f;
        for (int a = 0; a < 4; ++a) for (int q = 0; q < 16; ++q) s += acc[a][q];
        return s;
    }
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (KIND == S_EXP) r[k] = __builtin_amdgcn_exp2f(r[k]);
            if (KIND == S_RCP) r[k] = __builtin_amdgcn_rcpf(r[k]);
            if (KIND == S_FMA) r[k] = __builtin_fmaf(r[k], 1.0001f, 0.5f);
            if (KIND == S_PKFMA && (k & 1) == 0) {
                f32x2 v = {r[k], r[k + 1]};
                v = __builtin_elementwise_fma(v, (f32x2){1.0001f, 1.0002f}, (f32x2){0.5f, 0.25f});
                r[k] = v.x; r[k + 1] = v.y;
            }
            if (KIND == S_SILU) {           // one SiLU: fma, mul, exp, add, rcp, mul  (counted as ONE "instruction" of the stream)
                const float a = __builtin_fmaf(r[k], 1.01f, 0.1f);
                r[k] = a * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * a));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += r[i];
    return s;
}

template <int KA, int KB>
__global__ void k_pair(unsigned long long* out, int na, int nb, float seed) {
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    float s;
    if (wave < 4) s = stream<KA>(na, seed); else s = stream<KB>(nb, seed);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (s == 12345.f) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KA, int KB>
static void run(const char* name, int na, int nb) {
    const int blocks = 256;
    unsigned long long* d;
    (void)hipMalloc(&d, (1 + blocks * 8) * 8);
    (void)hipMemset(d, 0, (1 + blocks * 8) * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_pair<KA, KB>), dim3(blocks), dim3(512), 0, 0, d, na, nb, 1.0f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(1 + blocks * 8);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int bl = 0; bl < blocks; ++bl) for (int w = 0; w < 8; ++w) (w < 4 ? a : b) += (double)h[1 + bl * 8 + w];
    a /= blocks * 4; b /= blocks * 4;
    printf("%-34s A: %7.1f cycles per op", name, na ? a / na : 0.0);
    if (nb) printf("   B: %7.1f cycles per op", b / nb);
    printf("\n");
    (void)hipFree(d);
}

int main() {
    const int n = 4096;
    run<S_MFMA, S_NONE>("A = MFMA alone", n, 0);
    run<S_EXP, S_NONE>("A = v_exp_f32 alone", n, 0);
    run<S_RCP, S_NONE>("A = v_rcp_f32 alone", n, 0);
    run<S_FMA, S_NONE>("A = v_fma_f32 alone", n, 0);
    run<S_PKFMA, S_NONE>("A = v_pk_fma_f32 alone (per element)", n, 0);
    run<S_SILU, S_NONE>("A = SiLU (6 ops) alone", n, 0);
    run<S_MFMA, S_EXP>("A = MFMA beside B = v_exp_f32", n, 2 * n);
    run<S_MFMA, S_FMA>("A = MFMA beside B = v_fma_f32", n, 8 * n);
    run<S_MFMA, S_SILU>("A = MFMA beside B = SiLU", n, n);
    run<S_EXP, S_EXP>("A = B = v_exp_f32", n, n);
    run<S_MFMA, S_MFMA>("A = B = MFMA", n, n);
    return 0;
}
