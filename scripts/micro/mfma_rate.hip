// Microbenchmark: MFMA issue rate of one SIMD with one or two wavefronts feeding it (gfx950).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_rate.hip -o gpurun_out/mfma_rate && gpurun_out/mfma_rate
// Each wavefront runs `n` v_mfma_f32_32x32x16_bf16 on NACC rotating accumulators (no memory traffic); reported: shader cycles per
// MFMA per SIMD (32 = the pipe's rate).  Question it answers: do two wavefronts that issue MFMAs on the same SIMD at the same
// time interleave at full rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k_mfma(unsigned long long* out, int n, float seed) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = seed * (a + r);
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(seed + e); y[e] = (__bf16)(seed - e); }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i += NACC) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.f) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC>
static void run(int threads, int blocks, int n) {
    unsigned long long* d;
    hipMalloc(&d, (1 + blocks * 16) * 8);
    hipMemset(d, 0, (1 + blocks * 16) * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_mfma<NACC>, dim3(blocks), dim3(threads), 0, 0, d, n, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(1 + blocks * 16);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    const int waves = threads / 64;
    double mx = 0, sum = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < waves; ++w) { double c = (double)h[1 + b * 16 + w]; sum += c; if (c > mx) mx = c; }
    const double per_simd = (double)n * waves / 4.0;       // MFMAs a SIMD executes (waves spread over 4 SIMDs)
    printf("NACC %d, %d wave(s) per SIMD, %4d workgroups: %.1f cycles per MFMA per SIMD (slowest wavefront), %.1f (mean)\n", NACC, waves / 4, blocks,
           mx / per_simd, sum / (blocks * waves) / per_simd);
    hipFree(d);
}

int main() {
    const int n = 4096;
    for (int blocks : {64, 256}) {
        run<4>(256, blocks, n);
        run<4>(512, blocks, n);
        run<8>(256, blocks, n);
        run<8>(512, blocks, n);
        run<2>(512, blocks, n);
    }
    return 0;
}
