// Microbenchmark (round 5): the dense bf16 MFMA rate the board SUSTAINS, by operand data.  No memory traffic at all: every wavefront runs
// v_mfma_f32_32x32x16_bf16 back to back on eight rotating accumulators, two wavefronts per SIMD, one workgroup per CU, for `secs` seconds.
//   mode 0: constant operands (what scripts/micro/mfma_rate.hip and most peak-rate benchmarks do: the multipliers see the same bits every clock)
//   mode 1: operands change with every MFMA (eight register sets of pseudo-random bf16 values ~ N(0,1) per operand, rotated)
//   mode 2: like 1 but one operand is all zeros (products are zero: the adders and most of the multiplier array stay quiet)
// Prints TFLOP/s over the whole run; scripts/mfma_power.py samples rocm-smi (sclk, power) beside it.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_power.hip -o scripts/micro/bin/mfma_power && scripts/micro/bin/mfma_power <mode> <secs>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x16 __attribute__((ext_vector_type(16)));

static __device__ __forceinline__ float rnd(unsigned& s) {        // ~N(0,1): sum of four uniforms
    float a = 0.f;
    for (int i = 0; i < 4; ++i) { s = s * 1664525u + 1013904223u; a += (float)(s >> 8) * (1.0f / 16777216.0f); }
    return (a - 2.0f) * 1.7320508f;
}

__global__ __launch_bounds__(512, 1) void k_mfma(float* out, int n, int mode) {
    unsigned s = 12345u + threadIdx.x * 7919u + blockIdx.x * 104729u;
    bf16x8 x[8], y[8];
    for (int k = 0; k < 8; ++k)
        for (int e = 0; e < 8; ++e) {
            const float a = rnd(s), b = rnd(s);
            x[k][e] = (__bf16)(mode == 0 ? 1.0f : a);
            y[k][e] = (__bf16)(mode == 0 ? 0.5f : (mode == 2 ? 0.0f : b));
        }
    f32x16 acc[8];
    for (int a = 0; a < 8; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[a], y[(a * 3 + 1) & 7], acc[a], 0, 0, 0);
        // rotate the register sets so that every pipe input changes from one MFMA to the next (register moves only)
        const bf16x8 t = x[0];
#pragma unroll
        for (int a = 0; a < 7; ++a) x[a] = x[a + 1];
        x[7] = t;
    }
    float sum = 0.f;
    for (int a = 0; a < 8; ++a)
        for (int r = 0; r < 16; ++r) sum += acc[a][r];
    if (sum == 12345.678f) out[0] = sum;
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const double secs = argc > 2 ? atof(argv[2]) : 3.0;
    float* d;
    if (hipMalloc(&d, 64) != hipSuccess) return 1;
    const int n = 1 << 17, blocks = 256;
    hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(512), 0, 0, d, n, mode);
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double el = 0;
    do {
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(512), 0, 0, d, n, mode);
        if (hipDeviceSynchronize() != hipSuccess) return 1;
        launches += 8;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < secs);
    const double flop = (double)launches * blocks * 8 /*wavefronts*/ * n * 2.0 * 32 * 32 * 16;
    printf("mode %d: %.1f TFLOP/s over %.2f s (%ld launches)\n", mode, flop / el / 1e12, el, launches);
    return 0;
}
