// Microbenchmark: how fast can workgroups stream a weight matrix W[N][K] bf16 (K contiguous, the nn.Linear layout) in the access
// pattern of the small-batch decode-step GEMM -- and does a K-tile-major ("tiled") layout of the same bytes do better?
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/weight_stream.hip -o gpurun_out/weight_stream && gpurun_out/weight_stream
// One workgroup of 256 threads owns BN weight rows and walks K in steps of 64 columns (128 bytes per row), DEPTH steps of loads in
// flight, exactly the global-memory side of gemm_bf16_kernel<64, BN> (no LDS, no MFMA: the loaded values are folded into a checksum).
//   layout 0 (rows):  byte address of (row n, K-tile t) = n * K * 2 + t * 128                 -- BN scattered 128-B segments per step
//   layout 1 (tiled): byte address                     = ((n / BN) * KT + t) * BN * 128 + (n % BN) * 128   -- one contiguous block per step
// Weights rotate over several copies (as consecutive layers do), so nothing is served from the caches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int BN, int DEPTH, int LAYOUT>
__global__ __launch_bounds__(256) void k_stream(const char* w, int N, int K, unsigned* out) {
    constexpr int PER = BN * 128 / (256 * 16);            // 16-byte loads per thread per step (BN rows x 128 B / 4 KB)
    const int tid = threadIdx.x, KT = K / 64;
    const int n0 = blockIdx.x * BN;
    const int chunk = tid & 7, lrow = tid >> 3;           // 8 threads cover one 128-B row segment, 32 rows per pass
    size_t base[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int n = n0 + lrow + 32 * i;
        if (LAYOUT == 0) base[i] = (size_t)n * K * 2 + chunk * 16;
        else base[i] = ((size_t)(n / BN) * KT) * BN * 128 + (size_t)(n % BN) * 128 + chunk * 16;
    }
    const size_t step = LAYOUT == 0 ? 128 : (size_t)BN * 128;
    u32x4 buf[DEPTH][PER];
    unsigned acc = 0;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < PER; ++i) buf[d][i] = *(const u32x4*)(w + base[i] + (size_t)(d < KT ? d : KT - 1) * step);
    for (int t = 0; t < KT; t += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int i = 0; i < PER; ++i) acc += buf[d][i].x ^ buf[d][i].y ^ buf[d][i].z ^ buf[d][i].w;
            const int nt = t + d + DEPTH;
#pragma unroll
            for (int i = 0; i < PER; ++i) buf[d][i] = *(const u32x4*)(w + base[i] + (size_t)(nt < KT ? nt : KT - 1) * step);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int BN, int DEPTH, int LAYOUT>
static double run(const char* w, size_t copy_bytes, int copies, int N, int K, unsigned* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 40;
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL((k_stream<BN, DEPTH, LAYOUT>), dim3(N / BN), dim3(256), 0, 0, w + (size_t)(r % copies) * copy_bytes, N, K, out);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_stream<BN, DEPTH, LAYOUT>), dim3(N / BN), dim3(256), 0, 0, w + (size_t)(r % copies) * copy_bytes, N, K, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / reps;
}

int main() {
    const int copies = 24;
    struct Shape { const char* name; int N, K; } shapes[] = {{"qkv", 4608, 1536}, {"proj", 1536, 1536}, {"fc1", 6144, 1536}, {"fc2", 1536, 6144}};
    size_t maxb = (size_t)6144 * 1536 * 2;
    char* w;
    unsigned* out;
    hipMalloc(&w, maxb * copies);
    hipMemset(w, 1, maxb * copies);
    hipMalloc(&out, 64);
    for (auto& s : shapes) {
        const size_t bytes = (size_t)s.N * s.K * 2;
        printf("%-4s N=%5d K=%5d (%5.1f MB):\n", s.name, s.N, s.K, bytes / 1e6);
#define ROW(BN, DEPTH)                                                                                                               \
        {                                                                                                                            \
            const double a = run<BN, DEPTH, 0>(w, bytes, copies, s.N, s.K, out), b = run<BN, DEPTH, 1>(w, bytes, copies, s.N, s.K, out); \
            printf("   BN %3d (%4d workgroups), %d steps in flight: rows %6.2f us = %5.2f TB/s | tiled %6.2f us = %5.2f TB/s\n", BN,        \
                   s.N / BN, DEPTH, a, bytes / a / 1e6, b, bytes / b / 1e6);                                                           \
        }
        ROW(64, 2) ROW(64, 4) ROW(64, 8)
        ROW(32, 2) ROW(32, 4) ROW(32, 8)
        ROW(128, 2) ROW(128, 4)
#undef ROW
    }
    return 0;
}
