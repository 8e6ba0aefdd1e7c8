// Microbenchmark (round 5): how fast can every CU of the chip pull L2-resident operand panels at once, by instruction kind?
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/l2_intake.hip -o gpurun_out/l2_intake && gpurun_out/l2_intake
// One 512-thread workgroup per CU streams 16-KB units (two 16-byte pieces per thread: what the GEMM / conv kernels stage per unit) from a
// window that stays in its XCD's 4-MB L2, D units in flight:
//   dma  : global_load_lds_dwordx4 into an 8-slot LDS ring, counted vmcnt (the staging of gemm_p8 / the LDS-DMA tiles / the halo conv)
//   reg  : global_load_dwordx4 into registers (folded with xor: no LDS at all)
//   stage: global_load_dwordx4 -> ds_write_b128 (register staging, what the kernels did before LDS-DMA)
// share = how many workgroups of an XCD read the SAME window at the same time (GEMM panels are shared by 4-8 tiles of an XCD).
// The question: is the ~10-11 TB/s at which the 256 x 256 GEMM's and the mid-batch tiles' operand streams saturate a property of the
// LDS-DMA path, or of the L2 fabric under 256 simultaneous readers?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32;
struct __attribute__((aligned(16))) u128 { u32 x, y, z, w; };

static __device__ __forceinline__ void glds16(unsigned lds_base, const void* gsrc) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
template <int N> static __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

constexpr int UNIT = 16384;

template <int MODE, int D>
__global__ __launch_bounds__(512, 1) void k_intake(const char* buf, int win, int share, int n_units, int barrier, u32* sink, int ld) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int groups = (int)((gridDim.x >> 3) + share - 1) / share;              // windows per XCD
    const char* base = buf + ((size_t)xcd * groups + slot / share) * (size_t)win;
    const int upw = ld ? ld / 128 : win / UNIT;                               // units per window
    // ld == 0: a unit is 16 KB contiguous (a thread's two pieces 1 KB apart).  ld > 0: the window is a [rows][ld bytes] matrix (a GEMM operand
    // panel, K contiguous) and unit u = its 128 x 128-byte K-tile u: lane l of wavefront w fetches 16-byte chunk l & 7 of rows 16 w + (l >> 3)
    // and 16 w + 8 + (l >> 3) -- eight rows, i.e. eight cache lines ld bytes apart, per instruction: what gemm.h's staging does
    const unsigned toff = ld ? (unsigned)((wave * 16 + (lane >> 3)) * ld + (lane & 7) * 16) : (unsigned)(wave * 2048 + lane * 16);
    const unsigned ustep = ld ? 128u : (unsigned)UNIT, pstep = ld ? 8u * (unsigned)ld : 1024u;
    u32 acc = 0;
    if (MODE == 0) {
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem) +
                              (unsigned)__builtin_amdgcn_readfirstlane(wave) * 2048u;
        auto issue = [&](int u) {
            const char* g = base + (size_t)(u % upw) * ustep + toff;
            const unsigned dst = lds0 + (unsigned)(u & 7) * UNIT;
            glds16(dst, g);
            glds16(dst + 1024, g + pstep);
        };
        for (int u = 0; u < D - 1; ++u) issue(u);
        for (int u = 0; u < n_units; ++u) {
            issue(u + D - 1);
            wait_vmcnt<2 * (D - 1)>();
            if (barrier) __builtin_amdgcn_s_barrier();
        }
        wait_vmcnt<0>();
        acc = ((u32*)smem)[tid];
    } else {
        u128 r[D][2];
        auto ld = [&](int u, u128* d) {
            const char* g = base + (size_t)(u % upw) * ustep + toff;
            d[0] = *(const u128*)g;
            d[1] = *(const u128*)(g + pstep);
        };
#pragma unroll
        for (int u = 0; u < D; ++u) ld(u, r[u]);
        for (int u0 = 0; u0 < n_units; u0 += D) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                if (MODE == 1) {
                    acc ^= r[k][0].x ^ r[k][0].y ^ r[k][0].z ^ r[k][0].w ^ r[k][1].x ^ r[k][1].y ^ r[k][1].z ^ r[k][1].w;
                } else {
                    char* dst = smem + ((u0 + k) & 7) * UNIT + toff;
                    *(u128*)dst = r[k][0];
                    *(u128*)(dst + 1024) = r[k][1];
                }
                ld(u0 + k + D, r[k]);
                if (barrier) __builtin_amdgcn_s_barrier();
            }
        }
        if (MODE == 2) { __syncthreads(); acc = ((u32*)smem)[tid]; }
#pragma unroll
        for (int u = 0; u < D; ++u) acc ^= r[u][0].x ^ r[u][1].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int D>
static double run(const char* buf, u32* sink, int blocks, int win, int share, int n_units, int barrier, int ld = 0) {
    hipFuncSetAttribute((const void*)k_intake<MODE, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * UNIT);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_intake<MODE, D>), dim3(blocks), dim3(512), 8 * UNIT, 0, buf, win, share, n_units, barrier, sink, ld);
    hipEventRecord(e0);
    const int reps = 5;
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL((k_intake<MODE, D>), dim3(blocks), dim3(512), 8 * UNIT, 0, buf, win, share, n_units, barrier, sink, ld);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); exit(1); }
    return (double)blocks * n_units * UNIT * reps / (ms * 1e-3) / 1e12;       // TB/s
}

int main() {
    char* buf; u32* sink;
    const size_t total = 64u << 20;
    hipMalloc(&buf, total + (1u << 20));
    hipMemset(buf, 1, total + (1u << 20));
    hipMalloc(&sink, 64);
    const int n_units = 1536;                                                 // 24 MB per workgroup
    struct Cfg { int blocks, win, share; const char* what; };
    const Cfg cfgs[] = {
        {256, 384 << 10, 8, "256 CUs, 384-KB windows shared by 8 (1.5 MB per XCD)"},
        {256, 768 << 10, 8, "256 CUs, 768-KB windows shared by 8 (3 MB per XCD)"},
        {256, 64 << 10, 1, "256 CUs, private 64-KB windows (2 MB per XCD)"},
        {256, 512 << 10, 32, "256 CUs, one 512-KB window per XCD"},
        {128, 384 << 10, 8, "128 CUs, 384-KB windows shared by 8"},
        {64, 384 << 10, 8, " 64 CUs, 384-KB windows shared by 8"},
    };
    for (const Cfg& c : cfgs) {
        printf("== %s\n", c.what);
        for (int barrier = 0; barrier < 2; ++barrier) {
            printf("   %s  dma D=2 %6.2f  D=4 %6.2f  D=6 %6.2f | reg D=2 %6.2f  D=4 %6.2f  D=6 %6.2f | stage D=2 %6.2f  D=4 %6.2f  TB/s\n",
                   barrier ? "barrier per unit" : "no barrier      ",
                   run<0, 2>(buf, sink, c.blocks, c.win, c.share, n_units, barrier), run<0, 4>(buf, sink, c.blocks, c.win, c.share, n_units, barrier),
                   run<0, 6>(buf, sink, c.blocks, c.win, c.share, n_units, barrier),
                   run<1, 2>(buf, sink, c.blocks, c.win, c.share, n_units, barrier), run<1, 4>(buf, sink, c.blocks, c.win, c.share, n_units, barrier),
                   run<1, 6>(buf, sink, c.blocks, c.win, c.share, n_units, barrier),
                   run<2, 2>(buf, sink, c.blocks, c.win, c.share, n_units, barrier), run<2, 4>(buf, sink, c.blocks, c.win, c.share, n_units, barrier));
            fflush(stdout);
        }
    }
    // the same stream with the GEMM's addressing: 128-row panels of a K-contiguous matrix, row stride ld (window = 128 rows x ld bytes)
    for (int ld : {3072, 3200, 5120, 12288, 12416}) {
        for (int share : {8, 32}) {
            const int win = 128 * ld;
            printf("== 256 CUs, 128-row panels, row stride %d bytes (%d K-tiles), shared by %d (%.1f MB per XCD)\n", ld, ld / 128, share, 32.0 / share * win / 1048576.0);
            for (int barrier = 0; barrier < 2; ++barrier) {
                printf("   %s  dma D=2 %6.2f  D=4 %6.2f  D=6 %6.2f | reg D=2 %6.2f  D=4 %6.2f  D=6 %6.2f TB/s\n", barrier ? "barrier per unit" : "no barrier      ",
                       run<0, 2>(buf, sink, 256, win, share, n_units, barrier, ld), run<0, 4>(buf, sink, 256, win, share, n_units, barrier, ld),
                       run<0, 6>(buf, sink, 256, win, share, n_units, barrier, ld), run<1, 2>(buf, sink, 256, win, share, n_units, barrier, ld),
                       run<1, 4>(buf, sink, 256, win, share, n_units, barrier, ld), run<1, 6>(buf, sink, 256, win, share, n_units, barrier, ld));
                fflush(stdout);
            }
        }
    }
    // ---- cold operands: every workgroup streams its OWN 128-row panel ONCE (one pass over K, like a weight tile of a decode-step GEMM), from memory
    // nothing has touched since it left every cache (a 2-GB buffer, each launch on the next region).  Row-strided (the panel is 128 rows of a
    // K-contiguous matrix: a unit = 128 x 128 bytes, rows `ld` bytes apart) against the same bytes laid out unit by unit (16 KB contiguous).
    {
        char* big;
        const size_t big_bytes = (size_t)2048 << 20;
        if (hipMalloc(&big, big_bytes + (1u << 20)) != hipSuccess) { printf("cold test: no memory\n"); return 0; }
        hipMemset(big, 1, big_bytes);
        hipDeviceSynchronize();
        for (int ld : {3072, 12288}) {
          for (int share : {1, 4}) {                                     // 4: every panel is streamed by four workgroups of one XCD at once (the four m-tiles of a 500-row GEMM)
            for (int blocks : {144, 256}) {
                const int win = 128 * ld, units = ld / 128;
                const size_t per_launch = (size_t)blocks / share * win;  // windows, xcd-major
                const int launches = (int)(big_bytes / per_launch) < 24 ? (int)(big_bytes / per_launch) : 24;
                double tbs[2][2];
                for (int strided = 0; strided < 2; ++strided)
                    for (int mode = 0; mode < 2; ++mode) {
                        hipEvent_t e0, e1;
                        hipEventCreate(&e0); hipEventCreate(&e1);
                        // flush: touch another 512 MB so that the MALL holds none of the region (the buffer is 8 x the MALL anyway)
                        hipEventRecord(e0);
                        for (int l = 0; l < launches; ++l) {
                            const char* base = big + (size_t)l * per_launch;
                            if (mode == 0) hipLaunchKernelGGL((k_intake<0, 4>), dim3(blocks), dim3(512), 8 * UNIT, 0, base, win, share, units, 1, sink, strided ? ld : 0);
                            else if (strided) hipLaunchKernelGGL((k_intake<0, 6>), dim3(blocks), dim3(512), 8 * UNIT, 0, base, win, share, units, 1, sink, ld);   // (column 'reg' of the strided half: dma with 5 units in flight)
                            else hipLaunchKernelGGL((k_intake<0, 2>), dim3(blocks), dim3(512), 8 * UNIT, 0, base, win, share, units, 1, sink, 0);                  // (column 'reg' of the contiguous half: dma with 1 unit in flight)
                        }
                        hipEventRecord(e1);
                        hipEventSynchronize(e1);
                        float ms = 0;
                        hipEventElapsedTime(&ms, e0, e1);
                        tbs[strided][mode] = ms * 1e3 / launches;
                    }
                printf("== cold, %3d workgroups, %d per %d-KB panel (%d units), %.0f MB unique per launch: us per launch  contiguous units: dma D=4 %6.1f D=2 %6.1f | row-strided (ld %d): dma D=4 %6.1f D=6 %6.1f\n",
                       blocks, share, win >> 10, units, per_launch / 1048576.0, tbs[0][0], tbs[0][1], ld, tbs[1][0], tbs[1][1]);
                fflush(stdout);
            }
          }
        }
    }
    return 0;
}
