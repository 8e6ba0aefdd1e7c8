// Microbenchmark (round 4): what ONE dependent kernel of a decode-step chain costs on this part, by what it does -- the floor under
// the small-batch decode step.  A hipGraph of 512 dependent launches of the same kernel, replayed; reported
// per launch.  Kernels:
//   nop_struct   by-value 64-byte argument struct, no memory access
//   store        one thread stores a word (struct argument)
//   add          one thread: load, add, store -- two dependent round trips (the engine's position counter)
//   add_pre      the same with scalar arguments (-mllvm -amdgpu-kernarg-preload-count=8: arguments arrive in SGPRs, no s_load)
//   bcast_rows   256 workgroups x 256 threads each read the SAME 196 KB (64 rows x 1536 bf16: the A operand of a decode GEMM
//                written by the previous launch) and write 8 KB each -- the activation side of a GEMM without weights or MFMA
//   stream_w     256 workgroups each read their own 55 KB of a 14 MB matrix (rotating over 12 matrices) -- the weight side alone
//   both         bcast_rows + stream_w in one kernel
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=8 scripts/micro/launch_probe.hip -o scripts/micro/bin/launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct Args { int* p; const int* q; int v; int pad[11]; };

__global__ void k_nop(Args a) { if (a.v == 12345 && threadIdx.x == 999) a.p[0] = 1; }
__global__ void k_store(Args a) { if (threadIdx.x == 0) a.p[0] = a.v; }
__global__ void k_add(Args a) { if (threadIdx.x == 0) a.p[0] = a.q[0] + a.v; }
__global__ void k_add_pre(int* p, const int* q, int v) { if (threadIdx.x == 0) p[0] = q[0] + v; }

// mode bit0: read the shared rows (a_bytes, every workgroup the same), bit1: read this workgroup's own slice of w
__global__ __launch_bounds__(256) void k_rows(const char* a, int a_bytes, const char* w, long w_stride, int w_bytes, int mode, unsigned* out, int out_words) {
    const int tid = threadIdx.x;
    unsigned acc = 0;
    u32x4 r[12];
    if (mode & 2) {
        const char* src = w + (long)blockIdx.x * w_stride;
        for (int off = tid * 16; off < w_bytes; off += 256 * 16 * 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int o = off + i * 4096; r[i] = o < w_bytes ? *(const u32x4*)(src + o) : (u32x4){0, 0, 0, 0}; }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
        }
    }
    if (mode & 1) {
        for (int off = tid * 16; off < a_bytes; off += 256 * 16 * 12) {
#pragma unroll
            for (int i = 0; i < 12; ++i) { const int o = off + i * 4096; r[i] = o < a_bytes ? *(const u32x4*)(a + o) : (u32x4){0, 0, 0, 0}; }
#pragma unroll
            for (int i = 0; i < 12; ++i) acc += r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
        }
    }
    // 8 KB per workgroup of output (the next launch's shared rows when chained)
    for (int i = tid; i < out_words; i += 256) out[(long)blockIdx.x * out_words + i] = acc + i;
}

// the same traffic with EVERY load of the thread in flight before the first use (NA + NW 16-byte loads per thread, in registers):
// what one memory round trip + the CU's intake rate allow
template <int NA, int NWL>
__global__ __launch_bounds__(256) void k_rows_deep(const char* a, const char* w, long w_stride, unsigned* out, int out_words) {
    const int tid = threadIdx.x;
    u32x4 ra[NA > 0 ? NA : 1], rw[NWL > 0 ? NWL : 1];
    const char* src = w + (long)blockIdx.x * w_stride;
#pragma unroll
    for (int i = 0; i < NWL; ++i) rw[i] = *(const u32x4*)(src + tid * 16 + i * 4096);
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = *(const u32x4*)(a + tid * 16 + i * 4096);
    __builtin_amdgcn_sched_barrier(0);               // (the compiler otherwise sinks the loads between their uses)
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < NWL; ++i) acc += rw[i].x ^ rw[i].y ^ rw[i].z ^ rw[i].w;
#pragma unroll
    for (int i = 0; i < NA; ++i) acc += ra[i].x ^ ra[i].y ^ ra[i].z ^ ra[i].w;
    for (int i = tid; i < out_words; i += 256) out[(long)blockIdx.x * out_words + i] = acc + i;
}

// the same again through LDS-DMA (global_load_lds_dwordx4, what gemm_stream_kernel stages its operands with): NA + NWL 1-KB
// bursts per wavefront... per thread 16 bytes each, all issued before one s_waitcnt vmcnt(0); NA + NWL <= 36 (144 KB of LDS)
template <int NA, int NWL>
__global__ __launch_bounds__(256) void k_rows_dma(const char* a, const char* w, long w_stride, unsigned* out, int out_words) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)smem + (unsigned)wave * 1024u);
    const char* src = w + (long)blockIdx.x * w_stride;
    auto dma = [&](unsigned lds_base, const void* g) {
        unsigned keep;
        const unsigned lb = __builtin_amdgcn_readfirstlane(lds_base);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(lb) : "memory");
    };
#pragma unroll
    for (int i = 0; i < NWL; ++i) dma(lds0 + i * 4096, src + tid * 16 + i * 4096);
#pragma unroll
    for (int i = 0; i < NA; ++i) dma(lds0 + (NWL + i) * 4096, a + tid * 16 + i * 4096);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned acc = 0;
    for (int i = 0; i < NA + NWL; ++i) acc += ((const unsigned*)smem)[i * 1024 + tid];
    (void)lane;
    for (int i = tid; i < out_words; i += 256) out[(long)blockIdx.x * out_words + i] = acc + i;
}

// the shared rows once more, every load in flight, but addressed the way the GEMM's LDS-DMA addresses them: a wavefront instruction
// covers 8 ROWS x 128 bytes (row stride 3072 bytes = K 1536 bf16; lane -> row lane / 8, 16-byte chunk lane % 8), wavefront w takes
// K-tiles w, w + 4, ...: 48 loads per thread = the same 196 KB as k_rows_deep<48, 0>, which reads them as contiguous 4-KB blocks
__global__ __launch_bounds__(256) void k_rows_gather(const char* a, unsigned* out, int out_words) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane >> 3, lc = lane & 7;
    u32x4 r[48];
#pragma unroll
    for (int i = 0; i < 6; ++i)              // K-tile wave + 4 i
#pragma unroll
        for (int g = 0; g < 8; ++g)          // 8-row group
            r[i * 8 + g] = *(const u32x4*)(a + (long)(8 * g + lr) * 3072 + (wave + 4 * i) * 128 + lc * 16);
    __builtin_amdgcn_sched_barrier(0);
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 48; ++i) acc += r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
    for (int i = tid; i < out_words; i += 256) out[(long)blockIdx.x * out_words + i] = acc + i;
}

template <typename F> static double time_graph(F enqueue, int n_launch, int reps) {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n_launch; ++i) enqueue(st, i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(st));
    return ms * 1e3 / (reps * (double)n_launch);
}

int main() {
    int *p, *q;
    CK(hipMalloc(&p, 4096)); CK(hipMalloc(&q, 4096));
    CK(hipMemset(p, 0, 4096)); CK(hipMemset(q, 0, 4096));
    const int NL = 512, REPS = 20;
    Args a{p, p, 1, {}};
    printf("per dependent launch inside a replayed hipGraph of %d launches (us):\n", NL);
    printf("  nop_struct   %.2f\n", time_graph([&](hipStream_t s, int) { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s, a); }, NL, REPS));
    printf("  nop 256 wgs  %.2f\n", time_graph([&](hipStream_t s, int) { hipLaunchKernelGGL(k_nop, dim3(256), dim3(256), 0, s, a); }, NL, REPS));
    printf("  store        %.2f\n", time_graph([&](hipStream_t s, int) { hipLaunchKernelGGL(k_store, dim3(1), dim3(64), 0, s, a); }, NL, REPS));
    printf("  add          %.2f\n", time_graph([&](hipStream_t s, int) { hipLaunchKernelGGL(k_add, dim3(1), dim3(64), 0, s, a); }, NL, REPS));
    printf("  add_pre      %.2f\n", time_graph([&](hipStream_t s, int) { hipLaunchKernelGGL(k_add_pre, dim3(1), dim3(64), 0, s, p, (const int*)p, 1); }, NL, REPS));
    // GEMM-shaped traffic without arithmetic
    const int A_BYTES = 64 * 1536 * 2, W_BYTES = 3 * 1536 * 1536 * 2 / 256, NW = 12, OUT_WORDS = 2048;
    char* w;
    unsigned *o0, *o1;
    CK(hipMalloc(&w, (size_t)NW * 256 * W_BYTES + (1 << 20)));
    CK(hipMemset(w, 1, (size_t)NW * 256 * W_BYTES));
    CK(hipMalloc(&o0, 256 * OUT_WORDS * 4 + (1 << 20))); CK(hipMalloc(&o1, 256 * OUT_WORDS * 4 + (1 << 20)));
    CK(hipMemset(o0, 0, 256 * OUT_WORDS * 4)); CK(hipMemset(o1, 0, 256 * OUT_WORDS * 4));
    for (int wgs : {48, 144, 256}) {
        for (int mode : {1, 2, 3}) {
            const double t = time_graph([&](hipStream_t s, int i) {
                // chained: launch i reads the rows launch i-1 wrote (ping-pong), weights rotate over NW matrices
                hipLaunchKernelGGL(k_rows, dim3(wgs), dim3(256), 0, s, (const char*)((i & 1) ? o0 : o1), A_BYTES, (const char*)(w + (size_t)(i % NW) * 256 * W_BYTES),
                                   (long)W_BYTES, W_BYTES, mode, (i & 1) ? o1 : o0, OUT_WORDS);
            }, NL, REPS);
            printf("  %-10s %3d wgs  %.2f   (%s)\n", mode == 1 ? "bcast_rows" : mode == 2 ? "stream_w" : "both", wgs, t,
                   mode == 1 ? "196 KB shared rows per workgroup" : mode == 2 ? "55 KB own weights per workgroup" : "196 KB shared + 55 KB own");
        }
    }
    // every load in flight at once: 48 x 4 KB = 196 KB shared rows, 12 x 4 KB = 49 KB (a quarter of the rows: what a 4-way K split
    // or an activation-stationary workgroup reads), 14 x 4 KB = 56 KB own weights
    auto deep = [&](const char* label, auto kern, int wgs) {
        const double t = time_graph([&](hipStream_t s, int i) {
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, s, (const char*)((i & 1) ? o0 : o1), (const char*)(w + (size_t)(i % NW) * 256 * W_BYTES), (long)W_BYTES,
                               (i & 1) ? o1 : o0, OUT_WORDS);
        }, NL, REPS);
        printf("  %-34s %3d wgs  %.2f\n", label, wgs, t);
    };
    for (int wgs : {144, 256}) {
        deep("deep: 196 KB shared", k_rows_deep<48, 0>, wgs);
        deep("deep: 56 KB own", k_rows_deep<0, 14>, wgs);
        deep("deep: 196 KB shared + 56 KB own", k_rows_deep<48, 14>, wgs);
        deep("deep: 49 KB shared + 56 KB own", k_rows_deep<12, 14>, wgs);
        deep("deep: 49 KB shared + 16 KB own", k_rows_deep<12, 4>, wgs);
    }
    for (int wgs : {144, 256}) {
        const double t = time_graph([&](hipStream_t s, int i) {
            hipLaunchKernelGGL(k_rows_gather, dim3(wgs), dim3(256), 0, s, (const char*)((i & 1) ? o0 : o1), (i & 1) ? o1 : o0, OUT_WORDS);
        }, NL, REPS);
        printf("  %-34s %3d wgs  %.2f\n", "deep: 196 KB shared, 8 rows x 128 B", wgs, t);
    }
    // plain loads vs LDS-DMA at equal traffic: 128 KB shared rows + 16 KB own weights per workgroup, everything in flight
    auto dma = [&](const char* label, auto kern, int wgs, int smem) {
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        const double t = time_graph([&](hipStream_t s, int i) {
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), smem, s, (const char*)((i & 1) ? o0 : o1), (const char*)(w + (size_t)(i % NW) * 256 * W_BYTES), (long)W_BYTES,
                               (i & 1) ? o1 : o0, OUT_WORDS);
        }, NL, REPS);
        printf("  %-34s %3d wgs  %.2f\n", label, wgs, t);
    };
    for (int wgs : {144, 256}) {
        deep("plain: 128 KB shared + 16 KB own", k_rows_deep<32, 4>, wgs);
        dma("lds-dma: 128 KB shared + 16 KB own", k_rows_dma<32, 4>, wgs, 36 * 4096);
        deep("plain: 16 KB own", k_rows_deep<0, 4>, wgs);
        dma("lds-dma: 16 KB own", k_rows_dma<0, 4>, wgs, 4 * 4096);
    }
    return 0;
}
