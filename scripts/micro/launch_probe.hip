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
    CK(hipMalloc(&w, (size_t)NW * 256 * W_BYTES));
    CK(hipMemset(w, 1, (size_t)NW * 256 * W_BYTES));
    CK(hipMalloc(&o0, 256 * OUT_WORDS * 4)); CK(hipMalloc(&o1, 256 * OUT_WORDS * 4));
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
    return 0;
}
