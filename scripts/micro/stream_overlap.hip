// Microbenchmark: do kernels on two HIP streams overlap on this box?  A "spin" kernel of G workgroups busy-waits T microseconds
// (s_memrealtime); N launches per stream.  One stream: N * T.  Two streams: N * T when the work overlaps, 2 * N * T when the
// runtime / command processor serialises it.  Also through captured graphs (one graph per stream).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/stream_overlap.hip -o scripts/micro/bin/stream_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void spin(unsigned long long ticks, unsigned* sink, int lds_kb) {
    extern __shared__ unsigned char smem[];
    if (lds_kb && threadIdx.x == 0) smem[0] = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); }
    if (sink && threadIdx.x == 0 && blockIdx.x == 0 && ticks == 0) sink[0] = 1;
}

static double run(int nstreams, int G, int N, double us, int lds_kb, bool graph) {
    hipStream_t st[2];
    for (int s = 0; s < nstreams; ++s) hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking);
    const unsigned long long ticks = (unsigned long long)(us * 100.0);
    hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipGraphExec_t ge[2] = {nullptr, nullptr};
    if (graph) {
        for (int s = 0; s < nstreams; ++s) {
            hipGraph_t g;
            hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal);
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin, dim3(G), dim3(256), lds_kb * 1024, st[s], ticks, nullptr, lds_kb);
            hipStreamEndCapture(st[s], &g);
            hipGraphInstantiate(&ge[s], g, nullptr, nullptr, 0);
            hipGraphDestroy(g);
        }
    }
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipStreamWaitEvent(st[0], e0, 0);
    if (nstreams > 1) hipStreamWaitEvent(st[1], e0, 0);
    for (int s = 0; s < nstreams; ++s) {
        if (graph) hipGraphLaunch(ge[s], st[s]);
        else for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin, dim3(G), dim3(256), lds_kb * 1024, st[s], ticks, nullptr, lds_kb);
    }
    hipDeviceSynchronize();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    for (int s = 0; s < nstreams; ++s) { if (ge[s]) hipGraphExecDestroy(ge[s]); hipStreamDestroy(st[s]); }
    return ms;
}

int main() {
    const int N = 200;
    const double us = 20.0;
    for (int graph = 0; graph < 2; ++graph)
        for (int lds : {0, 100})
            for (int G : {32, 128, 256}) {
                const double a = run(1, G, N, us, lds, graph), b = run(2, G, N, us, lds, graph);
                printf("%s, %3d workgroups x 256 threads, %3d KB LDS each, %d launches of %.0f us per stream: one stream %6.2f ms, two streams %6.2f ms (%.2fx)\n",
                       graph ? "graphs " : "streams", G, lds, N, us, a, b, b / a);
            }
    return 0;
}
