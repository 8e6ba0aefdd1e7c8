// Microbenchmark: cost of a software grid barrier (one workgroup per CU, device-scope atomics) on gfx950, with and without a
// producer -> consumer exchange through global memory across it (each workgroup writes 4 KB, reads its neighbour's on another XCD).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/grid_barrier.hip -o grid_barrier && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned& epoch, unsigned G) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        epoch += G;
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { ok = false; break; }      // never hang the box
        }
    }
    __syncthreads();
    return ok;
}

// flag-array form: workgroup w publishes its epoch in flags[w] (one store, no read-modify-write on a shared line); thread t of
// every workgroup waits for flags[t], flags[t + 256], ...
__device__ __forceinline__ bool grid_barrier_flags(unsigned* flags, unsigned& epoch, unsigned G) {
    __syncthreads();
    ++epoch;
    if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x * 32, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = true;
    for (unsigned w = threadIdx.x; w < G; w += blockDim.x) {
        unsigned spins = 0;
        while (__hip_atomic_load(flags + w * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { ok = false; break; }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);      // (agent scope by default on this target)
    return __syncthreads_and(ok);
}

// two-level form: one counter per XCD (blockIdx & 7), the last arrival of each bumps the global counter
__device__ __forceinline__ bool grid_barrier_tree(unsigned* ctr, unsigned& epoch, unsigned G) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        ++epoch;
        const unsigned x = blockIdx.x & 7, nx = (G - x + 7) / 8;          // workgroups on this XCD
        const unsigned old = __hip_atomic_fetch_add(ctr + 32 * (1 + x), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == epoch * nx) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = epoch * (G < 8 ? G : 8);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { ok = false; break; }
        }
    }
    return __syncthreads_and(ok);
}

// no cache maintenance at all: the counter AND the exchanged data move with relaxed agent-scope atomics (sc1 accesses, coherent
// across the XCDs' L2s); s_waitcnt vmcnt(0) before the arrival orders the data stores ahead of it
__device__ __forceinline__ bool grid_barrier_relaxed(unsigned* ctr, unsigned& epoch, unsigned G) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        epoch += G;
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { ok = false; break; }
        }
    }
    ok = __syncthreads_and(ok);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return ok;
}

// the same without cache maintenance, arrivals spread over one counter per XCD (KIND 4) or one flag per workgroup (KIND 5)
__device__ __forceinline__ bool grid_barrier_relaxed_tree(unsigned* ctr, unsigned& epoch, unsigned G) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        ++epoch;
        const unsigned x = blockIdx.x & 7, nx = (G - x + 7) / 8;
        const unsigned old = __hip_atomic_fetch_add(ctr + 32 * (1 + x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == epoch * nx) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = epoch * (G < 8 ? G : 8);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { ok = false; break; }
        }
    }
    ok = __syncthreads_and(ok);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return ok;
}
__device__ __forceinline__ bool grid_barrier_relaxed_flags(unsigned* flags, unsigned& epoch, unsigned G) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    ++epoch;
    if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x * 32, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = true;
    for (unsigned w = threadIdx.x; w < G; w += blockDim.x) {
        unsigned spins = 0;
        while (__hip_atomic_load(flags + w * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { ok = false; break; }
        }
    }
    ok = __syncthreads_and(ok);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return ok;
}

template <int EXCH, int KIND>
__device__ __forceinline__ bool gbar(unsigned* ctr, unsigned& epoch, unsigned G) {
    return KIND == 0 ? grid_barrier(ctr, epoch, G) : KIND == 1 ? grid_barrier_flags(ctr, epoch, G) : KIND == 2 ? grid_barrier_tree(ctr, epoch, G) : KIND == 3 ? grid_barrier_relaxed(ctr, epoch, G) : KIND == 4 ? grid_barrier_relaxed_tree(ctr, epoch, G) : grid_barrier_relaxed_flags(ctr, epoch, G);
}

template <int EXCH, int KIND>
__global__ __launch_bounds__(256) void bar_kernel(unsigned* ctr, float* buf, int n, unsigned long long* cycles, int* bad) {
    extern __shared__ char smem[];          // > 80 KB requested by the host: one workgroup per CU
    const unsigned G = gridDim.x;
    unsigned epoch = 0;
    const int wg = blockIdx.x, tid = threadIdx.x;
    float acc = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        if (EXCH) {
            float4 v = make_float4(i + wg, tid, 1.f, 2.f);
            if (KIND >= 3) {
                unsigned long long* d = (unsigned long long*)buf + ((size_t)wg * 256 + tid) * 2;
                __hip_atomic_store(d, ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(d + 1, 0x400000003f800000ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else
            ((float4*)buf)[(size_t)wg * 256 + tid] = v;
        }
        if (!gbar<EXCH, KIND>(ctr, epoch, G)) { if (tid == 0) atomicAdd(bad, 1); return; }
        if (EXCH) {
            const int o = (wg + 1) % G;     // blockIdx -> XCD is round-robin: the neighbour lives on another XCD
            float4 v;
            if (KIND >= 3) {
                unsigned long long* d = (unsigned long long*)buf + ((size_t)o * 256 + tid) * 2;
                const unsigned long long a = __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(d + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v = make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b), __uint_as_float((unsigned)(b >> 32)));
            } else
            v = ((const float4*)buf)[(size_t)o * 256 + tid];
            if (v.x != (float)(i + o)) atomicAdd(bad, 1 << 8);
            acc += v.y;
            if (!gbar<EXCH, KIND>(ctr, epoch, G)) { if (tid == 0) atomicAdd(bad, 1); return; }      // before the buffer is overwritten
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cycles[wg] = t1 - t0;
    if (acc == -1.f) buf[0] = acc;
    (void)smem;
}

template <int EXCH, int KIND>
static void run(int G, int n) {
    unsigned* ctr; float* buf; unsigned long long* cyc; int* bad;
    hipMalloc(&ctr, 4 * 32 * 512); hipMalloc(&buf, (size_t)G * 4096); hipMalloc(&cyc, G * 8); hipMalloc(&bad, 4);
    hipFuncSetAttribute((const void*)bar_kernel<EXCH, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    int hbad = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(ctr, 0, 4 * 32 * 512); hipMemset(bad, 0, 4);
        hipEventRecord(e0);
        hipLaunchKernelGGL((bar_kernel<EXCH, KIND>), dim3(G), dim3(256), 96 * 1024, 0, ctr, buf, n, cyc, bad);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
    }
    const int nb = EXCH ? 2 * n : n;
    printf("%s %3d workgroups, %s: %7.3f us per barrier (%d barriers, %.1f us launch total)%s\n", KIND == 0 ? "one counter    " : KIND == 1 ? "flag per group " : KIND == 2 ? "counter per XCD" : KIND == 3 ? "relaxed + sc1  " : KIND == 4 ? "relaxed per XCD" : "relaxed flags  ", G, EXCH ? "4 KB exchange per workgroup + 2 barriers per round" : "barrier only",
           best * 1e3 / nb, nb, best * 1e3, hbad ? "  ** TIMEOUT / MISMATCH **" : "");
    if (hbad) printf("   bad = 0x%x\n", hbad);
    hipFree(ctr); hipFree(buf); hipFree(cyc); hipFree(bad);
}

int main() {
    int dev_cus = 0;
    hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("CUs: %d\n", dev_cus);
    for (int G : {64, 128, dev_cus}) {
        run<0, 0>(G, 2000); run<1, 0>(G, 1000);
        run<0, 1>(G, 2000); run<1, 1>(G, 1000);
        run<0, 2>(G, 2000); run<1, 2>(G, 1000);
        run<0, 3>(G, 2000); run<1, 3>(G, 1000);
        run<0, 4>(G, 2000); run<1, 4>(G, 1000);
        run<0, 5>(G, 2000); run<1, 5>(G, 1000);
    }
    return 0;
}
