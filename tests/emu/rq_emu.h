// rq_emu.h -- TEST INFRASTRUCTURE ONLY.  A host emulator for the subset of HIP that csrc/ uses.
//
// The container this repo is built in has no GPU; gpurun minutes are scarce.  To debug kernel index
// math (MFMA fragment maps, LDS swizzles, conv gathers, device-side step counters) locally, the
// kernel sources are compiled for the host with -DRQ_EMU and every GPU thread becomes a ucontext
// fiber.  Workgroups run one after another; inside a workgroup fibers run round-robin and switch only
// at __syncthreads() and at wave collectives (shuffles, MFMA), which exchange operands through a
// per-wave mailbox.  Semantics emulated: wave64, the gfx950 MFMA operand/result lane maps given in
// /opt/skills/guides/cdna_hip_programming.md §3, LDS as per-workgroup memory.  NOT emulated: memory
// model/races, occupancy, LDS capacity, alignment faults, timing.  The product (librqamd.so) is built
// by hipcc and never includes this file; nothing under rq-vae-transformer_amd/ routes here.
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace rqemu {

constexpr int kWave = 64;
constexpr size_t kStack = 256 * 1024;
constexpr int kMail = 160;   // bytes per lane per exchange

struct Wave {
    int nalive = 0;
    int arrived[2] = {0, 0};
    unsigned char mail[2][kWave][kMail];
};

// one lane's share (16 bytes) of an LDS-DMA instruction that has been issued and has not landed yet (RQ_EMU_DMA=late)
struct PendingDma {
    char* dst;
    unsigned char data[16];
};

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    dim3 tid;
    int lane = 0, wave = 0;
    unsigned gen = 0;   // per-fiber count of wave exchanges
    bool done = false;
    std::vector<PendingDma> dma;   // issue order; front = oldest
};

// RQ_EMU_DMA=late (environment, read at every launch): an LDS-DMA lands when the issuing lane's counted s_waitcnt vmcnt(N) retires it
// (or when its kernel ends) -- the LATEST moment the hardware allows -- so a ds_read that is not ordered behind the covering wait
// (and, for other wavefronts' reads, a barrier after it) returns the OLD LDS bytes, as it may on the GPU: a RAW hazard turns into
// wrong results.  Default (early): the DMA lands at issue, the EARLIEST moment, which is the worst case for WAR hazards (a unit
// refilled while some wavefront still has to read it).  A kernel's LDS-DMA schedule should pass in both modes.
extern bool g_dma_late;
inline void dma_land_until(Fiber& f, size_t keep) {
    size_t n = f.dma.size() > keep ? f.dma.size() - keep : 0;
    for (size_t i = 0; i < n; ++i) memcpy(f.dma[i].dst, f.dma[i].data, 16);
    if (n) f.dma.erase(f.dma.begin(), f.dma.begin() + (long)n);
}

struct Block {
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    int nalive = 0, bar_arrived = 0;
    unsigned bar_gen = 0;
    std::vector<unsigned char> dyn;
    unsigned long progress = 0;
};

extern ucontext_t g_main;
extern Fiber* g_cur;
extern Block* g_blk;
extern dim3 g_blockIdx, g_blockDim, g_gridDim;
extern std::function<void()> g_entry;

inline void yield() { swapcontext(&g_cur->ctx, &g_main); }

inline void block_barrier() {
    Block& b = *g_blk;
    unsigned my = b.bar_gen;
    b.progress++;
    if (++b.bar_arrived >= b.nalive) { b.bar_arrived = 0; b.bar_gen++; return; }
    while (b.bar_gen == my) yield();
}

// deposit `bytes` of this lane's operand; returns the mailbox of all 64 lanes for this exchange
inline const unsigned char (*wave_exchange(const void* mine, int bytes))[kMail] {
    Fiber& f = *g_cur;
    Wave& w = g_blk->waves[f.wave];
    int p = f.gen & 1;
    if (bytes > kMail) { fprintf(stderr, "rqemu: mailbox too small\n"); abort(); }
    memcpy(w.mail[p][f.lane], mine, bytes);
    f.gen++;
    g_blk->progress++;
    if (++w.arrived[p] >= w.nalive) w.arrived[1 - p] = 0;   // everyone is past the previous exchange
    while (w.arrived[p] < w.nalive) yield();
    return w.mail[p];
}

void run_grid(dim3 grid, dim3 block, size_t smem);
unsigned char* dyn_smem();
size_t dyn_smem_size();

template <typename K, typename... A>
inline void launch(K kern, dim3 grid, dim3 block, size_t smem, A... args) {
    g_entry = [=]() { kern(args...); };
    run_grid(grid, block, smem);
}

}  // namespace rqemu

#define threadIdx (rqemu::g_cur->tid)
#define blockIdx (rqemu::g_blockIdx)
#define blockDim (rqemu::g_blockDim)
#define gridDim (rqemu::g_gridDim)

#define RQ_DYN_SMEM(name) unsigned char* name = rqemu::dyn_smem()
#define RQ_LAUNCH(kern, grid, block, smem, stream, ...) rqemu::launch(kern, grid, block, smem, __VA_ARGS__)

// ---------------------------------------------------------------------------------------------
typedef float f32x4_emu __attribute__((ext_vector_type(4)));
typedef float f32x16_emu __attribute__((ext_vector_type(16)));
typedef short bf16x8_emu __attribute__((ext_vector_type(8)));

static inline void rq_syncthreads() { rqemu::block_barrier(); }
// LDS-DMA: each lane copies its 16 bytes to lds_base + lane * 16; the "address" is the host pointer itself
typedef uintptr_t rq_lds_t;
static inline uintptr_t rq_lds_addr(const void* p) { return (uintptr_t)p; }
static inline void rq_glds16(uintptr_t lds_base, const void* gsrc) {
    char* dst = (char*)lds_base + 16 * rqemu::g_cur->lane;
    if (dst < (char*)rqemu::dyn_smem() || dst + 16 > (char*)rqemu::dyn_smem() + rqemu::dyn_smem_size()) {
        fprintf(stderr, "rq_glds16: LDS destination %ld outside the %zu-byte dynamic segment\n", (long)(dst - (char*)rqemu::dyn_smem()), rqemu::dyn_smem_size());
        abort();
    }
    if (rqemu::g_dma_late) {
        rqemu::PendingDma pd;
        pd.dst = dst;
        memcpy(pd.data, gsrc, 16);
        rqemu::g_cur->dma.push_back(pd);
    } else {
        memcpy(dst, gsrc, 16);
    }
}
static inline void rq_glds16_nt(uintptr_t lds_base, const void* gsrc) { rq_glds16(lds_base, gsrc); }      // (cache policy: not modelled)
static inline void rq_glds16_s(uintptr_t lds_base, const void* sbase, unsigned voff) { rq_glds16(lds_base, (const char*)sbase + voff); }
template <int POL = 0>
static inline void rq_glds16_s2(uintptr_t lds_base, const void* sbase, unsigned voff0, unsigned voff1) {
    rq_glds16(lds_base, (const char*)sbase + voff0);
    rq_glds16(lds_base + 1024, (const char*)sbase + voff1);
}
// counted wait: the issuing lane's DMAs land, oldest first, until at most N are outstanding (hardware also counts the lane's other
// global loads there; the kernels that use counted waits issue none between their DMAs)
template <int N> static inline void rq_wait_vmcnt() { if (rqemu::g_dma_late) rqemu::dma_land_until(*rqemu::g_cur, (size_t)N); }
template <int ND, int NO> static inline void rq_wait_vmcnt_mixed() { rq_wait_vmcnt<ND>(); }   // (ordinary loads are synchronous here)
template <int N> static inline void rq_wait_lgkmcnt() {}
// A wavefront executes in lockstep: past any point of the program every lane has issued everything before it.  The emulator runs
// lanes as fibers, so kernels whose lanes hand data to each other through LDS WITHOUT a workgroup barrier (gemm_stream_kernel's
// wavefront-private LDS-DMA rings) mark those points with rq_wave_sync(): a wave-wide rendezvous here (a 1-byte exchange), nothing
// but a scheduling fence on the GPU.
static inline void rq_wave_sync() { unsigned char z = 0; (void)rqemu::wave_exchange(&z, 1); }
static inline void rq_barrier_raw() { rqemu::block_barrier(); }
#define rq_sched_barrier() ((void)0)
#define rq_setprio(x) ((void)0)
#define rq_sched_group(mask, n) ((void)0)
static inline int rq_uniform(int x) { return x; }
static inline unsigned long long rq_ballot(bool pred) {
    unsigned char mine = pred ? 1 : 0;
    auto mail = rqemu::wave_exchange(&mine, 1);
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) if (mail[l][0]) m |= 1ull << l;
    return m;
}
static inline int rq_popc64(unsigned long long m) { return __builtin_popcountll(m); }
static inline void rq_threadfence_block() {}
static inline void rq_opaque(int&) {}
static inline void rq_opaque_u(uint32_t&) {}
static inline float rq_dot2_bf16(uint32_t a, uint32_t b, float acc) {
    union { uint32_t u; float f; } al, ah, bl, bh;
    al.u = a << 16; ah.u = a & 0xffff0000u; bl.u = b << 16; bh.u = b & 0xffff0000u;
    return acc + (al.f * bl.f + ah.f * bh.f);
}
static inline void rq_opaque_acc(f32x16_emu&) {}
static inline void rq_opaque_f4(f32x4_emu&) {}
static inline void rq_use(unsigned, unsigned, unsigned, unsigned) {}
static inline void rq_use(float, float) {}
static inline void rq_trap() { abort(); }
static inline float rq_fast_rcp(float x) { return 1.0f / x; }
static inline float rq_med3(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }
static inline float rq_fast_exp2(float x) { return exp2f(x); }
template <int I> static inline float rq_ubyte_f32(uint32_t w) { return (float)((w >> (8 * I)) & 0xffu); }

static inline float rq_emu_bf16(short v) {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)(unsigned short)v) << 16;
    return c.f;
}

template <typename T>
static inline T rq_emu_shfl(T v, int src) {
    auto mail = rqemu::wave_exchange(&v, sizeof(T));
    T r;
    memcpy(&r, mail[src & 63], sizeof(T));
    return r;
}
static inline float rq_shfl_xor(float v, int m) { return rq_emu_shfl(v, rqemu::g_cur->lane ^ m); }
static inline int rq_shfl_xor_i(int v, int m) { return rq_emu_shfl(v, rqemu::g_cur->lane ^ m); }
static inline float rq_shfl(float v, int lane) { return rq_emu_shfl(v, lane); }
static inline int rq_shfl_i(int v, int lane) { return rq_emu_shfl(v, lane); }
static inline float rq_dpp_xor1(float v) { return rq_emu_shfl(v, rqemu::g_cur->lane ^ 1); }
static inline float rq_dpp_xor2(float v) { return rq_emu_shfl(v, rqemu::g_cur->lane ^ 2); }
static inline float rq_dpp_half_mirror(float v) { const int l = rqemu::g_cur->lane; return rq_emu_shfl(v, (l & ~7) | (7 - (l & 7))); }
static inline float rq_dpp_ror8(float v) { const int l = rqemu::g_cur->lane; return rq_emu_shfl(v, (l & ~15) | ((l + 8) & 15)); }
static inline float rq_readlane(float v, int lane) { return rq_emu_shfl(v, lane); }
static inline int rq_dpp_xor1_i(int v) { return rq_emu_shfl(v, rqemu::g_cur->lane ^ 1); }
static inline int rq_dpp_xor2_i(int v) { return rq_emu_shfl(v, rqemu::g_cur->lane ^ 2); }
static inline int rq_dpp_half_mirror_i(int v) { const int l = rqemu::g_cur->lane; return rq_emu_shfl(v, (l & ~7) | (7 - (l & 7))); }
static inline int rq_dpp_ror8_i(int v) { const int l = rqemu::g_cur->lane; return rq_emu_shfl(v, (l & ~15) | ((l + 8) & 15)); }
static inline int rq_readlane_i(int v, int lane) { return rq_emu_shfl(v, lane); }

// v_mfma_f32_32x32x16_bf16: A lane l holds A[i=l&31][k=8*(l>>5)+j], B lane l holds B[k=8*(l>>5)+j][n=l&31],
// C/D reg r of lane l is (row=(r&3)+8*(r>>2)+4*(l>>5), col=l&31).   (cdna_hip_programming.md §3)
static inline f32x16_emu rq_mfma_32x32x16_bf16(bf16x8_emu a, bf16x8_emu b, f32x16_emu c) {
    struct { short a[8], b[8]; } mine;
    for (int j = 0; j < 8; ++j) { mine.a[j] = a[j]; mine.b[j] = b[j]; }
    auto mail = rqemu::wave_exchange(&mine, sizeof(mine));
    int l = rqemu::g_cur->lane, col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int kh = 0; kh < 2; ++kh) {
            const short* pa = (const short*)mail[row + 32 * kh];
            const short* pb = (const short*)mail[col + 32 * kh] + 8;
            for (int j = 0; j < 8; ++j) acc = fmaf(rq_emu_bf16(pa[j]), rq_emu_bf16(pb[j]), acc);
        }
        c[r] = acc;
    }
    return c;
}
// v_mfma_f32_16x16x32_bf16: A lane l: A[i=l&15][k=8*(l>>4)+j]; B lane l: B[k=8*(l>>4)+j][n=l&15];
// C/D reg r: (row=4*(l>>4)+r, col=l&15).
static inline f32x4_emu rq_mfma_16x16x32_bf16(bf16x8_emu a, bf16x8_emu b, f32x4_emu c) {
    struct { short a[8], b[8]; } mine;
    for (int j = 0; j < 8; ++j) { mine.a[j] = a[j]; mine.b[j] = b[j]; }
    auto mail = rqemu::wave_exchange(&mine, sizeof(mine));
    int l = rqemu::g_cur->lane, col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int kq = 0; kq < 4; ++kq) {
            const short* pa = (const short*)mail[row + 16 * kq];
            const short* pb = (const short*)mail[col + 16 * kq] + 8;
            for (int j = 0; j < 8; ++j) acc = fmaf(rq_emu_bf16(pa[j]), rq_emu_bf16(pb[j]), acc);
        }
        c[r] = acc;
    }
    return c;
}
// v_mfma_f32_32x32x2_f32: A lane l: A[i=l&31][k=l>>5]; B lane l: B[k=l>>5][n=l&31]; C/D as 32x32 above.
// Exact f32: a k-ordered fmaf chain (guide §3 "Numerics").
static inline f32x16_emu rq_mfma_32x32x2_f32(float a, float b, f32x16_emu c) {
    struct { float a, b; } mine = {a, b};
    auto mail = rqemu::wave_exchange(&mine, sizeof(mine));
    int l = rqemu::g_cur->lane, col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av = ((const float*)mail[row + 32 * k])[0];
            float bv = ((const float*)mail[col + 32 * k])[1];
            acc = fmaf(av, bv, acc);
        }
        c[r] = acc;
    }
    return c;
}

// ---------------------------------------------------------------------------------------------
// device math / atomics
#define __expf expf
#define __logf logf
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }

// ---------------------------------------------------------------------------------------------
// host runtime subset
typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
typedef void* hipEvent_t;
#define hipSuccess 0
#define hipErrorNotSupported 801
#define hipErrorOutOfMemory 2
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); memset(*p, 0xFF, n); return *p ? 0 : 2; }
#define hipDeviceMallocUncached 0x3
static inline hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipFree(void* p) { free(p); return 0; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
    for (size_t i = 0; i < h; ++i) memcpy((char*)d + i * dp, (const char*)s + i * sp, w);
    return 0;
}
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*) { return hipErrorNotSupported; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return hipErrorNotSupported; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return 0; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return 0; }
#define hipFuncAttributeMaxDynamicSharedMemorySize 8
template <typename F> static inline hipError_t hipFuncSetAttribute(F, int, int) { return 0; }
#define hipDeviceAttributeMultiprocessorCount 63
static inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 256; return 0; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return 0; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
