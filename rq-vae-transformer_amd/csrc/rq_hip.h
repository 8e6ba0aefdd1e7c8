// rq_hip.h -- device/host prelude shared by every kernel file of librqamd (gfx950 only).
//
// Product builds go through hipcc (--offload-arch=gfx950).  tests/emu/ compiles the very same
// kernel sources for the host with -DRQ_EMU against a fiber-based wave64/workgroup emulator, so
// that index math can be checked in the CPU-only container; that build is test infrastructure and
// never ships (see tests/emu/README.md).
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef RQ_EMU
#include "rq_emu.h"
#else
#include <hip/hip_runtime.h>
#endif

// ---------------------------------------------------------------------------------------------
// vector / fragment types
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));   // 8 bf16 = one MFMA A/B fragment (4 VGPRs)
typedef unsigned short bf16_t;                                // raw bf16 bits

#define RQ_WAVE 64

// ---------------------------------------------------------------------------------------------
// The 16-bit storage type of activations / weights / KV cache.  Default: bfloat16 (what BASELINE.json asks for).  Compiled with
// -DRQ_F16=1 (librqamd_f16.so: the RQ-Transformer engine behind sample(amp=True) / forward(amp=True)) the SAME sources store IEEE
// fp16 instead -- what the reference's amp=True autocast computes in (transformers.py:21,206) -- through these few helpers: the
// conversions, the packed-pair unpack, the two MFMA wrappers and the packed dot product.  `bf16_t` / `bf16x8` keep their names
// (raw 16-bit words); accumulation, residual stream, LayerNorm, softmax and logits are fp32 either way.
#ifndef RQ_F16
#define RQ_F16 0
#endif
#if RQ_F16
static inline __host__ __device__ float bf16_to_f32(bf16_t v) {
    union { uint16_t u; _Float16 h; } c;
    c.u = v;
    return (float)c.h;
}
static inline __host__ __device__ bf16_t f32_to_bf16(float f) {      // round-to-nearest-even; beyond 65504 -> inf, as torch.float16 does
    union { uint16_t u; _Float16 h; } c;
    c.h = (_Float16)f;
    return c.u;
}
static inline __host__ __device__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}
#else
// bf16 <-> f32 (round-to-nearest-even), usable on host and device
static inline __host__ __device__ float bf16_to_f32(bf16_t v) {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)v) << 16;
    return c.f;
}
static inline __host__ __device__ bf16_t f32_to_bf16(float f) {
    union { uint32_t u; float f; } c;
    c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 converts in hardware (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN).  The bit-twiddling form
// above costs ~9 VALU ops per element and, through its NaN branch, an exec-mask diamond per element that also
// keeps the scheduler from moving anything across it (seen in the fused GroupNorm staging of conv_halo.hip).
static __device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef float rq_f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 rq_bf16x2_t __attribute__((ext_vector_type(2)));
    const rq_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, rq_bf16x2_t));
}
#else
static inline __host__ __device__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}
#endif
#endif
// 1.0 in both halves of a packed word of the 16-bit storage type
#if RQ_F16
#define RQ_ONE_X2 0x3c003c00u
#else
#define RQ_ONE_X2 0x3f803f80u
#endif
// the two 16-bit values of a packed word as floats (bf16: a shift and a mask)
static inline __host__ __device__ void rq_unpack2(uint32_t w, float& lo, float& hi) {
#if RQ_F16
    lo = bf16_to_f32((bf16_t)(w & 0xffffu));
    hi = bf16_to_f32((bf16_t)(w >> 16));
#else
    union { uint32_t u; float f; } a, b;
    a.u = w << 16;
    b.u = w & 0xffff0000u;
    lo = a.f;
    hi = b.f;
#endif
}

// ---------------------------------------------------------------------------------------------
// wave-collective wrappers (one spelling for hipcc and the emulator)
#ifndef RQ_EMU
#if RQ_F16
typedef _Float16 rq_f16x8 __attribute__((ext_vector_type(8)));
static __device__ __forceinline__ f32x16 rq_mfma_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rq_f16x8, a), __builtin_bit_cast(rq_f16x8, b), c, 0, 0, 0);
}
static __device__ __forceinline__ f32x4 rq_mfma_16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rq_f16x8, a), __builtin_bit_cast(rq_f16x8, b), c, 0, 0, 0);
}
#else
static __device__ __forceinline__ f32x16 rq_mfma_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
static __device__ __forceinline__ f32x4 rq_mfma_16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
#endif
static __device__ __forceinline__ f32x16 rq_mfma_32x32x2_f32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
static __device__ __forceinline__ float rq_shfl_xor(float v, int m) { return __shfl_xor(v, m, 64); }
static __device__ __forceinline__ int rq_shfl_xor_i(int v, int m) { return __shfl_xor(v, m, 64); }
static __device__ __forceinline__ float rq_shfl(float v, int lane) { return __shfl(v, lane, 64); }
static __device__ __forceinline__ int rq_shfl_i(int v, int lane) { return __shfl(v, lane, 64); }
static __device__ __forceinline__ void rq_syncthreads() { __syncthreads(); }
// ---- LDS-DMA (global_load_lds_dwordx4): 16 bytes per lane straight from global memory into LDS at
// lds_base + lane * 16 (lds_base wave-uniform), no VGPR staging and no ds_write.  hipcc does not count these loads
// in its s_waitcnt bookkeeping: pair every use with rq_wait_vmcnt<N>() + a barrier before the data is read.
typedef unsigned rq_lds_t;
static __device__ __forceinline__ unsigned rq_lds_addr(const void* p) {          // byte address inside the LDS aperture
    return __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p);
}
static __device__ __forceinline__ void rq_glds16(unsigned lds_base, const void* gsrc) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
// the same with the non-temporal policy (` nt`): for bytes that ONE compute unit reads once (a decode step's weight stream;
// MI355X_MICROARCH.md, row nt-weights) -- not for operands that other workgroups re-read from the L2
static __device__ __forceinline__ void rq_glds16_nt(unsigned lds_base, const void* gsrc) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
// Same DMA with the source as (wave-uniform 64-bit base in SGPRs) + (per-lane 32-bit byte offset): the per-lane offset is
// loop invariant and the K-tile advance is one scalar add on the base, so a staging instruction costs no VALU work at all.
static __device__ __forceinline__ void rq_glds16_s(unsigned lds_base, const void* sbase, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_base) : "memory");
}
// Two of them, to lds_base and lds_base + 1024 (the two 8-row groups a wavefront owns in a 16-KB unit), in one statement: M0 is
// saved and restored once.
// POL: cache-policy suffix of the loads (A/B switch of gemm_p8_kernel: 0 default, 1 " nt", 2 " sc1", 3 " sc0 sc1")
template <int POL = 0>
static __device__ __forceinline__ void rq_glds16_s2(unsigned lds_base, const void* sbase, unsigned voff0, unsigned voff1) {
    unsigned keep;
#define RQ_GLDS2(P) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3" P "\n\t" \
                                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3" P "\n\ts_mov_b32 m0, %0" \
                                 : "=&s"(keep) : "v"(voff0), "v"(voff1), "s"(sbase), "s"(lds_base) : "memory", "scc")
    if (POL == 1) RQ_GLDS2(" nt");
    else if (POL == 2) RQ_GLDS2(" sc1");
    else if (POL == 3) RQ_GLDS2(" sc0 sc1");
    else RQ_GLDS2("");
#undef RQ_GLDS2
}
template <int N> static __device__ __forceinline__ void rq_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
// counted wait in a kernel that also has ordinary global loads in flight: at most ND of the wavefront's LDS-DMAs and NO of its other
// (younger) vector-memory loads stay outstanding.  The hardware has one counter for both (they retire in issue order); the host
// emulator, whose ordinary loads are synchronous, counts the DMAs alone.
template <int ND, int NO> static __device__ __forceinline__ void rq_wait_vmcnt_mixed() { rq_wait_vmcnt<ND + NO>(); }
template <int N> static __device__ __forceinline__ void rq_wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N) : "memory"); }
static __device__ __forceinline__ void rq_barrier_raw() { __builtin_amdgcn_s_barrier(); }
// lockstep point of one wavefront (no instruction: a wavefront IS in lockstep; the host emulator, whose lanes are fibers, meets here)
static __device__ __forceinline__ void rq_wave_sync() { __builtin_amdgcn_wave_barrier(); }
// pins instruction order at this point (hipcc otherwise sinks independent global loads below LDS writes)
#define rq_sched_barrier() __builtin_amdgcn_sched_barrier(0)
#define rq_setprio(x) __builtin_amdgcn_s_setprio(x)
// scheduling pipeline hint: the next `n` instructions of class `mask` (0x8 MFMA, 0x2 VALU, 0x100 DS read, ...) form a group
#define rq_sched_group(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
// single-instruction reciprocal / exp2 (v_rcp_f32 / v_exp_f32, ~1 ulp): used where the result is rounded to bf16
// wave-uniform value -> SGPR (lets address arithmetic derived from it run on the scalar unit)
static __device__ __forceinline__ int rq_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
static __device__ __forceinline__ unsigned long long rq_ballot(bool pred) { return __ballot(pred); }
static __device__ __forceinline__ int rq_popc64(unsigned long long m) { return __popcll(m); }
static __device__ __forceinline__ void rq_threadfence_block() { __threadfence_block(); }
// DPP lane exchanges (one VALU op, no LDS round trip -- a ds_bpermute shuffle costs ~100 cycles of latency):
// xor 1 / xor 2 inside a quad, mirror inside 8 lanes (pairs quad 0 with quad 1: a valid "xor 4" once the quads are
// uniform), rotate by 8 inside a row of 16 (= xor 8), and a scalar read of one lane.
template <int CTRL> static __device__ __forceinline__ float rq_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
static __device__ __forceinline__ float rq_dpp_xor1(float v) { return rq_dpp<0xB1>(v); }          // quad_perm [1,0,3,2]
static __device__ __forceinline__ float rq_dpp_xor2(float v) { return rq_dpp<0x4E>(v); }          // quad_perm [2,3,0,1]
static __device__ __forceinline__ float rq_dpp_half_mirror(float v) { return rq_dpp<0x141>(v); }   // lane i <- lane 7 - i (per 8)
static __device__ __forceinline__ float rq_dpp_ror8(float v) { return rq_dpp<0x128>(v); }          // lane i <- lane (i + 8) % 16 (per 16)
static __device__ __forceinline__ float rq_readlane(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
template <int CTRL> static __device__ __forceinline__ int rq_dpp_int(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
static __device__ __forceinline__ int rq_dpp_xor1_i(int v) { return rq_dpp_int<0xB1>(v); }
static __device__ __forceinline__ int rq_dpp_xor2_i(int v) { return rq_dpp_int<0x4E>(v); }
static __device__ __forceinline__ int rq_dpp_half_mirror_i(int v) { return rq_dpp_int<0x141>(v); }
static __device__ __forceinline__ int rq_dpp_ror8_i(int v) { return rq_dpp_int<0x128>(v); }
static __device__ __forceinline__ int rq_readlane_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
// makes `x` opaque to the optimiser at this point (keeps loop-invariant address arithmetic from being hoisted out of a
// persistent tile loop, where it would hold dozens of registers across the whole main loop)
static __device__ __forceinline__ void rq_opaque(int& x) { asm volatile("" : "+v"(x)); }
// the same for a 32-bit word of data: no producer of `x` is scheduled below this point and no consumer above it (an asm
// statement is ordered against sched_barriers and other asm statements, register arithmetic is ordered only through operands)
static __device__ __forceinline__ void rq_opaque_u(uint32_t& x) { asm volatile("" : "+v"(x)); }
static __device__ __forceinline__ void rq_opaque_acc(f32x16& x) { asm volatile("" : "+v"(x)); }
static __device__ __forceinline__ void rq_opaque_f4(f32x4& x) { asm volatile("" : "+v"(x)); }
// "these values are needed now": makes the compiler place its wait for the loads that produce them here
static __device__ __forceinline__ void rq_use(unsigned a, unsigned b, unsigned c, unsigned d) { asm volatile("" :: "v"(a), "v"(b), "v"(c), "v"(d)); }
static __device__ __forceinline__ void rq_use(float a, float b) { asm volatile("" :: "v"(a), "v"(b)); }
// acc + a.lo * b.lo + a.hi * b.hi on packed bf16 pairs, fp32 accumulate (v_dot2c_f32_bf16)
static __device__ __forceinline__ float rq_dot2_bf16(uint32_t a, uint32_t b, float acc) {
#if RQ_F16
    typedef _Float16 rq_f16x2_v __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(rq_f16x2_v, a), __builtin_bit_cast(rq_f16x2_v, b), acc, false);      // v_dot2_f32_f16
#else
    typedef __bf16 rq_bf16x2_v __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(rq_bf16x2_v, a), __builtin_bit_cast(rq_bf16x2_v, b), acc, false);
#endif
}
// byte I (0..3) of a word as a float (the back end selects v_cvt_f32_ubyteI for this pattern: one instruction): the dequantisation
// of the opt-in 8-bit key cache
template <int I> static __device__ __forceinline__ float rq_ubyte_f32(uint32_t w) { return (float)((w >> (8 * I)) & 0xffu); }
static __device__ __forceinline__ void rq_trap() { __builtin_trap(); }
static __device__ __forceinline__ float rq_fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
static __device__ __forceinline__ float rq_med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }   // median: clamp in one op
static __device__ __forceinline__ float rq_fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
#define RQ_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#define RQ_LAUNCH(kern, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kern, grid, block, smem, (hipStream_t)(stream), __VA_ARGS__)
#endif

// ---------------------------------------------------------------------------------------------
// wave reductions
// Full wavefronts only (every kernel here exits whole wavefronts).  Four DPP steps make the 16 lanes of a row agree, four
// scalar lane reads combine the rows: no ds_bpermute round trips (~100 cycles each, six in a row, in the xor-shuffle form
// this replaces -- the sampler's top-k / top-p searches run one such reduction per iteration).
static __device__ __forceinline__ float wave_sum(float v) {
    v += rq_dpp_xor1(v); v += rq_dpp_xor2(v); v += rq_dpp_half_mirror(v); v += rq_dpp_ror8(v);
    return (rq_readlane(v, 0) + rq_readlane(v, 16)) + (rq_readlane(v, 32) + rq_readlane(v, 48));
}
static __device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, rq_dpp_xor1(v)); v = fmaxf(v, rq_dpp_xor2(v)); v = fmaxf(v, rq_dpp_half_mirror(v)); v = fmaxf(v, rq_dpp_ror8(v));
    return fmaxf(fmaxf(rq_readlane(v, 0), rq_readlane(v, 16)), fmaxf(rq_readlane(v, 32), rq_readlane(v, 48)));
}
static __device__ __forceinline__ int wave_sum_i(int v) {
    v += rq_dpp_xor1_i(v); v += rq_dpp_xor2_i(v); v += rq_dpp_half_mirror_i(v); v += rq_dpp_ror8_i(v);
    return (rq_readlane_i(v, 0) + rq_readlane_i(v, 16)) + (rq_readlane_i(v, 32) + rq_readlane_i(v, 48));
}

// 16-byte global/LDS access helpers
struct __attribute__((aligned(16))) rq_u128 { uint32_t x, y, z, w; };
static __device__ __forceinline__ rq_u128 ld128(const void* p) { return *(const rq_u128*)p; }
static __device__ __forceinline__ void st128(void* p, rq_u128 v) { *(rq_u128*)p = v; }
// non-temporal forms for data that streams through once per launch (KV cache, activations beyond any cache): same values, another cache policy
static __device__ __forceinline__ rq_u128 ld128_nt(const void* p) {
    typedef unsigned rq_v4u_t __attribute__((ext_vector_type(4)));
    const rq_v4u_t v = __builtin_nontemporal_load((const rq_v4u_t*)p);
    rq_u128 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
    return r;
}
static __device__ __forceinline__ void st128_nt(void* p, rq_u128 v) {
    typedef unsigned rq_v4u_t __attribute__((ext_vector_type(4)));
    rq_v4u_t u; u.x = v.x; u.y = v.y; u.z = v.z; u.w = v.w;
    __builtin_nontemporal_store(u, (rq_v4u_t*)p);
}
static __device__ __forceinline__ rq_u128 zero128() { rq_u128 z; z.x = z.y = z.z = z.w = 0; return z; }
static __device__ __forceinline__ bf16x8 as_bf16x8(rq_u128 v) {
    union { rq_u128 u; bf16x8 b; } c; c.u = v; return c.b;
}
