// rqt_kernels.hip -- see rqt_kernels.h for the reference call sites of each kernel.
#include "rqt_kernels.h"
#include "rq_common.h"
#include <mutex>
#include <stdlib.h>

// =================================================================================================
// residual add (+ split-K slab reduce + bias) fused with LayerNorm
// NS = number of split-K slabs (compile time: all slab loads are issued back to back, the kernel is
// latency-bound otherwise); each thread owns float4 columns tid, tid+256, ... of its row.
// 32 random bits -> uniform strictly inside (0,1): (2k+1) * 2^-24 for k = r >> 9 -- every value is exactly representable
// (an odd 24-bit integer times 2^-24), the smallest is 2^-24 and the largest 1 - 2^-24.  ((r >> 8) + 0.5) * 2^-24 is NOT safe:
// 16777215.5 rounds to 2^24, u = 1, -log(u) = -0 and the exponential race divides by it.)
static __device__ __forceinline__ float rq_u01(uint32_t r) { return fmaf((float)(r >> 9), 1.0f / 8388608.0f, 1.0f / 16777216.0f); }

template <int NS>
__global__ __launch_bounds__(256) void resid_ln_kernel(ResidLnArgs p) {
    __shared__ float red[8];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int E = p.E, E4 = E >> 2;
    const long base = (long)row * E;
    const long slab_stride = (long)p.rows * E;
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + 256 * i;
        v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (q < E4) v[i] = *(const f32x4*)(p.x_in + base + q * 4);
    }
    f32x4 t[NS > 0 ? NS : 1][4];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i;
            t[sl][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (q < E4) t[sl][i] = *(const f32x4*)(p.slabs + sl * slab_stride + base + q * 4);
        }
    // everything else the kernel reads is requested now as well -- bias / addvec / gamma / beta fetched where they are used were
    // three more dependent round trips (~1.4 us each on this part) in a kernel that is nothing but latency at small batches
    // (8448 launches of ~5 us per 64-image batch)
    f32x4 bv[4], av[4], gv[4], bev[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + 256 * i, qc = q < E4 ? q : 0;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        bv[i] = p.bias ? *(const f32x4*)(p.bias + qc * 4) : z;
        av[i] = p.addvec ? *(const f32x4*)(p.addvec + qc * 4) : z;
        gv[i] = p.gamma ? *(const f32x4*)(p.gamma + qc * 4) : z;
        bev[i] = p.gamma ? *(const f32x4*)(p.beta + qc * 4) : z;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + 256 * i;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) v[i] += t[sl][i];
        if (q < E4) {
            if (p.bias) v[i] += bv[i];
            if (p.addvec) v[i] += av[i];
            if (p.x_out) *(f32x4*)(p.x_out + base + q * 4) = v[i];
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    if (!p.gamma) return;   // uniform
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    rq_syncthreads();
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)E;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + 256 * i;
        if (q < E4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float d = v[i][e] - mean; s2 = fmaf(d, d, s2); }
        }
    }
    s2 = wave_sum(s2);
    if (lane == 0) red[4 + wave] = s2;
    rq_syncthreads();
    const float var = ((red[4] + red[5]) + (red[6] + red[7])) / (float)E;
    const float rstd = 1.0f / sqrtf(var + p.eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + 256 * i;
        if (q < E4) {
            const f32x4 g = gv[i], b = bev[i];
            const uint32_t lo = pack_bf16x2((v[i][0] - mean) * rstd * g[0] + b[0], (v[i][1] - mean) * rstd * g[1] + b[1]);
            const uint32_t hi = pack_bf16x2((v[i][2] - mean) * rstd * g[2] + b[2], (v[i][3] - mean) * rstd * g[3] + b[3]);
            *(uint32_t*)(p.y + base + q * 4) = lo;
            *(uint32_t*)(p.y + base + q * 4 + 2) = hi;
        }
    }
}

// Wave-per-row variant for E = 256 * VPL / ... (E/4 a multiple of 64): each lane owns VPL float4 columns
// lane, lane+64, ... of its row, every load is unconditional and issued up front, and both LayerNorm
// reductions are wave shuffles -- no LDS, no barriers.  (The block-per-row kernel above gives 256 threads
// 1.5 float4 each at E=1536 and crosses two barriers: 23 us per call at 4096 rows against 11 us of traffic.)
template <int VPL, int NS>
__global__ __launch_bounds__(256) void resid_ln_wave_kernel(ResidLnArgs p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + rq_uniform((int)(threadIdx.x >> 6));
    if (row >= p.rows) return;
    const int E = p.E;
    const long base = (long)row * E;
    const long slab_stride = (long)p.rows * E;
    f32x4 v[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) v[i] = *(const f32x4*)(p.x_in + base + (lane + 64 * i) * 4);
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
        f32x4 t[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) t[i] = *(const f32x4*)(p.slabs + sl * slab_stride + base + (lane + 64 * i) * 4);
#pragma unroll
        for (int i = 0; i < VPL; ++i) v[i] += t[i];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (p.bias) v[i] += *(const f32x4*)(p.bias + c);
        if (p.addvec) v[i] += *(const f32x4*)(p.addvec + c);
        if (p.x_out) *(f32x4*)(p.x_out + base + c) = v[i];
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    if (!p.gamma) return;   // uniform
    const float mean = wave_sum(s) / (float)E;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; s2 = fmaf(d, d, s2); }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)E + p.eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (lane + 64 * i) * 4;
        const f32x4 g = *(const f32x4*)(p.gamma + c), b = *(const f32x4*)(p.beta + c);
        struct __attribute__((aligned(8))) u64 { uint32_t a, b; } w;
        w.a = pack_bf16x2((v[i][0] - mean) * rstd * g[0] + b[0], (v[i][1] - mean) * rstd * g[1] + b[1]);
        w.b = pack_bf16x2((v[i][2] - mean) * rstd * g[2] + b[2], (v[i][3] - mean) * rstd * g[3] + b[3]);
        *(u64*)(p.y + base + c) = w;
    }
}

template <int VPL>
static int launch_resid_ln_wave(const ResidLnArgs& a, hipStream_t s) {
    const dim3 g((unsigned)((a.rows + 3) / 4)), b(256);
    switch (a.slabs ? a.n_slabs : 0) {
        case 0: RQ_LAUNCH((resid_ln_wave_kernel<VPL, 0>), g, b, 0, s, a); break;
        case 1: RQ_LAUNCH((resid_ln_wave_kernel<VPL, 1>), g, b, 0, s, a); break;
        case 2: RQ_LAUNCH((resid_ln_wave_kernel<VPL, 2>), g, b, 0, s, a); break;
        case 3: RQ_LAUNCH((resid_ln_wave_kernel<VPL, 3>), g, b, 0, s, a); break;
        case 4: RQ_LAUNCH((resid_ln_wave_kernel<VPL, 4>), g, b, 0, s, a); break;
        default: return 1;                        // more slabs: block-per-row kernel
    }
    return rq_check_launch("resid_ln_wave_kernel");
}

int rq_launch_resid_ln(const ResidLnArgs& a, hipStream_t s) {
    static const long wave_min_rows = getenv("RQAMD_LN_WAVE_MIN_ROWS") ? atol(getenv("RQAMD_LN_WAVE_MIN_ROWS")) : 512;      // A/B switch
    if ((long)a.rows * g_rq_row_scale >= wave_min_rows && a.E % 256 == 0) {       // plenty of rows: a wavefront per row keeps every CU busy
        int rc = 1;
        switch (a.E / 256) {
            case 4: rc = launch_resid_ln_wave<4>(a, s); break;     // E = 1024 (355M)
            case 5: rc = launch_resid_ln_wave<5>(a, s); break;     // 1280 (654M)
            case 6: rc = launch_resid_ln_wave<6>(a, s); break;     // 1536 (480M .. 1.4B)
            case 10: rc = launch_resid_ln_wave<10>(a, s); break;   // 2560 (3.8B)
            default: break;
        }
        if (rc <= 0) return rc;
    }
    if (a.E > 4096 || a.E % 4) return rq_fail(RQAMD_ERR_UNSUPPORTED, "resid_ln: embed_dim %d > 4096 or not a multiple of 4", a.E);
    const dim3 g(a.rows), b(256);
    switch (a.slabs ? a.n_slabs : 0) {
        case 0: RQ_LAUNCH(resid_ln_kernel<0>, g, b, 0, s, a); break;
        case 1: RQ_LAUNCH(resid_ln_kernel<1>, g, b, 0, s, a); break;
        case 2: RQ_LAUNCH(resid_ln_kernel<2>, g, b, 0, s, a); break;
        case 3: RQ_LAUNCH(resid_ln_kernel<3>, g, b, 0, s, a); break;
        case 4: RQ_LAUNCH(resid_ln_kernel<4>, g, b, 0, s, a); break;
        case 5: RQ_LAUNCH(resid_ln_kernel<5>, g, b, 0, s, a); break;
        case 6: RQ_LAUNCH(resid_ln_kernel<6>, g, b, 0, s, a); break;
        case 7: RQ_LAUNCH(resid_ln_kernel<7>, g, b, 0, s, a); break;
        case 8: RQ_LAUNCH(resid_ln_kernel<8>, g, b, 0, s, a); break;
        default: return rq_fail(RQAMD_ERR_INVALID, "resid_ln: %d slabs > 8", a.n_slabs);
    }
    return rq_check_launch("resid_ln_kernel");
}

// =================================================================================================
// KV-cache decode attention (attentions.py:60-105 of the reference, cached branch): one wavefront per (row, head),
// head_dim 64, up to NB*64 keys
static __device__ __forceinline__ void unpack8(rq_u128 u, float* f) {
    rq_unpack2(u.x, f[0], f[1]);
    rq_unpack2(u.y, f[2], f[3]);
    rq_unpack2(u.z, f[4], f[5]);
    rq_unpack2(u.w, f[6], f[7]);
}

// ---- opt-in 8-bit key cache (RQAMD_KV=int8k; AttnDecodeArgs::ksc != null; body stack only).  A cached key is 64 bytes + one fp32
// scale: component = (byte - 128) * scale, scale = max |k| / 127 over the key's 64 components (per token and head), byte =
// rint(k / scale) + 128 in 1 .. 255.  Costed on the reference model in round 4 (profiles/r04_kv_cache_precision_costing.txt: K alone
// adds 0.0053 max / 0.00063 mean to the logits, what bf16 storage itself adds); V stays bf16.  This token's own key is used as it
// comes out of the qkv GEMM (bf16), like the bf16 path; it is quantised only on its way into the cache.
struct __attribute__((aligned(8))) rq_u64w { uint32_t x, y; };
// Cache policy of the KV cache traffic.  A decode-step launch reads every cached key / value of a layer exactly once (4 GB at 10752 images) and
// nothing reads them again before the next position: with the default policy these lines displace the operands that ARE re-read (weights, the
// residual stream).  RQ_ATTN_NT=1 (default): non-temporal loads -- measured 4.9 -> 5.5 TB/s on the decode attention and +2.2 % on the whole step at
// 10752 images (profiles/r06_attn_nt_ab.txt).  RQ_ATTN_NT_ST: the appends likewise (A/B switch).
#ifndef RQ_ATTN_NT
#define RQ_ATTN_NT 1
#endif
#ifndef RQ_ATTN_NT_ST
#define RQ_ATTN_NT_ST 0
#endif
static __device__ __forceinline__ rq_u128 ld128_kv(const void* p) {
#if RQ_ATTN_NT
    return ld128_nt(p);
#else
    return ld128(p);
#endif
}
static __device__ __forceinline__ void st128_kv(void* p, rq_u128 v) {
#if RQ_ATTN_NT_ST
    st128_nt(p, v);
#else
    st128(p, v);
#endif
}
static __device__ __forceinline__ rq_u64w ld64(const void* p) { return *(const rq_u64w*)p; }
static __device__ __forceinline__ void st64(void* p, rq_u64w v) { *(rq_u64w*)p = v; }
static __device__ __forceinline__ rq_u64w ld64_kv(const void* p) {        // 8-bit cache rows (RQAMD_KV=int8k / int8kv): the same policy
#if RQ_ATTN_NT
    typedef unsigned rq_v2u_t __attribute__((ext_vector_type(2)));
    const rq_v2u_t v = __builtin_nontemporal_load((const rq_v2u_t*)p);
    rq_u64w r; r.x = v.x; r.y = v.y;
    return r;
#else
    return ld64(p);
#endif
}
// the 8 lanes of a key group hold its 64 components, 8 bf16 each: bytes of this lane's chunk + the key's scale (uniform in the group)
static __device__ __forceinline__ rq_u64w quant_key_chunk(rq_u128 kbf, float& scale) {
    float kf[8];
    unpack8(kbf, kf);
    float am = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf(kf[e]));
    am = fmaxf(am, rq_dpp_xor1(am));
    am = fmaxf(am, rq_dpp_xor2(am));
    am = fmaxf(am, rq_dpp_half_mirror(am));
    // a key with a NaN / inf component cannot be represented: its scale becomes NaN (every score against it is NaN, as with the bf16
    // cache) and its bytes the zero point -- never an int cast of a non-finite value
    float amf = am;
#pragma unroll
    for (int e = 0; e < 8; ++e) amf = kf[e] == kf[e] ? amf : __int_as_float(0x7fc00000);      // (fmaxf drops NaNs)
    amf = amf + rq_dpp_xor1(amf) * 0.f;
    amf = amf + rq_dpp_xor2(amf) * 0.f;
    amf = amf + rq_dpp_half_mirror(amf) * 0.f;
    const bool fin = amf <= 3.0e38f;                                   // false for NaN and inf
    scale = fin ? (am > 0.f ? am * (1.0f / 127.0f) : 1.0f) : __int_as_float(0x7fc00000);
    const float inv = (fin && am > 0.f) ? 127.0f / am : 0.f;
    uint32_t u[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) u[e] = (uint32_t)((int)rintf(fin ? kf[e] * inv : 0.f) + 128) & 0xffu;
    rq_u64w w;
    w.x = u[0] | (u[1] << 8) | (u[2] << 16) | (u[3] << 24);
    w.y = u[4] | (u[5] << 8) | (u[6] << 16) | (u[7] << 24);
    return w;
}
// this lane's share of sum_e q[e] * byte[e] over its 8 components
static __device__ __forceinline__ float dot_q_bytes(const float* qf, rq_u64w w) {
    float d = qf[0] * rq_ubyte_f32<0>(w.x);
    d = fmaf(qf[1], rq_ubyte_f32<1>(w.x), d); d = fmaf(qf[2], rq_ubyte_f32<2>(w.x), d); d = fmaf(qf[3], rq_ubyte_f32<3>(w.x), d);
    d = fmaf(qf[4], rq_ubyte_f32<0>(w.y), d); d = fmaf(qf[5], rq_ubyte_f32<1>(w.y), d);
    d = fmaf(qf[6], rq_ubyte_f32<2>(w.y), d); d = fmaf(qf[7], rq_ubyte_f32<3>(w.y), d);
    return d;
}
static __device__ __forceinline__ float group8_sum(float v) {
    v += rq_dpp_xor1(v);
    v += rq_dpp_xor2(v);
    v += rq_dpp_half_mirror(v);
    return v;
}

// Lane (g = lane >> 3, cc = lane & 7) owns 16-byte chunk cc of key / value row jj*8 + g of every 8-row block jj,
// for K and V alike (both caches are [row][head][Tcap][64]): a block is one 1 KB contiguous wavefront load, the
// number of load instructions follows the context length (2 * ceil((t+1)/8) + 2, not a fixed 25), the partial
// dot products are reduced over the 8 lanes of a group, and the softmax weight of a key already sits in the
// lanes that hold its value row.  Every launch at B=4096 issues 98 304 wavefronts; a 16-byte-per-lane load
// occupies the CU's address path for 16 cycles whatever the lanes address, which made the previous
// one-key-per-lane mapping cost ~100 us per launch even at t = 0 (profiles/r01_attn_trace.txt).
// NJ = number of 8-key blocks held in registers; DYN = skip blocks >= nblk at run time (long contexts).
// P = (row, head) pairs per wavefront (heads h0 .. h0+P-1 of one row), processed stage by stage so that the loads of
// all P pairs are in flight together: at short contexts a wavefront's lifetime is one memory round trip, and with
// 98 304 pairs per launch the launch time was 12 rounds of 8192 resident wavefronts x that latency (45 us at t = 0).
// VQ (round 6, opt-in RQAMD_KV=int8kv; implies KQ): the cached VALUES are bytes + one scale per (token, head) as well -- out = sum_j p_j
// scale_j (byte_j - 128) is accumulated as sum_j w_j byte_j - 128 sum_j w_j with w_j = p_j scale_j; this token's own value is used as
// it comes out of the qkv GEMM (bf16) and quantised only on its way into the cache.
template <int NJ, bool DYN, int P, bool KQ = false, bool VQ = false>
static __device__ __forceinline__ void attn_run(const AttnDecodeArgs& p, int lane, int b, int h0, int t) {
    static_assert(!VQ || KQ, "8-bit values come with 8-bit keys");
    const int E = p.E, Tcap = p.Tcap;
    const int cc = lane & 7, g = lane >> 3;
    const int nblk = (t >> 3) + 1;
    const int tprev = t > 0 ? t - 1 : 0;
    const float NEG_INF = -__int_as_float(0x7f800000);
    const long kvs = 64;                                // elements between consecutive positions of a pair's cache
    const bf16_t* qrow[P];
    bf16_t* kc[P];
    bf16_t* vc[P];
    unsigned char* kc8[P];                              // KQ: the same cache as bytes, and its scales
    float* ksc[P];
    unsigned char* vc8[P];                              // VQ: likewise for the values
    float* vsc[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const long pair = (long)b * p.nh + h0 + i;
        qrow[i] = p.qkv + (long)b * 3 * E + (h0 + i) * 64;
        kc[i] = p.kc + pair * Tcap * 64;                // pair-major [rows][nh][Tcap][64]: a position-major layout
        vc[i] = p.vc + pair * Tcap * 64;                // measured +10 % at long contexts and no gain at short ones
        kc8[i] = (unsigned char*)p.kc + pair * Tcap * 64;
        ksc[i] = KQ ? p.ksc + pair * Tcap : nullptr;
        vc8[i] = (unsigned char*)p.vc + pair * Tcap * 64;
        vsc[i] = VQ ? p.vsc + pair * Tcap : nullptr;
    }

    // every global load is issued before the first use (q, this token's k|v, all K and V blocks); rows j >= t
    // read this token's k / v straight from qkv (the cache row is written by this launch), clamped, unmasked
    rq_u128 qv[P], kv_new[P], kr[P][KQ ? 1 : NJ], vr[P][VQ ? 1 : NJ];
    rq_u128 kn_all[P];                                  // KQ: this token's key chunk in EVERY group (the self score)
    rq_u128 vn_all[P];                                  // VQ: this token's value chunk in every group (its own row of the weighted sum)
    rq_u64w kr8[P][KQ ? NJ : 1], vr8[P][VQ ? NJ : 1];
    float ksv[P][KQ ? NJ : 1], vsv[P][VQ ? NJ : 1];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        qv[i] = ld128(qrow[i] + cc * 8);
        kv_new[i] = ld128(qrow[i] + (lane < 8 ? E : 2 * E) + cc * 8);
        if (KQ) kn_all[i] = ld128(qrow[i] + E + cc * 8);
        if (VQ) vn_all[i] = ld128(qrow[i] + 2 * E + cc * 8);
    }
#pragma unroll
    for (int i = 0; i < P; ++i) {
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            if (DYN && jj >= nblk) continue;
            const int j = jj * 8 + g;
            const long off = (long)(j < t ? j : tprev) * kvs + cc * 8;
            if constexpr (KQ) {
                // cached keys only (rows j >= t are masked or served by the self score): 8 bytes per lane + the key's scale
                kr8[i][jj] = ld64_kv(kc8[i] + (long)(j < t ? j : tprev) * 64 + cc * 8);
                ksv[i][jj] = ksc[i][j < t ? j : tprev];
            } else {
                kr[i][jj] = ld128_kv((j >= t) ? (qrow[i] + E + cc * 8) : (kc[i] + off));
            }
        }
    }
#pragma unroll
    for (int i = 0; i < P; ++i) {
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            if (DYN && jj >= nblk) continue;
            const int j = jj * 8 + g;
            const long off = (long)(j < t ? j : tprev) * kvs + cc * 8;
            if constexpr (VQ) {
                vr8[i][jj] = ld64_kv(vc8[i] + (long)(j < t ? j : tprev) * 64 + cc * 8);
                vsv[i][jj] = vsc[i][j < t ? j : tprev];
            } else {
                vr[i][jj] = ld128_kv((j >= t) ? (qrow[i] + 2 * E + cc * 8) : (vc[i] + off));
            }
        }
    }
#pragma unroll
    for (int i = 0; i < P; ++i) {
        if constexpr (KQ) {
            // append: the key as 64 bytes + scale (lanes 0..7 hold it in kv_new), the value row as it is
            float s_app;
            const rq_u64w kb = quant_key_chunk(kv_new[i], s_app);              // (lanes >= 8 quantise their value chunk: unused)
            if (lane < 8) st64(kc8[i] + (long)t * 64 + cc * 8, kb);
            if (lane == 0) ksc[i][t] = s_app;
            if constexpr (VQ) {                                                  // (lanes 8 .. 15 just quantised the value chunk they hold)
                if (lane >= 8 && lane < 16) st64(vc8[i] + (long)t * 64 + cc * 8, kb);
                if (lane == 8) vsc[i][t] = s_app;
            } else if (lane >= 8 && lane < 16) st128_kv(vc[i] + (long)t * kvs + cc * 8, kv_new[i]);
        } else {
            if (lane < 8) st128_kv(kc[i] + (long)t * kvs + cc * 8, kv_new[i]);
            else if (lane < 16) st128_kv(vc[i] + (long)t * kvs + cc * 8, kv_new[i]);
        }
    }

    float sc[P][NJ], mx[P], inv[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        float qf[8];
        unpack8(qv[i], qf);
        mx[i] = NEG_INF;
        float s_self = 0.f, qsum128 = 0.f;
        if constexpr (KQ) {
            float kf[8];
            unpack8(kn_all[i], kf);
            float d = 0.f, qs = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { d = fmaf(qf[e], kf[e], d); qs += qf[e]; }
            s_self = group8_sum(d) * 0.125f;                        // q . k of this token, bf16 key (1/sqrt(64), attentions.py:87)
            qsum128 = 128.0f * group8_sum(qs);                      // the byte offset's share of every cached score
        }
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            float s = NEG_INF;
            if (!DYN || jj < nblk) {
                if constexpr (KQ) {
                    const float du = group8_sum(dot_q_bytes(qf, kr8[i][jj]));
                    const int j = jj * 8 + g;
                    if (j < t) s = (du - qsum128) * ksv[i][jj] * 0.125f;
                    else if (j == t) s = s_self;
                } else {
                    float kf[8];
                    unpack8(kr[i][jj], kf);
                    float dot = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) dot = fmaf(qf[e], kf[e], dot);
                    dot += rq_dpp_xor1(dot);                            // the 8 lanes of a key group: DPP, no LDS round trip
                    dot += rq_dpp_xor2(dot);
                    dot += rq_dpp_half_mirror(dot);
                    if (jj * 8 + g <= t) s = dot * 0.125f;              // 1/sqrt(64), attentions.py:87
                }
            }
            sc[i][jj] = s;
            mx[i] = fmaxf(mx[i], s);
        }
    }
    // scores are uniform inside a key group; across the 8 groups: rotate-by-8 inside each row of 16 lanes, then the
    // four rows through scalar lane reads (no ds_bpermute chain: its latency, not bandwidth, set the launch time)
#pragma unroll
    for (int i = 0; i < P; ++i) {
        float m = fmaxf(mx[i], rq_dpp_ror8(mx[i]));
        mx[i] = fmaxf(fmaxf(rq_readlane(m, 0), rq_readlane(m, 16)), fmaxf(rq_readlane(m, 32), rq_readlane(m, 48)));
    }
#pragma unroll
    for (int i = 0; i < P; ++i) {
        float sum = 0.f;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            sc[i][jj] = (sc[i][jj] == NEG_INF) ? 0.f : rq_fast_exp2((sc[i][jj] - mx[i]) * 1.4426950408889634f);
            sum += sc[i][jj];
        }
        sum += rq_dpp_ror8(sum);                                    // over the key groups only: the 8 lanes of a
        sum = (rq_readlane(sum, 0) + rq_readlane(sum, 16)) + (rq_readlane(sum, 32) + rq_readlane(sum, 48));   // group hold the same weights
        inv[i] = 1.0f / sum;
    }
#pragma unroll
    for (int i = 0; i < P; ++i) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        float wsum = 0.f;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            if (DYN && jj >= nblk) continue;
            const float pj = sc[i][jj] * inv[i];                    // 0 for j > t
            if constexpr (VQ) {
                const int j = jj * 8 + g;
                const float w = j < t ? pj * vsv[i][jj] : 0.f;      // cached rows: weight x scale on the bytes
                const rq_u64w vb = vr8[i][jj];
                acc[0] = fmaf(w, rq_ubyte_f32<0>(vb.x), acc[0]); acc[1] = fmaf(w, rq_ubyte_f32<1>(vb.x), acc[1]);
                acc[2] = fmaf(w, rq_ubyte_f32<2>(vb.x), acc[2]); acc[3] = fmaf(w, rq_ubyte_f32<3>(vb.x), acc[3]);
                acc[4] = fmaf(w, rq_ubyte_f32<0>(vb.y), acc[4]); acc[5] = fmaf(w, rq_ubyte_f32<1>(vb.y), acc[5]);
                acc[6] = fmaf(w, rq_ubyte_f32<2>(vb.y), acc[6]); acc[7] = fmaf(w, rq_ubyte_f32<3>(vb.y), acc[7]);
                wsum += w;
                const float ps = j < t ? 0.f : pj;                  // this token's own row (j == t; 0 beyond it), bf16
                float vf[8];
                unpack8(vn_all[i], vf);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(ps, vf[e], acc[e]);
            } else {
                float vf[8];
                unpack8(vr[i][jj], vf);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, vf[e], acc[e]);
            }
        }
        if constexpr (VQ) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(-128.0f, wsum, acc[e]);      // the bytes' zero point
        }
        // Reduce-scatter over the 8 key groups so that every lane ends with ONE of the 64 outputs (6 cross-row
        // shuffles instead of 24: the ds_bpermute count, 2.4 M per launch, was the LDS pipe's whole budget):
        //   rows {0,1} <-> {2,3}: keep e in 0..3 / 4..7;  row <-> row^1: keep 2 of those;  halves of a row: keep 1.
        const bool up32 = (lane & 32) != 0, up16 = (lane & 16) != 0, up8 = (lane & 8) != 0;
        float a4[4], a2[2];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float keep = up32 ? acc[e + 4] : acc[e], send = up32 ? acc[e] : acc[e + 4];
            a4[e] = keep + rq_shfl_xor(send, 32);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float keep = up16 ? a4[e + 2] : a4[e], send = up16 ? a4[e] : a4[e + 2];
            a2[e] = keep + rq_shfl_xor(send, 16);
        }
        const float keep1 = up8 ? a2[1] : a2[0], send1 = up8 ? a2[0] : a2[1];
        const float o1 = keep1 + rq_dpp_ror8(send1);
        const int eo = (up32 ? 4 : 0) + (up16 ? 2 : 0) + (up8 ? 1 : 0);            // which of the 8 channels of chunk cc
        p.y[(long)b * E + (h0 + i) * 64 + cc * 8 + eo] = (bf16_t)(pack_bf16x2(o1, 0.f) & 0xffffu);
    }
}

// One kernel per register-block count: the launch picks the smallest NJ that covers the host-known bound on t
// (engine_rqt.hip keeps one captured graph per NJ), so short contexts and the depth transformer (t < 8) run
// with few VGPRs and 8 wavefronts per SIMD instead of inheriting the 64-key variant's register budget.
template <int NJ, bool DYN, int P, bool KQ = false, bool VQ = false>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnDecodeArgs p) {
    const int lane = threadIdx.x & 63;
    const int wave = rq_uniform((int)(threadIdx.x >> 6));
    const int h0 = (blockIdx.x * 4 + wave) * P;
    if (h0 >= p.nh) return;                                        // whole wave exits together (nh % P == 0)
    const int t = (p.step ? *p.step : 0) + p.step_off;
    if ((t >> 3) >= NJ) rq_trap();                                 // host bound violated: never drop keys silently
    attn_run<NJ, DYN, P, KQ, VQ>(p, lane, (int)blockIdx.y, h0, t);
}

// Contexts of at most 8 keys (t <= 7: every step of the depth transformer and the first 8 spatial positions -- 44 % of all
// attention launches): EIGHT (row, head) pairs per wavefront, one per 8-lane group, lane = (pair slot, 16-byte chunk).
// In the kernel above a wavefront serves one or two pairs and most of a 64-lane load instruction fetches nothing new;
// its launch time at t = 0 was the texture-address path's instruction rate (7+ wavefront loads / stores per pair), not
// bytes.  Here a wavefront instruction moves 1 KB of useful data for 8 pairs, the softmax over <= 8 keys stays inside
// the lanes of a group (three DPP steps per dot product), and nothing crosses groups.
template <int T, bool KQ = false, bool VQ = false>      // number of cached keys (t), 0..7; KQ / VQ: 8-bit key / value cache (see quant_key_chunk)
static __device__ __forceinline__ void attn_small_run(const AttnDecodeArgs& p, long pair, bool valid, int cc) {
    const int E = p.E, Tcap = p.Tcap;
    const int b = (int)(pair / p.nh), h = (int)(pair - (long)b * p.nh);
    const bf16_t* qrow = p.qkv + (long)b * 3 * E + h * 64 + cc * 8;
    bf16_t* kc = p.kc + pair * Tcap * 64 + cc * 8;
    bf16_t* vc = p.vc + pair * Tcap * 64 + cc * 8;
    unsigned char* kc8 = (unsigned char*)p.kc + pair * Tcap * 64 + cc * 8;
    float* ksc = KQ ? p.ksc + pair * Tcap : nullptr;
    unsigned char* vc8 = (unsigned char*)p.vc + pair * Tcap * 64 + cc * 8;
    float* vsc = VQ ? p.vsc + pair * Tcap : nullptr;
    const rq_u128 qv = ld128(qrow), kn = ld128(qrow + E), vn = ld128(qrow + 2 * E);
    rq_u128 kr[(!KQ && T > 0) ? T : 1], vr[(!VQ && T > 0) ? T : 1];
    rq_u64w kr8[(KQ && T > 0) ? T : 1], vr8[(VQ && T > 0) ? T : 1];
    float ksv[(KQ && T > 0) ? T : 1], vsv[(VQ && T > 0) ? T : 1];
#pragma unroll
    for (int j = 0; j < T; ++j) {
        if constexpr (KQ) { kr8[j] = ld64_kv(kc8 + j * 64); ksv[j] = ksc[j]; }
        else kr[j] = ld128_kv(kc + j * 64);
    }
#pragma unroll
    for (int j = 0; j < T; ++j) {
        if constexpr (VQ) { vr8[j] = ld64_kv(vc8 + j * 64); vsv[j] = vsc[j]; }
        else vr[j] = ld128_kv(vc + j * 64);
    }
    if constexpr (KQ) {                            // append: key as bytes + scale (every lane takes part in the group reduction)
        float s_app;
        const rq_u64w kb = quant_key_chunk(kn, s_app);
        float s_appv = 0.f;
        rq_u64w vb = kb;
        if constexpr (VQ) vb = quant_key_chunk(vn, s_appv);
        if (valid) {
            st64(kc8 + T * 64, kb);
            if (cc == 0) ksc[T] = s_app;
            if constexpr (VQ) {
                st64(vc8 + T * 64, vb);
                if (cc == 0) vsc[T] = s_appv;
            } else st128_kv(vc + T * 64, vn);
        }
    } else if (valid) {                            // append this token's k / v
        st128_kv(kc + T * 64, kn);
        st128_kv(vc + T * 64, vn);
    }
    float qf[8], sc[T + 1];
    unpack8(qv, qf);
    float mx = -__int_as_float(0x7f800000);
    float qsum128 = 0.f;
    if constexpr (KQ) {
        float qs = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) qs += qf[e];
        qsum128 = 128.0f * group8_sum(qs);
    }
#pragma unroll
    for (int j = 0; j <= T; ++j) {
        if (KQ && j < T) {
            sc[j] = (group8_sum(dot_q_bytes(qf, kr8[j < T ? j : 0])) - qsum128) * ksv[j < T ? j : 0] * 0.125f;
        } else {
            float kf[8];
            unpack8((!KQ && j < T) ? kr[(!KQ && j < T) ? j : 0] : kn, kf);
            float dot = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) dot = fmaf(qf[e], kf[e], dot);
            dot += rq_dpp_xor1(dot);
            dot += rq_dpp_xor2(dot);
            dot += rq_dpp_half_mirror(dot);
            sc[j] = dot * 0.125f;                  // 1/sqrt(64), attentions.py:87
        }
        mx = fmaxf(mx, sc[j]);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j <= T; ++j) {
        sc[j] = rq_fast_exp2((sc[j] - mx) * 1.4426950408889634f);
        sum += sc[j];
    }
    const float inv = 1.0f / sum;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    float wsum = 0.f;
#pragma unroll
    for (int j = 0; j <= T; ++j) {
        const float pj = sc[j] * inv;
        if (VQ && j < T) {
            const float w = pj * vsv[j < T ? j : 0];
            const rq_u64w vb = vr8[j < T ? j : 0];
            acc[0] = fmaf(w, rq_ubyte_f32<0>(vb.x), acc[0]); acc[1] = fmaf(w, rq_ubyte_f32<1>(vb.x), acc[1]);
            acc[2] = fmaf(w, rq_ubyte_f32<2>(vb.x), acc[2]); acc[3] = fmaf(w, rq_ubyte_f32<3>(vb.x), acc[3]);
            acc[4] = fmaf(w, rq_ubyte_f32<0>(vb.y), acc[4]); acc[5] = fmaf(w, rq_ubyte_f32<1>(vb.y), acc[5]);
            acc[6] = fmaf(w, rq_ubyte_f32<2>(vb.y), acc[6]); acc[7] = fmaf(w, rq_ubyte_f32<3>(vb.y), acc[7]);
            wsum += w;
        } else {
            float vf[8];
            unpack8((!VQ && j < T) ? vr[(!VQ && j < T) ? j : 0] : vn, vf);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, vf[e], acc[e]);
        }
    }
    if constexpr (VQ) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(-128.0f, wsum, acc[e]);
    }
    if (valid) {
        rq_u128 o;
        o.x = pack_bf16x2(acc[0], acc[1]); o.y = pack_bf16x2(acc[2], acc[3]);
        o.z = pack_bf16x2(acc[4], acc[5]); o.w = pack_bf16x2(acc[6], acc[7]);
        st128(p.y + (long)b * E + h * 64 + cc * 8, o);
    }
}

template <bool KQ, bool VQ = false>
__global__ __launch_bounds__(256) void attn_small_kernel(AttnDecodeArgs p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long total = (long)p.rows * p.nh;
    long pair = ((long)blockIdx.x * 4 + wave) * 8 + (lane >> 3);
    const bool valid = pair < total;
    if (!valid) pair = total - 1;                  // clamped loads, masked stores
    const int t = (p.step ? *p.step : 0) + p.step_off;
    switch (t) {
        case 0: attn_small_run<0, KQ, VQ>(p, pair, valid, lane & 7); break;
        case 1: attn_small_run<1, KQ, VQ>(p, pair, valid, lane & 7); break;
        case 2: attn_small_run<2, KQ, VQ>(p, pair, valid, lane & 7); break;
        case 3: attn_small_run<3, KQ, VQ>(p, pair, valid, lane & 7); break;
        case 4: attn_small_run<4, KQ, VQ>(p, pair, valid, lane & 7); break;
        case 5: attn_small_run<5, KQ, VQ>(p, pair, valid, lane & 7); break;
        case 6: attn_small_run<6, KQ, VQ>(p, pair, valid, lane & 7); break;
        case 7: attn_small_run<7, KQ, VQ>(p, pair, valid, lane & 7); break;
        default: rq_trap();                        // host bound violated
    }
}

template <int NJ, bool DYN>
static void launch_attn(const AttnDecodeArgs& a, int pairs_per_wave, hipStream_t s) {
    const dim3 blk(256);
    // two pairs per wavefront exist only where rq_launch_attn_decode can ask for them (<= 4 register blocks, contexts <= 64):
    // the long-context forms with two pairs were never launched but were compiled, with 104 spilled registers at 32 blocks
    if constexpr (!DYN && NJ <= 4) {
        if (pairs_per_wave == 2) {
            if (a.vsc) RQ_LAUNCH((attn_decode_kernel<NJ, DYN, 2, true, true>), dim3((unsigned)((a.nh / 2 + 3) / 4), (unsigned)a.rows), blk, 0, s, a);
            else if (a.ksc) RQ_LAUNCH((attn_decode_kernel<NJ, DYN, 2, true>), dim3((unsigned)((a.nh / 2 + 3) / 4), (unsigned)a.rows), blk, 0, s, a);
            else RQ_LAUNCH((attn_decode_kernel<NJ, DYN, 2>), dim3((unsigned)((a.nh / 2 + 3) / 4), (unsigned)a.rows), blk, 0, s, a);
            return;
        }
    }
    if (a.vsc) RQ_LAUNCH((attn_decode_kernel<NJ, DYN, 1, true, true>), dim3((unsigned)((a.nh + 3) / 4), (unsigned)a.rows), blk, 0, s, a);
    else if (a.ksc) RQ_LAUNCH((attn_decode_kernel<NJ, DYN, 1, true>), dim3((unsigned)((a.nh + 3) / 4), (unsigned)a.rows), blk, 0, s, a);
    else RQ_LAUNCH((attn_decode_kernel<NJ, DYN, 1>), dim3((unsigned)((a.nh + 3) / 4), (unsigned)a.rows), blk, 0, s, a);
}

// =================================================================================================
// Head sizes other than 64 (attentions.py:44-57 takes any embed_dim / n_head, and the two stacks carry their own n_head; every released
// config has 64, which the kernels above are written for).  One wavefront per (row, head): q in LDS as fp32, lane = key for the scores
// (fp32 dot products of bf16 operands, scale 1/sqrt(head_dim), attentions.py:87), a wavefront-wide softmax, then lane = output component
// for the weighted sum of the value rows (coalesced).  The same arithmetic as the 64-wide kernels in a plain form -- an HBM-bound pass
// over the pair's cached keys and values; no released model takes it.
static __device__ __forceinline__ void attn_generic_core(const bf16_t* q, const bf16_t* kpast, long kstride, const bf16_t* vpast, long vstride,
                                                         int n_past, const bf16_t* kself, const bf16_t* vself, bf16_t* y, int hd, int lane,
                                                         float* sQ, float* sP) {
    for (int d = lane; d < hd; d += 64) sQ[d] = bf16_to_f32(q[d]);
    rq_syncthreads();
    const float NEG_INF = -__int_as_float(0x7f800000);
    const float scale = 1.0f / sqrtf((float)hd);
    float mx = NEG_INF;
    for (int j = lane; j <= n_past; j += 64) {
        const bf16_t* kr = j < n_past ? kpast + (long)j * kstride : kself;
        float dot = 0.f;
        for (int d = 0; d < hd; ++d) dot = fmaf(sQ[d], bf16_to_f32(kr[d]), dot);
        dot *= scale;
        sP[j] = dot;
        mx = fmaxf(mx, dot);
    }
    mx = wave_max(mx);
    float l = 0.f;
    for (int j = lane; j <= n_past; j += 64) {
        const float w = rq_fast_exp2((sP[j] - mx) * 1.4426950408889634f);
        sP[j] = w;
        l += w;
    }
    l = wave_sum(l);
    rq_syncthreads();
    const float inv = 1.0f / l;
    for (int d = lane; d < hd; d += 64) {
        float acc = 0.f;
        for (int j = 0; j < n_past; ++j) acc = fmaf(sP[j], bf16_to_f32(vpast[(long)j * vstride + d]), acc);
        acc = fmaf(sP[n_past], bf16_to_f32(vself[d]), acc);
        y[d] = f32_to_bf16(acc * inv);
    }
}

// decode step: append this token's key / value at position t, attend over positions 0..t (AttnDecodeArgs as the kernels above; bf16 cache only)
__global__ __launch_bounds__(64) void attn_generic_decode_kernel(AttnDecodeArgs p) {
    __shared__ float sQ[256], sP[264];
    const int lane = threadIdx.x;
    const long pair = blockIdx.x;
    const int b = (int)(pair / p.nh), hh = (int)(pair - (long)b * p.nh), hd = p.E / p.nh;
    const int t = (p.step ? *p.step : 0) + p.step_off;
    if (t >= p.Tcap) rq_trap();                                    // host bound violated: never write past the cache
    const bf16_t* q = p.qkv + (long)b * 3 * p.E + hh * hd;
    const bf16_t *k = q + p.E, *v = q + 2 * p.E;
    bf16_t* kc = p.kc + pair * p.Tcap * hd;
    bf16_t* vc = p.vc + pair * p.Tcap * hd;
    for (int d = lane; d < hd; d += 64) { kc[(long)t * hd + d] = k[d]; vc[(long)t * hd + d] = v[d]; }
    attn_generic_core(q, kc, hd, vc, hd, t, k, v, p.y + (long)b * p.E + hh * hd, hd, lane, sQ, sP);
}

// conditioning prefix (AttnPrefillArgs as attn_prefill_kernel): one wavefront per (image, head, token i); keys / values 0..i straight from
// the qkv rows of the image, token i's own pair appended to the cache at position i
__global__ __launch_bounds__(64) void attn_generic_prefill_kernel(AttnPrefillArgs p) {
    __shared__ float sQ[256], sP[264];
    const int lane = threadIdx.x;
    const long blk = blockIdx.x;
    const int i = (int)(blk % p.P);
    const long pair = blk / p.P;
    const int img = (int)(pair / p.nh), hh = (int)(pair - (long)img * p.nh), hd = p.E / p.nh;
    const bf16_t* q0 = p.qkv + (long)img * p.P * 3 * p.E + hh * hd;       // token 0 of the image, this head
    const bf16_t* q = q0 + (long)i * 3 * p.E;
    const bf16_t *k = q + p.E, *v = q + 2 * p.E;
    bf16_t* kc = p.kc + (pair * p.Tcap + i) * hd;
    bf16_t* vc = p.vc + (pair * p.Tcap + i) * hd;
    for (int d = lane; d < hd; d += 64) { kc[d] = k[d]; vc[d] = v[d]; }
    attn_generic_core(q, q0 + p.E, 3L * p.E, q0 + 2 * p.E, 3L * p.E, i, k, v, p.y + ((long)img * p.P + i) * p.E + hh * hd, hd, lane, sQ, sP);
}

static int attn_generic_check(int E, int nh, int Tcap, const void* ksc) {
    if (nh < 1 || E % nh) return rq_fail(RQAMD_ERR_INVALID, "attention: embed_dim %d is not a multiple of n_head %d", E, nh);
    if (E / nh > 256) return rq_fail(RQAMD_ERR_UNSUPPORTED, "attention: head_dim %d > 256", E / nh);
    if (Tcap > 256) return rq_fail(RQAMD_ERR_UNSUPPORTED, "attention: context %d > 256", Tcap);
    if (ksc) return rq_fail(RQAMD_ERR_UNSUPPORTED, "attention: the 8-bit cache formats (RQAMD_KV) are written for head_dim 64 (E=%d, n_head=%d)", E, nh);
    return RQAMD_OK;
}

int rq_launch_attn_decode(const AttnDecodeArgs& a, hipStream_t s) {
    if (a.nh < 1 || a.E != a.nh * 64) {             // any other head size: the plain kernel
        RQ_TRY(attn_generic_check(a.E, a.nh, a.Tcap, a.ksc));
        const long pairs = (long)a.rows * a.nh;
        if (pairs > 0x7fffffffL) return rq_fail(RQAMD_ERR_UNSUPPORTED, "attention: %ld (row, head) pairs", pairs);
        RQ_LAUNCH(attn_generic_decode_kernel, dim3((unsigned)pairs), dim3(64), 0, s, a);
        return rq_check_launch("attn_generic_decode_kernel");
    }
    if (a.rows > 65535) return rq_fail(RQAMD_ERR_UNSUPPORTED, "attention: %d rows > 65535", a.rows);
    if (a.vsc && !a.ksc) return rq_fail(RQAMD_ERR_INVALID, "attention: 8-bit values come with 8-bit keys");
    const int nj_cap = (a.Tcap + 7) / 8;
    int nj = a.t_max >= 0 ? (a.t_max >> 3) + 1 : nj_cap;
    if (nj > nj_cap) nj = nj_cap;
    // two heads per wavefront while the register blocks are small (latency-bound regime) and the heads pair up
    const int ppw = (a.nh % 2 == 0 && nj <= 4 && a.Tcap <= 64 && (long)a.rows * g_rq_row_scale * a.nh >= 16384) ? 2 : 1;
    static const bool no_small = getenv("RQAMD_NO_ATTN_SMALL") != nullptr;      // A/B switch
    if (nj == 1 && !no_small) {                   // at most 8 keys: eight pairs per wavefront
        const long pairs = (long)a.rows * a.nh;
        if (a.vsc) RQ_LAUNCH((attn_small_kernel<true, true>), dim3((unsigned)((pairs + 31) / 32)), dim3(256), 0, s, a);
        else if (a.ksc) RQ_LAUNCH(attn_small_kernel<true>, dim3((unsigned)((pairs + 31) / 32)), dim3(256), 0, s, a);
        else RQ_LAUNCH(attn_small_kernel<false>, dim3((unsigned)((pairs + 31) / 32)), dim3(256), 0, s, a);
        return rq_check_launch("attn_small_kernel");
    }
    if (a.Tcap <= 64) {
        switch (nj) {
            case 1: launch_attn<1, false>(a, ppw, s); break;
            case 2: launch_attn<2, false>(a, ppw, s); break;
            case 3: launch_attn<3, false>(a, ppw, s); break;
            case 4: launch_attn<4, false>(a, ppw, s); break;
            case 5: launch_attn<5, false>(a, 1, s); break;
            case 6: launch_attn<6, false>(a, 1, s); break;
            case 7: launch_attn<7, false>(a, 1, s); break;
            default: launch_attn<8, false>(a, 1, s); break;
        }
    } else if (a.Tcap <= 128) launch_attn<16, true>(a, 1, s);
    else if (a.Tcap <= 256) launch_attn<32, true>(a, 1, s);
    else return rq_fail(RQAMD_ERR_UNSUPPORTED, "attention: context %d > 256", a.Tcap);
    return rq_check_launch("attn_decode_kernel");
}


// =================================================================================================
// Prefill attention over the conditioning prefix: P tokens per image enter the body stack at once (reference:
// transformers.py:235-239 -> MultiSelfAttention.forward on (B, P, E) with the causal mask, attentions.py:60-104).
// One wavefront per (image, head): K and V of the pair are staged once in LDS (and appended to the KV cache at positions
// 0..P-1 on the way), then lane i owns query i -- scores against keys 0..i with an online softmax in fp32; every lane
// reads the same K / V row at a time (LDS broadcast, conflict-free).  Same arithmetic as the decode kernels above: bf16
// q / k / v, fp32 dot products, scale 1/8, fp32 softmax weights on bf16 values, one bf16 rounding of the output.
// The prefix is processed once per sample() call (the decode kernels run 64 times), P <= 255, so this is a VALU kernel.
__global__ __launch_bounds__(64) void attn_prefill_kernel(AttnPrefillArgs p) {
    RQ_DYN_SMEM(smem);
    const int lane = threadIdx.x;
    const int pair = blockIdx.x, img = pair / p.nh, hh = pair - img * p.nh;
    const int P = p.P, E = p.E;
    bf16_t* sK = (bf16_t*)smem;
    bf16_t* sV = sK + (size_t)P * 64;
    const bf16_t* q0 = p.qkv + (long)img * P * 3 * E + hh * 64;
    bf16_t* kc = p.kc + (long)pair * p.Tcap * 64;
    bf16_t* vc = p.vc + (long)pair * p.Tcap * 64;
    unsigned char* kc8 = (unsigned char*)p.kc + (long)pair * p.Tcap * 64;
    float* ksc = p.ksc ? p.ksc + (long)pair * p.Tcap : nullptr;
    unsigned char* vc8 = (unsigned char*)p.vc + (long)pair * p.Tcap * 64;
    float* vsc = p.vsc ? p.vsc + (long)pair * p.Tcap : nullptr;
    for (int i0 = 0; i0 < P * 8; i0 += 64) {          // (whole wavefront per pass: the 8-bit append reduces over the 8 lanes of a key)
        const int idx = i0 + lane;
        const bool in = idx < P * 8;                  // whole 8-lane groups: P * 8 is a multiple of 8
        const int j = in ? idx >> 3 : P - 1, c = idx & 7;
        const rq_u128 kv = ld128(q0 + (long)j * 3 * E + E + c * 8), vv = ld128(q0 + (long)j * 3 * E + 2 * E + c * 8);
        if (ksc) {                                    // uniform: opt-in 8-bit key cache (the prefix attention itself runs on the bf16 keys)
            float s_app;
            const rq_u64w kb = quant_key_chunk(kv, s_app);
            if (in) {
                st64(kc8 + j * 64 + c * 8, kb);
                if (c == 0) ksc[j] = s_app;
            }
        } else if (in) {
            st128(kc + j * 64 + c * 8, kv);
        }
        if (vsc) {                                    // uniform: opt-in 8-bit value cache (likewise: the prefix attention runs on the bf16 values)
            float s_app;
            const rq_u64w vb = quant_key_chunk(vv, s_app);
            if (in) {
                st64(vc8 + j * 64 + c * 8, vb);
                if (c == 0) vsc[j] = s_app;
            }
        } else if (in) {
            st128(vc + j * 64 + c * 8, vv);
        }
        if (in) {
            st128(sK + j * 64 + c * 8, kv);
            st128(sV + j * 64 + c * 8, vv);
        }
    }
    rq_syncthreads();
    const float NEG_INF = -__int_as_float(0x7f800000);
    for (int i0 = 0; i0 < P; i0 += 64) {
        const int i = i0 + lane;
        const bool live = i < P;
        const int iq = live ? i : P - 1;
        float qf[64], acc[64];
#pragma unroll
        for (int c = 0; c < 8; ++c) unpack8(ld128(q0 + (long)iq * 3 * E + c * 8), qf + c * 8);
#pragma unroll
        for (int e = 0; e < 64; ++e) acc[e] = 0.f;
        float m = NEG_INF, l = 0.f;
        const int jend = i0 + 63 < P - 1 ? i0 + 63 : P - 1;       // wave-uniform trip count; lanes mask keys j > i
        for (int j = 0; j <= jend; ++j) {
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float kf[8];
                unpack8(ld128(sK + j * 64 + c * 8), kf);
#pragma unroll
                for (int e = 0; e < 8; ++e) dot = fmaf(qf[c * 8 + e], kf[e], dot);
            }
            const float sc = j <= iq ? dot * 0.125f : NEG_INF;    // 1/sqrt(64), attentions.py:87; causal mask :88-91
            const float mn = fmaxf(m, sc);
            const float alpha = (m == NEG_INF) ? 0.f : rq_fast_exp2((m - mn) * 1.4426950408889634f);
            const float w = (sc == NEG_INF) ? 0.f : rq_fast_exp2((sc - mn) * 1.4426950408889634f);
            l = l * alpha + w;
            m = mn;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float vf[8];
                unpack8(ld128(sV + j * 64 + c * 8), vf);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[c * 8 + e] = fmaf(w, vf[e], acc[c * 8 + e] * alpha);
            }
        }
        if (live) {
            const float inv = 1.0f / l;
            bf16_t* o = p.y + ((long)img * P + i) * E + hh * 64;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                rq_u128 u;
                u.x = pack_bf16x2(acc[c * 8 + 0] * inv, acc[c * 8 + 1] * inv); u.y = pack_bf16x2(acc[c * 8 + 2] * inv, acc[c * 8 + 3] * inv);
                u.z = pack_bf16x2(acc[c * 8 + 4] * inv, acc[c * 8 + 5] * inv); u.w = pack_bf16x2(acc[c * 8 + 6] * inv, acc[c * 8 + 7] * inv);
                st128(o + c * 8, u);
            }
        }
    }
}

int rq_launch_attn_prefill(const AttnPrefillArgs& a, hipStream_t s) {
    if (a.nh >= 1 && a.E != a.nh * 64) {            // any other head size: the plain kernel
        RQ_TRY(attn_generic_check(a.E, a.nh, a.Tcap, a.ksc));
        if (a.P < 1 || a.P > a.Tcap) return rq_fail(RQAMD_ERR_UNSUPPORTED, "prefill attention: %d tokens (cache %d)", a.P, a.Tcap);
        const long blocks = (long)a.n_img * a.nh * a.P;
        if (blocks > 0x7fffffffL) return rq_fail(RQAMD_ERR_UNSUPPORTED, "prefill attention: %ld (image, head, token) triples", blocks);
        RQ_LAUNCH(attn_generic_prefill_kernel, dim3((unsigned)blocks), dim3(64), 0, s, a);
        return rq_check_launch("attn_generic_prefill_kernel");
    }
    if (a.nh < 1 || a.P < 1 || a.P > a.Tcap || a.P > 255) return rq_fail(RQAMD_ERR_UNSUPPORTED, "prefill attention: %d tokens (cache %d, max 255)", a.P, a.Tcap);
    const size_t smem = (size_t)a.P * 64 * 2 * 2;
    RQ_LAUNCH(attn_prefill_kernel, dim3((unsigned)(a.n_img * a.nh)), dim3(64), smem, s, a);
    return rq_check_launch("attn_prefill_kernel");
}

// =================================================================================================
// embedding of the newest position: sum over depths [0, n_depth) of codebook rows -> bf16 GEMM operand
__global__ void embed_tokens_kernel(EmbedTokArgs p) {
    const int per_row = p.dim / 8;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)p.rows * per_row) return;
    const int b = (int)(gid / per_row), c = (int)(gid - (long)b * per_row);
    const int pos = (p.pos ? *p.pos : 0) + p.pos_off;
    const int64_t* codes = p.xs + ((long)b * p.HW + pos) * p.D;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int d = p.depth_lo; d < p.n_depth; ++d) {
        long code = codes[d];
        if (code < 0 || code >= p.K[d]) continue;                  // padding row (index K) embeds to zero
        const float* src = p.cb[d] + code * p.dim + c * 8;
        f32x4 a = *(const f32x4*)src, bq = *(const f32x4*)(src + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[e] += a[e]; acc[4 + e] += bq[e]; }
    }
    rq_u128 o;
    o.x = pack_bf16x2(acc[0], acc[1]); o.y = pack_bf16x2(acc[2], acc[3]);
    o.z = pack_bf16x2(acc[4], acc[5]); o.w = pack_bf16x2(acc[6], acc[7]);
    st128(p.out + (long)b * p.dim + c * 8, o);
}

int rq_launch_embed_tokens(const EmbedTokArgs& a, hipStream_t s) {
    if (a.dim % 8 != 0) return rq_fail(RQAMD_ERR_UNSUPPORTED, "embed: dim %d %% 8 != 0", a.dim);
    const long n = (long)a.rows * (a.dim / 8);
    RQ_LAUNCH(embed_tokens_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    return rq_check_launch("embed_tokens_kernel");
}

__global__ void tok_embed_kernel(TokEmbedArgs p) {
    const int per_row = p.E / 4;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)p.rows * per_row) return;
    const int b = (int)(gid / per_row), c = (int)(gid - (long)b * per_row);
    const int pv = p.pos ? *p.pos : 0;
    const int64_t* codes = p.xs + ((long)b * p.HW + pv + p.pos_off) * p.D;
    const int arow = (p.add_by_pos ? pv : 0) + p.add_row;
    f32x4 acc = *(const f32x4*)(p.add + (long)arow * p.E + c * 4);
    for (int d = p.d_lo; d < p.d_hi; ++d) {
        long code = codes[d];
        if (code < 0 || code >= p.V[d]) rq_trap();          // index error in the reference (nn.Embedding)
        acc = acc + *(const f32x4*)(p.table + ((long)p.offs[d] + code) * p.E + c * 4);
    }
    *(f32x4*)(p.out + (long)b * p.E + c * 4) = acc;
}
int rq_launch_tok_embed(const TokEmbedArgs& a, hipStream_t s) {
    const long n = (long)a.rows * (a.E / 4);
    RQ_LAUNCH(tok_embed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    return rq_check_launch("tok_embed_kernel");
}
__global__ void mask_logits_kernel(float* logits, int rows, int V, int v_lo) {
    const int w = V - v_lo;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)rows * w) return;
    const int r = (int)(gid / w), c = (int)(gid - (long)r * w);
    logits[(long)r * V + v_lo + c] = -__int_as_float(0x7f800000);
}
int rq_launch_mask_logits(float* logits, int rows, int V, int v_lo, hipStream_t s) {
    if (v_lo >= V) return RQAMD_OK;
    const long n = (long)rows * (V - v_lo);
    RQ_LAUNCH(mask_logits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, logits, rows, V, v_lo);
    return rq_check_launch("mask_logits_kernel");
}

__global__ void cond_embed_kernel(const int64_t* cond, int cond_stride, int cond_idx, const float* cond_emb, int vocab_cond,
                                  const float* pos_emb_cond, float* x, int rows, int E) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)rows * E) return;
    const int b = (int)(gid / E), e = (int)(gid - (long)b * E);
    long c = cond ? cond[(long)b * cond_stride + cond_idx] : 0;
    if (c < 0) c = 0;
    if (c >= vocab_cond) c = vocab_cond - 1;
    x[gid] = cond_emb[c * E + e] + pos_emb_cond[(long)cond_idx * E + e];
}

int rq_launch_cond_embed(const int64_t* cond, int cond_stride, int cond_idx, const float* cond_emb, int vocab_cond,
                         const float* pos_emb_cond, float* x, int rows, int E, hipStream_t s) {
    const long n = (long)rows * E;
    RQ_LAUNCH(cond_embed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, cond, cond_stride, cond_idx,
              cond_emb, vocab_cond, pos_emb_cond, x, rows, E);
    return rq_check_launch("cond_embed_kernel");
}

__global__ void cond_embed_multi_kernel(const int64_t* cond, int cond_stride, int n_tok, const float* cond_emb, int vocab_cond,
                                        const float* pos_emb_cond, float* x, int n_img, int E) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)n_img * n_tok * E) return;
    const long row = gid / E;
    const int e = (int)(gid - row * E);
    const int img = (int)(row / n_tok), i = (int)(row - (long)img * n_tok);
    long c = cond ? cond[(long)img * cond_stride + i] : 0;
    if (c < 0) c = 0;
    if (c >= vocab_cond) c = vocab_cond - 1;
    x[gid] = cond_emb[c * E + e] + pos_emb_cond[(long)i * E + e];
}

int rq_launch_cond_embed_multi(const int64_t* cond, int cond_stride, int n_tok, const float* cond_emb, int vocab_cond,
                               const float* pos_emb_cond, float* x, int n_img, int E, hipStream_t s) {
    const long n = (long)n_img * n_tok * E;
    RQ_LAUNCH(cond_embed_multi_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, cond, cond_stride, n_tok,
              cond_emb, vocab_cond, pos_emb_cond, x, n_img, E);
    return rq_check_launch("cond_embed_multi_kernel");
}

// =================================================================================================
// small utilities
__global__ void cvt_bf16_kernel(const float* src, bf16_t* dst, long n) {
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        f32x4 v = *(const f32x4*)(src + i);
        uint32_t lo = pack_bf16x2(v[0], v[1]), hi = pack_bf16x2(v[2], v[3]);
        *(uint32_t*)(dst + i) = lo;
        *(uint32_t*)(dst + i + 2) = hi;
    } else {
        for (; i < n; ++i) dst[i] = f32_to_bf16(src[i]);
    }
}
int rq_launch_cvt_bf16(const float* src, bf16_t* dst, long n, hipStream_t s) {
    if (n <= 0) return RQAMD_OK;
    if (((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return rq_fail(RQAMD_ERR_INVALID, "cvt_bf16: misaligned pointers");
    const long nt = (n + 3) / 4;
    RQ_LAUNCH(cvt_bf16_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, src, dst, n);
    return rq_check_launch("cvt_bf16_kernel");
}
__global__ void cvt_bf16_transpose_kernel(const float* src, bf16_t* dst, int R, int Cc) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)R * Cc) return;
    const int c = (int)(gid / R), r = (int)(gid - (long)c * R);           // consecutive threads write consecutive dst elements
    dst[gid] = f32_to_bf16(src[(long)r * Cc + c]);
}
int rq_launch_cvt_bf16_transpose(const float* src, bf16_t* dst, int R, int Cc, hipStream_t s) {
    const long n = (long)R * Cc;
    RQ_LAUNCH(cvt_bf16_transpose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, R, Cc);
    return rq_check_launch("cvt_bf16_transpose_kernel");
}
__global__ void set_int_kernel(int* p, int v) { if (threadIdx.x == 0) *p = v; }
__global__ void add_int_kernel(int* p, int v) { if (threadIdx.x == 0) *p += v; }
int rq_launch_set_int(int* p, int v, hipStream_t s) {
    RQ_LAUNCH(set_int_kernel, dim3(1), dim3(64), 0, s, p, v);
    return rq_check_launch("set_int_kernel");
}
int rq_launch_add_int(int* p, int v, hipStream_t s) {
    RQ_LAUNCH(add_int_kernel, dim3(1), dim3(64), 0, s, p, v);
    return rq_check_launch("add_int_kernel");
}

// =================================================================================================
// on-device sampler: temperature, top-k, NaN scrub, softmax, top-p, renormalise, one draw per row
#ifndef RQ_SMP_NT             // A/B switch: the logits rows of sample_topk_kernel (read once) with the non-temporal policy
#define RQ_SMP_NT 0
#endif
constexpr int SMP_T = 256;    // threads per row: 4 wavefronts keep block barriers cheap (the first version used 1024
                              // threads and spent ~5 us per search iteration in 16-wave barriers)
constexpr int SMP_VPT = 64;   // probabilities per thread held in registers during the top-p search (V <= 16384)

static __device__ __forceinline__ float blk_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    if (lane == 0) red[wave] = v;
    rq_syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < SMP_T / 64; ++w) t += red[w];
    rq_syncthreads();
    return t;
}
// one barrier per call: partials alternate between two 16-float halves of `red2` (a wave can only reach
// the next write of a half after every wave has passed the barrier that follows the previous read of it)
static __device__ __forceinline__ float blk_sum_pp(float v, float* red2, int parity) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* red = red2 + (parity & 1) * 16;
    v = wave_sum(v);
    if (lane == 0) red[wave] = v;
    rq_syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < SMP_T / 64; ++w) t += red[w];
    return t;
}
static __device__ __forceinline__ float blk_max(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_max(v);
    if (lane == 0) red[wave] = v;
    rq_syncthreads();
    float t = red[0];
#pragma unroll
    for (int w = 1; w < SMP_T / 64; ++w) t = fmaxf(t, red[w]);
    rq_syncthreads();
    return t;
}
static __device__ __forceinline__ float blk_min(float v, float* red) { return -blk_max(-v, red); }

// exclusive prefix sum of one int per thread over the block (thread order); also returns the total
static __device__ __forceinline__ int blk_excl_scan(int v, int* redi, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int o = rq_shfl_i(inc, lane >= off ? lane - off : lane);
        if (lane >= off) inc += o;
    }
    if (lane == 63) redi[wave] = inc;
    rq_syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SMP_T / 64; ++w) {
        if (w < wave) base += redi[w];
        tot += redi[w];
    }
    rq_syncthreads();
    *total = tot;
    return base + inc - v;
}

static __device__ __forceinline__ unsigned order_key(float f) {   // monotone float -> uint, NaN on top (torch.topk)
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

static __device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                                     unsigned k0, unsigned k1, unsigned* out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ __launch_bounds__(256) void sample_kernel(SampleArgs p) {
    RQ_DYN_SMEM(smem);
    float* sx = (float*)smem;                  // [V] logits -> probabilities
    float* red = sx + p.V;                     // [16]
    int* redi = (int*)(red + 16);              // [16]
    float* red2 = (float*)(redi + 16) + 256 + 4; // [32] ping-pong partials (after hist[256] and bcast[4])
    unsigned* hist = (unsigned*)(redi + 16);   // [256]
    unsigned* bcast = hist + 256;              // [4]
    const int tid = threadIdx.x, V = p.V, row = blockIdx.x;
    if (p.redo && !p.redo[row]) return;        // second pass after sample_topk_kernel: only the rows it handed back
    const float* lg = p.logits + (long)row * V;
    const float NEG_INF = -__int_as_float(0x7f800000);

    for (int i = tid; i < V; i += SMP_T) sx[i] = lg[i] / p.temperature;          // utils.py:96-97
    rq_syncthreads();

    // ---- top-k: radix-select the k-th largest key, drop everything strictly below it (utils.py:60-64)
    if (p.top_k > 0 && p.top_k < V) {
        unsigned prefix = 0;
        int remaining = p.top_k;
        for (int pass = 3; pass >= 0; --pass) {
            for (int i = tid; i < 256; i += SMP_T) hist[i] = 0;
            rq_syncthreads();
            const int shift = pass * 8;
            for (int i = tid; i < V; i += SMP_T) {
                const unsigned k = order_key(sx[i]);
                if (pass == 3 || (k >> (shift + 8)) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
            }
            rq_syncthreads();
            // suffix scan over bins 255..0 by the first 256 threads (all threads run the collectives)
            const int bin = 255 - tid;
            const int cnt = (tid < 256) ? (int)hist[bin] : 0;
            int total;
            const int excl = blk_excl_scan(cnt, redi, &total);
            if (tid < 256 && excl < remaining && excl + cnt >= remaining) { bcast[0] = (unsigned)bin; bcast[1] = (unsigned)(remaining - excl); }
            rq_syncthreads();
            prefix = (prefix << 8) | bcast[0];
            remaining = (int)bcast[1];
            rq_syncthreads();
        }
        // k-th largest is NaN (torch.topk ranks NaN first): `out < NaN` is false everywhere -> nothing dropped
        if (prefix != 0xffffffffu)
            for (int i = tid; i < V; i += SMP_T)
                if (order_key(sx[i]) < prefix) sx[i] = NEG_INF;
        rq_syncthreads();
    }

    // ---- NaN scrub (utils.py:103-105) + softmax (:108)
    float mx = NEG_INF;
    for (int i = tid; i < V; i += SMP_T) {
        float v = sx[i];
        if (v != v) { v = NEG_INF; sx[i] = v; }
        mx = fmaxf(mx, v);
    }
    mx = blk_max(mx, red);
    float z = 0.f;
    for (int i = tid; i < V; i += SMP_T) {
        const float e = expf(sx[i] - mx);
        sx[i] = e;
        z += e;
    }
    z = blk_sum(z, red);
    for (int i = tid; i < V; i += SMP_T) sx[i] = sx[i] / z;
    rq_syncthreads();

    // ---- top-p (utils.py:67-79): keep the sorted prefix up to and including the first token whose
    // inclusive cumulative mass reaches p.  tau = largest value v with mass{prob >= v} >= p, found by a
    // bitwise search on the (monotone) float bit pattern -- no sort, deterministic reductions.
    // p >= 1 keeps every token: the reference's `cum_probs >= 1.0` can only fire through fp32 cumsum rounding
    // (mass of a few ulp), which no reordering of the sum reproduces -- the filter is skipped.
    if (p.top_p >= 0.f && p.top_p < 1.0f) {
        unsigned cur = 0;
        const bool in_regs = V <= SMP_T * SMP_VPT;
        float pv[SMP_VPT];
#pragma unroll
        for (int k = 0; k < SMP_VPT; ++k) {
            const int i = tid + k * SMP_T;
            pv[k] = (in_regs && i < V) ? sx[i] : 0.f;
        }
        for (int bit = 30; bit >= 0; --bit) {
            const unsigned cand = cur | (1u << bit);
            const float cv = __uint_as_float(cand);
            if (cand > 0x3f800000u) continue;           // probabilities never exceed 1.0 (uniform skip)
            float g = 0.f;
            if (in_regs) {
                float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
#pragma unroll
                for (int k = 0; k < SMP_VPT; k += 4) {
                    g0 += pv[k] >= cv ? pv[k] : 0.f;
                    g1 += pv[k + 1] >= cv ? pv[k + 1] : 0.f;
                    g2 += pv[k + 2] >= cv ? pv[k + 2] : 0.f;
                    g3 += pv[k + 3] >= cv ? pv[k + 3] : 0.f;
                }
                g = (g0 + g1) + (g2 + g3);
            } else {
                for (int i = tid; i < V; i += SMP_T) { const float q = sx[i]; if (q >= cv) g += q; }
            }
            g = blk_sum_pp(g, red2, bit);
            if (g >= p.top_p) cur = cand;
        }
        const float tau = __uint_as_float(cur);
        // boundary value, strict mass above it, number of ties at it
        float vmin = 2.0f;
        for (int i = tid; i < V; i += SMP_T) { const float q = sx[i]; if (q >= tau) vmin = fminf(vmin, q); }
        vmin = blk_min(vmin, red);
        float gs = 0.f;
        int nt = 0;
        const int per = (V + SMP_T - 1) / SMP_T, i0 = tid * per, i1 = (i0 + per < V) ? i0 + per : V;
        for (int i = i0; i < i1; ++i) { const float q = sx[i]; if (q > vmin) gs += q; else if (q == vmin) ++nt; }
        gs = blk_sum(gs, red);
        int ntie;
        int rank = blk_excl_scan(nt, redi, &ntie);
        int need = ntie;
        if (vmin > 0.f) {
            float m = ceilf((p.top_p - gs) / vmin);
            if (m < 1.f) m = 1.f;
            if (m < (float)ntie) need = (int)m;
        }
        float kept = 0.f;
        for (int i = i0; i < i1; ++i) {
            float q = sx[i];
            if (q < vmin) q = 0.f;
            else if (q == vmin) { if (rank >= need) q = 0.f; ++rank; }   // ties: lowest indices survive
            sx[i] = q;
            kept += q;
        }
        kept = blk_sum(kept, red);
        for (int i = tid; i < V; i += SMP_T) sx[i] = sx[i] / kept;
        rq_syncthreads();
    }

    if (p.probs_out)
        for (int i = tid; i < V; i += SMP_T) p.probs_out[(long)row * V + i] = sx[i];
    if (!p.out) return;

    // ---- one multinomial draw: argmax_i prob_i / E_i, E_i ~ Exp(1) (exponential race)
    const int slot = p.pos ? (*p.pos) * p.D + p.d : 0;
    const uint64_t seed = p.rng ? p.rng[0] : p.seed;
    const uint64_t off = (p.rng ? p.rng[1] : p.offset) + (uint64_t)slot;
    float best = -1.f;
    int besti = 0x7fffffff;
    for (int i4 = tid; i4 * 4 < V; i4 += SMP_T) {
        unsigned r[4];
        philox4x32_10((unsigned)i4, (unsigned)row, (unsigned)off, (unsigned)(off >> 32), (unsigned)seed, (unsigned)(seed >> 32), r);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = i4 * 4 + e;
            if (i < V) {
                const float u = rq_u01(r[e]);
                const float sc = sx[i] / (-logf(u));
                if (sc > best) { best = sc; besti = i; }
            }
        }
    }
    // block argmax, lowest index on ties
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = rq_shfl_xor(best, m);
        const int oi = rq_shfl_xor_i(besti, m);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if (lane == 0) { red[wave] = best; redi[wave] = besti; }
    rq_syncthreads();
    if (tid == 0) {
        for (int w = 1; w < SMP_T / 64; ++w)
            if (red[w] > best || (red[w] == best && redi[w] < besti)) { best = red[w]; besti = redi[w]; }
        if (besti >= V) besti = 0;
        p.out[(long)row * p.out_stride + slot] = (int64_t)besti;
    }
}

// -------------------------------------------------------------------------------------------------
// Register-resident sampler for V <= 16384, V % 4 == 0 (every released vocabulary): same arithmetic as
// sample_kernel above, different data movement.
//  * the row lives in registers (64 values per thread, float4 loads) -- no 64 KB LDS copy, so 4 workgroups per CU
//    instead of 2 and no LDS pass per phase;
//  * top-k threshold = k-th largest order_key by a bitwise search: the top 16 bits block-wide (count{key >= cand}
//    with one wavefront ballot + popcount per register), then the handful of keys that share that 16-bit prefix go
//    to LDS and ONE wavefront settles the low 16 bits without block barriers (falls back to 16 more block-wide
//    iterations when more than SMP_UND keys share the prefix, e.g. constant logits);
//  * after top-k the <= SMP_CAP survivors are compacted (deterministic scan order) to 8 per thread, and softmax,
//    the top-p search (31 block reductions) and the Philox draw touch only those; rows with more survivors
//    (ties, or top-k off) run the same tail on all 64 registers;
//  * ties at the top-p boundary keep the lowest vocabulary indices (as the stable sort of the oracle): the cut
//    index is found by a 15-step bitwise search over the index, so no phase depends on storage order.
// (Measured at 4096 x 16384, k=1024 / p=0.95: sample_kernel 1117 us.)
constexpr int SMP_CAP = 2048;              // survivors handled by the compact tail
constexpr int SMP_CV = SMP_CAP / SMP_T;    // 8 per thread
constexpr int SMP_UND = 256;               // keys sharing the 16-bit prefix settled by one wavefront

static __device__ __forceinline__ float key_to_float(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

struct SmpShared {
    float red[16];
    int redi[16];
    float red2[32];
    int cnt[2][4];
    unsigned bc[4];
    unsigned und[SMP_UND];
    int und_n;
    float val[SMP_CAP];
    int idx[SMP_CAP];
};

// block-wide count of set predicates accumulated per thread in `c` (ping-pong slots: one barrier per call)
static __device__ __forceinline__ int blk_count_pp(int wave_cnt, SmpShared& sh, int parity) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh.cnt[parity & 1][wave] = wave_cnt;
    rq_syncthreads();
    return (sh.cnt[parity & 1][0] + sh.cnt[parity & 1][1]) + (sh.cnt[parity & 1][2] + sh.cnt[parity & 1][3]);
}

// softmax -> top-p -> renormalise -> (probs_out) -> draw, on NV scaled logits per thread (absent entries: idx < 0)
template <int NV, typename IdxF>
static __device__ __forceinline__ void sample_tail(const SampleArgs& p, float (&q)[NV], IdxF idx_of, int row, SmpShared& sh) {
    const int tid = threadIdx.x, V = p.V;
    const float NEG_INF = -__int_as_float(0x7f800000);
    // NaN scrub (utils.py:103-105) + softmax (:108)
    float mx = NEG_INF;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float v = idx_of(k) >= 0 ? q[k] : NEG_INF;
        if (v != v) v = NEG_INF;
        q[k] = v;
        mx = fmaxf(mx, v);
    }
    mx = blk_max(mx, sh.red);
    float z = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const float e = idx_of(k) >= 0 ? expf(q[k] - mx) : 0.f;
        q[k] = e;
        z += e;
    }
    z = blk_sum(z, sh.red);
#pragma unroll
    for (int k = 0; k < NV; ++k) q[k] = q[k] / z;

    if (p.top_p >= 0.f && p.top_p < 1.0f) {     // utils.py:67-79, see sample_kernel
        unsigned cur = 0;
        for (int bit = 30; bit >= 0; --bit) {
            const unsigned cand = cur | (1u << bit);
            const float cv = __uint_as_float(cand);
            if (cand > 0x3f800000u) continue;           // probabilities never exceed 1.0 (uniform skip)
            float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
#pragma unroll
            for (int k = 0; k < NV; k += 4) {
                g0 += q[k] >= cv ? q[k] : 0.f;
                g1 += q[k + 1] >= cv ? q[k + 1] : 0.f;
                g2 += q[k + 2] >= cv ? q[k + 2] : 0.f;
                g3 += q[k + 3] >= cv ? q[k + 3] : 0.f;
            }
            const float g = blk_sum_pp((g0 + g1) + (g2 + g3), sh.red2, bit);
            if (g >= p.top_p) cur = cand;
        }
        const float tau = __uint_as_float(cur);
        float vmin = 2.0f;
#pragma unroll
        for (int k = 0; k < NV; ++k) if (q[k] >= tau) vmin = fminf(vmin, q[k]);
        vmin = blk_min(vmin, sh.red);
        float gs = 0.f;
        int nt = 0;
#pragma unroll
        for (int k = 0; k < NV; ++k) { if (q[k] > vmin) gs += q[k]; else if (q[k] == vmin && idx_of(k) >= 0) ++nt; }
        gs = blk_sum(gs, sh.red);
        int ntie;
        (void)blk_excl_scan(nt, sh.redi, &ntie);
        int need = ntie;
        if (vmin > 0.f) {
            float m = ceilf((p.top_p - gs) / vmin);
            if (m < 1.f) m = 1.f;
            if (m < (float)ntie) need = (int)m;
        }
        int cut = 0x7fffffff;                       // ties with index <= cut survive
        if (need < ntie) {                          // uniform; rare: the need-th smallest index among the ties
            unsigned ci = 0;
            for (int bit = 14; bit >= 0; --bit) {
                const unsigned cand = ci | (1u << bit);
                int c = 0;
#pragma unroll
                for (int k = 0; k < NV; ++k) c += (q[k] == vmin && idx_of(k) >= 0 && (unsigned)idx_of(k) < cand) ? 1 : 0;
                c = wave_sum_i(c);
                if (blk_count_pp(c, sh, bit) < need) ci = cand;
            }
            cut = (int)ci;
        }
        float kept = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            float v = q[k];
            if (v < vmin || (v == vmin && idx_of(k) > cut)) v = 0.f;
            q[k] = v;
            kept += v;
        }
        kept = blk_sum(kept, sh.red);
#pragma unroll
        for (int k = 0; k < NV; ++k) q[k] = q[k] / kept;
    }

    if (p.probs_out) {
#pragma unroll
        for (int k = 0; k < NV; ++k)
            if (idx_of(k) >= 0) p.probs_out[(long)row * V + idx_of(k)] = q[k];
    }
    if (!p.out) return;

    // one multinomial draw: argmax_i prob_i / E_i, E_i ~ Exp(1); Philox counter = (index / 4, row), word index % 4
    const int slot = p.pos ? (*p.pos) * p.D + p.d : 0;
    const uint64_t seed = p.rng ? p.rng[0] : p.seed;
    const uint64_t off = (p.rng ? p.rng[1] : p.offset) + (uint64_t)slot;
    float best = -1.f;
    int besti = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = idx_of(k);
        if (i < 0) continue;
        unsigned r[4];
        philox4x32_10((unsigned)(i >> 2), (unsigned)row, (unsigned)off, (unsigned)(off >> 32), (unsigned)seed, (unsigned)(seed >> 32), r);
        const unsigned w = (i & 3) == 0 ? r[0] : (i & 3) == 1 ? r[1] : (i & 3) == 2 ? r[2] : r[3];
        const float u = rq_u01(w);
        const float sc = q[k] / (-logf(u));
        if (sc > best || (sc == best && i < besti)) { best = sc; besti = i; }
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = rq_shfl_xor(best, m);
        const int oi = rq_shfl_xor_i(besti, m);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if (lane == 0) { sh.red[wave] = best; sh.redi[wave] = besti; }
    rq_syncthreads();
    if (tid == 0) {
        for (int w = 1; w < SMP_T / 64; ++w)
            if (sh.red[w] > best || (sh.red[w] == best && sh.redi[w] < besti)) { best = sh.red[w]; besti = sh.redi[w]; }
        if (besti >= V) besti = 0;
        p.out[(long)row * p.out_stride + slot] = (int64_t)besti;
    }
}

__global__ __launch_bounds__(SMP_T) void sample_topk_kernel(SampleArgs p) {
    __shared__ SmpShared sh;
    const int tid = threadIdx.x, lane = tid & 63, V = p.V, row = blockIdx.x;
    const float* lg = p.logits + (long)row * V;
    const int V4 = V >> 2;
    // thread t owns float4 groups t, t+256, ...: value 4j+e is vocabulary index (t + 256 j) * 4 + e
    unsigned ks[SMP_VPT];
    const bool scale = p.temperature != 1.0f;
#pragma unroll
    for (int j = 0; j < SMP_VPT / 4; ++j) {
        const int i4 = tid + SMP_T * j;
#if RQ_SMP_NT
        const f32x4 v = __builtin_nontemporal_load((const f32x4*)(lg + (long)(i4 < V4 ? i4 : V4 - 1) * 4));
#else
        const f32x4 v = *(const f32x4*)(lg + (long)(i4 < V4 ? i4 : V4 - 1) * 4);
#endif
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = scale ? v[e] / p.temperature : v[e];            // utils.py:96-97
            ks[4 * j + e] = i4 < V4 ? order_key(x) : 0u;                     // key 0 = absent (never >= a candidate)
        }
    }

    // ---- top-k: k-th largest key (utils.py:60-64); the launcher guarantees 0 < top_k < V
    unsigned kth = 0;
    {
        unsigned cur = 0;
        for (int bit = 31; bit >= 16; --bit) {
            const unsigned cand = cur | (1u << bit);
            int c = 0;
#pragma unroll
            for (int k = 0; k < SMP_VPT; ++k) c += ks[k] >= cand ? 1 : 0;      // v_cmp + v_addc; a ballot per key
            if (blk_count_pp(wave_sum_i(c), sh, bit) >= p.top_k) cur = cand;     // spilled 255 SGPRs
        }
        // keys sharing the 16-bit prefix: collect (order is irrelevant for a selection), count the keys above it
        if (tid == 0) sh.und_n = 0;
        rq_syncthreads();
        const unsigned pre = cur >> 16;
        int above = 0;
#pragma unroll
        for (int k = 0; k < SMP_VPT; ++k) {
            above += (ks[k] >> 16) > pre ? 1 : 0;
            if ((ks[k] >> 16) == pre) {
                const int pos = atomicAdd(&sh.und_n, 1);
                if (pos < SMP_UND) sh.und[pos] = ks[k];
            }
        }
        above = blk_count_pp(wave_sum_i(above), sh, 0);
        const int und_n = sh.und_n;
        const int need = p.top_k - above;                // >= 1: the k-th largest lies among the prefix keys
        if (und_n <= SMP_UND) {
            if (tid < 64) {                              // one wavefront, no block barriers
                unsigned uk[SMP_UND / 64];
#pragma unroll
                for (int m = 0; m < SMP_UND / 64; ++m) uk[m] = (lane + 64 * m < und_n) ? sh.und[lane + 64 * m] : 0u;
                for (int bit = 15; bit >= 0; --bit) {
                    const unsigned cand = cur | (1u << bit);
                    int c = 0;
#pragma unroll
                    for (int m = 0; m < SMP_UND / 64; ++m) c += rq_popc64(rq_ballot(uk[m] >= cand));
                    if (c >= need) cur = cand;
                }
                if (lane == 0) sh.bc[0] = cur;
            }
            rq_syncthreads();
            cur = sh.bc[0];
        } else {
            for (int bit = 15; bit >= 0; --bit) {
                const unsigned cand = cur | (1u << bit);
                int c = 0;
#pragma unroll
                for (int k = 0; k < SMP_VPT; ++k) c += ks[k] >= cand ? 1 : 0;
                if (blk_count_pp(wave_sum_i(c), sh, bit) >= p.top_k) cur = cand;
            }
        }
        kth = cur;
    }
    // k-th largest is NaN (torch.topk ranks NaN first): `out < NaN` is false everywhere -> nothing dropped
    const bool cutk = kth != 0xffffffffu;
    int mine = 0;
#pragma unroll
    for (int k = 0; k < SMP_VPT; ++k) mine += ks[k] >= kth ? 1 : 0;
    int total = 0;
    const int base = blk_excl_scan(mine, sh.redi, &total);
    const bool compact = cutk && total <= SMP_CAP;
    if (tid == 0) p.redo[row] = compact ? 0 : 1;         // ties past SMP_CAP / NaN threshold: sample_kernel redoes the row
    if (!compact) return;                                // uniform

    // ---- survivors in scan order, 8 per thread
    int pos = base;
#pragma unroll
    for (int k = 0; k < SMP_VPT; ++k)
        if (ks[k] >= kth) { sh.val[pos] = key_to_float(ks[k]); sh.idx[pos] = (tid + SMP_T * (k >> 2)) * 4 + (k & 3); ++pos; }
    if (p.probs_out) {                      // dropped entries are zeros; survivors overwrite theirs below
#pragma unroll
        for (int j = 0; j < SMP_VPT / 4; ++j)
            if (tid + SMP_T * j < V4) *(f32x4*)(p.probs_out + (long)row * V + (tid + SMP_T * j) * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
        rq_threadfence_block();
    }
    rq_syncthreads();
    float q[SMP_CV];
    int qi[SMP_CV];
#pragma unroll
    for (int k = 0; k < SMP_CV; ++k) {
        const int s = tid + SMP_T * k;
        q[k] = s < total ? sh.val[s] : 0.f;
        qi[k] = s < total ? sh.idx[s] : -1;
    }
    rq_syncthreads();
    sample_tail<SMP_CV>(p, q, [&](int k) -> int { return qi[k]; }, row, sh);
}

// Unfiltered draw (top_k covers the vocabulary, top_p >= 1: the reference's defaults, transformers.py:309-323):
// softmax + multinomial collapse to one streaming pass, argmax_i (logit_i / T + Gumbel_i), with
// Gumbel_i = -log(-log u_i) from the same Philox counters as sample_kernel -- no max / sum reductions, no
// LDS copy of the row, so occupancy is set by registers only (the general kernel holds V floats in LDS:
// 2 workgroups per CU, 830 us per call at 4096 x 16384; this one is bound by reading the logits once).
__global__ __launch_bounds__(256) void sample_gumbel_kernel(SampleArgs p) {
    __shared__ float red[4];
    __shared__ int redi[4];
    const int tid = threadIdx.x, V = p.V, row = blockIdx.x;
    const float* lg = p.logits + (long)row * V;
    const float inv_t = 1.0f / p.temperature;
    const int slot = p.pos ? (*p.pos) * p.D + p.d : 0;
    const uint64_t seed = p.rng ? p.rng[0] : p.seed;
    const uint64_t off = (p.rng ? p.rng[1] : p.offset) + (uint64_t)slot;
    const float NEG_INF = -__int_as_float(0x7f800000);
    float best = NEG_INF;
    int besti = 0x7fffffff;
    const bool vec = (V & 3) == 0;
    for (int i4 = tid; i4 * 4 < V; i4 += 256) {
        float v[4];
        if (vec) {
            const f32x4 q = *(const f32x4*)(lg + i4 * 4);
            v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (i4 * 4 + e < V) ? lg[i4 * 4 + e] : NEG_INF;
        }
        unsigned r[4];
        philox4x32_10((unsigned)i4, (unsigned)row, (unsigned)off, (unsigned)(off >> 32), (unsigned)seed, (unsigned)(seed >> 32), r);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = i4 * 4 + e;
            const float u = rq_u01(r[e]);
            float x = v[e] * inv_t;
            if (x != x) x = NEG_INF;                                            // NaN scrub (utils.py:103-105)
            const float sc = x - __logf(-__logf(u));
            if (i < V && sc > best) { best = sc; besti = i; }
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = rq_shfl_xor(best, m);
        const int oi = rq_shfl_xor_i(besti, m);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if (lane == 0) { red[wave] = best; redi[wave] = besti; }
    rq_syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (red[w] > best || (red[w] == best && redi[w] < besti)) { best = red[w]; besti = redi[w]; }
        if (besti >= V) besti = 0;
        p.out[(long)row * p.out_stride + slot] = (int64_t)besti;
    }
}

int rq_launch_sample(const SampleArgs& a, hipStream_t s) {
    if (a.V < 1 || a.V > 36000) return rq_fail(RQAMD_ERR_UNSUPPORTED, "sampler: vocab %d not in 1..36000", a.V);
    if (!(a.temperature > 0.f)) return rq_fail(RQAMD_ERR_INVALID, "sampler: temperature must be > 0");
    if ((a.top_k <= 0 || a.top_k >= a.V) && (a.top_p < 0.f || a.top_p >= 1.0f) && !a.probs_out && a.out) {
        RQ_LAUNCH(sample_gumbel_kernel, dim3(a.rows), dim3(256), 0, s, a);
        return rq_check_launch("sample_gumbel_kernel");
    }
    const size_t smem = (size_t)a.V * 4 + 16 * 4 + 16 * 4 + 256 * 4 + 4 * 4 + 32 * 4;
    static RqDeviceOnce attr_once;      // kernel attributes are per device
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    SampleArgs b = a;
    static const bool env_lds_only = getenv("RQAMD_SAMPLER_LDS") != nullptr;      // A/B switch
    if (a.top_k > 0 && a.top_k < a.V && a.V <= SMP_T * SMP_VPT && a.V % 4 == 0 && a.redo && !env_lds_only) {
        // top-k on: register-resident kernel; rows it cannot finish (more than SMP_CAP keys tied into the top k, NaN
        // threshold) are flagged in a.redo and redone by the general kernel, whose other workgroups exit at once
        RQ_LAUNCH(sample_topk_kernel, dim3(a.rows), dim3(SMP_T), 0, s, a);
        RQ_TRY(rq_check_launch("sample_topk_kernel"));
    } else {
        b.redo = nullptr;
    }
    RQ_LAUNCH(sample_kernel, dim3(a.rows), dim3(SMP_T), smem, s, b);
    return rq_check_launch("sample_kernel");
}
// one unfiltered multinomial draw per row of `logits` (rows x vocab, contiguous) -> out[row * out_stride]; used by the
// stochastic soft codes of the quantiser (csrc/quantize.hip)
int rq_launch_sample_rows(const float* logits, int rows, int vocab, uint64_t seed, uint64_t offset, int64_t* out, long out_stride, hipStream_t s) {
    SampleArgs a{};
    a.logits = logits; a.rows = rows; a.V = vocab; a.temperature = 1.0f; a.top_k = 0; a.top_p = -1.0f;
    a.seed = seed; a.offset = offset; a.out = out; a.out_stride = out_stride; a.D = 1;
    return rq_launch_sample(a, s);
}

extern "C" int rqamd_sample_logits(const float* logits, int rows, int vocab, float temperature, int top_k, float top_p,
                                   uint64_t seed, uint64_t offset, int64_t* samples_out, float* probs_out, int* row_flags, void* stream) {
    if (!logits || rows < 0) return rq_fail(RQAMD_ERR_INVALID, "sample_logits: bad argument");
    if (rows == 0) return RQAMD_OK;
    SampleArgs a{};
    a.logits = logits; a.rows = rows; a.V = vocab; a.temperature = temperature; a.top_k = top_k; a.top_p = top_p;
    a.seed = seed; a.offset = offset; a.out = samples_out; a.out_stride = 1; a.probs_out = probs_out; a.D = 1;
    a.redo = row_flags;     // caller-owned (rows ints) or NULL: without it every row takes the general kernel
    return rq_launch_sample(a, (hipStream_t)stream);
}
