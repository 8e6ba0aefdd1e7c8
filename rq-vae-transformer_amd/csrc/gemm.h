// gemm.h -- bf16 MFMA GEMM with fused epilogues; also the implicit-GEMM conv (gfx950).
//
//   C[M,N] = A[M,K] . W[N,K]^T  (+bias, +GELU | +residual | fp32 | split-K partial slabs)
//
// W is always a torch nn.Linear / repacked Conv2d weight: [N][K] row-major bf16 (K contiguous), so
// both operands are K-contiguous and fragments come out of LDS as single ds_read_b128.
// A is either a dense [M][lda] bf16 matrix (transformer decode steps: M = batch rows) or an NHWC
// bf16 feature map gathered on the fly (conv 3x3 / 1x1, stride 1 / 2, optional folded nearest-2x
// upsample): row m = output pixel, K = taps x Cin, one 64-wide K tile never straddles a tap
// (Cin % 64 == 0).
//
// Replaces, on the reference side: nn.Linear in MultiSelfAttention / AttentionBlock.mlp / classifier /
// input_mlp / head_mlp (rqvae/models/rqtransformer/attentions.py:48-55,117-122, transformers.py:68-94)
// and nn.Conv2d in Encoder/Decoder/ResnetBlock/AttnBlock/Upsample/Downsample
// (rqvae/models/rqvae/layers.py:20-182, modules.py:23,67,123,165), F.interpolate(nearest, x2)
// (layers.py:32) and F.pad(0,1,0,1) (layers.py:52-53).
//
// Tiling: WGM x WGN wavefronts (2x2 = 256 threads, or 4x2 = 512 threads for the 256x128 tile), block tile
// BM x BN x 64, wave tile (BM/WGM) x (BN/WGN) built from
// v_mfma_f32_32x32x16_bf16; LDS double-buffered, register-staged (global_load_dwordx4 -> ds_write_b128)
// with the loads of the next TWO K-tiles in flight behind the current tile's MFMAs; XCD-aware tile order; 16-byte-chunk XOR swizzle
// (chunk ^ ((row>>1)&7)) makes both the b128 writes and the fragment reads bank-conflict free.
#pragma once
#include <type_traits>
#include "rq_hip.h"

#ifndef RQ_EPI_ST_NT
#define RQ_EPI_ST_NT 0
#endif
#ifndef RQ_TILED_W_NT        // weight DMAs of the LDS-DMA tiled kernels with the non-temporal policy (A/B switch)
#define RQ_TILED_W_NT 0
#endif
#ifndef RQ_GEMM_VARIANT
#define RQ_GEMM_VARIANT 0
#endif
#ifndef RQ_GEMM_PF           // fragment reads one K-tile ahead of the MFMAs in the LDS-DMA tiles (A/B switch; measured, off: see the loop)
#define RQ_GEMM_PF 0
#endif
#ifndef RQ_GEMM_PF_MINW      // ... from this many wavefronts per tile
#define RQ_GEMM_PF_MINW 8
#endif

enum GemmEpi {
    EPI_BF16 = 0,         // out bf16 = acc + bias
    EPI_BF16_GELU = 1,    // out bf16 = gelu(acc + bias)
    EPI_BF16_RESID = 2,   // out bf16 = acc + bias + resid(bf16)
    EPI_F32 = 3,          // out f32  = acc + bias
    EPI_F32_PARTIAL = 4,  // out f32 slab[blockIdx.z] = acc (split-K partial; consumer reduces)
};

struct GemmArgs {
    const bf16_t* A;
    const bf16_t* W;
    int M, N, K;
    int lda;
    // conv gather (conv != 0): A is NHWC [n_img][Hs][Ws][Cin], Hs = Hin >> ups
    int conv, Hin, Win, Cin, Hout, Wout, ksize, stride, pad, ups;
    int cin_shift;           // log2(Cin) when Cin is a power of two, else -1 (filled by rq_gemm_launch)
    // epilogue
    int epi, gelu_v2;
    const float* bias;       // [N] (or [steps][N] when bias_step != nullptr)
    const int* bias_step;    // device-side step index selecting the bias row
    int bias_stride;
    void* out;
    int ldo;
    const bf16_t* resid;
    int ldr;
    int splitk;
    int vsplit;              // > 1 (conv kernels, bf16 epilogues, splitk == 1): "virtual split-K" -- the K loop runs as vsplit chunks of
                             // K / 64 / vsplit tiles (an even count), each accumulated from zero and added to a running total in chunk
                             // order: bit for bit what a real split into vsplit fp32 slabs + splitk_reduce computes, without the slabs.
                             // It makes a conv's result independent of whether the engine divided its K loop over workgroups (few
                             // images) or not (many) -- see engine_vae.hip.
    int accum;               // EPI_F32_PARTIAL with splitk == 1 only: out[m][n] = (out[m][n] + acc) + bias[n] -- the fp32 residual stream
                             // updated in place by the GEMM that produces the branch (same order of additions as slab + resid_ln)
    int sched, sched_gm;     // tile schedule (filled by rq_gemm_launch): 0 linear, 1 n-ranges per XCD, 2 m-bands per XCD
    int dbg;                 // diagnostics only: bit0 = skip the epilogue (ablation in scripts/gemm_bench.py)
    int glds;                // 0 = register-staged operands; 2..3 = LDS-DMA ring with that many stages (dense only)
};

// GELU of the transformer MLP (attentions.py:17-22 of the reference): v1 = x * Phi(x), v2 = x * sigmoid(1.702 x).
// Phi(x) = (1 + erf(x / sqrt 2)) / 2 with erf(z) = z * P(z^2), P of degree 8 fitted on |z| <= 3 (|error| < 1.7e-5; beyond,
// z * P(z^2) grows monotonically past 1 and is clamped, |1 - erf| < 2.3e-5 there): |GELU error| < 5e-5 absolute, against
// 2^-9 relative for the bf16 rounding that follows.  No transcendental, and every step is an fma / mul that exists in packed
// form (v_pk_fma_f32, two elements per issue slot): the fc1 epilogue applies it to 66 M elements per launch with no MFMA
// running, where the previous form (Abramowitz-Stegun 7.1.26: v_rcp + v_exp + 12 VALU per element) cost as much as the
// stores (9.1 of 19 us per round of tiles, profiles/r02_gemm_p8_epilogue_ablation.txt).  T = float or f32x2: the same operation
// sequence per element, so scalar and packed call sites (and every kernel variant) round identically.
typedef float f32x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ float rq_clamp_unit(float v) { return rq_med3(v, -1.0f, 1.0f); }
static __device__ __forceinline__ f32x2 rq_clamp_unit(f32x2 v) { return (f32x2){rq_med3(v.x, -1.0f, 1.0f), rq_med3(v.y, -1.0f, 1.0f)}; }
template <typename T> static __device__ __forceinline__ T rq_gelu_erf(T x) {
    const T z = x * 0.70710678118654752440f;
    const T t = z * z;
    T p = (T)(4.074052874e-08f);
    p = __builtin_elementwise_fma(p, t, (T)(-1.944763019e-06f));
    p = __builtin_elementwise_fma(p, t, (T)(4.105959529e-05f));
    p = __builtin_elementwise_fma(p, t, (T)(-5.110292166e-04f));
    p = __builtin_elementwise_fma(p, t, (T)(4.235391654e-03f));
    p = __builtin_elementwise_fma(p, t, (T)(-2.510276678e-02f));
    p = __builtin_elementwise_fma(p, t, (T)(1.110792036e-01f));
    p = __builtin_elementwise_fma(p, t, (T)(-3.753147936e-01f));
    p = __builtin_elementwise_fma(p, t, (T)(1.128268411e+00f));
    const T e = rq_clamp_unit(z * p);
    const T hx = x * 0.5f;
    return __builtin_elementwise_fma(hx, e, hx);
}
static __device__ __forceinline__ float rq_gelu(float x, int v2) {
    if (v2) return x * rq_fast_rcp(1.0f + rq_fast_exp2(-1.702f * 1.4426950408889634f * x));
    return rq_gelu_erf<float>(x);
}
// four consecutive outputs of one lane (the epilogues' unit of work), as two packed pairs
static __device__ __forceinline__ void rq_gelu4(float (&v)[4], int v2) {
    if (v2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rq_gelu(v[e], 1);
        return;
    }
    const f32x2 a = rq_gelu_erf<f32x2>((f32x2){v[0], v[1]}), b = rq_gelu_erf<f32x2>((f32x2){v[2], v[3]});
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}

static __device__ __forceinline__ int swz_off(int row, int chunk) {   // element offset in a [rows][64] bf16 tile
    return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3);
}

// MODE 0: dense A; 1: conv gather; 2: conv gather through a folded nearest-2x upsample.
// TR 1: accumulate the transposed tile (4 consecutive output columns per lane -> 16-byte fp32 stores); used
// for the split-K partial slabs, whose rows are short (N = E).  TR 0: bf16 outputs go through an LDS transpose,
// wide fp32 outputs (logits) are stored as 2 x 128 contiguous bytes per wavefront store.
// Epilogue shared by the GEMM kernels: bias / GELU / residual / bf16 packing / fp32 and split-K slab stores from the
// accumulators of a WGM x WGN wavefront grid over a BM x BN tile.  `smem` is the workgroup's dynamic LDS segment of
// SMEM_BYTES bytes (the operand buffers, free once every wave has passed the barrier inside); TR as in the kernels.
#ifdef RQ_GL_TRACE
// Diagnostics build only (scripts/gl_trace.py): shader-clock stamps of workgroup RQ_GL_TRACE of the LDS-DMA tiled kernel (plain loop)
__device__ unsigned long long g_gl_trace[16 * 32];
#define RQ_GLT(slot) do { if (blockIdx.x == RQ_GL_TRACE && blockIdx.z == 0 && (threadIdx.x & 63) == 0) g_gl_trace[(threadIdx.x >> 6) * 32 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define RQ_GLT(slot) do { } while (0)
#endif
#ifdef RQ_GEMM_TRACE
// Diagnostics build only (scripts/gemm_trace.sh): shader-clock stamps of one workgroup's epilogue phases.
__device__ unsigned long long g_gemm_trace[8 * 16];
#define RQ_GT(slot) do { if (blockIdx.x == RQ_GEMM_TRACE && blockIdx.z == 0 && (threadIdx.x & 63) == 0) g_gemm_trace[(threadIdx.x >> 6) * 16 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define RQ_GT(slot) do { } while (0)
#endif
// bias[n .. n+3] without control flow (columns >= N give 0): the epilogues fetch all of a lane's bias vectors up front, back
// to back -- inside their store loops, behind per-column bounds checks, every fetch was a serialised L2 round trip (32 of them:
// 12 000 of the 17 000 cycles of the eight-phase kernel's bf16 epilogue, profiles/r02_gemm_p8_epilogue_timeline.txt)
static __device__ __forceinline__ f32x4 rq_bias4(const float* bias, int n, int N) {
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (!bias) return b;                                   // uniform
    if ((N & 3) == 0) {                                    // uniform: n is a multiple of 4, so n < N <=> n + 3 < N
        const f32x4 v = *(const f32x4*)(bias + (n < N ? n : N - 4));
        if (n < N) b = v;
        return b;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float v = bias[n + e < N ? n + e : N - 1];
        b[e] = n + e < N ? v : 0.f;
    }
    return b;
}

// FULL (256 x 256 kernel): the tile lies inside the problem and every vector-store condition holds (checked once per workgroup by
// the caller), so no row / column test stands between the stores -- behind one, every 16-byte store of the bf16 stream-out waited
// for its own LDS read (16 serialised round trips per thread, ~4500 cycles of an 80 000-cycle tile) and every in-place residual
// vector sat in an exec-mask diamond of its own.
// EK (256 x 256 kernel): the epilogue family as a compile-time constant -- a GemmEpi value, or 5 = EPI_F32_PARTIAL with the in-place
// residual update (GemmArgs::accum), 6 = EPI_BF16_GELU with the sigmoid form (GemmArgs::gelu_v2); -1 = read p.epi / p.accum at run time (the other kernels).  With the family a run-time value
// every vector of the in-place path carried a v_cndmask per element, a 64-bit slab-offset multiply and two scalar branches
// (~45 instructions per 16-byte store).
template <int BM, int BN, int TR, int WGM, int WGN, int SMEM_BYTES, bool FULL = false, int EK = -1>
static __device__ __forceinline__ void rq_gemm_epilogue(const GemmArgs& p, f32x16 (&acc)[BM / WGM / 32][BN / WGN / 32],
                                                        unsigned char* smem, int m0, int n0) {
    constexpr int NTH = 64 * WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int MI = WM / 32, NI = WN / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    // ------------------------------------------------------------------ epilogue
    if (p.dbg & 1) {          // ablation: keep the accumulators live, store one value per wave
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) keep += acc[i][j][r];
        if (keep == 12345.678f) ((float*)p.out)[0] = keep;
        return;
    }
    const float* bias = p.bias;
    if (bias && p.bias_step) bias += (long)(*p.bias_step) * p.bias_stride;
    const int epi = EK < 0 ? p.epi : EK == 5 ? (int)EPI_F32_PARTIAL : EK == 6 ? (int)EPI_BF16_GELU : EK;       // (6: GELU v2)
    const int gelu_v2 = EK == 1 ? 0 : EK == 6 ? 1 : p.gelu_v2;
    // a lane's bias vector for columns n .. n + 3: inside a FULL tile one 16-byte load (rq_bias4's column tests cost an exec-mask
    // diamond per vector)
    auto bias4 = [&](const float* b, int n) -> f32x4 {
        if (FULL) return b ? *(const f32x4*)(b + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
        return rq_bias4(b, n, p.N);
    };
    if (TR && epi <= EPI_BF16_RESID) {
        // bf16 outputs: each lane packs its 4 consecutive columns to 8 bytes and writes them into a padded
        // [BM][BN] bf16 tile in LDS (over the operand buffers); the tile is then streamed out as 16-byte,
        // row-contiguous global stores, the residual being read the same way.  (Ablation on the decoder convs:
        // the first, scattered 2-byte epilogue cost +33..80 %; an fp32 LDS transpose +24 %.)
        constexpr int LDR = BN * 2 + 16;           // padded row stride in bytes (16-byte aligned rows)
        static_assert(BM * LDR <= SMEM_BYTES, "bf16 epilogue tile must fit the operand buffers");
        char* sT = (char*)smem;
        f32x4 bvec[NI][4];                          // the lane's bias vectors (the same for every row block i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) bvec[j][q] = bias4(bias, n0 + wn * WN + j * 32 + 8 * q + 4 * (lane >> 5));
        // likewise the lane's residual values (8 bytes per 4 columns), fetched up front with clamped addresses (rows / columns
        // outside the problem are computed but never stored): inside the loop below they were one global round trip each
        const bool res_vec = epi == EPI_BF16_RESID && (p.ldr & 3) == 0 && (p.N & 3) == 0;
        uint32_t rres[MI][NI][4][2];
        if (res_vec) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        int m = m0 + wm * WM + i * 32 + (lane & 31), n = n0 + wn * WN + j * 32 + 8 * q + 4 * (lane >> 5);
                        if (!FULL) {
                            m = m < p.M ? m : p.M - 1;
                            n = n < p.N ? n : p.N - 4;
                        }
                        const uint32_t* rp = (const uint32_t*)(p.resid + (long)m * p.ldr + n);
                        rres[i][j][q][0] = rp[0];
                        rres[i][j][q][1] = rp[1];
                    }
        }
        RQ_GT(1);
        rq_syncthreads();                          // every wave is done reading the operand buffers
        RQ_GT(2);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int ml = wm * WM + i * 32 + (lane & 31);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nl = wn * WN + j * 32 + 8 * q + 4 * (lane >> 5);
                    const int n = n0 + nl;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e] + bvec[j][q][e];
                    if (epi == EPI_BF16_GELU) {
                        rq_gelu4(v, gelu_v2);
                    }
                    if (res_vec) {
                        // residual added in fp32 BEFORE the single bf16 rounding (as the reference's x + h)
                        const uint32_t r0 = rres[i][j][q][0], r1 = rres[i][j][q][1];
                        { float ra_, rb_; rq_unpack2(r0, ra_, rb_); v[0] += ra_; v[1] += rb_; rq_unpack2(r1, ra_, rb_); v[2] += ra_; v[3] += rb_; }
                    } else if (epi == EPI_BF16_RESID) {
                        const int m = m0 + ml;
                        if (m < p.M && n < p.N) {
                            const bf16_t* rp = p.resid + (long)m * p.ldr + n;
                            if ((p.ldr & 3) == 0 && n + 3 < p.N) {
                                const uint32_t r0 = ((const uint32_t*)rp)[0], r1 = ((const uint32_t*)rp)[1];
                                { float ra_, rb_; rq_unpack2(r0, ra_, rb_); v[0] += ra_; v[1] += rb_; rq_unpack2(r1, ra_, rb_); v[2] += ra_; v[3] += rb_; }
                            } else {
                                for (int e = 0; e < 4 && n + e < p.N; ++e) v[e] += bf16_to_f32(rp[e]);
                            }
                        }
                    }
                    struct __attribute__((aligned(8))) u64 { uint32_t a, b; } w;
                    w.a = pack_bf16x2(v[0], v[1]);
                    w.b = pack_bf16x2(v[2], v[3]);
                    *(u64*)(sT + ml * LDR + nl * 2) = w;
                }
            }
        }
        RQ_GT(3);
        rq_syncthreads();
        RQ_GT(4);
        constexpr int CPR = BN / 8;                // 16-byte chunks per row
        const bool v16 = (p.N & 7) == 0 && (p.ldo & 7) == 0;
        if constexpr (FULL) {
            // piece k of a thread = 16-byte chunk tid % CPR of tile row tid / CPR + (NTH / CPR) k: all reads, then all stores
            static_assert(NTH % CPR == 0 && (BM * CPR) % NTH == 0, "whole rows per pass");
            constexpr int IT = BM * CPR / NTH, RSTEP = NTH / CPR;
            const char* src = sT + (tid / CPR) * LDR + (tid % CPR) * 16;
            bf16_t* o = (bf16_t*)p.out + (long)(m0 + tid / CPR) * p.ldo + n0 + (tid % CPR) * 8;
            rq_u128 u[IT];
#pragma unroll
            for (int k = 0; k < IT; ++k) u[k] = ld128(src + k * (RSTEP * LDR));
#pragma unroll
#if RQ_EPI_ST_NT            // A/B switch: whole-tile bf16 outputs (qkv, fc1 at large batch: read once by the next kernel) stored with the non-temporal policy
            for (int k = 0; k < IT; ++k) st128_nt(o + (long)k * RSTEP * p.ldo, u[k]);
#else
            for (int k = 0; k < IT; ++k) st128(o + (long)k * RSTEP * p.ldo, u[k]);
#endif
            RQ_GT(5);
            return;
        }
#pragma unroll 4
        for (int c = tid; c < BM * CPR; c += NTH) {
            const int ml = c / CPR, nl = (c - ml * CPR) * 8;
            const int m = m0 + ml, n = n0 + nl;
            if (m >= p.M || n >= p.N) continue;
            rq_u128 u = ld128(sT + ml * LDR + nl * 2);
            bf16_t* o = (bf16_t*)p.out + (long)m * p.ldo + n;
            if (v16) {
                st128(o, u);
            } else {
                const bf16_t* t = (const bf16_t*)(sT + ml * LDR + nl * 2);
                for (int e = 0; e < 8 && n + e < p.N; ++e) o[e] = t[e];
            }
        }
        RQ_GT(5);
        return;
    }
    if (TR) {
    // TR: the MFMAs above computed the TRANSPOSED tile (weight fragment as the "A" operand), so in the C/D
    // register map (row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31) the row index is the output
    // column n and the col index is the output row m: every lane owns, for its row m, four groups of
    // FOUR CONSECUTIVE n -- one 8-byte (bf16) or 16-byte (fp32) vector store each, and the residual is
    // read the same way.  (The first version stored one scattered 2-byte element per register: +33..80 %
    // time on the decoder convs; an LDS-staged transpose was 19 %.)
    const bool vec_ok = FULL || ((p.ldo & 3) == 0 && (epi != EPI_BF16_RESID || (p.ldr & 3) == 0));
    const bool accum = EK < 0 ? (epi == EPI_F32_PARTIAL && p.accum) : EK == 5;      // (launcher: splitk == 1, N % 4 == 0, ldo % 4 == 0)
    f32x4 bvec[NI][4];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            bvec[j][q] = bias4((epi != EPI_F32_PARTIAL || accum) ? bias : nullptr, n0 + wn * WN + j * 32 + 8 * q + 4 * (lane >> 5));
    f32x4 xr2[2][NI][4];
    // fp32 outputs: the lane's first vector of the tile (row m0 + wm WM + lane % 32, column n0 + wn WN + 4 (lane / 32), in the slab of
    // this K split); vector (i, j, q) lies 32 i rows and 32 j + 8 q columns on
    float* const o_lane = (float*)p.out + (epi == EPI_F32_PARTIAL ? (long)blockIdx.z * p.M * p.ldo : 0) +
                          (long)(m0 + wm * WM + (lane & 31)) * p.ldo + n0 + wn * WN + 4 * (lane >> 5);
    const long o_istep = 32l * p.ldo;
    auto load_x = [&](int i, f32x4 (&dst)[NI][4]) {
        const int m = m0 + wm * WM + i * 32 + (lane & 31);
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * WN + j * 32 + 8 * q + 4 * (lane >> 5);
                dst[j][q] = (FULL || (m < p.M && n < p.N)) ? *(const f32x4*)(o_lane + i * o_istep + j * 32 + 8 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
    };
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * WM + i * 32 + (lane & 31);
        // in-place residual update: the lane's vectors of a 32-row block are requested together, one block ahead of their use
        if (accum && i == 0) load_x(0, xr2[0]);
        if (accum && i + 1 < MI) load_x(i + 1, xr2[(i + 1) & 1]);
        f32x4 (&xr)[NI][4] = xr2[i & 1];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * WN + j * 32 + 8 * q + 4 * (lane >> 5);
                if (!FULL && (m >= p.M || n >= p.N)) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = accum ? (xr[j][q][e] + acc[i][j][4 * q + e]) + bvec[j][q][e] : acc[i][j][4 * q + e] + bvec[j][q][e];
                const bool full4 = FULL || (vec_ok && n + 3 < p.N);
                if (epi == EPI_BF16_GELU) {
                    rq_gelu4(v, gelu_v2);
                }
                if (epi <= EPI_BF16_RESID) {
                    bf16_t* o = (bf16_t*)p.out + (long)m * p.ldo + n;
                    if (full4) {
                        if (epi == EPI_BF16_RESID) {
                            const uint32_t* rp = (const uint32_t*)(p.resid + (long)m * p.ldr + n);
                            const uint32_t r0 = rp[0], r1 = rp[1];
                            { float ra_, rb_; rq_unpack2(r0, ra_, rb_); v[0] += ra_; v[1] += rb_; rq_unpack2(r1, ra_, rb_); v[2] += ra_; v[3] += rb_; }
                        }
                        struct __attribute__((aligned(8))) u64 { uint32_t a, b; } w;
                        w.a = pack_bf16x2(v[0], v[1]);
                        w.b = pack_bf16x2(v[2], v[3]);
                        *(u64*)o = w;
                    } else {
                        for (int e = 0; e < 4 && n + e < p.N; ++e) {
                            float x = v[e];
                            if (epi == EPI_BF16_RESID) x += bf16_to_f32(p.resid[(long)m * p.ldr + n + e]);
                            o[e] = f32_to_bf16(x);
                        }
                    }
                } else {
                    float* o = o_lane + i * o_istep + j * 32 + 8 * q;
                    if (full4) {
                        *(f32x4*)o = (f32x4){v[0], v[1], v[2], v[3]};
                    } else {
                        for (int e = 0; e < 4 && n + e < p.N; ++e) o[e] = v[e];
                    }
                }
            }
        }
    }
        return;
    }
    // TR == 0: untransposed accumulators (row = output row m, col = output column n)
    if (epi <= EPI_BF16_RESID) {
        // bf16 outputs: stage the fp32 tile through LDS (reusing the operand buffers) so that global
        // stores -- and the residual reads -- are row-contiguous 8-byte accesses instead of one scattered
        // 2-byte access per accumulator register (ablation: the scattered epilogue cost 33-80 % on top of
        // the main loop on the decoder convs).
        float* sC = (float*)smem;                  // [HB][BN] fp32 slab of the tile, reusing the operand buffers
        constexpr int SM_BYTES = SMEM_BYTES;
        constexpr int NH = (BM * BN * 4 + SM_BYTES - 1) / SM_BYTES;     // passes (1, or 2 for the 256-row tile)
        constexpr int HB = BM / NH;                                     // rows per pass (multiple of WM)
        static_assert(HB % WM == 0 && HB * BN * 4 <= SM_BYTES, "epilogue slab must fit the operand buffers");
        constexpr int QPR = BN / 4;                // float4 groups per row
        const bool n_vec_ok = (p.N & 3) == 0 && (p.ldo & 3) == 0 && (epi != EPI_BF16_RESID || (p.ldr & 3) == 0);
        float bvj[NI];                             // the lane's bias values, fetched up front (see rq_bias4)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wn * WN + j * 32 + (lane & 31);
            const float b = bias ? bias[n < p.N ? n : p.N - 1] : 0.f;
            bvj[j] = (bias && n < p.N) ? b : 0.f;
        }
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
            rq_syncthreads();                      // operand buffers / previous slab are free
            if (wm * WM >= hh * HB && wm * WM < (hh + 1) * HB) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const int nl = wn * WN + j * 32 + (lane & 31);
                        const float bv = bvj[j];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ml = wm * WM - hh * HB + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                            float v = acc[i][j][r] + bv;
                            if (epi == EPI_BF16_GELU) v = rq_gelu(v, gelu_v2);
                            sC[ml * BN + nl] = v;
                        }
                    }
            }
            rq_syncthreads();
#pragma unroll 4
            for (int q = tid; q < HB * QPR; q += NTH) {
                const int ml = q / QPR, nl = (q - ml * QPR) * 4;
                const int m = m0 + hh * HB + ml, n = n0 + nl;
                if (m >= p.M || n >= p.N) continue;
                f32x4 v = *(const f32x4*)(sC + ml * BN + nl);
                bf16_t* o = (bf16_t*)p.out + (long)m * p.ldo + n;
                if (n_vec_ok) {
                    if (epi == EPI_BF16_RESID) {
                        const uint32_t* rp = (const uint32_t*)(p.resid + (long)m * p.ldr + n);
                        const uint32_t r0 = rp[0], r1 = rp[1];
                        { float ra_, rb_; rq_unpack2(r0, ra_, rb_); v[0] += ra_; v[1] += rb_; rq_unpack2(r1, ra_, rb_); v[2] += ra_; v[3] += rb_; }
                    }
                    uint32_t* op = (uint32_t*)o;
                    const uint32_t w0 = pack_bf16x2(v[0], v[1]), w1 = pack_bf16x2(v[2], v[3]);
                    op[0] = w0;
                    op[1] = w1;
                } else {
                    for (int e = 0; e < 4 && n + e < p.N; ++e) {
                        float x = v[e];
                        if (epi == EPI_BF16_RESID) x += bf16_to_f32(p.resid[(long)m * p.ldr + n + e]);
                        o[e] = f32_to_bf16(x);
                    }
                }
            }
        }
        return;
    }
    // fp32 outputs: a wavefront store already covers 2 x 128 contiguous bytes
    float bvj[NI];                                 // the lane's bias values, fetched up front (see rq_bias4)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * WN + j * 32 + (lane & 31);
        const bool on = bias && epi != EPI_F32_PARTIAL;
        const float b = on ? bias[n < p.N ? n : p.N - 1] : 0.f;
        bvj[j] = (on && n < p.N) ? b : 0.f;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wn * WN + j * 32 + (lane & 31);
            const float bv = bvj[j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (!FULL && (m >= p.M || n >= p.N)) continue;
                const float v = acc[i][j][r] + bv;
                const long o = (long)m * p.ldo + n;
                if (epi == EPI_F32) ((float*)p.out)[o] = v;
                else ((float*)p.out)[(long)blockIdx.z * p.M * p.ldo + o] = v;
            }
        }
}

// GL 0: operands staged global -> registers -> LDS (all modes).  GL = 2..4 (dense operands only): a ring of GL LDS
// stages filled by LDS-DMA (global_load_lds_dwordx4, 1 KB = 8 tile rows per wavefront instruction): no staging
// registers, no ds_write pass; the XOR swizzle moves to the per-lane SOURCE address (the DMA writes lane-linear),
// the loads are counted by hand (s_waitcnt vmcnt(N) before the barrier that publishes a stage).
template <int BM, int BN, int MODE, int TR, int WGM = 2, int WGN = 2, int GL = 0, int VS = 0>   // WGM x WGN wavefronts per workgroup
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_bf16_kernel(GemmArgs p) {
    static_assert(!VS || (GL == 0 && MODE != 0), "virtual split-K: register-staged conv kernels only");
    constexpr int NTH = 64 * WGM * WGN;      // threads per workgroup
    constexpr int RP = NTH / 8;              // tile rows staged per pass (8 threads x 16 B per 64-wide row)
    constexpr bool CONV = MODE != 0;
    constexpr int BK = 64;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int A_IT = BM / RP, B_IT = BN / RP;
    RQ_DYN_SMEM(smem);
    // GL 0: sA[2][BM*64] then sB[2][BN*64].  GL > 0: GL stages of { A tile, B tile }.
    constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
    constexpr int A_STRIDE = GL ? STAGE_BYTES : BM * BK * 2;      // bytes between buffers of the A / B tile
    constexpr int B_STRIDE = GL ? STAGE_BYTES : BN * BK * 2;
    bf16_t* sA = (bf16_t*)smem;
    bf16_t* sB = sA + (GL ? 1 : 2) * BM * BK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    // tile id -> (m-tile, n-tile).  Workgroups are dispatched round-robin over the 8 XCDs, each with a private
    // 4 MiB L2 (block b runs on XCD b % 8 -- a speed assumption only, never a correctness one).  PMC showed
    // fabric-side fetch traffic at 3-5x the algorithmic bytes with a naive order, so the schedule gives every
    // XCD a contiguous slice of the tile space and walks it so that what is re-read stays in that XCD's L2:
    //  * sched 1 (NT >= 8, decode GEMMs): XCD x owns a contiguous range of n-tiles; inside it the m-tiles are
    //    visited in groups of `sched_gm` whose A panel fits L2, all n-tiles of the range per group.
    //  * sched 2 (NT < 8, convs): XCD x owns a contiguous band of m-tiles (adjacent output rows share their
    //    3x3 halo rows), n fastest so an A tile is reused by all its n-tiles at once.
    // The grid is padded to 8 x (largest slice); surplus workgroups exit here, before any barrier.
    const int MT = (p.M + BM - 1) / BM, NT = (p.N + BN - 1) / BN;
    int mt, nt, zs = blockIdx.z;
    {
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
        if (p.sched == 1) {
            const int nb = NT >> 3, rem = NT & 7;
            const int nn = nb + (xcd < rem ? 1 : 0);
            const int n_lo = xcd * nb + (xcd < rem ? xcd : rem);
            if (slot >= nn * MT) return;
            const int per_group = p.sched_gm * nn;
            const int mg = slot / per_group, r = slot - mg * per_group;
            int gm = MT - mg * p.sched_gm;
            gm = gm < p.sched_gm ? gm : p.sched_gm;
            nt = n_lo + r / gm;
            mt = mg * p.sched_gm + (r - (r / gm) * gm);
        } else if (p.sched == 2) {
            const int mm = (MT + 7) >> 3;
            mt = xcd * mm + slot / NT;
            nt = slot - (slot / NT) * NT;
            if (slot >= mm * NT || mt >= MT) return;
        } else if (p.sched == 4) {
            // split-K slab GEMMs with 2, 4 or 8 K slices (round 6): a K SLICE per group of 8 / splitk XCDs, the n-tiles of the slice divided
            // over the XCDs of the group, m fastest.  With n-ranges per XCD (sched 1) every XCD walks all K slices and pulls ALL of A through
            // its L2 (fc2 at 500 rows: 8 x 6.1 MB on the fabric for 25 MB of operands); here an XCD reads its slice of A and of W once.
            // The grid is one-dimensional: the slab index comes from the XCD, not from blockIdx.z.
            const int xper = 8 / p.splitk;
            zs = xcd / xper;
            const int sub = xcd - zs * xper;
            const int nb = NT / xper, rem = NT - nb * xper;
            const int nn = nb + (sub < rem ? 1 : 0);
            const int n_lo = sub * nb + (sub < rem ? sub : rem);
            if (slot >= nn * MT) return;
            nt = n_lo + slot / MT;
            mt = slot - (slot / MT) * MT;
            p.out = (float*)p.out + (long)zs * p.M * p.ldo;      // (blockIdx.z is 0 in the epilogue's slab offset)
        } else {
            mt = id / NT;
            nt = id - mt * NT;
        }
    }
    const int m0 = mt * BM, n0 = nt * BN;
    const int kt_total = p.K / BK;
    const int per = (kt_total + p.splitk - 1) / p.splitk;
    const int kt0 = zs * per;
    const int kt1 = (kt0 + per < kt_total) ? kt0 + per : kt_total;

    const int chunk = tid & 7, lrow = tid >> 3;

    // ---------------------------------------------------------------------------------------------
    // Addressing.  Everything the main loop touches is either loop-invariant or one 32-bit add:
    //  * LDS: the XOR swizzle of a fragment read depends only on (lane&31)>>1 and of a staging write only
    //    on (tid>>3)>>1, so each thread keeps a few byte offsets; tile rows / buffers are immediates.
    //  * global: 32-bit BYTE offsets from the (wave-uniform) tensor bases (launcher checks < 4 GiB).
    //  * Loads are UNCONDITIONAL: a predicated load becomes an exec-masked branch and hipcc then drains
    //    vmcnt(0) at every join, serialising the pipeline.  Rows >= M / >= N only feed outputs that are
    //    never stored.  Conv padding taps must contribute zeros: each row has a 9-bit validity mask
    //    (computed once); an invalid tap loads the row's centre pixel instead and is zeroed when the
    //    registers go to LDS (a branch taken only on border tiles).
    // (PMC on the first version: 13 VALU + 8 SALU per MFMA -- the issue port, not the matrix pipe, was busy.)
    const char* gA = (const char*)p.A;
    const char* gW = (const char*)p.W;
    unsigned a_off[A_IT];    // dense: byte offset of A[m][chunk*8]; conv: byte offset of tap (0,0) (may wrap below 0)
    unsigned a_safe[A_IT];   // conv: byte offset of an in-bounds pixel of the same image
    int a_par[A_IT];         // MODE 2: (oy&1) | (ox&1)<<1
    unsigned a_valid[A_IT];  // conv: bit t = tap t lies inside the image
    const int Hs = p.Hin >> p.ups, Ws = p.Win >> p.ups;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int m = m0 + lrow + RP * i;
        if (m > p.M - 1) m = p.M - 1;
        a_par[i] = 0;
        a_valid[i] = 0x1ffu;
        a_safe[i] = 0;
        if (CONV) {
            const int hw = p.Hout * p.Wout;
            const int img = m / hw, rem = m - img * hw;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            const unsigned img_base = ((unsigned)img * Hs * Ws * p.Cin + chunk * 8) * 2u;
            // validity of the 3 (or 1) rows and columns of the stencil, combined into 9 bits
            unsigned vy = 0, vx = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int iy = oy * p.stride + k - p.pad, ix = ox * p.stride + k - p.pad;
                if (k < p.ksize && iy >= 0 && iy < p.Hin) vy |= 1u << k;
                if (k < p.ksize && ix >= 0 && ix < p.Win) vx |= 1u << k;
            }
            unsigned valid = 0;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
                if ((vy >> ky) & 1u) valid |= vx << (ky * p.ksize);
            a_valid[i] = valid;
            if (MODE == 2) {
                // source pixel of tap (ky,kx): ((oy+ky-1)>>1, (ox+kx-1)>>1) = (oy>>1, ox>>1) + ((par+k-1)>>1)
                a_par[i] = (oy & 1) | ((ox & 1) << 1);
                a_off[i] = img_base + (unsigned)(((oy >> 1) * Ws + (ox >> 1)) * p.Cin) * 2u;
                a_safe[i] = a_off[i];
            } else {
                a_off[i] = img_base + (unsigned)(((oy * p.stride - p.pad) * Ws + (ox * p.stride - p.pad)) * p.Cin) * 2u;
                int cy = oy * p.stride, cx = ox * p.stride;
                cy = cy < p.Hin ? cy : p.Hin - 1;
                cx = cx < p.Win ? cx : p.Win - 1;
                a_safe[i] = img_base + (unsigned)((cy * Ws + cx) * p.Cin) * 2u;
            }
        } else {
            a_off[i] = ((unsigned)m * p.lda + chunk * 8) * 2u;
        }
    }
    unsigned w_off[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        int n = n0 + lrow + RP * i;
        if (n > p.N - 1) n = p.N - 1;
        w_off[i] = ((unsigned)n * p.K + chunk * 8) * 2u;
    }
    constexpr unsigned FULL = (1u << A_IT) - 1u;

    // returns the validity mask of the A rows (bit i = this K-tile's tap is inside the image for row i)
    auto load_tile = [&](int kt, rq_u128* ra, rq_u128* rb) -> unsigned {
        const unsigned kb = (unsigned)kt * (BK * 2);          // byte offset of the K-tile (dense operands)
        unsigned mask = FULL;
        if (CONV) {
            // wave-uniform tap decode (scalar unit)
            const int k0 = kt * BK;
            const int tap = p.cin_shift >= 0 ? (k0 >> p.cin_shift) : (k0 / p.Cin);
            const int ci0 = k0 - tap * p.Cin;
            const int ky = p.ksize == 3 ? (tap * 11) >> 5 : 0;      // tap / 3 for tap < 9
            const int kx = tap - ky * p.ksize;
            const unsigned cb = (unsigned)ci0 * 2u;
            mask = 0;
            if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < A_IT; ++i) {
                    const bool ok = (a_valid[i] >> tap) & 1u;
                    const int dsy = ((a_par[i] & 1) + ky - 1) >> 1, dsx = ((a_par[i] >> 1) + kx - 1) >> 1;
                    const unsigned off = a_off[i] + (unsigned)((dsy * Ws + dsx) * p.Cin) * 2u;
                    mask |= (ok ? 1u : 0u) << i;
                    ra[i] = ld128(gA + ((ok ? off : a_safe[i]) + cb));
                }
            } else {
                const unsigned toff = (unsigned)((ky * Ws + kx) * p.Cin) * 2u;       // same for every row
#pragma unroll
                for (int i = 0; i < A_IT; ++i) {
                    const bool ok = (a_valid[i] >> tap) & 1u;
                    mask |= (ok ? 1u : 0u) << i;
                    ra[i] = ld128(gA + ((ok ? a_off[i] + toff : a_safe[i]) + cb));
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) ra[i] = ld128(gA + (a_off[i] + kb));
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) rb[i] = ld128(gW + (w_off[i] + kb));
        return mask;
    };
    // staging writes: row lrow + 32 i, chunk swizzled by (lrow>>1)&7 (32 i does not change it)
    char* wr_a = (char*)sA + (lrow * 64 + ((chunk ^ ((lrow >> 1) & 7)) << 3)) * 2;
    char* wr_b = (char*)sB + (lrow * 64 + ((chunk ^ ((lrow >> 1) & 7)) << 3)) * 2;
    auto store_tile = [&](int buf, rq_u128* ra, const rq_u128* rb, unsigned mask) {
        if (CONV && mask != FULL) {             // border tiles only
#pragma unroll
            for (int i = 0; i < A_IT; ++i)
                if (!((mask >> i) & 1u)) ra[i] = zero128();
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) st128(wr_a + buf * (BM * BK * 2) + i * (RP * 64 * 2), ra[i]);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) st128(wr_b + buf * (BN * BK * 2) + i * (RP * 64 * 2), rb[i]);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment reads: row = wave tile base + 32 i + (lane&31); the swizzle term is per-thread constant
    const int frow = lane & 31, fk = lane >> 5;
    const char* rd_a[4];
    const char* rd_b[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int c = (((ks * 2 + fk) ^ ((frow >> 1) & 7)) << 3);
        rd_a[ks] = (const char*)sA + ((wm * WM + frow) * 64 + c) * 2;
        rd_b[ks] = (const char*)sB + ((wn * WN + frow) * 64 + c) * 2;
    }
    auto compute = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[MI], bfr[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = as_bf16x8(ld128(rd_a[ks] + buf * A_STRIDE + i * (32 * 64 * 2)));
#pragma unroll
            for (int j = 0; j < NI; ++j) bfr[j] = as_bf16x8(ld128(rd_b[ks] + buf * B_STRIDE + j * (32 * 64 * 2)));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = TR ? rq_mfma_32x32x16_bf16(bfr[j], af[i], acc[i][j]) : rq_mfma_32x32x16_bf16(af[i], bfr[j], acc[i][j]);
        }
    };

    // Two register sets keep two K-tiles of global loads in flight (tile t+1 landing, tile t+2 issued)
    // while tile t is multiplied out of LDS.  The loop body is branch-free apart from the trip count: loads
    // past the last tile re-read the last tile (harmless), so the compiler can use counted vmcnt waits.
    const int nk = kt1 - kt0, last = kt1 - 1;
    if constexpr (GL > 0) {
        static_assert(MODE == 0 && GL >= 2 && GL <= 6, "LDS-DMA staging: dense operands, 2..6 stages");
        constexpr int NW = WGM * WGN;
        constexpr int A_PER = BM / 8 / NW, B_PER = BN / 8 / NW;      // 8-row (1 KB) groups per wavefront and tile
        constexpr int PER = A_PER + B_PER;
        static_assert(A_PER * 8 * NW == BM && B_PER * 8 * NW == BN, "tile rows must split evenly over the wavefronts");
        const int uw = rq_uniform(wave);
        const rq_lds_t lds0 = rq_lds_addr(smem);
        // lane l of a group fills LDS position (row lr, chunk lc): it must fetch the chunk that the swizzled read
        // expects there, lc ^ ((row >> 1) & 7) -- the same involution as the register-staged path's write side
        const int lr = lane >> 3, lc = lane & 7;
        unsigned ga[A_PER], gb[B_PER];
#pragma unroll
        for (int q = 0; q < A_PER; ++q) {
            const int row = (uw * A_PER + q) * 8 + lr;
            int m = m0 + row;
            if (m > p.M - 1) m = p.M - 1;
            ga[q] = ((unsigned)m * p.lda + ((lc ^ ((row >> 1) & 7)) << 3)) * 2u;
        }
#pragma unroll
        for (int q = 0; q < B_PER; ++q) {
            const int row = (uw * B_PER + q) * 8 + lr;
            int n = n0 + row;
            if (n > p.N - 1) n = p.N - 1;
            gb[q] = ((unsigned)n * p.K + ((lc ^ ((row >> 1) & 7)) << 3)) * 2u;
        }
        auto issue = [&](int kt, int st) {
            const unsigned kb = (unsigned)kt * (BK * 2);
            const rq_lds_t base = lds0 + (rq_lds_t)(st * STAGE_BYTES);
#pragma unroll
            for (int q = 0; q < A_PER; ++q) rq_glds16(base + (rq_lds_t)((uw * A_PER + q) * 1024), gA + (ga[q] + kb));
#pragma unroll
            for (int q = 0; q < B_PER; ++q) {
#if RQ_TILED_W_NT
                rq_glds16_nt(base + (rq_lds_t)(BM * 128 + (uw * B_PER + q) * 1024), gW + (gb[q] + kb));
#else
                rq_glds16(base + (rq_lds_t)(BM * 128 + (uw * B_PER + q) * 1024), gW + (gb[q] + kb));
#endif
            }
        };
        static_assert((GL - 2) * PER <= 63, "vmcnt range");
        // diagnostics (scripts only): dbg bit1 = no staging at all (MFMAs + fragment reads + barriers on whatever the LDS holds),
        // bit2 = staging, waits and barriers only (no fragment reads, no MFMAs): which side of the loop a tile shape is bound by
        const bool no_dma = p.dbg & 2, no_mma = p.dbg & 4;
        // Round 6: fragment reads one K-tile AHEAD of the MFMAs (eight / sixteen-wavefront tiles, >= 3 stages, where two fragment sets
        // fit the register budget).  The loop below this one reads tile t's fragments right after the barrier that publishes tile t and
        // only then multiplies: with every wavefront of the tile leaving that barrier together, the matrix pipes sit idle for one LDS
        // round trip per K-tile and then compete with the reads for the LDS (a 128 x 128 tile on sixteen wavefronts reads 128 KB per
        // K-tile = 512 LDS cycles for 512 MFMA cycles per SIMD: 14.8 us of 'MFMA side' for 5.9 us of pipe time,
        // profiles/r05_gemm_mid_probes.txt).  Here the barrier at the top of tile t publishes tile t + 1; its fragments are requested
        // into the second register set and travel while the MFMAs of tile t run on the first.  Tile k lives in stage k % GL; the stage
        // of tile t is free once every wavefront has its fragments in registers (lgkmcnt(0) before the barrier) and is refilled with
        // tile t + GL right after it.  Same MFMA order per accumulator as the plain loop: bit-identical results.
        // MEASURED (profiles/r06_gemm_pf_ab.txt, MI355X, 500 rows, in-graph, cold weights): no gain on the shipped tiles -- 128 x 128 on
        // sixteen wavefronts 18.5 -> 18.9 us (MFMA side alone 14.9 -> 15.6), eight wavefronts 19.0 -> 19.3; the four-wavefront forms gain
        // (2 x 2: 25.8 -> 21.1 us) and still lose to sixteen.  The launch is bound by its cold weight stream (staging alone 18.0-18.5 us),
        // not by the order of reads and MFMAs.  Off by default (-DRQ_GEMM_PF=1 -DRQ_GEMM_PF_MINW=4 builds it).
        constexpr bool PF = RQ_GEMM_PF && GL >= 3 && NW >= RQ_GEMM_PF_MINW && ((MI + NI) * 32 + MI * NI * 16 + 40 <= 2048 / NW);
        if constexpr (PF) {
            bf16x8 fa[2][4][MI], fb[2][4][NI];
            auto read_frags = [&](int stg, int set) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) fa[set][ks][i] = as_bf16x8(ld128(rd_a[ks] + stg * A_STRIDE + i * (32 * 64 * 2)));
#pragma unroll
                    for (int j = 0; j < NI; ++j) fb[set][ks][j] = as_bf16x8(ld128(rd_b[ks] + stg * B_STRIDE + j * (32 * 64 * 2)));
                }
            };
            auto mma = [&](int set) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = TR ? rq_mfma_32x32x16_bf16(fb[set][ks][j], fa[set][ks][i], acc[i][j])
                                           : rq_mfma_32x32x16_bf16(fa[set][ks][i], fb[set][ks][j], acc[i][j]);
            };
            // own DMAs of tile `need` have landed: at most min(GL - 2, last issued - need) newer tiles stay in flight
            auto wait_tile = [&](int newer) {
                if (GL >= 6 && newer >= 4) rq_wait_vmcnt<(GL >= 6 ? 4 : 0) * PER>();
                else if (GL >= 5 && newer >= 3) rq_wait_vmcnt<(GL >= 5 ? 3 : 0) * PER>();
                else if (GL >= 4 && newer >= 2) rq_wait_vmcnt<(GL >= 4 ? 2 : 0) * PER>();
                else if (newer >= 1) rq_wait_vmcnt<PER>();
                else rq_wait_vmcnt<0>();
            };
            if (nk > 0) {
#pragma unroll
                for (int s0 = 0; s0 < GL - 1; ++s0)
                    if (s0 < nk && !no_dma) issue(kt0 + s0, s0);
                wait_tile(nk - 1 < GL - 2 ? nk - 1 : GL - 2);
                rq_barrier_raw();                                      // publishes tile 0
                if (GL - 1 < nk && !no_dma) issue(kt0 + GL - 1, GL - 1);
                if (!no_mma) read_frags(0, 0);
                int st = 0;
                // one step: tile t multiplied from register set SET while tile t + 1 is read into the other
                auto step = [&](int t, auto set_c) {
                    constexpr int SET = decltype(set_c)::value;
                    const int nx = st + 1 == GL ? 0 : st + 1;
                    if (t + 1 < nk) {
                        const int issued_last = t + GL - 1 < nk - 1 ? t + GL - 1 : nk - 1;
                        wait_tile(issued_last - (t + 1));
                        rq_wait_lgkmcnt<0>();                          // tile t's fragments are in registers: its stage may be refilled
                        rq_barrier_raw();                              // publishes tile t + 1
                        if (t + GL < nk && !no_dma) issue(kt0 + t + GL, st);
                        if (!no_mma) read_frags(nx, SET ^ 1);
                        rq_sched_barrier();
                    }
                    if (!no_mma) mma(SET);
                    st = nx;
                };
                int t = 0;
                for (; t + 1 < nk; t += 2) {
                    step(t, std::integral_constant<int, 0>{});
                    step(t + 1, std::integral_constant<int, 1>{});
                }
                if (t < nk) step(t, std::integral_constant<int, 0>{});
            }
        } else
        if (nk > 0) {
            RQ_GLT(0);
#pragma unroll
            for (int s0 = 0; s0 < GL - 1; ++s0)
                if (s0 < nk && !no_dma) issue(kt0 + s0, s0);
            RQ_GLT(1);
            int st = 0;
            for (int t = 0; t < nk; ++t) {
                // tile t has landed once at most (tiles issued after it: up to GL - 2) * PER loads are outstanding
                const int newer = nk - 1 - t;
                if (GL >= 6 && newer >= 4) rq_wait_vmcnt<(GL >= 6 ? 4 : 0) * PER>();
                else if (GL >= 5 && newer >= 3) rq_wait_vmcnt<(GL >= 5 ? 3 : 0) * PER>();
                else if (GL >= 4 && newer >= 2) rq_wait_vmcnt<(GL >= 4 ? 2 : 0) * PER>();
                else if (GL >= 3 && newer >= 1) rq_wait_vmcnt<PER>();
                else rq_wait_vmcnt<0>();
                if (t < 6) RQ_GLT(2 + 4 * t);
                rq_barrier_raw();            // publishes stage st; every wave is past its reads of stage st-1 (refilled next)
                if (t < 6) RQ_GLT(3 + 4 * t);
                if (t + GL - 1 < nk && !no_dma) issue(kt0 + t + GL - 1, st == 0 ? GL - 1 : st - 1);
                if (t < 6) RQ_GLT(4 + 4 * t);
                if (!no_mma) compute(st);
                if (t < 6) RQ_GLT(5 + 4 * t);
                st = st + 1 == GL ? 0 : st + 1;
            }
            RQ_GLT(26);
        }
    } else {
    rq_u128 ra0[A_IT], rb0[B_IT], ra1[A_IT], rb1[B_IT];
    unsigned mk0 = 0, mk1 = 0;
    if (nk > 0) {
        mk0 = load_tile(kt0, ra0, rb0);
        mk1 = load_tile(kt0 + 1 < kt1 ? kt0 + 1 : last, ra1, rb1);
        store_tile(0, ra0, rb0, mk0);
        rq_syncthreads();
        // whole pairs, no exit from the middle of the body: a mid-loop break adds a CFG edge into the loop
        // header along which hipcc's waitcnt pass sees the first half's loads pending, and it then drains
        // them at the top of every iteration (vmcnt(1) in the ISA) -- that serialised the pipeline.
        const int npair = nk >> 1;
        if (p.dbg & 6) {
            // ablations (scripts only): bit1 = MFMA + LDS reads + barriers, no staging at all;
            // bit2 = global loads issued but never written to LDS
            for (int pi = 0; pi < npair; ++pi) {
                if (p.dbg & 4) { mk0 = load_tile(kt0 + 2 * pi + 2 < kt1 ? kt0 + 2 * pi + 2 : last, ra0, rb0); rq_sched_barrier(); }
                compute(0);
                rq_syncthreads();
                if (p.dbg & 4) { mk1 = load_tile(kt0 + 2 * pi + 3 < kt1 ? kt0 + 2 * pi + 3 : last, ra1, rb1); rq_sched_barrier(); }
                compute(1);
                rq_syncthreads();
            }
            if (p.dbg & 4) { store_tile(0, ra0, rb0, mk0); store_tile(1, ra1, rb1, mk1); }
        } else {
        auto pair = [&](int pi) {
            const int i = 2 * pi;
            // even tile in LDS buffer 0; set 0 is free, set 1 holds tile i+1
            mk0 = load_tile(kt0 + i + 2 < kt1 ? kt0 + i + 2 : last, ra0, rb0);
            rq_sched_barrier();      // keep the loads above the MFMAs: hipcc sinks them below the LDS writes otherwise
            compute(0);
            store_tile(1, ra1, rb1, mk1);
            rq_syncthreads();
            // odd tile in LDS buffer 1; set 1 is free, set 0 holds tile i+2
            mk1 = load_tile(kt0 + i + 3 < kt1 ? kt0 + i + 3 : last, ra1, rb1);
            rq_sched_barrier();
            compute(1);
            store_tile(0, ra0, rb0, mk0);
            rq_syncthreads();
        };
        if constexpr (VS) {
            // virtual split-K (GemmArgs::vsplit): the same pipelined pair loop, cut into chunks of ppc pairs; at a chunk boundary
            // the accumulators move into the running total (chunk 0: assignment, as the slab reduce starts from slab 0) and restart
            // from zero.  The boundary sits between two pair bodies, so the loads in flight across it are untouched.
            // One flat loop over the pairs (the nested chunk / pair loops of the first version cost 14-16 % on these kernels: 2-3
            // pairs per chunk left the software pipeline little to run on between loop boundaries); the fold hangs off the END of a
            // pair body, where both ways into the next iteration carry the same loads in flight.  tot starts at zero: chunk 0 is
            // then 0 + acc -- except that (+0) + (-0) = +0 where the slab reduce, which starts FROM slab 0, keeps -0; the sign of
            // an exact zero is restored below so that the two forms stay bit-identical.
            f32x16 tot[MI][NI];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tot[i][j][r] = -0.f;      // (-0) + x == x for every x, including x = -0 and x = +0
            const int ppc = npair / p.vsplit;              // launcher: nk == 2 * ppc * vsplit
            int left = ppc;
            for (int pi = 0; pi < npair; ++pi) {
                pair(pi);
                if (--left == 0) {
                    left = ppc;
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                tot[i][j][r] += acc[i][j][r];
                                acc[i][j][r] = 0.f;
                            }
                }
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = tot[i][j];
        } else {
            for (int pi = 0; pi < npair; ++pi) pair(pi);
        }
        }
        if (nk & 1) compute(0);      // odd tile count: the last tile already sits in buffer 0 (never with VS: nk is even)
    }
    }   // GL == 0

    rq_gemm_epilogue<BM, BN, TR, WGM, WGN, (GL ? GL : 2) * (BM + BN) * 64 * 2>(p, acc, smem, m0, n0);
    if (GL > 0) RQ_GLT(27);
}

// -------------------------------------------------------------------------------------------------
// Register-blocked dense GEMM: 256 x 128 tile, FOUR wavefronts (2 x 2) of 128 x 64 each, K-steps of 32 through a ring
// of NS 24-KB LDS stages filled by LDS-DMA.  The 64 x 64 wave tile of the kernel above reads 1 KB of fragments from
// LDS per MFMA -- exactly the LDS port's rate at full MFMA rate, so that loop tops out near 50 % of peak; 128 x 64
// reads 0.75 KB.  128 accumulator registers per lane are affordable here only because LDS-DMA needs no staging
// registers; the half-depth stage keeps LDS at 72 KB (= the bf16 epilogue tile) so that two workgroups share a CU and
// one's barrier / LDS latency is covered by the other.  Tile rows are 64 bytes: 16-byte chunk c of row r sits at
// slot c ^ ((r >> 2) & 3) (conflict-free ds_read_b128 over 16 consecutive rows); the DMA writes lane-linear, so the
// permutation is applied to each lane's source address.
// BM = 256: wave tiles 128 x 64 (the register-blocked form described above).  BM = 128: wave tiles 64 x 64 with the same
// half-depth stages -- 16 KB per stage, three stages = 48 KB, so THREE workgroups (12 wavefronts) share a CU where the
// BK = 64 kernel above fits two.
template <int BM, int TR, int NS>
__global__ __launch_bounds__(256, BM == 256 ? 2 : 3) void gemm_rb_kernel(GemmArgs p) {
    constexpr int BN = 128, BK = 32, WGM = 2, WGN = 2, NW = 4;
    constexpr int WM = BM / WGM, WN = BN / WGN, MI = WM / 32, NI = WN / 32;
    constexpr int STAGE_BYTES = (BM + BN) * BK * 2;                      // 24 KB
    constexpr int SMEM_BYTES = NS * STAGE_BYTES > BM * (BN * 2 + 16) ? NS * STAGE_BYTES : BM * (BN * 2 + 16);
    constexpr int A_PER = BM / 16 / NW, B_PER = BN / 16 / NW, PER = A_PER + B_PER;      // 16-row (1 KB) groups per wave
    RQ_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = rq_uniform(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int MT = (p.M + BM - 1) / BM, NT = (p.N + BN - 1) / BN;
    int mt, nt;
    {   // same XCD-aware schedules as gemm_bf16_kernel
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
        if (p.sched == 1) {
            const int nb = NT >> 3, rem = NT & 7;
            const int nn = nb + (xcd < rem ? 1 : 0);
            const int n_lo = xcd * nb + (xcd < rem ? xcd : rem);
            if (slot >= nn * MT) return;
            const int per_group = p.sched_gm * nn;
            const int mg = slot / per_group, r = slot - mg * per_group;
            int gm = MT - mg * p.sched_gm;
            gm = gm < p.sched_gm ? gm : p.sched_gm;
            nt = n_lo + r / gm;
            mt = mg * p.sched_gm + (r - (r / gm) * gm);
        } else if (p.sched == 2) {
            const int mm = (MT + 7) >> 3;
            mt = xcd * mm + slot / NT;
            nt = slot - (slot / NT) * NT;
            if (slot >= mm * NT || mt >= MT) return;
        } else {
            mt = id / NT;
            nt = id - mt * NT;
        }
    }
    const int m0 = mt * BM, n0 = nt * BN;
    const int kt_total = p.K / BK;
    const int per = (kt_total + p.splitk - 1) / p.splitk;
    const int kt0 = blockIdx.z * per;
    const int kt1 = (kt0 + per < kt_total) ? kt0 + per : kt_total;
    const int nk = kt1 - kt0;

    const char* gA = (const char*)p.A;
    const char* gW = (const char*)p.W;
    const rq_lds_t lds0 = rq_lds_addr(smem);
    const int lr = lane >> 2, lc = lane & 3;             // lane fills (row lr of its group, slot lc)
    unsigned ga[A_PER], gb[B_PER];
#pragma unroll
    for (int q = 0; q < A_PER; ++q) {
        const int row = (wave * A_PER + q) * 16 + lr;
        int m = m0 + row;
        if (m > p.M - 1) m = p.M - 1;
        ga[q] = ((unsigned)m * p.lda + ((lc ^ ((row >> 2) & 3)) << 3)) * 2u;
    }
#pragma unroll
    for (int q = 0; q < B_PER; ++q) {
        const int row = (wave * B_PER + q) * 16 + lr;
        int n = n0 + row;
        if (n > p.N - 1) n = p.N - 1;
        gb[q] = ((unsigned)n * p.K + ((lc ^ ((row >> 2) & 3)) << 3)) * 2u;
    }
    auto issue = [&](int kt, int st) {
        const unsigned kb = (unsigned)kt * (BK * 2);
        const rq_lds_t base = lds0 + (rq_lds_t)(st * STAGE_BYTES);
#pragma unroll
        for (int q = 0; q < A_PER; ++q) rq_glds16(base + (rq_lds_t)((wave * A_PER + q) * 1024), gA + (ga[q] + kb));
#pragma unroll
        for (int q = 0; q < B_PER; ++q) rq_glds16(base + (rq_lds_t)(BM * BK * 2 + (wave * B_PER + q) * 1024), gW + (gb[q] + kb));
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment reads: row = wave tile base + 32 i + (lane & 31), chunk = 2 ks + (lane >> 5)
    const int frow = lane & 31, fk = lane >> 5;
    unsigned rd_a[2], rd_b[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const unsigned c = (unsigned)(((ks * 2 + fk) ^ ((frow >> 2) & 3)) << 4);
        rd_a[ks] = (unsigned)((wm * WM + frow) * (BK * 2)) + c;
        rd_b[ks] = (unsigned)(BM * BK * 2 + (wn * WN + frow) * (BK * 2)) + c;
    }
    auto compute = [&](int st) {
        const char* sb = (const char*)smem + st * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[MI], bfr[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = as_bf16x8(ld128(sb + rd_a[ks] + i * (32 * BK * 2)));
#pragma unroll
            for (int j = 0; j < NI; ++j) bfr[j] = as_bf16x8(ld128(sb + rd_b[ks] + j * (32 * BK * 2)));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = TR ? rq_mfma_32x32x16_bf16(bfr[j], af[i], acc[i][j]) : rq_mfma_32x32x16_bf16(af[i], bfr[j], acc[i][j]);
        }
    };

    if (nk > 0) {
#pragma unroll
        for (int s0 = 0; s0 < NS - 1; ++s0)
            if (s0 < nk) issue(kt0 + s0, s0);
        int st = 0;
        for (int t = 0; t < nk; ++t) {
            const int newer = nk - 1 - t;                // stages issued after tile t that may still be in flight
            if (NS >= 4 && newer >= 2) rq_wait_vmcnt<2 * PER>();
            else if (NS >= 3 && newer >= 1) rq_wait_vmcnt<PER>();
            else rq_wait_vmcnt<0>();
            rq_barrier_raw();
            if (t + NS - 1 < nk) issue(kt0 + t + NS - 1, st == 0 ? NS - 1 : st - 1);
            compute(st);
            st = st + 1 == NS ? 0 : st + 1;
        }
    }
    rq_gemm_epilogue<BM, BN, TR, WGM, WGN, SMEM_BYTES>(p, acc, smem, m0, n0);
}


// -------------------------------------------------------------------------------------------------
// tile id -> (m-tile, n-tile) under the XCD-aware schedules described in gemm_bf16_kernel; false = surplus workgroup of the
// padded grid (must return before any barrier).
static __device__ __forceinline__ bool rq_gemm_tile_coords(const GemmArgs& p, int BM, int BN, int& mt, int& nt) {
    const int MT = (p.M + BM - 1) / BM, NT = (p.N + BN - 1) / BN;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    if (p.sched == 1) {
        const int nb = NT >> 3, rem = NT & 7;
        const int nn = nb + (xcd < rem ? 1 : 0);
        const int n_lo = xcd * nb + (xcd < rem ? xcd : rem);
        if (slot >= nn * MT) return false;
        const int per_group = p.sched_gm * nn;
        const int mg = slot / per_group, r = slot - mg * per_group;
        int gm = MT - mg * p.sched_gm;
        gm = gm < p.sched_gm ? gm : p.sched_gm;
        nt = n_lo + r / gm;
        mt = mg * p.sched_gm + (r - (r / gm) * gm);
    } else if (p.sched == 2) {
        const int mm = (MT + 7) >> 3;
        mt = xcd * mm + slot / NT;
        nt = slot - (slot / NT) * NT;
        if (slot >= mm * NT || mt >= MT) return false;
    } else if (p.sched == 3) {
        // balanced: the tile space is walked in groups of sched_gm m-tiles x all n-tiles (m fastest inside a group, so
        // consecutive tiles share a W panel and neighbouring A panels), and XCD x owns a CONTIGUOUS range of that walk whose
        // length differs by at most one tile between XCDs -- an XCD is a 32-CU machine of its own, so the per-XCD counts,
        // not the total, decide how many rounds a launch takes (42 m-tiles in bands of 6 left one XCD idle).
        const int T = MT * NT, q = T >> 3, r = T & 7;
        const int cnt = q + (xcd < r ? 1 : 0);
        if (slot >= cnt) return false;
        const int L = xcd * q + (xcd < r ? xcd : r) + slot;
        const int G = p.sched_gm * NT;
        const int mg = L / G, rem = L - mg * G;
        int gm = MT - mg * p.sched_gm;
        gm = gm < p.sched_gm ? gm : p.sched_gm;
        nt = rem / gm;
        mt = mg * p.sched_gm + (rem - nt * gm);
    } else {
        mt = id / NT;
        nt = id - mt * NT;
        if (mt >= MT) return false;
    }
    return true;
}

// -------------------------------------------------------------------------------------------------
// 256 x 256 x 64 "eight-phase" dense GEMM for the compute-bound decode steps (M >= ~2048 rows): 8 wavefronts as 2 (M) x 4 (N),
// wave tile 128 x 64 = acc[4][2] of v_mfma_f32_32x32x16_bf16 (128 accumulator registers), 128 KB of LDS as a ring of EIGHT 16-KB
// units filled by LDS-DMA, counted vmcnt waits, raw s_barriers, s_setprio around the MFMA bursts
// (/opt/skills/guides/cdna_hip_programming.md, "The 256^2 8-phase template"; written from that description).
//
// Why: the 128x128 kernel above reads 1 KB of fragments from LDS per MFMA and drains its staging at a barrier every K-tile
// (measured: 0.30 of the MFMA peak, 33-47 % of wave time parked at s_barrier / s_waitcnt).  Here a wave reads 0.75 KB per MFMA,
// the two wave rows run half a phase apart -- one row's MFMA burst covers the other row's ds_reads and DMA issue on the same
// SIMDs -- and no wait ever drains the DMA queue: four units (64 KB) stay in flight across every barrier.
//
// K-tile t lives in LDS buffer t & 1 as four units, each [128 rows][64 k] bf16 with the 16-byte-chunk swizzle
// chunk ^ ((row >> 1) & 7) (conflict-free ds_read_b128; the DMA writes lane-linear, so the permutation is applied to the
// per-lane source address):
//   AH0 / AH1: A rows  m0 + wm*128 + mh*64 + [0,64)   for wm = 0,1   (unit row u = wm*64 + r)
//   BH0 / BH1: W rows  n0 + wn*64  + nb*32 + [0,32)   for wn = 0..3  (unit row u = wn*32 + r)
// so every unit is consumed by ALL waves in exactly ONE phase.  Phase j of tile t (global phase k = 4t + j):
//   j   ds_read (this wave)          MFMAs (8 = 2 m-blocks x 4 k-steps)   DMA issued (one unit, 2 instructions per wave)
//   1   BH0 -> B(n0), AH0 -> A(mh0)  (mh0, n0)                            BH1 of tile t+1
//   2   BH1 -> B(n1)                 (mh0, n1)                            AH1 of tile t+1
//   3   AH1 -> A(mh1)                (mh1, n1)                            AH0 of tile t+2
//   4   --  (B(n0) kept in registers) (mh1, n0)                           BH0 of tile t+2
// Each phase is  { ds_reads; DMA issue; s_waitcnt vmcnt(8); s_barrier }  { s_waitcnt lgkmcnt(0); 8 MFMAs; s_barrier },
// and wave row 1 executes one extra s_barrier up front, so it always runs one half-phase behind wave row 0.
// Hazards, with that stagger (intervals between consecutive workgroup barriers; row 0 loads in interval 2k-1 and computes in
// 2k, row 1 loads in 2k and computes in 2k+1):
//   RAW  a unit waited for in phase k (vmcnt before the barrier that ends the load half) is read in phase >= k+1.  vmcnt(8)
//        after this phase's issue leaves the four newest units in flight and retires the one issued four phases ago, which is
//        exactly the unit the next phase reads (check: BH1(t+1) issued at 4t+1, retired at 4t+5, read at 4t+6 = phase 2 of t+1).
//   WAR  a unit is refilled >= 2 phases after its last ds_read (AH0: read phase 1, refilled phase 3; BH0 1 -> 4; BH1 2 -> 5;
//        AH1 3 -> 6): row 1 retires its phase-k reads (lgkmcnt(0)) in interval 2k+1, row 0 issues the refill in interval
//        2(k+2)-1 = 2k+3.
// The last two K-tiles issue nothing new and wait with the matching smaller counts.  Needs K/64 (per split) >= 2.
// PH = phases per K-tile.  4 (round 2's first schedule): every 16-KB unit is consumed in its own phase, 8 MFMAs per wavefront
// between barriers.  2: phase I reads AH0, BH0, BH1 and runs the 16 MFMAs of m-half 0, phase II reads AH1 and runs m-half 1 (B stays
// in registers) -- half the barriers per K-tile (a phase's two barriers and waits cost ~300 cycles whatever its length, and at 8
// MFMAs the non-MFMA half of a phase was longer than the other wavefront row's MFMA burst that is meant to cover it).  Staging for
// PH = 2: phase I(t) refills [AH0, BH0, BH1] of the other buffer with tile t+1 (last read two phases ago, in I(t-1)), phase II(t)
// refills its AH1; the wait that ends I(t) leaves three units in flight (vmcnt(6): AH1(t) has landed for II(t)), the wait that
// ends II(t) one (vmcnt(2): [AH0, BH0, BH1](t+1) have landed for I(t+1)).
#ifndef RQ_P8_A_POL          // cache policy of the operand DMAs (A/B switches: ` nt` measured +7 ... +9 %, profiles/r03_gemm_p8_cache_policy.txt)
#define RQ_P8_A_POL 0
#endif
#ifndef RQ_P8_W_POL
#define RQ_P8_W_POL 0
#endif
template <int TR, int PH = 4, int EK = -1>
__global__ __launch_bounds__(512) void gemm_p8_kernel(GemmArgs p) {
    constexpr int BM = 256, BN = 256, BK = 64, WGM = 2, WGN = 4;
    constexpr int UNIT = 128 * BK * 2;                 // 16 KB
    constexpr int BUF = 4 * UNIT;                      // one K-tile: AH0, BH0, BH1, AH1
    constexpr int U_AH0 = 0, U_BH0 = UNIT, U_BH1 = 2 * UNIT, U_AH1 = 3 * UNIT;
    constexpr int SMEM_BYTES = BM * (BN * 2 + 16);     // the bf16 epilogue tile (135 168 B) >= the 128-KB operand ring
    RQ_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = rq_uniform(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    int mt, nt;
    if (!rq_gemm_tile_coords(p, BM, BN, mt, nt)) return;
    RQ_GT(0);
    const int m0 = mt * BM, n0 = nt * BN;
    const int kt_total = p.K / BK;
    const int per = (kt_total + p.splitk - 1) / p.splitk;
    const int kt0 = blockIdx.z * per;
    const int kt1 = (kt0 + per < kt_total) ? kt0 + per : kt_total;
    const int nk = kt1 - kt0;

    // ---- DMA source offsets: wave w fills 8-row groups 2w and 2w+1 of every unit; lane (lr, lc) fills (row 8g + lr, chunk lc)
    const int lr = lane >> 3, lc = lane & 7;
    unsigned a_off[2][2], b_off[2][2];                                // [mh | nb][q]; rows >= M / >= N are clamped (never stored)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int u = (2 * wave + q) * 8 + lr;                       // unit row 0..127
        const unsigned sw = (unsigned)((lc ^ ((u >> 1) & 7)) << 4);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int m = m0 + (u >> 6) * 128 + h * 64 + (u & 63);
            int n = n0 + (u >> 5) * 64 + h * 32 + (u & 31);
            m = m < p.M - 1 ? m : p.M - 1;
            n = n < p.N - 1 ? n : p.N - 1;
            a_off[h][q] = (unsigned)m * (unsigned)p.lda * 2u + sw;
            b_off[h][q] = (unsigned)n * (unsigned)p.K * 2u + sw;
        }
    }
    const char* gA = (const char*)p.A;
    const char* gW = (const char*)p.W;
    const rq_lds_t lds0 = rq_lds_addr(smem);
    const rq_lds_t my_grp = (rq_lds_t)(2 * wave) * 1024;
    // unit `uo` (byte offset inside a buffer) of K-tile kt (absolute index) into buffer (kt - kt0) & 1
    auto stage_a = [&](int kt, int uo, int mh) {
        const char* base = gA + (size_t)kt * (BK * 2);
        const rq_lds_t dst = lds0 + (rq_lds_t)(((kt - kt0) & 1) * BUF + uo) + my_grp;
        rq_glds16_s2<RQ_P8_A_POL>(dst, base, a_off[mh][0], a_off[mh][1]);
    };
    auto stage_b = [&](int kt, int uo, int nb) {
        const char* base = gW + (size_t)kt * (BK * 2);
        const rq_lds_t dst = lds0 + (rq_lds_t)(((kt - kt0) & 1) * BUF + uo) + my_grp;
        rq_glds16_s2<RQ_P8_W_POL>(dst, base, b_off[nb][0], b_off[nb][1]);
    };

    // K-tile 0 is requested HERE, ahead of the fragment addresses and the 128 accumulator moves below: the first DMA used to be
    // instruction ~500 of the kernel, i.e. a tile's 24 K-tiles started ~2000 cycles after the workgroup did
    if (nk >= 2) { stage_a(kt0, U_AH0, 0); stage_b(kt0, U_BH0, 0); stage_b(kt0, U_BH1, 1); stage_a(kt0, U_AH1, 1); }
    rq_sched_barrier();

    // ---- fragment read offsets: row = (wave block) + (lane & 31), 16-byte chunk 2 ks + (lane >> 5), swizzled
    const int frow = lane & 31, fk = lane >> 5;
    unsigned rd_a[4], rd_b[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const unsigned c = (unsigned)(((ks * 2 + fk) ^ ((frow >> 1) & 7)) << 4);
        rd_a[ks] = (unsigned)((wm * 64 + frow) * (BK * 2)) + c;      // + (i & 1) * 32 rows, unit AH[i >> 1]
        rd_b[ks] = (unsigned)((wn * 32 + frow) * (BK * 2)) + c;      // unit BH[j]
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fa[2][4], fb0[4], fb1[4];          // A fragments of the current m-half, B fragments of n-block 0 / 1

    auto read_a = [&](const char* sb, int uo) {        // 8 x ds_read_b128
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fa[ii][ks] = as_bf16x8(ld128(sb + uo + rd_a[ks] + ii * (32 * BK * 2)));
    };
    auto read_b = [&](const char* sb, int uo, bf16x8 (&fb)[4]) {   // 4 x ds_read_b128
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb[ks] = as_bf16x8(ld128(sb + uo + rd_b[ks]));
    };
    auto mma = [&](int ih, int j, const bf16x8 (&fb)[4]) {          // 8 MFMAs: m-blocks 2 ih, 2 ih + 1 x n-block j x 4 k-steps
        rq_sched_barrier();
        rq_wait_lgkmcnt<0>();
        rq_sched_barrier();            // keeps the register-only MFMAs below the wait (hipcc hoists them past inline-asm waits)
        rq_setprio(1);                         // (measured null against no priority raise, profiles/r02_gemm_p8_prio.txt; the run-time
                                               // A/B switch that used to sit here cost a VALU compare + branch per MFMA burst)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
                acc[2 * ih + ii][j] = TR ? rq_mfma_32x32x16_bf16(fb[ks], fa[ii][ks], acc[2 * ih + ii][j])
                                         : rq_mfma_32x32x16_bf16(fa[ii][ks], fb[ks], acc[2 * ih + ii][j]);
        rq_setprio(0);
        rq_sched_barrier();
        rq_barrier_raw();
        rq_sched_barrier();
    };
    // end of a load half: the DMA wait (N = 2 x units allowed to stay in flight), then the barrier that publishes it
#define RQ_P8_LOAD_END(N)          \
    rq_sched_barrier();            \
    rq_wait_vmcnt<N>();            \
    rq_barrier_raw();              \
    rq_sched_barrier();

    if (PH == 2 && nk >= 2) {
        // (Measured and not kept, profiles/r03_gemm_p8_bh1_in_burst.txt: the DMA of BH1(t+1) issued from inside phase I's MFMA burst
        // instead of in its load half, which carries 16 ds_reads + 6 DMA instructions against 8 + 2 in phase II -- bit-identical,
        // qkv / proj / fc1 -0.3 ... -0.6 %, fc2 +7 %, classifier +2 %.)
        auto mma16 = [&](int ih, bool n1_first) {      // 16 MFMAs: m-blocks 2 ih, 2 ih + 1 x both n-blocks x 4 k-steps
            rq_sched_barrier();
            rq_wait_lgkmcnt<0>();
            rq_sched_barrier();
            rq_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = n1_first ? 1 - jj : jj;
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        const bf16x8 b = j ? fb1[ks] : fb0[ks];
                        acc[2 * ih + ii][j] = TR ? rq_mfma_32x32x16_bf16(b, fa[ii][ks], acc[2 * ih + ii][j])
                                                 : rq_mfma_32x32x16_bf16(fa[ii][ks], b, acc[2 * ih + ii][j]);
                    }
                }
            }
            rq_setprio(0);
            rq_sched_barrier();
            rq_barrier_raw();
            rq_sched_barrier();
        };
        // prologue: tile 0 whole (requested above); AH0, BH0, BH1 must have landed before phase I reads them (AH1 may still be in flight)
        RQ_P8_LOAD_END(2)
        if (wm == 1) rq_barrier_raw();           // wave row 1 runs half a phase behind
        rq_sched_barrier();
        // one K-tile; MORE (compile time): tile t + 1 exists and is staged -- the last tile runs from a copy of the body without the
        // staging (as a run-time `if (t + 1 < nk)` inside one loop body every phase carried two scalar branches and their mask set-up)
        auto k_tile = [&](int t, auto more_tag) {
            constexpr bool more = decltype(more_tag)::value;
            const char* sb = (const char*)smem + (t & 1) * BUF;
            const int kt = kt0 + t;
            // ---- phase I
            read_b(sb, U_BH0, fb0);
            read_b(sb, U_BH1, fb1);
            rq_sched_barrier();
            read_a(sb, U_AH0);
            if (more) { stage_a(kt + 1, U_AH0, 0); stage_b(kt + 1, U_BH0, 0); stage_b(kt + 1, U_BH1, 1); RQ_P8_LOAD_END(6) }
            else { RQ_P8_LOAD_END(0) }
            mma16(0, false);
            // ---- phase II (B is still in registers)
            read_a(sb, U_AH1);
            if (more) { stage_a(kt + 1, U_AH1, 1); RQ_P8_LOAD_END(2) }
            else { RQ_P8_LOAD_END(0) }
            mma16(1, true);
        };
        for (int t = 0; t + 1 < nk; ++t) k_tile(t, std::true_type{});
        k_tile(nk - 1, std::false_type{});
        if (wm == 0) rq_barrier_raw();           // wave row 0 catches up
        rq_sched_barrier();
    } else
    if (nk >= 2) {
        // prologue: tile 0 whole (requested above), AH0 / BH0 of tile 1; the first two units must have landed before phase 1 reads them
        stage_a(kt0 + 1, U_AH0, 0); stage_b(kt0 + 1, U_BH0, 0);
        RQ_P8_LOAD_END(8)
        if (wm == 1) rq_barrier_raw();           // wave row 1 runs half a phase behind (uniform per wave: wm is scalar)
        rq_sched_barrier();
        for (int t = 0; t < nk; ++t) {
            const char* sb = (const char*)smem + (t & 1) * BUF;
            const int kt = kt0 + t;
            const int ahead = nk - 1 - t;        // K-tiles after this one
            if (t == 10) RQ_GT(6);
            // ---- phase 1
            read_b(sb, U_BH0, fb0);
            rq_sched_barrier();
            read_a(sb, U_AH0);
            if (ahead >= 1) { stage_b(kt + 1, U_BH1, 1); RQ_P8_LOAD_END(8) }          // in flight: AH1(t+1) .. this one
            else { RQ_P8_LOAD_END(2) }                                                   // last tile: only AH1(t) may still be pending
            mma(0, 0, fb0);
            if (t == 10) RQ_GT(7);
            // ---- phase 2
            read_b(sb, U_BH1, fb1);
            if (ahead >= 1) { stage_a(kt + 1, U_AH1, 1); RQ_P8_LOAD_END(8) }
            else { RQ_P8_LOAD_END(0) }
            mma(0, 1, fb1);
            if (t == 10) RQ_GT(8);
            // ---- phase 3
            read_a(sb, U_AH1);
            if (ahead >= 2) { stage_a(kt + 2, U_AH0, 0); RQ_P8_LOAD_END(8) }
            else if (ahead == 1) { RQ_P8_LOAD_END(6) }                                   // BH0(t+1), BH1(t+1), AH1(t+1) stay in flight
            else { RQ_P8_LOAD_END(0) }
            mma(1, 1, fb1);
            if (t == 10) RQ_GT(9);
            // ---- phase 4 (no LDS reads: B(n0) is still in registers)
            if (ahead >= 2) { stage_b(kt + 2, U_BH0, 0); RQ_P8_LOAD_END(8) }
            else if (ahead == 1) { RQ_P8_LOAD_END(4) }                                   // BH1(t+1), AH1(t+1) stay in flight
            else { RQ_P8_LOAD_END(0) }
            mma(1, 0, fb0);
            if (t == 10) RQ_GT(10);
        }
        if (wm == 0) rq_barrier_raw();           // wave row 0 catches up: every wave has executed the same number of barriers
        rq_sched_barrier();
    }
#undef RQ_P8_LOAD_END
    // interior tile with every vector-store condition met (all tiles of the benchmark shapes): the check-free epilogue
    const bool full = m0 + BM <= p.M && n0 + BN <= p.N && (p.N & 7) == 0 && (p.ldo & 7) == 0 &&
                      (p.epi != EPI_BF16_RESID || (p.ldr & 3) == 0) && !(p.dbg & 1);
    if (full) rq_gemm_epilogue<BM, BN, TR, WGM, WGN, SMEM_BYTES, true, EK>(p, acc, smem, m0, n0);
    else rq_gemm_epilogue<BM, BN, TR, WGM, WGN, SMEM_BYTES, false, EK>(p, acc, smem, m0, n0);
}

// -------------------------------------------------------------------------------------------------
// Skinny GEMM for the small-batch decode steps (M <= 64 rows: BASELINE configs[3]'s per-GPU share of 64 images):
// out[M, N] = x[M, K] . W[N, K]^T is one pass over W at HBM speed with almost no arithmetic, and what bounded the tiled
// kernels there was the 24-step K chain of a workgroup (barrier + LDS round trip per step: 10 us for 14 MB of weights).
// Here nothing goes through LDS until the end: a workgroup owns 32 weight rows, its EIGHT wavefronts each take one eighth of
// the K range (in-workgroup split-K), every wavefront issues ALL its loads up front -- W fragments straight from HBM (each
// lane 64 contiguous bytes of one row per 64-wide step; a lane pair covers a full 128-byte line), x fragments from L2 (x is
// <= 200 KB) -- runs its MFMAs (transposed tile D[n][m]), and the eight partial tiles are summed through LDS by all 512
// threads, which apply bias / GELU and write row-contiguous 8- or 16-byte pieces.  The k-order inside an MFMA is permuted
// identically for both operands (lane group g of step s feeds k = 32 g + 8 i .. + 7 to k-step i), which a dot product ignores.
// blockIdx.z = global split-K slice (fp32 partial slabs, residual-producing GEMMs only).  Needs (K / splitk) % 512 == 0.
template <int UNUSED = 0>      // (a template only so that the header may be included by several translation units)
__global__ __launch_bounds__(512) void gemm_skinny_kernel(GemmArgs p) {
    constexpr int CH = 3;                         // 64-wide steps loaded per batch (36 x 16 bytes per lane in flight)
    RQ_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = rq_uniform(tid >> 6);
    const int n0 = blockIdx.x * 32;
    const int Kz = p.K / p.splitk, Kw = Kz >> 3;          // K range of this workgroup / of one wavefront
    const int kbeg = blockIdx.z * Kz + wave * Kw;
    const int steps = Kw >> 6;
    const int fr = lane & 31, kg = lane >> 5;
    int n = n0 + fr;
    n = n < p.N - 1 ? n : p.N - 1;
    int m_a = fr < p.M - 1 ? fr : p.M - 1, m_b = fr + 32 < p.M - 1 ? fr + 32 : p.M - 1;
    const char* wp = (const char*)p.W + ((size_t)n * p.K + kbeg + kg * 32) * 2;
    const char* xa = (const char*)p.A + ((size_t)m_a * p.lda + kbeg + kg * 32) * 2;
    const char* xb = (const char*)p.A + ((size_t)m_b * p.lda + kbeg + kg * 32) * 2;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    for (int s0 = 0; s0 < steps; s0 += CH) {
        rq_u128 wf[CH][4], xf0[CH][4], xf1[CH][4];
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const int so = (s0 + s < steps ? s0 + s : steps - 1) * 128;          // past the end: reload the last step (not used)
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[s][i] = ld128(wp + so + i * 16);
        }
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const int so = (s0 + s < steps ? s0 + s : steps - 1) * 128;
#pragma unroll
            for (int i = 0; i < 4; ++i) { xf0[s][i] = ld128(xa + so + i * 16); xf1[s][i] = ld128(xb + so + i * 16); }
        }
        rq_sched_barrier();          // every load of the batch is in flight before the first MFMA waits for one (memory-level
                                     // parallelism, not instruction order, is what this kernel lives on)
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            if (s0 + s < steps) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc0 = rq_mfma_32x32x16_bf16(as_bf16x8(wf[s][i]), as_bf16x8(xf0[s][i]), acc0);
                    acc1 = rq_mfma_32x32x16_bf16(as_bf16x8(wf[s][i]), as_bf16x8(xf1[s][i]), acc1);
                }
            }
        }
    }
    // ---- partial tiles -> LDS [wave][m 64][n 32 (+4 pad)] fp32.  C/D map of the transposed tile: column = lane & 31 = m,
    // rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) = n: four groups of four consecutive n per lane
    constexpr int LDP = 36;
    float* sP = (float*)smem;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int nl = 8 * g + 4 * kg;
        *(f32x4*)(sP + ((wave * 64 + fr) * LDP + nl)) = (f32x4){acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]};
        *(f32x4*)(sP + ((wave * 64 + 32 + fr) * LDP + nl)) = (f32x4){acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]};
    }
    rq_syncthreads();
    const int m = tid >> 3, nq = (tid & 7) * 4;
    f32x4 v = *(const f32x4*)(sP + (m * LDP + nq));
#pragma unroll
    for (int w = 1; w < 8; ++w) v = v + *(const f32x4*)(sP + ((w * 64 + m) * LDP + nq));
    const int nn = n0 + nq;
    if (m >= p.M || nn >= p.N) return;
    const int epi = p.epi;
    const float* bias = p.bias;
    if (bias && p.bias_step) bias += (long)(*p.bias_step) * p.bias_stride;
    const bool full4 = nn + 3 < p.N && (p.ldo & 3) == 0;
    if (bias && epi != EPI_F32_PARTIAL) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (nn + e < p.N) v[e] += bias[nn + e];
    }
    if (epi == EPI_BF16_GELU) {
        float g[4] = {v[0], v[1], v[2], v[3]};
        rq_gelu4(g, p.gelu_v2);
        v = (f32x4){g[0], g[1], g[2], g[3]};
    }
    if (epi <= EPI_BF16_RESID) {
        bf16_t* o = (bf16_t*)p.out + (long)m * p.ldo + nn;
        if (epi == EPI_BF16_RESID) {
            for (int e = 0; e < 4 && nn + e < p.N; ++e) v[e] += bf16_to_f32(p.resid[(long)m * p.ldr + nn + e]);
        }
        if (full4) {
            struct __attribute__((aligned(8))) u64 { uint32_t a, b; } w2;
            w2.a = pack_bf16x2(v[0], v[1]);
            w2.b = pack_bf16x2(v[2], v[3]);
            *(u64*)o = w2;
        } else {
            for (int e = 0; e < 4 && nn + e < p.N; ++e) o[e] = f32_to_bf16(v[e]);
        }
    } else {
        float* o = (float*)p.out + (epi == EPI_F32_PARTIAL ? (long)blockIdx.z * p.M * p.ldo : 0) + (long)m * p.ldo + nn;
        if (full4) *(f32x4*)o = v;
        else for (int e = 0; e < 4 && nn + e < p.N; ++e) o[e] = v[e];
    }
}

// host-side launcher (gemm.hip)
int rq_gemm_launch(const GemmArgs& a, int bm, int bn, hipStream_t stream);
// picks (BM, BN, splitk) for a weight-streaming decode GEMM; returns splitk actually used via args
// glds (may be null): receives the number of LDS-DMA stages to use (0 = register-staged operands)
void rq_gemm_pick_tile(int M, int N, int K, bool allow_splitk, int* bm, int* bn, int* splitk, int* glds);


// -------------------------------------------------------------------------------------------------
// Weight-streaming GEMM for the small-batch decode steps (M <= 128 rows per m-tile: the per-GPU batches of 64 / 100 that
// SURVEY 8d names).  At these sizes a GEMM is one pass over W with almost no arithmetic, and what decides its time is how
// many bytes a workgroup keeps in flight and how long its dependent chain per K-tile is (scripts/micro/weight_stream.hip,
// profiles/r03_weight_stream.txt: the qkv weights stream in 9.6 us with 2 K-tiles of 64 rows in flight per workgroup --
// what gemm_bf16_kernel<64, 64> does and takes -- in 5.3 us with 4, in 4.1 us with 32-row tiles and 8; the layout of W in
// memory makes no difference).  The tiled kernels add a workgroup barrier, an LDS round trip and a chain of four dependent
// MFMAs per K-tile on top (~600 cycles x 24 K-tiles); the round-2 skinny kernel avoided LDS but fetched the activations in
// fragment layout straight from L2 (32 scattered 32-byte pieces per instruction) and lost at 64 rows.
// Here: a workgroup owns 32 weight rows and BM activation rows; its FOUR wavefronts each take every fourth K-tile and run
// WITHOUT workgroup barriers -- each wavefront has a private ring of NS LDS slots { A tile [BM][64], W tile [32][64] }
// filled by its own LDS-DMA instructions (coalesced 1-KB bursts, no staging registers), waits with counted vmcnt, reads
// its fragments, refills the slot and runs 4 MFMAs per 32-row block (independent accumulator chains per block).  12 slots of
// 12 KB (BM = 64: 144 KB in flight per workgroup) or 8 of 20 KB (BM = 128).  The four partial tiles are summed through LDS
// in wavefront order (deterministic) by all 256 threads, which apply the epilogue and write row-contiguous 16 / 32-byte
// pieces.  blockIdx.z = split-K slice (fp32 partial slabs, residual-producing GEMMs).  The arithmetic of an output element
// (K-tiles of a slice dealt round-robin to four accumulators, MFMA order inside a tile, reduction order) does not depend on
// BM, so the 64- and 128-row forms agree bit for bit.
#ifndef RQ_STREAM_W_NT       // weight DMAs of gemm_stream_kernel with the non-temporal policy (A/B: profiles/r05_stream_nt_ab.txt)
#define RQ_STREAM_W_NT 0
#endif
#ifdef RQ_STREAM_TRACE
// Diagnostics build only (scripts/stream_trace.py): constant-clock (100 MHz) stamps of every workgroup's phases -- entry, first
// DMA burst issued, first K-tile landed, main loop done, after the barrier, partial tiles in LDS, stores issued, stores
// acknowledged, and (slots 8 ..) tile landed / MFMAs issued for the first eight K-tiles of wavefronts 0 and 3
static __device__ unsigned long long g_stream_trace[1024 * 2 * 24];
#define RQ_ST(slot) do { if ((threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 3) && blockIdx.y == 0) { \
        const int wg_ = blockIdx.x + gridDim.x * blockIdx.z;                                                                          \
        if (wg_ < 1024) g_stream_trace[(wg_ * 2 + ((threadIdx.x >> 6) == 3)) * 24 + (slot)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define RQ_ST(slot) do { } while (0)
#endif
template <int BM, int BN = 32>
__global__ __launch_bounds__(256) void gemm_stream_kernel(GemmArgs p) {
    RQ_ST(0);
    // BN = 64 (BM = 64 only, round 4): for GEMMs whose 32-row tiles would not fit one round of the 256 CUs -- fc1 of the E = 2560
    // models (N = 10240: 320 workgroups, 23 us against ~12 for one round) and their fc2 (K = 10240: two K slices of 80 tiles each)
    // -- a workgroup owns 64 weight rows: half the workgroups, each reading the activation rows once for twice the columns.  The
    // arithmetic of an output element is the same as with 32-row tiles (same K-tile -> wavefront assignment, same reduction order).
    constexpr int BK = 64, NW = 4, NB = BN / 32;
    static_assert(BN == 32 || (BN == 64 && BM == 64), "tile shapes: BM x 32, or 64 x 64");
    constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, SLOT = A_BYTES + W_BYTES;
#ifdef RQ_STREAM_NS
    constexpr int NS = RQ_STREAM_NS;                   // (diagnostics: ring-depth sensitivity; 2 == 3, profiles/r04_stream_trace_ring2.txt)
#else
    constexpr int NS = (BM == 64 && BN == 32) ? 3 : 2; // slots per wavefront
#endif
    constexpr int A_G = BM / 8, W_G = BN / 8, PER = A_G + W_G;      // 1-KB (8-row) DMA groups per K-tile
    constexpr int MI = BM / 32;
    constexpr int RS = BN + 1;                         // row stride (floats) of a partial tile in LDS
    static_assert(NW * NS * SLOT <= 160 * 1024 && NW * BM * RS * 4 <= NW * NS * SLOT, "LDS budget");
    static_assert(2 * PER <= 63, "vmcnt range");
    RQ_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = rq_uniform(tid >> 6);
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int kt_total = p.K / BK;
    const int per = (kt_total + p.splitk - 1) / p.splitk;
    const int kt0 = blockIdx.z * per;
    const int kt1 = (kt0 + per < kt_total) ? kt0 + per : kt_total;
    const int nk = kt1 - kt0;
    const int n_mine = nk > wave ? (nk - wave + NW - 1) / NW : 0;     // this wavefront's K-tiles: kt0 + wave + 4 i

    const char* gA = (const char*)p.A;
    const char* gW = (const char*)p.W;
    const rq_lds_t lds_w = rq_lds_addr(smem) + (rq_lds_t)(wave * NS * SLOT);
    // DMA: lane (lr, lc) of 8-row group g fills LDS position (row 8 g + lr, chunk lc); the 16-byte-chunk swizzle
    // chunk ^ ((row >> 1) & 7) of the fragment reads is applied to the per-lane SOURCE address
    const int lr = lane >> 3, lc = lane & 7;
    unsigned ga[A_G], gb[W_G];
#pragma unroll
    for (int g = 0; g < A_G; ++g) {
        const int row = 8 * g + lr;
        int m = m0 + row;
        m = m < p.M - 1 ? m : p.M - 1;
        ga[g] = ((unsigned)m * (unsigned)p.lda + (unsigned)((lc ^ ((row >> 1) & 7)) << 3)) * 2u;
    }
#pragma unroll
    for (int g = 0; g < W_G; ++g) {
        const int row = 8 * g + lr;
        int n = n0 + row;
        n = n < p.N - 1 ? n : p.N - 1;
        gb[g] = ((unsigned)n * (unsigned)p.K + (unsigned)((lc ^ ((row >> 1) & 7)) << 3)) * 2u;
    }
    auto issue = [&](int i, int slot) {
        const unsigned kb = (unsigned)(kt0 + wave + i * NW) * (BK * 2);
        const rq_lds_t base = lds_w + (rq_lds_t)(slot * SLOT);
#pragma unroll
        for (int g = 0; g < A_G; ++g) rq_glds16(base + (rq_lds_t)(g * 1024), gA + (ga[g] + kb));
#pragma unroll
        for (int g = 0; g < W_G; ++g) {
#if RQ_STREAM_W_NT
            rq_glds16_nt(base + (rq_lds_t)(A_BYTES + g * 1024), gW + (gb[g] + kb));
#else
            rq_glds16(base + (rq_lds_t)(A_BYTES + g * 1024), gW + (gb[g] + kb));
#endif
        }
    };
    // fragment reads: row = 32 i + (lane & 31), chunk 2 ks + (lane >> 5), swizzled
    const int frow = lane & 31, fk = lane >> 5;
    unsigned rd[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rd[ks] = (unsigned)(frow * (BK * 2) + (((ks * 2 + fk) ^ ((frow >> 1) & 7)) << 4));

    f32x16 acc[MI][NB];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int s0 = 0; s0 < NS; ++s0)
        if (s0 < n_mine) issue(s0, s0);
    RQ_ST(1);
    // what the epilogue needs from global memory (bias, and the fp32 residual rows of the in-place update) is requested now, under the
    // streaming loop: fetched after it, each was a dependent L2 / HBM round trip (~1.5 us) at the very end of a ~7-us kernel
    const float* bias = p.bias;
    if (bias && p.bias_step) bias += (long)(*p.bias_step) * p.bias_stride;
    const int epi = p.epi;
    const bool accum = epi == EPI_F32_PARTIAL && p.accum;
    if (epi == EPI_F32_PARTIAL && !accum) bias = nullptr;
    const int c0 = (tid & 3) * 8;                      // this thread's 8 columns inside every 32-column block
    float bv[NB][8], xr[BM / 64][NB][8];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
        const int n = n0 + 32 * cb + c0;
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[cb][e] = (bias && n + e < p.N) ? bias[n + e] : 0.f;
#pragma unroll
        for (int rr = 0; rr < BM / 64; ++rr) {
            const int m = m0 + (tid >> 2) + 64 * rr;
#pragma unroll
            for (int e = 0; e < 8; ++e) xr[rr][cb][e] = (accum && m < p.M && n + e < p.N) ? ((const float*)p.out)[(long)m * p.ldo + n + e] : 0.f;
        }
    }
    int slot = 0;
    for (int i = 0; i < n_mine; ++i) {
        const int newer = n_mine - 1 - i;            // tiles issued after tile i that may still be in flight (at most NS - 1)
        if (NS >= 3 && newer >= 2) rq_wait_vmcnt<2 * PER>();
        else if (newer >= 1) rq_wait_vmcnt<PER>();
        else rq_wait_vmcnt<0>();
        rq_wave_sync();                              // every lane's share of the tile has landed
        if (i == 0) RQ_ST(2);
        if (i < 8) RQ_ST(8 + 2 * i);
        const char* sb = (const char*)smem + (wave * NS + slot) * SLOT;
        bf16x8 af[MI][4], bfr[NB][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) af[mi][ks] = as_bf16x8(ld128(sb + rd[ks] + mi * (32 * BK * 2)));
#pragma unroll
            for (int j = 0; j < NB; ++j) bfr[j][ks] = as_bf16x8(ld128(sb + A_BYTES + rd[ks] + j * (32 * BK * 2)));
        }
        rq_wait_lgkmcnt<0>();                        // the fragments are in registers: the slot may be refilled
        rq_wave_sync();
        rq_sched_barrier();
        if (i + NS < n_mine) issue(i + NS, slot);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[mi][j] = rq_mfma_32x32x16_bf16(af[mi][ks], bfr[j][ks], acc[mi][j]);
#ifdef RQ_STREAM_TRACE
        if (i < 8) { rq_opaque_acc(acc[0][0]); RQ_ST(9 + 2 * i); }
#endif
        slot = slot + 1 == NS ? 0 : slot + 1;
    }

    // ---- cross-wavefront reduction (wavefront order) + epilogue
    RQ_ST(3);
    rq_syncthreads();                                // every wavefront is done with its ring
    RQ_ST(4);
    float* sRed = (float*)smem;                      // [NW][BM][RS]
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * fk;
                sRed[(wave * BM + row) * RS + 32 * j + frow] = acc[mi][j][r];
            }
    rq_syncthreads();
    RQ_ST(5);
#pragma unroll
    for (int rr = 0; rr < BM / 64; ++rr) {
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            const int row = (tid >> 2) + 64 * rr, cc = 32 * cb + c0;
            const int m = m0 + row, n = n0 + cc;
            if (m >= p.M || n >= p.N) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float a = sRed[row * RS + cc + e];
#pragma unroll
                for (int w = 1; w < NW; ++w) a += sRed[(w * BM + row) * RS + cc + e];
                v[e] = a;
            }
            const bool full = n + 7 < p.N;
            if (epi <= EPI_BF16_RESID) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bv[cb][e];
                if (epi == EPI_BF16_GELU) {
                    float lo[4] = {v[0], v[1], v[2], v[3]}, hi[4] = {v[4], v[5], v[6], v[7]};
                    rq_gelu4(lo, p.gelu_v2);
                    rq_gelu4(hi, p.gelu_v2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
                }
                if (epi == EPI_BF16_RESID)
                    for (int e = 0; e < 8 && n + e < p.N; ++e) v[e] += bf16_to_f32(p.resid[(long)m * p.ldr + n + e]);
                bf16_t* o = (bf16_t*)p.out + (long)m * p.ldo + n;
                if (full && (p.ldo & 7) == 0) {
                    rq_u128 u;
                    u.x = pack_bf16x2(v[0], v[1]); u.y = pack_bf16x2(v[2], v[3]); u.z = pack_bf16x2(v[4], v[5]); u.w = pack_bf16x2(v[6], v[7]);
                    st128(o, u);
                } else {
                    for (int e = 0; e < 8 && n + e < p.N; ++e) o[e] = f32_to_bf16(v[e]);
                }
            } else {
                float* o = (float*)p.out + ((epi == EPI_F32_PARTIAL && !accum) ? (long)blockIdx.z * p.M * p.ldo : 0) + (long)m * p.ldo + n;
                if (full && (p.ldo & 3) == 0) {
                    f32x4 lo, hi;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        lo[e] = accum ? (xr[rr][cb][e] + v[e]) + bv[cb][e] : v[e] + bv[cb][e];
                        hi[e] = accum ? (xr[rr][cb][4 + e] + v[4 + e]) + bv[cb][4 + e] : v[4 + e] + bv[cb][4 + e];
                    }
                    *(f32x4*)o = lo;
                    *(f32x4*)(o + 4) = hi;
                } else {
                    for (int e = 0; e < 8 && n + e < p.N; ++e) o[e] = accum ? (xr[rr][cb][e] + v[e]) + bv[cb][e] : v[e] + bv[cb][e];
                }
            }
        }
    }
    RQ_ST(6);
#ifdef RQ_STREAM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stores have been acknowledged
    RQ_ST(7);
#endif
}

