// rqt_kernels.hip -- see rqt_kernels.h for the reference call sites of each kernel.
#include "rqt_kernels.h"
#include "rq_common.h"

// =================================================================================================
// residual add (+ split-K slab reduce + bias) fused with LayerNorm
// NS = number of split-K slabs (compile time: all slab loads are issued back to back, the kernel is
// latency-bound otherwise); each thread owns float4 columns tid, tid+256, ... of its row.
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
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + 256 * i;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) v[i] += t[sl][i];
        if (q < E4) {
            if (p.bias) v[i] += *(const f32x4*)(p.bias + q * 4);
            if (p.addvec) v[i] += *(const f32x4*)(p.addvec + q * 4);
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
            const f32x4 g = *(const f32x4*)(p.gamma + q * 4), b = *(const f32x4*)(p.beta + q * 4);
            const uint32_t lo = pack_bf16x2((v[i][0] - mean) * rstd * g[0] + b[0], (v[i][1] - mean) * rstd * g[1] + b[1]);
            const uint32_t hi = pack_bf16x2((v[i][2] - mean) * rstd * g[2] + b[2], (v[i][3] - mean) * rstd * g[3] + b[3]);
            *(uint32_t*)(p.y + base + q * 4) = lo;
            *(uint32_t*)(p.y + base + q * 4 + 2) = hi;
        }
    }
}

int rq_launch_resid_ln(const ResidLnArgs& a, hipStream_t s) {
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
// KV-cache decode attention: one wavefront per (row, head), head_dim 64, up to NB*64 keys
static __device__ __forceinline__ void unpack8(rq_u128 u, float* f) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}

template <int NB>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnDecodeArgs p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long pair = (long)blockIdx.x * 4 + wave;
    if (pair >= (long)p.rows * p.nh) return;                       // whole wave exits together
    const int b = (int)(pair / p.nh), h = (int)(pair - (long)b * p.nh);
    const int t = (p.step ? *p.step : 0) + p.step_off;
    const int E = p.E, Tcap = p.Tcap;
    const bf16_t* qrow = p.qkv + (long)b * 3 * E + h * 64;
    const bf16_t* krow = qrow + E;
    const bf16_t* vrow = qrow + 2 * E;
    bf16_t* kc = p.kc + pair * 8 * Tcap * 8;
    bf16_t* vc = p.vc + pair * Tcap * 64;

    // append this token's k / v (the current key is always read back from qkv, never from the cache)
    if (lane < 8) st128(kc + ((long)lane * Tcap + t) * 8, ld128(krow + lane * 8));
    else if (lane < 16) st128(vc + (long)t * 64 + (lane - 8) * 8, ld128(vrow + (lane - 8) * 8));

    float q[64];
#pragma unroll
    for (int c = 0; c < 8; ++c) unpack8(ld128(qrow + c * 8), q + c * 8);

    float sc[NB];
    float mx = -__int_as_float(0x7f800000);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int j = nb * 64 + lane;
        float s = -__int_as_float(0x7f800000);
        if (j <= t) {
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const bf16_t* src = (j == t) ? (krow + c * 8) : (kc + ((long)c * Tcap + j) * 8);
                float kf[8];
                unpack8(ld128(src), kf);
#pragma unroll
                for (int e = 0; e < 8; ++e) dot = fmaf(q[c * 8 + e], kf[e], dot);
            }
            s = dot * 0.125f;                                       // 1/sqrt(64), attentions.py:87
        }
        sc[nb] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int j = nb * 64 + lane;
        sc[nb] = (j <= t) ? expf(sc[nb] - mx) : 0.f;
        sum += sc[nb];
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;

    const int cc = lane & 7, g = lane >> 3;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int jl = jj * 8 + g;
            const int j = nb * 64 + jl;
            const float pj = rq_shfl(sc[nb], jl) * inv;
            if (j <= t) {
                const bf16_t* src = (j == t) ? (vrow + cc * 8) : (vc + (long)j * 64 + cc * 8);
                float vf[8];
                unpack8(ld128(src), vf);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, vf[e], acc[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        acc[e] += rq_shfl_xor(acc[e], 8);
        acc[e] += rq_shfl_xor(acc[e], 16);
        acc[e] += rq_shfl_xor(acc[e], 32);
    }
    if (g == 0) {
        rq_u128 o;
        o.x = pack_bf16x2(acc[0], acc[1]); o.y = pack_bf16x2(acc[2], acc[3]);
        o.z = pack_bf16x2(acc[4], acc[5]); o.w = pack_bf16x2(acc[6], acc[7]);
        st128(p.y + (long)b * E + h * 64 + cc * 8, o);
    }
}

int rq_launch_attn_decode(const AttnDecodeArgs& a, hipStream_t s) {
    if (a.E != a.nh * 64) return rq_fail(RQAMD_ERR_UNSUPPORTED, "attention: head_dim must be 64 (E=%d, n_head=%d)", a.E, a.nh);
    const long pairs = (long)a.rows * a.nh;
    dim3 grid((unsigned)((pairs + 3) / 4));
    if (a.Tcap <= 64) RQ_LAUNCH(attn_decode_kernel<1>, grid, dim3(256), 0, s, a);
    else if (a.Tcap <= 128) RQ_LAUNCH(attn_decode_kernel<2>, grid, dim3(256), 0, s, a);
    else if (a.Tcap <= 256) RQ_LAUNCH(attn_decode_kernel<4>, grid, dim3(256), 0, s, a);
    else return rq_fail(RQAMD_ERR_UNSUPPORTED, "attention: context %d > 256", a.Tcap);
    return rq_check_launch("attn_decode_kernel");
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
    for (int d = 0; d < p.n_depth; ++d) {
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
    if (p.top_p >= 0.f) {
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
                const float u = ((float)(r[e] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
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

int rq_launch_sample(const SampleArgs& a, hipStream_t s) {
    if (a.V < 1 || a.V > 36000) return rq_fail(RQAMD_ERR_UNSUPPORTED, "sampler: vocab %d not in 1..36000", a.V);
    if (!(a.temperature > 0.f)) return rq_fail(RQAMD_ERR_INVALID, "sampler: temperature must be > 0");
    const size_t smem = (size_t)a.V * 4 + 16 * 4 + 16 * 4 + 256 * 4 + 4 * 4 + 32 * 4;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    RQ_LAUNCH(sample_kernel, dim3(a.rows), dim3(SMP_T), smem, s, a);
    return rq_check_launch("sample_kernel");
}

extern "C" int rqamd_sample_logits(const float* logits, int rows, int vocab, float temperature, int top_k, float top_p,
                                   uint64_t seed, uint64_t offset, int64_t* samples_out, float* probs_out, void* stream) {
    if (!logits || rows < 0) return rq_fail(RQAMD_ERR_INVALID, "sample_logits: bad argument");
    if (rows == 0) return RQAMD_OK;
    SampleArgs a{};
    a.logits = logits; a.rows = rows; a.V = vocab; a.temperature = temperature; a.top_k = top_k; a.top_p = top_p;
    a.seed = seed; a.offset = offset; a.out = samples_out; a.out_stride = 1; a.probs_out = probs_out; a.D = 1;
    return rq_launch_sample(a, (hipStream_t)stream);
}
