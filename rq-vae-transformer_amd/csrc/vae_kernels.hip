// vae_kernels.hip -- non-GEMM kernels of the RQ-VAE encoder/decoder (gfx950).  Activations are
// NHWC bf16 between layers (fp32 accumulation everywhere), so channel vectors are contiguous 16-B
// chunks and every kernel below moves whole chunks.
//
// Reference call sites (rqvae/models/rqvae/):
//   gn_stats / gn_apply  <- Normalize = GroupNorm(32, C, eps=1e-6) + nonlinearity = SiLU (layers.py:11-17)
//   vae_attn_kernel      <- AttnBlock.forward (layers.py:158-182): softmax(q k^T * C^-0.5) v, single head
//   conv_in3_kernel      <- Encoder.conv_in (modules.py:23-27): 3 -> ch, reads the NCHW fp32 image
//   conv_out3_kernel     <- Decoder.conv_out (modules.py:165-169): ch -> 3, writes the NCHW fp32 image
//   repack_conv_kernel   <- nn.Conv2d weight (O,I,kh,kw) fp32 -> [O][kh][kw][I] bf16 (GEMM "W[N][K]")
#include "rq_common.h"
#include "vae_kernels.h"

static __device__ __forceinline__ void unpack8v(rq_u128 u, float* f) {
    rq_unpack2(u.x, f[0], f[1]);
    rq_unpack2(u.y, f[2], f[3]);
    rq_unpack2(u.z, f[4], f[5]);
    rq_unpack2(u.w, f[6], f[7]);
}

// -------------------------------------------------------------------------------------------------
// GroupNorm statistics: partial (sum, sumsq) per (image, pixel-chunk, group); deterministic order.
// grid (nchunk, B), 256 threads; thread t owns channel chunk cc = t % (C/8) and pixels prow, prow+PR, ...
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16_t* x, float* part, int HW, int C, int nchunk) {
    __shared__ float sp[256][8];
    const int tid = threadIdx.x, CC = C / 8, PR = 256 / CC;
    const int cc = tid % CC, prow = tid / CC;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int per = (HW + nchunk - 1) / nchunk;
    const int p0 = chunk * per, p1 = (p0 + per < HW) ? p0 + per : HW;
    const int gs = C / 32;                      // channels per group: 2, 4, 8, 16, ...
    float s[8], ss[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
    if (prow < PR)
        for (int px = p0 + prow; px < p1; px += PR) {
            float f[8];
            unpack8v(ld128(x + ((long)b * HW + px) * C + cc * 8), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += f[e]; ss[e] = fmaf(f[e], f[e], ss[e]); }
        }
    // fold the 8 channels into this thread's groups: npt = max(1, 8/gs) (sum, sumsq) pairs
    const int sub = gs >= 8 ? 8 : gs;           // channels per pair
    const int npt = 8 / sub;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float a = 0.f, q = 0.f;
        if (k < npt)
            for (int e = 0; e < sub; ++e) { a += s[k * sub + e]; q += ss[k * sub + e]; }
        sp[tid][2 * k] = a;
        sp[tid][2 * k + 1] = q;
    }
    rq_syncthreads();
    if (tid < 32) {
        const int g = tid;
        float a = 0.f, q = 0.f;
        if (gs >= 8) {
            const int c0 = g * gs / 8, c1 = c0 + gs / 8;      // channel chunks of this group
            for (int r = 0; r < PR; ++r)
                for (int c = c0; c < c1; ++c) { a += sp[r * CC + c][0]; q += sp[r * CC + c][1]; }
        } else {
            const int c = g * gs / 8, k = (g * gs % 8) / gs;
            for (int r = 0; r < PR; ++r) { a += sp[r * CC + c][2 * k]; q += sp[r * CC + c][2 * k + 1]; }
        }
        float* o = part + (((long)b * nchunk + chunk) * 32 + g) * 2;
        o[0] = a;
        o[1] = q;
    }
}

// apply: y = (x - mean) * rstd * gamma + beta, optional SiLU; grid (nblk, B)
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* x, const float* part, const float* gamma, const float* beta,
                                                       bf16_t* y, int HW, int C, int nchunk, float eps, int silu) {
    __shared__ float smean[32], srstd[32];
    const int tid = threadIdx.x, b = blockIdx.y;
    if (tid < 32) {
        float a = 0.f, q = 0.f;
        for (int c = 0; c < nchunk; ++c) {
            const float* o = part + (((long)b * nchunk + c) * 32 + tid) * 2;
            a += o[0];
            q += o[1];
        }
        const float n = (float)HW * (float)(C / 32);
        const float mean = a / n;
        float var = q / n - mean * mean;
        if (var < 0.f) var = 0.f;
        smean[tid] = mean;
        srstd[tid] = 1.0f / sqrtf(var + eps);
    }
    rq_syncthreads();
    const int CC = C / 8, gs = C / 32;
    const long nvec = (long)HW * CC;
    // the grid stride (gridDim.x*256) is a multiple of CC, so a thread always sees the same 8 channels:
    // fold mean / rstd / gamma / beta into one scale and shift per channel, once
    const int cc = tid % CC;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = cc * 8 + e, g = ch / gs;
        sc[e] = srstd[g] * gamma[ch];
        sh[e] = beta[ch] - smean[g] * sc[e];
    }
    for (long i = (long)blockIdx.x * 256 + tid; i < nvec; i += (long)gridDim.x * 256) {
        const long off = ((long)b * HW) * C + i * 8;
        float f[8];
        unpack8v(ld128(x + off), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = fmaf(f[e], sc[e], sh[e]);
            if (silu) v = v / (1.0f + __expf(-v));
            f[e] = v;
        }
        rq_u128 o;
        o.x = pack_bf16x2(f[0], f[1]); o.y = pack_bf16x2(f[2], f[3]); o.z = pack_bf16x2(f[4], f[5]); o.w = pack_bf16x2(f[6], f[7]);
        st128(y + off, o);
    }
}

int rq_launch_gn_stats(const bf16_t* x, float* part, int B, int HW, int C, int* nchunk_out, hipStream_t s) {
    if (C % 32 != 0 || C % 8 != 0 || 256 % (C / 8) != 0 || C > 2048)
        return rq_fail(RQAMD_ERR_UNSUPPORTED, "groupnorm: channels %d unsupported (need 64,128,256,512,1024,2048)", C);
    int nchunk = HW / 64;
    if (nchunk < 1) nchunk = 1;
    if (nchunk > RQ_GN_MAX_CHUNK) nchunk = RQ_GN_MAX_CHUNK;
    RQ_LAUNCH(gn_stats_kernel, dim3(nchunk, B), dim3(256), 0, s, x, part, HW, C, nchunk);
    *nchunk_out = nchunk;
    return rq_check_launch("gn_stats_kernel");
}

int rq_launch_groupnorm(const bf16_t* x, bf16_t* y, float* part, const float* gamma, const float* beta, int B, int HW, int C,
                        int silu, hipStream_t s) {
    int nchunk = 0;
    RQ_TRY(rq_launch_gn_stats(x, part, B, HW, C, &nchunk, s));
    long nvec = (long)HW * C / 8;
    int nblk = (int)((nvec + 1023) / 1024);
    if (nblk < 1) nblk = 1;
    if (nblk > 256) nblk = 256;
    RQ_LAUNCH(gn_apply_kernel, dim3(nblk, B), dim3(256), 0, s, x, part, gamma, beta, y, HW, C, nchunk, 1e-6f, silu);
    return rq_check_launch("gn_apply_kernel");
}

// -------------------------------------------------------------------------------------------------
// single-head spatial attention: one wavefront per query token.  qkv [B*T][3C] bf16 -> out [B*T][C]
template <int NB>   // key blocks of 64: T <= 64*NB
__global__ __launch_bounds__(256) void vae_attn_kernel(const bf16_t* qkv, bf16_t* out, int B, int T, int C, float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long qi = (long)blockIdx.x * 4 + wave;        // global query index
    if (qi >= (long)B * T) return;
    const long b = qi / T;
    const bf16_t* base = qkv + b * T * 3 * C;
    const bf16_t* q = qkv + qi * 3 * C;
    float sc[NB];
    float mx = -__int_as_float(0x7f800000);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int j = nb * 64 + lane;
        float s = -__int_as_float(0x7f800000);
        if (j < T) {
            const bf16_t* k = base + (long)j * 3 * C + C;
            float dot = 0.f;
            for (int c = 0; c < C; c += 8) {
                float qf[8], kf[8];
                unpack8v(ld128(q + c), qf);
                unpack8v(ld128(k + c), kf);
#pragma unroll
                for (int e = 0; e < 8; ++e) dot = fmaf(qf[e], kf[e], dot);
            }
            s = dot * scale;
        }
        sc[nb] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int j = nb * 64 + lane;
        sc[nb] = (j < T) ? expf(sc[nb] - mx) : 0.f;
        sum += sc[nb];
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    // out[c] = sum_j p_j v_j[c]; lane owns channel chunks lane, lane+64, ... (8 channels each)
    const int nch = C / 8;
    for (int ch0 = 0; ch0 < nch; ch0 += 64) {
        const int ch = ch0 + lane;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            for (int jl = 0; jl < 64; ++jl) {
                const int j = nb * 64 + jl;
                const float pj = rq_shfl(sc[nb], jl) * inv;
                if (j < T && ch < nch) {
                    float vf[8];
                    unpack8v(ld128(base + (long)j * 3 * C + 2 * C + ch * 8), vf);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, vf[e], acc[e]);
                }
            }
        }
        if (ch < nch) {
            rq_u128 o;
            o.x = pack_bf16x2(acc[0], acc[1]); o.y = pack_bf16x2(acc[2], acc[3]);
            o.z = pack_bf16x2(acc[4], acc[5]); o.w = pack_bf16x2(acc[6], acc[7]);
            st128(out + qi * C + ch * 8, o);
        }
    }
}

// The same attention for T = 64 tokens (the 8 x 8 AttnBlocks of the released models) on the matrix pipe, one workgroup per image.
// The wavefront-per-query kernel above reads a different K row per lane (64 cache lines per load instruction: the CU's address
// path, not arithmetic, set its 119 us per 128 images, profiles/r02_decode_timeline_b128.txt).  Here:
//   1. S = Q K^T: wavefront w owns the 32 x 32 block (w >> 1, w & 1); both operands are K-contiguous rows of qkv, loaded from
//      global memory in MFMA fragment layout (16 bytes per lane), fp32 scores * C^-0.5 into LDS;
//   2. softmax per row in fp32 (4 threads per row), P rounded to bf16 into LDS;
//   3. O^T = V^T P^T: A = V^T gathered out of a row-major LDS copy of V (8 two-byte reads per fragment; the row pitch C + 4 puts
//      the two 8-row groups of a read on disjoint banks), B = P rows; D[channel][query] leaves 4 consecutive channels per lane:
//      8-byte stores.  V's staging loads are issued first and land under steps 1-2.
constexpr int VA_T = 64;
__global__ __launch_bounds__(256) void vae_attn_mfma_kernel(const bf16_t* qkv, bf16_t* out, int C, float scale) {
    RQ_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = rq_uniform(tid >> 6);
    const int l31 = lane & 31, kg = lane >> 5;
    const int VP = C + 4;                                  // V row pitch (elements)
    bf16_t* sV = (bf16_t*)smem;                            // [64][VP]
    float* sS = (float*)(smem + (size_t)VA_T * VP * 2);    // [64][65]
    bf16_t* sP = (bf16_t*)(sS + VA_T * 65);                // [64][72]
    const bf16_t* base = qkv + (long)blockIdx.x * VA_T * 3 * C;

    // ---- V -> LDS (row-major, 8-byte stores: the pitch is not a multiple of 16 bytes); at most 16 chunks per thread
    const int CC = C >> 3, n_chunk = VA_T * CC;
    rq_u128 vr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int id = tid + 256 * i;
        if (id < n_chunk) vr[i] = ld128(base + (long)(id / CC) * 3 * C + 2 * C + (id % CC) * 8);
    }
    // ---- 1. scores
    {
        const int mb = wave >> 1, nb = wave & 1, nks = C >> 4;
        const bf16_t* qp = base + (long)(mb * 32 + l31) * 3 * C + kg * 8;
        const bf16_t* kp = base + (long)(nb * 32 + l31) * 3 * C + C + kg * 8;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int k0 = 0; k0 < nks; k0 += 8) {
            rq_u128 a[8], b[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a[j] = zero128(); b[j] = zero128();
                if (k0 + j < nks) { a[j] = ld128(qp + (k0 + j) * 16); b[j] = ld128(kp + (k0 + j) * 16); }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (k0 + j < nks) acc = rq_mfma_32x32x16_bf16(as_bf16x8(a[j]), as_bf16x8(b[j]), acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sS[(mb * 32 + (r >> 2) * 8 + kg * 4 + (r & 3)) * 65 + nb * 32 + l31] = acc[r] * scale;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int id = tid + 256 * i;
        if (id < n_chunk) {
            uint64_t* d = (uint64_t*)(sV + (id / CC) * VP + (id % CC) * 8);
            d[0] = (uint64_t)vr[i].x | ((uint64_t)vr[i].y << 32);
            d[1] = (uint64_t)vr[i].z | ((uint64_t)vr[i].w << 32);
        }
    }
    rq_syncthreads();
    // ---- 2. softmax (layers.py:171-172): 4 threads per row, 16 keys each
    {
        const int row = tid >> 2, part = tid & 3;
        float v[16], mx = -__int_as_float(0x7f800000);
#pragma unroll
        for (int i = 0; i < 16; ++i) { v[i] = sS[row * 65 + part * 16 + i]; mx = fmaxf(mx, v[i]); }
        mx = fmaxf(mx, rq_shfl_xor(mx, 1));
        mx = fmaxf(mx, rq_shfl_xor(mx, 2));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { v[i] = expf(v[i] - mx); sum += v[i]; }
        sum += rq_shfl_xor(sum, 1);
        sum += rq_shfl_xor(sum, 2);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int i = 0; i < 16; i += 2) *(uint32_t*)(sP + row * 72 + part * 16 + i) = pack_bf16x2(v[i] * inv, v[i + 1] * inv);
    }
    rq_syncthreads();
    // ---- 3. O^T = V^T P^T
    rq_u128 pf[2][4];                                      // B operand: P[q = qb*32 + l31][ks*16 + kg*8 ..]
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) pf[qb][ks] = ld128(sP + (qb * 32 + l31) * 72 + ks * 16 + kg * 8);
    for (int cb = wave; cb < (C >> 5); cb += 4) {
        f32x16 acc[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[qb][r] = 0.f;
        const bf16_t* vcol = sV + cb * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint32_t w4[4];
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const uint32_t lo = vcol[(ks * 16 + kg * 8 + e) * VP], hi = vcol[(ks * 16 + kg * 8 + e + 1) * VP];
                w4[e >> 1] = lo | (hi << 16);
            }
            rq_u128 af;
            af.x = w4[0]; af.y = w4[1]; af.z = w4[2]; af.w = w4[3];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) acc[qb] = rq_mfma_32x32x16_bf16(as_bf16x8(af), as_bf16x8(pf[qb][ks]), acc[qb]);
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            bf16_t* o = out + ((long)blockIdx.x * VA_T + qb * 32 + l31) * C + cb * 32 + kg * 4;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                *(uint64_t*)(o + rq * 8) = (uint64_t)pack_bf16x2(acc[qb][rq * 4], acc[qb][rq * 4 + 1]) | ((uint64_t)pack_bf16x2(acc[qb][rq * 4 + 2], acc[qb][rq * 4 + 3]) << 32);
        }
    }
}

int rq_launch_vae_attn(const bf16_t* qkv, bf16_t* out, int B, int T, int C, hipStream_t s) {
    if (C % 8 != 0) return rq_fail(RQAMD_ERR_UNSUPPORTED, "vae attention: C %d", C);
    const float scale = 1.0f / sqrtf((float)C);      // int(c)**(-0.5), layers.py:170
    static const bool no_mfma = getenv("RQAMD_VAE_ATTN_VALU") != nullptr;      // A/B switch
    if (T == VA_T && C % 64 == 0 && C <= 512 && !no_mfma) {
        const size_t smem = (size_t)VA_T * (C + 4) * 2 + (size_t)VA_T * 65 * 4 + (size_t)VA_T * 72 * 2;
        static RqDeviceOnce attr_once;
        if (attr_once.first()) (void)hipFuncSetAttribute((const void*)vae_attn_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        RQ_LAUNCH(vae_attn_mfma_kernel, dim3((unsigned)B), dim3(256), smem, s, qkv, out, C, scale);
        return rq_check_launch("vae_attn_mfma_kernel");
    }
    dim3 grid((unsigned)(((long)B * T + 3) / 4));
    if (T <= 64) RQ_LAUNCH(vae_attn_kernel<1>, grid, dim3(256), 0, s, qkv, out, B, T, C, scale);
    else if (T <= 256) RQ_LAUNCH(vae_attn_kernel<4>, grid, dim3(256), 0, s, qkv, out, B, T, C, scale);
    else if (T <= 1024) RQ_LAUNCH(vae_attn_kernel<16>, grid, dim3(256), 0, s, qkv, out, B, T, C, scale);
    else return rq_fail(RQAMD_ERR_UNSUPPORTED, "vae attention: %d tokens > 1024", T);
    return rq_check_launch("vae_attn_kernel");
}

// -------------------------------------------------------------------------------------------------
// Encoder.conv_in: x NCHW fp32 (Cin <= 4) -> y NHWC bf16 (Cout).  w: [ky][kx][ci][co] fp32.
__global__ __launch_bounds__(256) void conv_in3_kernel(const float* x, const float* w, const float* bias, bf16_t* y,
                                                       int B, int H, int W, int Cin, int Cout) {
    RQ_DYN_SMEM(smem);
    float* sw = (float*)smem;                      // [9*Cin][Cout]
    for (int i = threadIdx.x; i < 9 * Cin * Cout; i += 256) sw[i] = w[i];
    rq_syncthreads();
    const int CC = Cout / 8;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)B * H * W * CC) return;
    const int cc = (int)(gid % CC);
    const long pix = gid / CC;
    const int ox = (int)(pix % W), oy = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = bias[cc * 8 + e];
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy + ky - 1, ix = ox + kx - 1;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            for (int ci = 0; ci < Cin; ++ci) {
                const float v = x[(((long)b * Cin + ci) * H + iy) * W + ix];
                const float* wr = sw + ((ky * 3 + kx) * Cin + ci) * Cout + cc * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(v, wr[e], acc[e]);
            }
        }
    rq_u128 o;
    o.x = pack_bf16x2(acc[0], acc[1]); o.y = pack_bf16x2(acc[2], acc[3]); o.z = pack_bf16x2(acc[4], acc[5]); o.w = pack_bf16x2(acc[6], acc[7]);
    st128(y + pix * Cout + cc * 8, o);
}

int rq_launch_conv_in3(const float* x, const float* w, const float* bias, bf16_t* y, int B, int H, int W, int Cin, int Cout, hipStream_t s) {
    if (Cout % 8 != 0 || Cin > 4 || 9 * Cin * Cout * 4 > 60000) return rq_fail(RQAMD_ERR_UNSUPPORTED, "conv_in: Cin %d Cout %d", Cin, Cout);
    const long n = (long)B * H * W * (Cout / 8);
    RQ_LAUNCH(conv_in3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), (size_t)9 * Cin * Cout * 4, s, x, w, bias, y, B, H, W, Cin, Cout);
    return rq_check_launch("conv_in3_kernel");
}

// Decoder.conv_out: x NHWC bf16 (Cin) -> y NCHW fp32 (Cout <= 4).  w: [co][ky][kx][ci] fp32.
// 8 lanes per output pixel, each owning channel chunks l8, l8+8, ...; partials reduced by xor-shuffles.
__global__ __launch_bounds__(256) void conv_out3_kernel(const bf16_t* x, const float* w, const float* bias, float* y,
                                                        int B, int H, int W, int Cin, int Cout) {
    RQ_DYN_SMEM(smem);
    float* sw = (float*)smem;                      // [Cout][9][Cin]
    for (int i = threadIdx.x; i < Cout * 9 * Cin; i += 256) sw[i] = w[i];
    rq_syncthreads();
    const int l8 = threadIdx.x & 7;
    long pix = (long)blockIdx.x * 32 + (threadIdx.x >> 3);
    const bool ok = pix < (long)B * H * W;
    if (!ok) pix = 0;
    const int ox = (int)(pix % W), oy = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int nch = Cin / 8;
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy + ky - 1, ix = ox + kx - 1;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            const bf16_t* src = x + (((long)b * H + iy) * W + ix) * Cin;
            for (int ch = l8; ch < nch; ch += 8) {
                float f[8];
                unpack8v(ld128(src + ch * 8), f);
                for (int co = 0; co < Cout; ++co) {
                    const float* wr = sw + (co * 9 + ky * 3 + kx) * Cin + ch * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[co] = fmaf(f[e], wr[e], acc[co]);
                }
            }
        }
#pragma unroll
    for (int co = 0; co < 4; ++co) {
        acc[co] += rq_shfl_xor(acc[co], 1);
        acc[co] += rq_shfl_xor(acc[co], 2);
        acc[co] += rq_shfl_xor(acc[co], 4);
    }
    if (ok && l8 < Cout) y[(((long)b * Cout + l8) * H + oy) * W + ox] = acc[l8] + bias[l8];
}

int rq_launch_conv_out3(const bf16_t* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin, int Cout, hipStream_t s) {
    if (Cin % 8 != 0 || Cout > 4 || Cout * 9 * Cin * 4 > 60000) return rq_fail(RQAMD_ERR_UNSUPPORTED, "conv_out: Cin %d Cout %d", Cin, Cout);
    const long n = (long)B * H * W;
    RQ_LAUNCH(conv_out3_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), (size_t)Cout * 9 * Cin * 4, s, x, w, bias, y, B, H, W, Cin, Cout);
    return rq_check_launch("conv_out3_kernel");
}

// -------------------------------------------------------------------------------------------------
// weight repacks.  mode 0: (O,I,kh,kw) fp32 -> [O][kh][kw][I] bf16; mode 1: -> [kh][kw][I][O] fp32 (conv_in);
// mode 2: -> [O][kh][kw][I] fp32 (conv_out)
__global__ void repack_conv_kernel(const float* src, void* dst, int O, int I, int kh, int kw, int mode) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)O * I * kh * kw;
    if (gid >= n) return;
    // gid enumerates the source layout
    const int x = (int)(gid % kw), yk = (int)((gid / kw) % kh), i = (int)((gid / ((long)kw * kh)) % I), o = (int)(gid / ((long)kw * kh * I));
    const float v = src[gid];
    if (mode == 0) ((bf16_t*)dst)[(((long)o * kh + yk) * kw + x) * I + i] = f32_to_bf16(v);
    else if (mode == 1) ((float*)dst)[(((long)yk * kw + x) * I + i) * O + o] = v;
    else ((float*)dst)[(((long)o * kh + yk) * kw + x) * I + i] = v;
}

int rq_launch_repack_conv(const float* src, void* dst, int O, int I, int kh, int kw, int mode, hipStream_t s) {
    const long n = (long)O * I * kh * kw;
    RQ_LAUNCH(repack_conv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, O, I, kh, kw, mode);
    return rq_check_launch("repack_conv_kernel");
}


// =================================================================================================
// Small-batch mode (<= 8 images per call: the drivers' one-image-per-call decode and the rFID loop): a low-resolution
// conv (8x8 .. 32x32 pixels, K = 9 * Cin up to 4608) has only a handful of output tiles, so its K loop is divided over
// blockIdx.z into fp32 partial slabs and this kernel finishes it: slab sum (fixed order) + bias + residual, one rounding.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* slabs, int n_slabs, long MN, int N, const float* bias,
                                                            const bf16_t* resid, void* out, int out_f32) {
    const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= MN) return;
    f32x4 acc = *(const f32x4*)(slabs + i4);
    for (int z = 1; z < n_slabs; ++z) acc = acc + *(const f32x4*)(slabs + (long)z * MN + i4);
    const int n = (int)(i4 % N);
    const f32x4 bv = *(const f32x4*)(bias + n);
    acc = acc + bv;
    if (resid) {
        const uint32_t* rp = (const uint32_t*)(resid + i4);
        const uint32_t r0 = rp[0], r1 = rp[1];
        { float lo_, hi_; rq_unpack2(r0, lo_, hi_); acc[0] += lo_; acc[1] += hi_; }
        { float lo_, hi_; rq_unpack2(r1, lo_, hi_); acc[2] += lo_; acc[3] += hi_; }
    }
    if (out_f32) {
        *(f32x4*)((float*)out + i4) = acc;
    } else {
        struct __attribute__((aligned(8))) u64 { uint32_t a, b; } w;
        w.a = pack_bf16x2(acc[0], acc[1]);
        w.b = pack_bf16x2(acc[2], acc[3]);
        *(u64*)((bf16_t*)out + i4) = w;
    }
}

int rq_launch_splitk_reduce(const float* slabs, int n_slabs, int M, int N, const float* bias, const bf16_t* resid, void* out, int out_f32,
                            hipStream_t s) {
    if (N % 4 != 0) return rq_fail(RQAMD_ERR_UNSUPPORTED, "splitk_reduce: N %d %% 4 != 0", N);
    const long MN = (long)M * N;
    RQ_LAUNCH(splitk_reduce_kernel, dim3((unsigned)((MN / 4 + 255) / 256)), dim3(256), 0, s, slabs, n_slabs, MN, N, bias, resid, out, out_f32);
    return rq_check_launch("splitk_reduce_kernel");
}


// -------------------------------------------------------------------------------------------------
// resamp_with_conv = False (rqvae/models/rqvae/layers.py:20-57 of the reference; no released config): Upsample is the bare
// F.interpolate(scale_factor=2, mode='nearest'), Downsample is F.avg_pool2d(kernel_size=2, stride=2).  NHWC bf16, 16-byte pieces.
__global__ void upsample2_nearest_kernel(const rq_u128* x, rq_u128* y, long n_out, int Ho, int Wo, int C8) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_out) return;
    const int c = (int)(gid % C8);
    long pix = gid / C8;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const long b = pix / Ho;
    y[gid] = x[((b * (Ho >> 1) + (oy >> 1)) * (Wo >> 1) + (ox >> 1)) * C8 + c];
}
__global__ void avgpool2_kernel(const rq_u128* x, rq_u128* y, long n_out, int Ho, int Wo, int C8) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_out) return;
    const int c = (int)(gid % C8);
    long pix = gid / C8;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const long b = pix / Ho;
    const int Wi = Wo * 2;
    const rq_u128* s0 = x + ((b * (Ho * 2) + 2 * oy) * Wi + 2 * ox) * C8 + c;
    const rq_u128 p00 = s0[0], p01 = s0[C8], p10 = s0[(long)Wi * C8], p11 = s0[(long)Wi * C8 + C8];
    const uint32_t a[4] = {p00.x, p00.y, p00.z, p00.w}, bq[4] = {p01.x, p01.y, p01.z, p01.w};
    const uint32_t cq[4] = {p10.x, p10.y, p10.z, p10.w}, d[4] = {p11.x, p11.y, p11.z, p11.w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float a0, a1, b0, b1, c0, c1, d0, d1;
        rq_unpack2(a[e], a0, a1); rq_unpack2(bq[e], b0, b1); rq_unpack2(cq[e], c0, c1); rq_unpack2(d[e], d0, d1);
        // window order of avg_pool2d: row 0 (left, right), then row 1; fp32 sum, one rounding
        o[e] = pack_bf16x2((((a0 + b0) + c0) + d0) * 0.25f, (((a1 + b1) + c1) + d1) * 0.25f);
    }
    rq_u128 r;
    r.x = o[0]; r.y = o[1]; r.z = o[2]; r.w = o[3];
    y[gid] = r;
}
int rq_launch_upsample2(const bf16_t* x, bf16_t* y, int B, int Ho, int Wo, int C, hipStream_t s) {
    if (C % 8 || (Ho & 1) || (Wo & 1)) return rq_fail(RQAMD_ERR_UNSUPPORTED, "upsample2: C %% 8 and even output sizes needed");
    const long n = (long)B * Ho * Wo * (C / 8);
    RQ_LAUNCH(upsample2_nearest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const rq_u128*)x, (rq_u128*)y, n, Ho, Wo, C / 8);
    return rq_check_launch("upsample2_nearest_kernel");
}
int rq_launch_avgpool2(const bf16_t* x, bf16_t* y, int B, int Ho, int Wo, int C, hipStream_t s) {
    if (C % 8) return rq_fail(RQAMD_ERR_UNSUPPORTED, "avgpool2: C %% 8 needed");
    const long n = (long)B * Ho * Wo * (C / 8);
    RQ_LAUNCH(avgpool2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const rq_u128*)x, (rq_u128*)y, n, Ho, Wo, C / 8);
    return rq_check_launch("avgpool2_kernel");
}
