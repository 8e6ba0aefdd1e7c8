// gemm_ks.h -- decode-step GEMM for 1 .. 512 rows: K split over the wavefronts of a workgroup, activations through private
// LDS-DMA rings, weights straight into registers from a fragment-packed copy (gfx950).
//
//   C[M,N] = A[M,K] . W[N,K]^T  (+bias, +GELU | fp32 | split-K partial slabs | fp32 residual stream updated in place)
//
// Stands where gemm.h's kernels stand -- nn.Linear of MultiSelfAttention / AttentionBlock.mlp in a cached decoding step
// (rqvae/models/rqtransformer/attentions.py:48-55,117-122,134-142 of the reference) -- for the per-GPU batches SURVEY 8d names
// (64 = BASELINE configs[3] / [4] per GPU, 100 / 200 / 500 = the reference's measure_throughput and Figure 4).
//
// What round 4 measured about this regime (profiles/r04_launch_probe.txt, r04_stream_trace.txt):
//  * a dependent launch inside a replayed hipGraph costs 1.53 us whatever it does; a launch that only READS what a 64-row qkv
//    GEMM reads (196 KB of activations shared by all workgroups + 56 KB of weights each, everything in flight at once, no
//    arithmetic) costs 5.2 us; with a quarter of the activation rows 3.5 us.  LDS-DMA and plain loads move the same bytes in the
//    same time; more workgroups reading the same rows cost more (the shared rows come through the L2s at ~17 TB/s in total).
//  * gemm_stream_kernel (round 3) spends 1.7 us before its first byte is requested, then ~0.84 us per K-tile per wavefront
//    -- 0.55 us of it a serial chain (fragment reads, the refill's 12 DMA instructions issued against three other wavefronts,
//    8 MFMAs) that nothing overlaps, because a SIMD holds ONE wavefront; the ring depth makes no difference (2 slots = 3 slots).
//  * the tiled kernels (gemm_bf16_kernel, 129 .. 2047 rows) cross a workgroup barrier per K-tile: ~0.5 us x 24 tiles whatever the
//    tile shape or ring depth (profiles/r02_gemm_deep_ring_mid_batch.txt) -- 14-26 us per GEMM at 256-500 rows for traffic that
//    takes 5-7 us to move.
// Hence this kernel:
//  * a workgroup owns BM activation rows x BN weight rows; its NWAVE wavefronts each take every NWAVE-th K-tile and never meet
//    until the end (no barrier in the main loop, as in gemm_stream_kernel);
//  * W does not go through LDS at all: a fragment-packed copy Wp[n / 32][k / 64][4][64 lanes][8] (made once per parameter,
//    rq_pack_w) puts the 16 bytes an MFMA lane needs for (32 weight rows, 16 k) at lane * 16 of a contiguous 1-KB block, so a
//    K-tile of 32 weight rows is four fully coalesced loads straight into the B operand registers, double-buffered;
//  * with W out of the LDS, a ring slot is the A tile alone (BM x 128 bytes): two slots per wavefront and EIGHT wavefronts fit
//    at 64 rows (128 KB) -- two wavefronts per SIMD, so one's fragment reads / DMA issue / MFMA dependencies hide behind the
//    other's -- and 128 x 96 tiles fit at 128 rows per m-tile (four wavefronts): 192 workgroups cover a 500-row qkv GEMM in ONE
//    round of the 256 CUs with 688 KB of operands each, where 128 x 64 tiles need two rounds and 128 x 128 leave 112 CUs idle;
//  * the wavefronts' partial tiles are summed pairwise through LDS (a fixed tree: wavefront w + wavefront w + half), the total is
//    laid out row-major in LDS and all threads apply the epilogue on row-contiguous 16 / 32-byte pieces.
// The result of a row does not depend on the batch it sits in as long as the kernel choice (a function of the row count class,
// N and K: rq_gemm_pick_ks) is the same.
#pragma once
#include "gemm.h"

// bytes of one (32 weight rows x 64 k) block of the packed layout: [ks 0..3][lane 0..63][8 bf16]
#define RQ_WP_BLOCK 4096

// W[N][K] row-major bf16 -> Wp[ceil(N / 32)][K / 64][4][64][8]; rows >= N are zero.  One thread per 16-byte piece.
__global__ void rq_pack_w_kernel(const bf16_t* W, bf16_t* Wp, int N, int K) {
    const long pc = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int KT = K / 64;
    const long total = (long)((N + 31) / 32) * KT * 256;
    if (pc >= total) return;
    const int lane = (int)(pc & 63), ks = (int)((pc >> 6) & 3);
    const long blk = pc >> 8;
    const int kt = (int)(blk % KT), nb = (int)(blk / KT);
    const int n = nb * 32 + (lane & 31), k = kt * 64 + ks * 16 + (lane >> 5) * 8;
    rq_u128 v = zero128();
    if (n < N) v = ld128(W + (long)n * K + k);
    st128(Wp + pc * 8, v);
}

template <int BM, int BN, int NWAVE>
__global__ __launch_bounds__(NWAVE * 64) void gemm_ks_kernel(GemmArgs p) {
    constexpr int BK = 64, NS = 2;
    constexpr int MI = BM / 32, NB = BN / 32;
    constexpr int A_BYTES = BM * BK * 2;               // one A tile = one ring slot
    constexpr int A_G = BM / 8;                        // 1-KB (8-row) DMA groups per A tile
    constexpr int W_L = 4 * NB;                        // 16-byte loads per lane per W tile
    constexpr int NT = NWAVE * 64;
    constexpr int RS = BN + 1;                         // row stride (floats) of the reduced tile in LDS
    constexpr int FRAG = MI * NB * 16 * 64;            // floats of one wavefront's accumulators
    constexpr int PC = (BM * BN / 8 + NT - 1) / NT;    // 8-column output pieces per thread
    static_assert(NWAVE * NS * A_BYTES <= 160 * 1024, "LDS budget (rings)");
    static_assert((NWAVE / 2) * FRAG * 4 <= 160 * 1024 && BM * RS * 4 <= 160 * 1024, "LDS budget (reduction)");
    static_assert(W_L + A_G + W_L <= 63, "vmcnt range");
    RQ_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = rq_uniform(tid >> 6);
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int kt_total = p.K / BK;
    const int per = (kt_total + p.splitk - 1) / p.splitk;
    const int kt0 = blockIdx.z * per;
    const int kt1 = (kt0 + per < kt_total) ? kt0 + per : kt_total;
    const int nk = kt1 - kt0;
    const int n_mine = nk > wave ? (nk - wave + NWAVE - 1) / NWAVE : 0;      // this wavefront's K-tiles: kt0 + wave + NWAVE i

    const char* gA = (const char*)p.A;
    const rq_lds_t lds_w = rq_lds_addr(smem) + (rq_lds_t)(wave * NS * A_BYTES);
    // A by LDS-DMA: lane (lr, lc) of 8-row group g fills LDS position (row 8 g + lr, chunk lc); the 16-byte-chunk swizzle
    // chunk ^ ((row >> 1) & 7) of the fragment reads is applied to the per-lane SOURCE address (as in gemm_stream_kernel)
    const int lr = lane >> 3, lc = lane & 7;
    unsigned ga[A_G];
#pragma unroll
    for (int g = 0; g < A_G; ++g) {
        const int row = 8 * g + lr;
        int m = m0 + row;
        m = m < p.M - 1 ? m : p.M - 1;
        ga[g] = ((unsigned)m * (unsigned)p.lda + (unsigned)((lc ^ ((row >> 1) & 7)) << 3)) * 2u;
    }
    // W from the packed copy: n-block j of this workgroup (clamped: a ragged last workgroup re-reads the last block, its columns
    // are masked in the epilogue), K-tile kt, k-step ks: 1 KB at ((nb * KT + kt) * 4 + ks) * 1024, this lane's 16 bytes at lane * 16
    const int nb_total = (p.N + 31) >> 5;
    const char* gw[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        int nb = (n0 >> 5) + j;
        nb = nb < nb_total - 1 ? nb : nb_total - 1;
        gw[j] = (const char*)p.W + ((long)nb * kt_total) * RQ_WP_BLOCK + lane * 16;
    }
    rq_u128 wq0[NB][4], wq1[NB][4];
    auto issue = [&](int i, int b, rq_u128 (&wq)[NB][4]) {
        const int kt = kt0 + wave + i * NWAVE;
        const unsigned kb = (unsigned)kt * (BK * 2);
        const rq_lds_t base = lds_w + (rq_lds_t)(b * A_BYTES);
#pragma unroll
        for (int g = 0; g < A_G; ++g) rq_glds16(base + (rq_lds_t)(g * 1024), gA + (ga[g] + kb));
        rq_sched_barrier();                          // the order A (DMA), then W (plain loads) is what the counted waits below assume
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wq[j][ks] = ldg128_counted(gw[j] + (long)kt * RQ_WP_BLOCK + ks * 1024);
        rq_sched_barrier();
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

    // what the epilogue needs from global memory is requested under the main loop when a thread owns few pieces (the small tiles,
    // where a launch is a handful of microseconds and a dependent round trip at its end shows); the large tiles fetch it there
    const float* bias = p.bias;
    if (bias && p.bias_step) bias += (long)(*p.bias_step) * p.bias_stride;
    const int epi = p.epi;
    const bool accum = epi == EPI_F32_PARTIAL && p.accum;
    if (epi == EPI_F32_PARTIAL && !accum) bias = nullptr;
    constexpr bool EARLY = PC <= 2;
    float bv[PC][8], xr[PC][8];
    // (no divergent control flow around these loads: a value that is "loaded or zero" behind a branch has to be waited for at the
    // join -- vmcnt(0), with the first K-tiles in the queue; addresses are clamped instead and the result selected afterwards)
    auto fetch_piece = [&](int q) {
        int pc = tid + q * NT;
        pc = pc < BM * BN / 8 ? pc : BM * BN / 8 - 1;
        const int row = pc / (BN / 8), c0 = (pc % (BN / 8)) * 8;
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;
        const int n = n0 + c0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ne = n + e < p.N ? n + e : p.N - 1;
            bv[q][e] = bias ? bias[ne] : 0.f;                                       // uniform condition
            xr[q][e] = accum ? ((const float*)p.out)[(long)m * p.ldo + ne] : 0.f;   // uniform condition
        }
    };
    if (EARLY) {
#pragma unroll
        for (int q = 0; q < PC; ++q) fetch_piece(q);
    }

    // ---- main loop, two K-tiles per trip (static register names for the two W buffers).  Loads in flight behind tile t when it
    // is awaited, oldest first: A(t) | W(t) | A(t+1) | W(t+1) (| the epilogue's early fetches, once): the counted wait leaves
    // everything after A(t) outstanding.  The compiler places its own wait for the W registers; it does not see the DMAs, which
    // makes that wait also cover A(t+1) -- issued together with W(t) and served from L2: nothing to lose.
    auto consume = [&](int b, rq_u128 (&wq)[NB][4], bool next_in_flight) {
        if (next_in_flight) rq_wait_vmcnt<W_L + A_G + W_L>();
        else rq_wait_vmcnt<W_L>();
        rq_wave_sync();                              // every lane's share of the A tile has landed
        const char* sb = (const char*)smem + (wave * NS + b) * A_BYTES;
        if constexpr (MI * NB <= 4) {
            // small tiles: all fragments first (one LDS round trip), then the MFMAs
            bf16x8 af[MI][4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) af[mi][ks] = as_bf16x8(ld128(sb + rd[ks] + mi * (32 * BK * 2)));
            rq_wait_lgkmcnt<0>();                    // the fragments are in registers: the slot may be refilled
            rq_wave_sync();
            rq_sched_barrier();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int j = 0; j < NB; ++j) acc[mi][j] = rq_mfma_32x32x16_bf16(af[mi][ks], as_bf16x8(wq[j][ks]), acc[mi][j]);
        } else {
            // large tiles (up to 12 accumulator blocks): fragments per k-step, the reads of step ks + 1 behind the MFMAs of step ks
            // (the compiler orders them; 16 fragment registers live instead of 64)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 af[MI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) af[mi] = as_bf16x8(ld128(sb + rd[ks] + mi * (32 * BK * 2)));
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int j = 0; j < NB; ++j) acc[mi][j] = rq_mfma_32x32x16_bf16(af[mi], as_bf16x8(wq[j][ks]), acc[mi][j]);
            }
            rq_wait_lgkmcnt<0>();                    // (all fragment reads retired: the slot may be refilled)
            rq_wave_sync();
        }
        rq_sched_barrier();
    };
    // Straight-line code per tile count: the compiler's own waits (for the W registers) are exact only where it can see which loads
    // are in flight -- behind a conditional issue it assumes the worst ("nothing newer") and waits for the NEXT tile as well, which
    // is the end of double buffering (seen in the first build of this kernel).  So the short counts (1, 2, 3 tiles: 1536-wide K over
    // eight wavefronts is 3) are spelled out, and the general case is an unconditional steady-state loop plus the same tails.
    // (The epilogue's early fetches are newer than everything the first waits count: more outstanding loads than a wait assumes
    // only makes it wait for more.)
    auto tail2 = [&](int t) { consume(0, wq0, true); consume(1, wq1, false); (void)t; };
    auto tail3 = [&](int t) { consume(0, wq0, true); issue(t + 2, 0, wq0); consume(1, wq1, true); consume(0, wq0, false); };
    if (n_mine >= 4) {
        issue(0, 0, wq0); issue(1, 1, wq1);
        int t = 0;
        for (; t + 3 < n_mine; t += 2) {             // tiles t (buffer 0) and t + 1 (buffer 1) are in flight; both refills exist
            consume(0, wq0, true);
            issue(t + 2, 0, wq0);
            consume(1, wq1, true);
            issue(t + 3, 1, wq1);
        }
        if (n_mine - t == 2) tail2(t);
        else tail3(t);
    } else if constexpr (NWAVE == 8) {               // the 64-row tile: 3 tiles per wavefront is ITS common case (K = 1536)
        if (n_mine == 1) {
            issue(0, 0, wq0);
            consume(0, wq0, false);
        } else if (n_mine == 2) {
            issue(0, 0, wq0); issue(1, 1, wq1);
            tail2(0);
        } else if (n_mine == 3) {
            issue(0, 0, wq0); issue(1, 1, wq1);
            tail3(0);
        }
    } else {
        // the 128-row tiles are never chosen with fewer than 4 K-tiles per wavefront (rq_gemm_pick_ks); for completeness, one tile
        // at a time
        for (int t = 0; t < n_mine; ++t) {
            issue(t, 0, wq0);
            consume(0, wq0, false);
        }
    }

    // ---- cross-wavefront reduction: a fixed pairwise tree in fragment order (lane-contiguous, conflict-free), then the total in
    // row-major order for the epilogue
    float* sRed = (float*)smem;
    rq_syncthreads();                                // every wavefront is done with its ring
#pragma unroll
    for (int half = NWAVE / 2; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
            float* dst = sRed + (long)(wave - half) * FRAG + lane;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[((mi * NB + j) * 16 + r) * 64] = acc[mi][j][r];
        }
        rq_syncthreads();
        if (wave < half) {
            const float* src = sRed + (long)wave * FRAG + lane;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][j][r] += src[((mi * NB + j) * 16 + r) * 64];
        }
        rq_syncthreads();
    }
    if (wave == 0) {
        // C/D map: register r of lane (frow, fk) holds (row 32 mi + (r & 3) + 8 (r >> 2) + 4 fk, column 32 j + frow)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * fk;
                    sRed[row * RS + 32 * j + frow] = acc[mi][j][r];
                }
    }
    rq_syncthreads();

    if (!EARLY) {
#pragma unroll
        for (int q = 0; q < PC; ++q) fetch_piece(q);
    }
#pragma unroll
    for (int q = 0; q < PC; ++q) {
        const int pc = tid + q * NT;
        if (pc >= BM * BN / 8) continue;
        const int row = pc / (BN / 8), c0 = (pc % (BN / 8)) * 8;
        const int m = m0 + row, n = n0 + c0;
        if (m >= p.M || n >= p.N) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = sRed[row * RS + c0 + e];
        const bool full = n + 7 < p.N;
        if (epi <= EPI_BF16_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bv[q][e];
            if (epi == EPI_BF16_GELU) {
                float lo[4] = {v[0], v[1], v[2], v[3]}, hi[4] = {v[4], v[5], v[6], v[7]};
                rq_gelu4(lo, p.gelu_v2);
                rq_gelu4(hi, p.gelu_v2);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
            }
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
                    lo[e] = accum ? (xr[q][e] + v[e]) + bv[q][e] : v[e] + bv[q][e];
                    hi[e] = accum ? (xr[q][4 + e] + v[4 + e]) + bv[q][4 + e] : v[4 + e] + bv[q][4 + e];
                }
                *(f32x4*)o = lo;
                *(f32x4*)(o + 4) = hi;
            } else {
                for (int e = 0; e < 8 && n + e < p.N; ++e) o[e] = accum ? (xr[q][e] + v[e]) + bv[q][e] : v[e] + bv[q][e];
            }
        }
    }
}

// (host-side entry points: gemm.h)
