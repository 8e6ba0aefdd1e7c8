// quantize.hip -- residual quantiser: fused nearest-codebook search over all depths (gfx950).
//
// Reference: RQBottleneck.quantize (rqvae/models/rqvae/quantizations.py:237-271) =
//   depth x { VQEmbedding.compute_distances :43-62 (||r||^2 + ||c||^2 - 2 r.c^T via addmm, fp32),
//             find_nearest_embedding :64-69 (argmin, first minimum), embed :144-146,
//             residual.sub_(quant); aggregated.add_(quant) :264-265 }
// plus embed_code :297-311 / embed_code_with_depth :313-334 (rq_embed_kernel below).
//
// The reference materialises an (N x K) fp32 distance matrix per depth (268 MB at 64 images) and
// re-reads it for argmin.  Here one workgroup owns 64 vectors for the whole depth loop: the residual
// lives in LDS (fp32, exact), the codebook streams through LDS in [128 codes][64 dims] chunks
// (double-buffered, coalesced 256-B row segments), the r.c^T contraction runs on the exact-f32
// matrix pipe (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain), every lane keeps a running
// (distance, index) minimum for its code column, and the minimum is wavefront-reduced with
// lowest-index tie-break.  Distances never leave registers; HBM traffic is the algorithmic minimum
// (x in, codes/quants out, codebook from L2/MALL).
//
// Lane map of one 4-MFMA group (k-permutation is free as long as A and B agree): lane (i = l&31,
// half = l>>5) loads float4 r[i][8q+4*half .. +3] and c[j][8q+4*half .. +3]; MFMA m of the group
// consumes component m of both, so half 0 contributes dim 8q+m and half 1 dim 8q+4+m.
#include <stdlib.h>
#include "rq_common.h"

#define RQ_MAX_DEPTH 8
struct RqQuantArgs {
    const float* x;                    // (n_vec, dim): the input; split mode, depth > 0: the residual buffer
    const float* cb[RQ_MAX_DEPTH];
    const float* cn[RQ_MAX_DEPTH];     // ||c||^2 per code
    int K[RQ_MAX_DEPTH];
    int depth, dim;
    long n_vec;
    int64_t* codes;
    float* quant_cum;                  // (depth, n_vec, dim) or null
    // split mode (few vectors: one launch per depth, the codebook divided over blockIdx.y)
    int dep, n_split, tiles_per_split;
    float* part_v;                     // [n_vec][n_split] partial minima
    int* part_i;
    float* resid;                      // [n_vec][dim] residual between the per-depth launches
    float* logit_out;                  // split mode, or null: [n_vec][K] receives -distance * inv_temp (soft codes)
    float inv_temp;
    int use_codes;                     // combine: take the code from p.codes (written by the sampler) instead of the partial minima
};

constexpr int QT_M = 64;     // vectors per workgroup
constexpr int QT_N = 128;    // codes per tile (4 code groups x 32)
constexpr int QT_K = 64;     // dims per staged chunk
constexpr int QT_NTH = 512;  // 8 wavefronts: 2 vector halves x 4 code groups, one 32 x 32 accumulator each
constexpr int QT_NST = 4;    // ring stages of [128 codes][64 dims] fp32 (32 KB each), filled by LDS-DMA
constexpr int QT_PER = 5;    // LDS-DMA instructions per wavefront and step: 4 x (4 code rows x 256 B) + this code group's 32 norms
constexpr int QT_STAGE_BYTES = QT_N * QT_K * 4;
constexpr size_t QT_SMEM = (size_t)QT_NST * QT_STAGE_BYTES + (size_t)QT_NST * 4 * 1024 + (QT_M + 4 * QT_M) * sizeof(float) + (4 * QT_M + QT_M) * sizeof(int);

__global__ void rq_code_norm_kernel(const float* cb, int K, int dim, float* cn) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const float* c = cb + (long)k * dim;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int d = 0; d < dim; d += 4) {
        s0 = fmaf(c[d], c[d], s0);
        s1 = fmaf(c[d + 1], c[d + 1], s1);
        s2 = fmaf(c[d + 2], c[d + 2], s2);
        s3 = fmaf(c[d + 3], c[d + 3], s3);
    }
    cn[k] = (s0 + s1) + (s2 + s3);
}

// SPLIT = 0: all depths in one launch.  SPLIT = 1: depth p.dep only, codes of tiles [blockIdx.y * tiles_per_split, ...) only; the
// per-vector partial minimum goes to part_v / part_i and rq_split_combine_kernel finishes the depth.  NCH = dim / 64.
//
// Round 6 structure (rounds 1-5: residual in LDS, codebook chunks staged global -> registers -> ds_write into two LDS buffers, one
// barrier per 64 dims behind a vmcnt(0): 0.57-0.61 of the fp32 MFMA rate with 38.5 % of the wave-cycles parked, profiles/r05_rq_sqpmc.txt):
//  * the RESIDUAL LIVES IN REGISTERS: lane (i = l & 31, half = l >> 5) of a wavefront of vector half vh holds
//    r[32 vh + i][64 c + 8 q + 4 half + m] in ra[(8 c + q) * 4 + m] -- exactly the A operand of MFMA m of group q of chunk c, so the
//    main loop reads no A fragment at all (half of the LDS read traffic gone, 66 KB of LDS freed); the four code-group wavefronts
//    of a vector half hold the same 32 vectors.  The residual update gathers c[code] in the same layout and subtracts in registers.
//  * the CODEBOOK goes from global memory straight into a ring of four 32-KB LDS stages by LDS-DMA (global_load_lds_dwordx4: no
//    staging registers, no ds_write); a 16-byte chunk of a row lands in slot (chunk ^ (row & 15)) of its 256-byte LDS row (the
//    swizzle is applied to the per-lane SOURCE address, the DMA writes lane-linear), which makes the ds_read_b128 of 32 rows x
//    one chunk conflict-free.  The ||c||^2 of the tile ride along as a fifth DMA per wavefront and step (128 bytes, into a 1-KB
//    slot per stage and code group).  Counted waits: the DMAs of step s were issued three steps earlier; before the barrier that
//    publishes step s a wavefront waits until at most 2 x 5 of its DMAs (steps s + 1, s + 2) are in flight.
//  * one barrier per step as before, but nothing else stands at it: no vmcnt(0), no ds_write pass.
// Arithmetic unchanged: the same MFMA sequence per accumulator, the same ||r||^2 summation order (rq_norm_regs reproduces the
// eight-threads-per-vector order of rounds 1-5 bit for bit), the same distance expression and tie-breaks.
template <int NCH>
static __device__ __forceinline__ float rq_norm_regs(const float (&ra)[NCH * 32]) {
    // partial sums as the 8 threads of a vector formed them: thread useg = 2 u + half took float4s useg, useg + 8, ... of the row,
    // i.e. (chunk c ascending; q = u, then q = u + 4), an fmaf chain over its elements; then the xor-1 / 2 / 4 shuffle tree
    float s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float v = ra[(8 * c + u + 4 * hi) * 4 + m];
                    ss = fmaf(v, v, ss);
                }
        s[u] = ss;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] += rq_shfl_xor(s[u], 32);      // useg ^ 1: the other half
    return (s[0] + s[1]) + (s[2] + s[3]);                          // useg ^ 2, useg ^ 4
}

template <int SPLIT, int NCH>
__global__ __launch_bounds__(QT_NTH) void rq_quantize_kernel(RqQuantArgs p) {
    RQ_DYN_SMEM(smem);
    constexpr int D = NCH * 64;
    float* sC = (float*)smem;                                          // [NST][128][64] swizzled
    float* sCn = sC + QT_NST * QT_N * QT_K;                            // [NST][4][256] (the first 32 floats of a slot are used)
    float* sXn = sCn + QT_NST * 4 * 256;                               // [64]
    float* sRedV = sXn + QT_M;                                         // [4][64]
    int* sRedI = (int*)(sRedV + 4 * QT_M);                             // [4][64]
    int* sCode = sRedI + 4 * QT_M;                                     // [64]

    const int tid = threadIdx.x, lane = tid & 63, wave = rq_uniform(tid >> 6);
    const int vh = wave >> 2, cw = wave & 3;          // vector half (32 rows), code group (32 columns of the tile)
    const long v0 = (long)blockIdx.x * QT_M;
    const int fi = lane & 31, fh = lane >> 5;
    const long myvec = v0 + vh * 32 + fi;
    const bool vok = myvec < p.n_vec;

    // ---- this lane's slice of its vector: 32 float4s per chunk pair ... (8 NCH float4s), ||x||^2
    float ra[NCH * 32];
    {
        const float* src = p.x + myvec * D + 4 * fh;
#pragma unroll
        for (int g = 0; g < NCH * 8; ++g) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (vok) v = *(const f32x4*)(src + 8 * g);
#pragma unroll
            for (int m = 0; m < 4; ++m) ra[g * 4 + m] = v[m];
        }
    }
    rq_sched_barrier();
    float xn_mine = rq_norm_regs<NCH>(ra);
    if (cw == 0 && fh == 0) sXn[vh * 32 + fi] = xn_mine;

    // ---- LDS-DMA addressing.  Instruction i of this wavefront fills tile rows 4 g .. 4 g + 3, g = 4 wave + i: lane l -> row
    // 4 g + (l >> 4), LDS slot l & 15, source chunk (l & 15) ^ (row & 15).
    const rq_lds_t lds0 = rq_lds_addr(smem);
    const rq_lds_t ldsCn = lds0 + (rq_lds_t)(QT_NST * QT_STAGE_BYTES);
    int drow[4];
    unsigned dchunk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        drow[i] = (4 * wave + i) * 4 + (lane >> 4);
        dchunk[i] = (unsigned)(((lane & 15) ^ (drow[i] & 15)) << 4);
    }
    // B fragment of group q: row 32 cw + fi, chunk 2 q + half -> slot (2 q) ^ (half ^ (row & 15))
    unsigned brd[8];
    {
        const int row = cw * 32 + fi;
#pragma unroll
        for (int q = 0; q < 8; ++q) brd[q] = (unsigned)(row * 256 + (((2 * q) ^ (fh ^ (row & 15))) << 4));
    }

    const int dep_lo = SPLIT ? p.dep : 0, dep_hi = SPLIT ? p.dep + 1 : p.depth;
    for (int dep = dep_lo; dep < dep_hi; ++dep) {
        const float* cb = p.cb[dep];
        const float* cn = p.cn[dep];
        const int K = p.K[dep];
        const int ntile_all = (K + QT_N - 1) / QT_N;
        const int tile_lo = SPLIT ? blockIdx.y * p.tiles_per_split : 0;
        int tile_hi = SPLIT ? tile_lo + p.tiles_per_split : ntile_all;
        tile_hi = tile_hi < ntile_all ? tile_hi : ntile_all;
        const int ntile = tile_hi > tile_lo ? tile_hi - tile_lo : 0;
        const int nstep = ntile * NCH;

        float bestv[16];
        int besti[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { bestv[r] = __int_as_float(0x7f800000); besti[r] = 0x7fffffff; }

        // DMA of step (tile, c) into stage st: the tile's rows beyond the codebook re-read its last row (never selected: `valid`)
        auto issue = [&](int tile, int c, int st) {
            const int row_max = K - 1 - tile * QT_N;                   // >= 0
            const char* base = (const char*)(cb + (long)tile * QT_N * D + c * QT_K);
            const rq_lds_t dst = lds0 + (rq_lds_t)(st * QT_STAGE_BYTES + wave * 4096);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = drow[i] < row_max ? drow[i] : row_max;
                rq_glds16_s(dst + (rq_lds_t)(i * 1024), base, (unsigned)r * (unsigned)(D * 4) + dchunk[i]);
            }
            // the 32 norms of this code group: lanes 0 .. 7 carry them (16 bytes each), the others repeat them into the rest of the slot;
            // a ragged last tile clamps the chunk inside the array (its misplaced values belong to codes >= K or are re-read below)
            int e = tile * QT_N + cw * 32 + 4 * (lane & 7);
            e = e <= K - 4 ? e : (K >= 4 ? K - 4 : 0);
            rq_glds16_s(ldsCn + (rq_lds_t)(st * 4096 + cw * 1024), K >= 4 ? (const char*)cn : (const char*)cb, (unsigned)e * 4u);      // (K < 4: any 16 readable bytes)
        };
        // steps are numbered s = (tile - tile_lo) * NCH + c and live in stage s % NST
        auto issue_step = [&](int s, int st) {
            const int t = s / NCH;
            issue(tile_lo + t, s - t * NCH, st);
        };

        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        rq_syncthreads();                                              // sXn of this depth is complete; the ring is free

#pragma unroll
        for (int s0 = 0; s0 < QT_NST - 1; ++s0)
            if (s0 < nstep) issue_step(s0, s0);
        int st = 0;
        for (int t = 0; t < ntile; ++t) {
            const int tile = tile_lo + t;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int s = t * NCH + c;
                // step s has landed once at most the steps issued after it are outstanding: min(NST - 2, nstep - 1 - s) of them
                const int newer = nstep - 1 - s;
                if (newer >= 2) rq_wait_vmcnt<2 * QT_PER>();
                else if (newer == 1) rq_wait_vmcnt<QT_PER>();
                else rq_wait_vmcnt<0>();
                rq_barrier_raw();                                      // publishes stage st; stage st - 1 is free for step s + NST - 1
                if (s + QT_NST - 1 < nstep) issue_step(s + QT_NST - 1, st == 0 ? QT_NST - 1 : st - 1);
                const char* cT = (const char*)sC + st * QT_STAGE_BYTES;
                // the B fragment of group q + 1 is requested before the four MFMAs of group q issue
                f32x4 b = *(const f32x4*)(cT + brd[0]);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    f32x4 bn = b;
                    if (q < 7) bn = *(const f32x4*)(cT + brd[q + 1]);
                    rq_sched_barrier();
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc = rq_mfma_32x32x2_f32(ra[(8 * c + q) * 4 + m], b[m], acc);
                    rq_sched_barrier();
                    b = bn;
                }
                if (c == NCH - 1) {
                    // distances of this lane's code column against its 16 rows
                    const int code = tile * QT_N + cw * 32 + fi;
                    const bool valid = code < K;
                    float cnv = sCn[st * 1024 + cw * 256 + fi];
                    if (tile * QT_N + QT_N > K) cnv = valid ? cn[code] : 0.f;      // ragged last tile (the DMA chunk may be clamped)
                    // ||r||^2 of the 16 rows this lane scores: rows (r & 3) + 8 (r >> 2) + 4 half + 32 vh
                    f32x4 xn4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) xn4[j] = *(const f32x4*)(sXn + vh * 32 + 8 * j + 4 * fh);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * fh + 32 * vh;
                        float d0 = fmaf(-2.0f, acc[r], xn4[r >> 2][r & 3] + cnv);
                        if (valid && d0 < bestv[r]) { bestv[r] = d0; besti[r] = code; }
                        if (SPLIT && p.logit_out && valid && v0 + row < p.n_vec)       // soft codes: softmax(-d / temp) logits
                            p.logit_out[(v0 + row) * K + code] = -d0 * p.inv_temp;
                        acc[r] = 0.f;
                    }
                }
                st = st + 1 == QT_NST ? 0 : st + 1;
            }
        }

        // ---- wavefront (value, index) min-reduction over the 32 code columns, lowest index on ties
        int rbase = cw * QT_M + 4 * fh + 32 * vh;      // row (r & 3) + 8 (r >> 2) + 4 half + 32 vh of code group cw
        rq_opaque(rbase);                              // (formed here: hoisted out of the depth loop, the 32 addresses below were spilled)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = bestv[r];
            int ix = besti[r];
#pragma unroll
            for (int m = 16; m >= 1; m >>= 1) {
                float ov = rq_shfl_xor(v, m);
                int oi = rq_shfl_xor_i(ix, m);
                if (ov < v || (ov == v && oi < ix)) { v = ov; ix = oi; }
            }
            if (fi == 0) {
                sRedV[rbase + (r & 3) + 8 * (r >> 2)] = v;
                sRedI[rbase + (r & 3) + 8 * (r >> 2)] = ix;
            }
        }
        rq_syncthreads();
        if (tid < QT_M) {
            float v = sRedV[tid];
            int ix = sRedI[tid];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                float ov = sRedV[w * QT_M + tid];
                int oi = sRedI[w * QT_M + tid];
                if (ov < v || (ov == v && oi < ix)) { v = ov; ix = oi; }
            }
            // a row without a single finite distance (NaN / inf in the encoder output) never satisfies `d < best`: its index is still
            // the 0x7fffffff seed, which would index the codebook ~2 TB out of bounds below.  Such a row gets code 0 -- what
            // torch.argmin returns when the distances are all NaN or all +inf (first NaN / first of equal infinities); for a row that
            // mixes NaN and inf distances torch returns its first NaN instead: garbage in either case, but always a valid index.
            // (SPLIT: the combine kernel clamps after merging the splits.)
            if (!SPLIT && (unsigned)ix >= (unsigned)K) ix = 0;
            if (SPLIT) {
                if (v0 + tid < p.n_vec) {
                    p.part_v[(v0 + tid) * p.n_split + blockIdx.y] = v;
                    p.part_i[(v0 + tid) * p.n_split + blockIdx.y] = ix;
                }
            } else {
                sCode[tid] = ix;
                if (v0 + tid < p.n_vec) p.codes[(v0 + tid) * p.depth + dep] = (int64_t)ix;
            }
        }
        if (SPLIT) return;
        rq_syncthreads();

        // ---- residual -= c[code]; aggregated += c[code]; new ||r||^2   (quantizations.py:264-267).  Every wavefront updates its own
        // copy of the residual (the four code groups of a vector half compute identical values); the aggregated output -- read-modify-
        // write of the previous depth's row -- is divided between them: code group cw takes groups q = 2 cw, 2 cw + 1 of every chunk.
        {
            const float* qrow = cb + (long)sCode[vh * 32 + fi] * D + 4 * fh;
#pragma unroll
            for (int g = 0; g < NCH * 8; ++g) {
                const f32x4 qv = *(const f32x4*)(qrow + 8 * g);
#pragma unroll
                for (int m = 0; m < 4; ++m) ra[g * 4 + m] -= qv[m];
                if (p.quant_cum && vok && ((g & 7) >> 1) == cw) {
                    f32x4 agg = qv;
                    if (dep > 0) agg = *(const f32x4*)(p.quant_cum + ((long)(dep - 1) * p.n_vec + myvec) * D + 8 * g + 4 * fh) + qv;
                    *(f32x4*)(p.quant_cum + ((long)dep * p.n_vec + myvec) * D + 8 * g + 4 * fh) = agg;
                }
                if ((g & 7) == 7) rq_sched_barrier();              // eight row pieces in flight at a time (the residual holds 32 NCH registers)
            }
            xn_mine = rq_norm_regs<NCH>(ra);
            if (cw == 0 && fh == 0) sXn[vh * 32 + fi] = xn_mine;
        }
        // (the barrier at the top of the next depth publishes sXn and frees the ring)
    }
}

// split mode, second half of a depth: combine the per-split partial minima (splits are ascending code ranges: the lowest
// index wins a tie), write the code, residual -= c[code] (into p.resid), aggregated += c[code].  8 threads per vector.
__global__ __launch_bounds__(512) void rq_split_combine_kernel(RqQuantArgs p) {
    const int tid = threadIdx.x;
    const int urow = tid >> 3, useg = tid & 7;
    const long vec = (long)blockIdx.x * 64 + urow;
    if (vec >= p.n_vec) return;
    const int D = p.dim, dep = p.dep;
    int ix;
    if (p.use_codes) {
        ix = (int)p.codes[vec * p.depth + dep];        // stochastic soft codes: the sampler already drew this depth's code
    } else {
        float v = p.part_v[vec * p.n_split];
        ix = p.part_i[vec * p.n_split];
        for (int s2 = 1; s2 < p.n_split; ++s2) {
            const float ov = p.part_v[vec * p.n_split + s2];
            const int oi = p.part_i[vec * p.n_split + s2];
            if (ov < v || (ov == v && oi < ix)) { v = ov; ix = oi; }
        }
        if ((unsigned)ix >= (unsigned)p.K[dep]) ix = 0;        // no finite distance in the row (see rq_quantize_kernel)
        if (useg == 0) p.codes[vec * p.depth + dep] = (int64_t)ix;
    }
    const float* q = p.cb[dep] + (long)ix * D;
    const float* src = p.x + vec * D;                  // depth 0: the input; later: the residual buffer (== p.resid)
    for (int i = 0; i < D / 32; ++i) {
        const int f = useg + 8 * i;
        const f32x4 qv = *(const f32x4*)(q + f * 4);
        f32x4 rv = *(const f32x4*)(src + f * 4);
        rv = rv - qv;
        *(f32x4*)(p.resid + vec * D + f * 4) = rv;
        if (p.quant_cum) {
            f32x4 agg = qv;
            if (dep > 0) agg = *(const f32x4*)(p.quant_cum + ((long)(dep - 1) * p.n_vec + vec) * D + f * 4) + qv;
            *(f32x4*)(p.quant_cum + ((long)dep * p.n_vec + vec) * D + f * 4) = agg;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// embed_code (mode 0: sum over depth), embed_code_with_depth (mode 1), depth-cumsum (mode 2)
struct RqEmbedArgs {
    const int64_t* codes;
    const float* cb[RQ_MAX_DEPTH];
    int K[RQ_MAX_DEPTH];
    int depth, dim, mode;
    long n_vec;
    float* out;
};

__global__ void rq_embed_kernel(RqEmbedArgs p) {
    const int f4_per_vec = p.dim / 4;
    long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= p.n_vec * f4_per_vec) return;
    long v = gid / f4_per_vec;
    int f = (int)(gid - v * f4_per_vec);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < p.depth; ++d) {
        long code = p.codes[v * p.depth + d];
        f32x4 e = {0.f, 0.f, 0.f, 0.f};
        // row K is the zero padding row of VQEmbedding (quantizations.py:28); anything else outside [0, K] is an index
        // error in the reference (F.embedding's device-side assert) and a trap here -- never silent zeros
        if (code < 0 || code > p.K[d]) rq_trap();
        if (code < p.K[d]) e = *(const f32x4*)(p.cb[d] + code * p.dim + f * 4);
        if (p.mode == 1) {
            *(f32x4*)(p.out + ((v * p.depth + d) * p.dim) + f * 4) = e;
        } else {
            acc = (d == 0) ? e : acc + e;          // depth-ordered fp32 sum, as cat(...).sum(-2) / cumsum
            if (p.mode == 2) *(f32x4*)(p.out + ((v * p.depth + d) * p.dim) + f * 4) = acc;
        }
    }
    if (p.mode == 0) *(f32x4*)(p.out + v * p.dim + f * 4) = acc;
}

// -------------------------------------------------------------------------------------------------
// one launch of the quantiser kernel for a.dim = 64 NCH
template <int SPLIT, int NCH>
static int rq_launch_quant_n(const RqQuantArgs& a, dim3 grid, hipStream_t st) {
    static RqDeviceOnce attr_once;      // kernel attributes are per device
    if (attr_once.first())
        (void)hipFuncSetAttribute((const void*)rq_quantize_kernel<SPLIT, NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)QT_SMEM);
    RQ_LAUNCH((rq_quantize_kernel<SPLIT, NCH>), grid, dim3(QT_NTH), QT_SMEM, st, a);
    return rq_check_launch(SPLIT ? "rq_quantize_kernel<split>" : "rq_quantize_kernel");
}
template <int SPLIT>
static int rq_launch_quant(const RqQuantArgs& a, dim3 grid, hipStream_t st) {
    switch (a.dim) {
        case 64: return rq_launch_quant_n<SPLIT, 1>(a, grid, st);
        case 128: return rq_launch_quant_n<SPLIT, 2>(a, grid, st);
        case 192: return rq_launch_quant_n<SPLIT, 3>(a, grid, st);
        case 256: return rq_launch_quant_n<SPLIT, 4>(a, grid, st);
        default: return rq_fail(RQAMD_ERR_UNSUPPORTED, "rq_quantize: dim %d must be 64, 128, 192 or 256", a.dim);
    }
}

// -------------------------------------------------------------------------------------------------
// ||c||^2 of every code of one codebook (the ||e||^2 term of VQEmbedding.compute_distances, quantizations.py:51-52), computed
// ONCE per codebook version by the caller and passed to every rqamd_rq_quantize on it (it used to be recomputed per call into a
// process-global scratch buffer, which was neither stream- nor thread-safe).
extern "C" int rqamd_rq_code_norms(const float* codebook, int n_embed, int dim, float* norms_out, void* stream) {
    if (!codebook || !norms_out || n_embed < 1) return rq_fail(RQAMD_ERR_INVALID, "rq_code_norms: bad argument");
    if (dim % 4 != 0) return rq_fail(RQAMD_ERR_UNSUPPORTED, "rq_code_norms: dim %d %% 4 != 0", dim);
    RQ_LAUNCH(rq_code_norm_kernel, dim3((n_embed + 255) / 256), dim3(256), 0, (hipStream_t)stream, codebook, n_embed, dim, norms_out);
    return rq_check_launch("rq_code_norm_kernel");
}

extern "C" int rqamd_rq_quantize(const float* x, const float* const* codebooks, const float* const* code_norms, const int* n_embed,
                                 int depth, int64_t n_vec, int dim, int64_t* codes, float* quant_cum, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
    if (n_vec == 0) return RQAMD_OK;
    if (!x || !codebooks || !code_norms || !n_embed || !codes) return rq_fail(RQAMD_ERR_INVALID, "rq_quantize: null argument");
    if (depth < 1 || depth > RQ_MAX_DEPTH) return rq_fail(RQAMD_ERR_UNSUPPORTED, "rq_quantize: depth %d not in 1..%d", depth, RQ_MAX_DEPTH);
    if (dim % 64 != 0 || dim < 64 || dim > 256)
        return rq_fail(RQAMD_ERR_UNSUPPORTED, "rq_quantize: dim %d must be 64, 128, 192 or 256", dim);
    if (n_vec < 0) return rq_fail(RQAMD_ERR_INVALID, "rq_quantize: n_vec < 0");
    hipStream_t st = (hipStream_t)stream;
    RqQuantArgs a{};
    int kmin = 1 << 30;
    for (int d = 0; d < depth; ++d) {
        if (n_embed[d] < 1 || !codebooks[d] || !code_norms[d]) return rq_fail(RQAMD_ERR_INVALID, "rq_quantize: empty codebook / missing norms");
        a.cb[d] = codebooks[d];
        a.K[d] = n_embed[d];
        a.cn[d] = code_norms[d];
        kmin = n_embed[d] < kmin ? n_embed[d] : kmin;
    }
    a.x = x; a.depth = depth; a.dim = dim; a.n_vec = n_vec; a.codes = codes; a.quant_cum = quant_cum;
    const long ntiles = (n_vec + QT_M - 1) / QT_M;
    // Few vectors (the per-image rFID / get_codes calls: 64 vectors = ONE workgroup scanning a 16.8 MB codebook four times):
    // divide the codebook over blockIdx.y, one launch pair per depth.  Needs the caller's workspace (residual + partials).
    static const bool no_split = getenv("RQAMD_RQ_NO_SPLIT") != nullptr;
    if (!no_split && workspace && ntiles * g_rq_row_scale < 96 && kmin >= 1024) {
        const int tiles_k = (kmin + QT_N - 1) / QT_N;
        int S = (int)(512 / ntiles);
        S = S > 64 ? 64 : S;
        S = S > tiles_k ? tiles_k : S;
        if (S >= 2) {
            const size_t need = (size_t)n_vec * dim * 4 + (size_t)n_vec * S * 8;
            if ((size_t)workspace_bytes >= need) {
                a.resid = (float*)workspace;
                a.part_v = a.resid + (size_t)n_vec * dim;
                a.part_i = (int*)(a.part_v + (size_t)n_vec * S);
                for (int d = 0; d < depth; ++d) {
                    const int tk = (n_embed[d] + QT_N - 1) / QT_N;
                    a.dep = d;
                    a.tiles_per_split = (tk + S - 1) / S;
                    a.n_split = (tk + a.tiles_per_split - 1) / a.tiles_per_split;
                    a.x = d == 0 ? x : a.resid;
                    RQ_TRY(rq_launch_quant<1>(a, dim3((unsigned)ntiles, (unsigned)a.n_split), st));
                    RQ_LAUNCH(rq_split_combine_kernel, dim3((unsigned)ntiles), dim3(512), 0, st, a);
                    RQ_TRY(rq_check_launch("rq_split_combine_kernel"));
                }
                return RQAMD_OK;
            }
        }
    }
    return rq_launch_quant<0>(a, dim3((unsigned)ntiles), st);
}


// -------------------------------------------------------------------------------------------------
// Soft codes: RQBottleneck.get_soft_codes (quantizations.py:371-400) = per depth softmax(-distances / temp) over the
// codebook, the code of that depth (argmin, or one multinomial draw from the soft code when stochastic), residual update.
// Row softmax of the logits written by the split-mode quantiser kernel: one workgroup per (vector), K <= 65536.
__global__ __launch_bounds__(256) void rq_softmax_rows_kernel(const float* logits, int K, float* out, long out_stride) {
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* src = logits + (long)blockIdx.x * K;
    float* dst = out + (long)blockIdx.x * out_stride;
    float m = -__int_as_float(0x7f800000);
    for (int k = tid; k < K; k += 256) m = fmaxf(m, src[k]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    rq_syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int k = tid; k < K; k += 256) s += rq_fast_exp2((src[k] - m) * 1.4426950408889634f);
    s = wave_sum(s);
    if (lane == 0) red[4 + wave] = s;
    rq_syncthreads();
    const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
    for (int k = tid; k < K; k += 256) dst[k] = rq_fast_exp2((src[k] - m) * 1.4426950408889634f) * inv;
}

int rq_launch_sample_rows(const float* logits, int rows, int vocab, uint64_t seed, uint64_t offset, int64_t* out, long out_stride, hipStream_t s);

extern "C" int rqamd_rq_soft_codes(const float* x, const float* const* codebooks, const float* const* code_norms, const int* n_embed,
                                   int depth, int64_t n_vec, int dim, float temp, int stochastic, uint64_t seed, uint64_t offset,
                                   float* soft_out, int64_t* codes, void* workspace, int64_t workspace_bytes, void* stream) {
    if (n_vec == 0) return RQAMD_OK;
    if (!x || !codebooks || !code_norms || !n_embed || !codes || !soft_out || !workspace) return rq_fail(RQAMD_ERR_INVALID, "rq_soft_codes: null argument");
    if (depth < 1 || depth > RQ_MAX_DEPTH) return rq_fail(RQAMD_ERR_UNSUPPORTED, "rq_soft_codes: depth %d not in 1..%d", depth, RQ_MAX_DEPTH);
    if (dim % 64 != 0 || dim < 64 || dim > 256) return rq_fail(RQAMD_ERR_UNSUPPORTED, "rq_soft_codes: dim %d must be 64, 128, 192 or 256", dim);
    if (!(temp > 0.f)) return rq_fail(RQAMD_ERR_INVALID, "rq_soft_codes: temp must be > 0");
    const int K = n_embed[0];
    for (int d = 0; d < depth; ++d)
        if (n_embed[d] != K || !codebooks[d] || !code_norms[d]) return rq_fail(RQAMD_ERR_UNSUPPORTED, "rq_soft_codes: codebooks of one size needed (the reference concatenates the soft codes along depth)");
    hipStream_t st = (hipStream_t)stream;
    const long ntiles = (n_vec + QT_M - 1) / QT_M;
    const int tiles_k = (K + QT_N - 1) / QT_N;
    int S = (int)(512 / ntiles);
    S = S < 1 ? 1 : (S > 64 ? 64 : S);
    S = S > tiles_k ? tiles_k : S;
    const size_t need = (size_t)n_vec * dim * 4 + (size_t)n_vec * 64 * 8 + (size_t)n_vec * K * 4;
    if ((size_t)workspace_bytes < need) return rq_fail(RQAMD_ERR_INVALID, "rq_soft_codes: workspace of %zu bytes needed", need);
    RqQuantArgs a{};
    for (int d = 0; d < depth; ++d) { a.cb[d] = codebooks[d]; a.K[d] = K; a.cn[d] = code_norms[d]; }
    a.depth = depth; a.dim = dim; a.n_vec = n_vec; a.codes = codes; a.quant_cum = nullptr;
    a.resid = (float*)workspace;
    a.part_v = a.resid + (size_t)n_vec * dim;
    a.part_i = (int*)(a.part_v + (size_t)n_vec * 64);
    a.logit_out = (float*)(a.part_i + (size_t)n_vec * 64);
    a.inv_temp = 1.0f / temp;
    a.use_codes = stochastic ? 1 : 0;
    a.tiles_per_split = (tiles_k + S - 1) / S;
    a.n_split = (tiles_k + a.tiles_per_split - 1) / a.tiles_per_split;
    for (int d = 0; d < depth; ++d) {
        a.dep = d;
        a.x = d == 0 ? x : a.resid;
        RQ_TRY(rq_launch_quant<1>(a, dim3((unsigned)ntiles, (unsigned)a.n_split), st));
        if (stochastic)      // torch.multinomial(soft_code, 1) (quantizations.py:388-390): one draw per vector from softmax(logits)
            RQ_TRY(rq_launch_sample_rows(a.logit_out, (int)n_vec, K, seed, offset + 4ull * d, codes + d, depth, st));
        RQ_LAUNCH(rq_split_combine_kernel, dim3((unsigned)ntiles), dim3(512), 0, st, a);
        RQ_TRY(rq_check_launch("rq_split_combine_kernel"));
        RQ_LAUNCH(rq_softmax_rows_kernel, dim3((unsigned)n_vec), dim3(256), 0, st, (const float*)a.logit_out, K, soft_out + (size_t)d * K, (long)depth * K);
        RQ_TRY(rq_check_launch("rq_softmax_rows_kernel"));
    }
    return RQAMD_OK;
}

// VQEmbedding.compute_distances (quantizations.py:43-62) as a stand-alone call: the split-mode kernel's logit output with
// inv_temp = -1 is the distance itself (-d * -1, exact); the partial minima it also writes go to the workspace and are not used.
extern "C" int rqamd_rq_distances(const float* x, const float* codebook, const float* code_norms, int n_embed, int64_t n_vec, int dim,
                                  float* dist_out, void* workspace, int64_t workspace_bytes, void* stream) {
    if (n_vec == 0) return RQAMD_OK;
    if (!x || !codebook || !code_norms || !dist_out || !workspace) return rq_fail(RQAMD_ERR_INVALID, "rq_distances: null argument");
    if (dim % 64 != 0 || dim < 64 || dim > 256) return rq_fail(RQAMD_ERR_UNSUPPORTED, "rq_distances: dim %d must be 64, 128, 192 or 256", dim);
    if (n_embed < 1) return rq_fail(RQAMD_ERR_INVALID, "rq_distances: empty codebook");
    const long ntiles = (n_vec + QT_M - 1) / QT_M;
    const int tiles_k = (n_embed + QT_N - 1) / QT_N;
    int S = (int)(512 / ntiles);
    S = S < 1 ? 1 : (S > 64 ? 64 : S);
    S = S > tiles_k ? tiles_k : S;
    const size_t need = (size_t)n_vec * 64 * 8;
    if ((size_t)workspace_bytes < need) return rq_fail(RQAMD_ERR_INVALID, "rq_distances: workspace of %zu bytes needed", need);
    RqQuantArgs a{};
    a.cb[0] = codebook; a.K[0] = n_embed; a.cn[0] = code_norms;
    a.depth = 1; a.dim = dim; a.n_vec = n_vec; a.codes = nullptr; a.quant_cum = nullptr; a.x = x; a.dep = 0;
    a.part_v = (float*)workspace;
    a.part_i = (int*)(a.part_v + (size_t)n_vec * 64);
    a.logit_out = dist_out;
    a.inv_temp = -1.0f;
    a.tiles_per_split = (tiles_k + S - 1) / S;
    a.n_split = (tiles_k + a.tiles_per_split - 1) / a.tiles_per_split;
    return rq_launch_quant<1>(a, dim3((unsigned)ntiles, (unsigned)a.n_split), (hipStream_t)stream);
}

extern "C" int rqamd_rq_embed(const int64_t* codes, const float* const* codebooks, const int* n_embed, int depth,
                              int64_t n_vec, int dim, int mode, float* out, void* stream) {
    if (n_vec == 0) return RQAMD_OK;
    if (!codes || !codebooks || !n_embed || !out) return rq_fail(RQAMD_ERR_INVALID, "rq_embed: null argument");
    if (depth < 1 || depth > RQ_MAX_DEPTH) return rq_fail(RQAMD_ERR_UNSUPPORTED, "rq_embed: depth %d not in 1..%d", depth, RQ_MAX_DEPTH);
    if (dim % 4 != 0) return rq_fail(RQAMD_ERR_UNSUPPORTED, "rq_embed: dim %d must be a multiple of 4", dim);
    if (mode < 0 || mode > 2) return rq_fail(RQAMD_ERR_INVALID, "rq_embed: mode %d", mode);
    if (n_vec == 0) return RQAMD_OK;
    RqEmbedArgs a{};
    for (int d = 0; d < depth; ++d) { a.cb[d] = codebooks[d]; a.K[d] = n_embed[d]; }
    a.codes = codes; a.depth = depth; a.dim = dim; a.mode = mode; a.n_vec = n_vec; a.out = out;
    long work = n_vec * (dim / 4);
    RQ_LAUNCH(rq_embed_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return rq_check_launch("rq_embed_kernel");
}


// =================================================================================================
// EMA codebook update of stage-1 training (VQEmbedding._update_buffers / _update_embedding, quantizations.py:80-129).
// The reference builds a dense (n_embed x n_vectors) one-hot matrix and multiplies it with the vectors (16384 x 8192 fp32 =
// 512 MB per depth for a 128-image batch) to get per-code counts and vector sums; here a wavefront owns one code, scans the index
// list 64 entries at a time (ballot) and adds the matching vectors in ascending vector order -- deterministic, no atomics,
// nothing materialised.  The EMA / dead-code restart / normalisation steps are elementwise.
__global__ __launch_bounds__(256) void rq_ema_accumulate_kernel(const float* x, const int64_t* idx, long n_vec, int D, int K,
                                                                float* count_out, float* sum_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.x * 4 + wave;
    if (k >= K) return;
    float acc[16];                                   // D <= 1024: lane owns dims lane, lane + 64, ...
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int nd = (D + 63) / 64;
    int count = 0;
    for (long v0 = 0; v0 < n_vec; v0 += 64) {
        const long v = v0 + lane;
        const bool mine = v < n_vec && idx[v] == (int64_t)k;
        unsigned long long mask = rq_ballot(mine);
        while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const float* row = x + (v0 + j) * D;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (i < nd && lane + 64 * i < D) acc[i] += row[lane + 64 * i];
            ++count;
        }
    }
    if (lane == 0) count_out[k] = (float)count;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < nd && lane + 64 * i < D) sum_out[(long)k * D + lane + 64 * i] = acc[i];
}

// cluster_size_ema.mul_(decay).add_(count, alpha = 1 - decay); embed_ema likewise; then (restart != null) the dead-code restart:
// usage = cluster_size_ema >= 1; embed_ema = embed_ema * usage + restart * (1 - usage); cluster_size_ema = cluster_size_ema *
// usage + (1 - usage)   (quantizations.py:103-118).  One thread per (code, dim).
// The arithmetic mirrors torch's two steps: mul_(decay) rounds the product, add_(x, alpha) is one fused multiply-add with alpha =
// float(1 - decay) taken in DOUBLE (1 - 0.99 -> 0.01f, not 1.0f - 0.99f = 0.00999999), so that values next to the restart
// threshold cs >= 1 fall on the reference's side of it.
__global__ void rq_ema_update_kernel(float* cs_ema, float* embed_ema, const float* count, const float* sum, const float* restart,
                                     int K, int D, float decay, float alpha) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)K * D) return;
    const int k = (int)(gid / D);
    float cs = fmaf(alpha, count[k], cs_ema[k] * decay);
    float e = fmaf(alpha, sum[gid], embed_ema[gid] * decay);
    if (restart) {
        const float usage = cs >= 1.0f ? 1.0f : 0.0f;
        e = e * usage + restart[gid] * (1.0f - usage);
        cs = cs * usage + (1.0f - usage);
    }
    embed_ema[gid] = e;      // cluster_size_ema[k] is read by every thread of the row: it is updated by rq_ema_cs_kernel, after this launch
}
__global__ void rq_ema_cs_kernel(float* cs_ema, const float* count, int K, float decay, float alpha, int restart) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    float cs = fmaf(alpha, count[k], cs_ema[k] * decay);
    if (restart) {
        const float usage = cs >= 1.0f ? 1.0f : 0.0f;
        cs = cs * usage + (1.0f - usage);
    }
    cs_ema[k] = cs;
}

// weight[k][:] = embed_ema[k][:] / (n * (cluster_size_ema[k] + eps) / (n + n_embed * eps)), n = sum(cluster_size_ema)
// (quantizations.py:120-129); *n_total is a device scalar (the caller's cluster_size_ema.sum()).
__global__ void rq_ema_normalize_kernel(const float* cs_ema, const float* embed_ema, const float* n_total, int K, int D, float eps,
                                        float* weight) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)K * D) return;
    const int k = (int)(gid / D);
    const float n = *n_total;
    const float norm = n * (cs_ema[k] + eps) / (n + (float)K * eps);
    weight[gid] = embed_ema[gid] / norm;
}

extern "C" int rqamd_rq_ema_accumulate(const float* x, const int64_t* idx, int64_t n_vec, int dim, int n_embed, float* count_out,
                                       float* sum_out, void* stream) {
    if (!x || !idx || !count_out || !sum_out) return rq_fail(RQAMD_ERR_INVALID, "rq_ema_accumulate: null argument");
    if (n_vec < 0 || n_embed < 1 || dim < 1 || dim > 1024) return rq_fail(RQAMD_ERR_UNSUPPORTED, "rq_ema_accumulate: n_vec >= 0, n_embed >= 1, 1 <= dim <= 1024 needed");
    RQ_LAUNCH(rq_ema_accumulate_kernel, dim3((n_embed + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, idx, (long)n_vec, dim, n_embed, count_out, sum_out);
    return rq_check_launch("rq_ema_accumulate_kernel");
}

extern "C" int rqamd_rq_ema_update(float* cluster_size_ema, float* embed_ema, const float* count, const float* sum, const float* restart_vectors,
                                   int n_embed, int dim, double decay, void* stream) {
    if (!cluster_size_ema || !embed_ema || !count || !sum) return rq_fail(RQAMD_ERR_INVALID, "rq_ema_update: null argument");
    if (n_embed < 1 || dim < 1) return rq_fail(RQAMD_ERR_INVALID, "rq_ema_update: bad shape");
    const long n = (long)n_embed * dim;
    // embed_ema first (it reads the OLD cluster_size_ema for the restart decision), then cluster_size_ema itself
    RQ_LAUNCH(rq_ema_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cluster_size_ema, embed_ema, count, sum,
              restart_vectors, n_embed, dim, (float)decay, (float)(1.0 - decay));
    RQ_TRY(rq_check_launch("rq_ema_update_kernel"));
    RQ_LAUNCH(rq_ema_cs_kernel, dim3((n_embed + 255) / 256), dim3(256), 0, (hipStream_t)stream, cluster_size_ema, count, n_embed, (float)decay,
              (float)(1.0 - decay), restart_vectors ? 1 : 0);
    return rq_check_launch("rq_ema_cs_kernel");
}

extern "C" int rqamd_rq_ema_normalize(const float* cluster_size_ema, const float* embed_ema, const float* n_total, int n_embed, int dim, float eps,
                                      float* weight_out, void* stream) {
    if (!cluster_size_ema || !embed_ema || !n_total || !weight_out) return rq_fail(RQAMD_ERR_INVALID, "rq_ema_normalize: null argument");
    const long n = (long)n_embed * dim;
    RQ_LAUNCH(rq_ema_normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cluster_size_ema, embed_ema, n_total,
              n_embed, dim, eps, weight_out);
    return rq_check_launch("rq_ema_normalize_kernel");
}
