// conv_halo.hip -- 3x3 / stride 1 / pad 1 convolution for the high-resolution RQ-VAE layers (gfx950).
//
// Same arithmetic as the implicit-GEMM conv of gemm.h (bf16 MFMA, fp32 accumulate, bias / residual
// epilogue), different data movement.  The implicit GEMM re-gathers the activation tile once per tap
// (9x) and needs a separate GroupNorm-apply pass in front of it.  Here a workgroup owns an 8 x 32 pixel
// output tile (256 GEMM rows) x 128 output channels and, per 64-channel chunk of the input:
//   * stages the (8+2) x (32+2) HALO patch once (43.5 KB) and reuses it for all 9 taps -- 3.0x less
//     activation staging than 9 separate gathers (ablation in DESIGN.md: staging, not MFMA, bounds the conv);
//   * applies GroupNorm (per image/channel scale+shift prepared from the statistics) and SiLU while the
//     patch goes from registers to LDS, i.e. once per element, so ResnetBlock's norm -> swish -> conv
//     (rqvae/models/rqvae/layers.py:100-120 of the reference) needs no normalised copy of the activation in HBM;
//   * streams the 9 weight tiles W[cout][tap][chunk] (16 KB each) through a double buffer.
// Zero padding is applied AFTER normalisation (F.conv2d pads the normalised tensor), so out-of-image halo
// pixels are stored as zeros.  Fragment reads address the patch at (ty+ky, tx+kx): a per-tap constant
// shift of the pixel index, XOR-swizzled like gemm.h's tiles (conflict-free ds_read_b128).
// 8 wavefronts (4 x 2), wave tile 64 pixels x 64 channels, transposed accumulators and the packed-bf16
// LDS epilogue of gemm.h.  Used when H % 8 == 0, W % 32 == 0, Cin % 64 == 0, Cout % 128 == 0.
#include <stdlib.h>
#include <type_traits>
#include "gemm.h"
#include "rq_common.h"
#include "vae_kernels.h"
#ifndef RQ_CONV_NT            // A/B switch, cache policy of the activations (each layer's output is GBs per chunk, read once by the next layer):
#define RQ_CONV_NT 0          // 1 = output stores non-temporal, 2 = + the halo-piece and residual loads
#endif
#ifndef RQ_CONVOUT_NT         // A/B switch: conv_out's input (4.3 GB per 255-image chunk, read once + halo) with the non-temporal policy
#define RQ_CONVOUT_NT 0
#endif
static __device__ __forceinline__ rq_u128 ld128_act(const void* p) {
#if RQ_CONV_NT >= 2
    return ld128_nt(p);
#else
    return ld128(p);
#endif
}
static __device__ __forceinline__ void st128_act(void* p, rq_u128 v) {
#if RQ_CONV_NT >= 1
    st128_nt(p, v);
#else
    st128(p, v);
#endif
}

struct ConvHaloArgs {
    const bf16_t* x;        // NHWC [B][H][W][Cin] raw (pre-norm) activation, or already-activated when gn == null
    const bf16_t* w;        // [Cout][3][3][Cin]
    const float* bias;      // [Cout]
    const float* gn;        // [B][Cin][2] (scale, shift) or null: y = silu(x*scale + shift) applied on staging
    const bf16_t* resid;    // NHWC [B][H][W][Cout] or null
    bf16_t* out;            // NHWC [B][H][W][Cout]
    float* stats;           // GroupNorm partials of `out`: [B][tiles][32][2] (sum, sumsq) as gn_stats_kernel writes them, or null
    int B, H, W, Cin, Cout;
};

// SiLU of an argument that arrives pre-multiplied by log2 e (the GroupNorm scale / shift are scaled once per chunk):
// t = a log2 e  ->  a / (1 + e^-a) = t / (log2 e + log2 e * 2^-t): fma, v_exp (negated input), fma, v_rcp, mul -- one VALU
// instruction per element less than a * rcp(1 + exp2(-log2e * a)), in the issue-bound fused tap loop.
#define RQ_LOG2E 1.4426950408889634f
static __device__ __forceinline__ float rq_silu_l2(float t) { return t * rq_fast_rcp(fmaf(rq_fast_exp2(-t), RQ_LOG2E, RQ_LOG2E)); }

// GroupNorm partial sums of one 16-byte piece (8 bf16 channels) on its way out of the epilogue: sum and sum of squares per
// channel PAIR straight from the packed words (v_dot2c_f32_bf16: acc += lo * lo' + hi * hi' in fp32, products of two bf16 are
// exact) -- 8 instructions per piece instead of 8 unpacks + 8 adds + 8 fmas (the statistics were ~2400 cycles of a fused tile's
// epilogue, profiles/r02_conv_halo_barrier_timeline.txt).
static __device__ __forceinline__ void rq_stats_piece(const rq_u128& u, float (&sum)[4], float (&sq)[4]) {
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sum[e] = rq_dot2_bf16(w[e], RQ_ONE_X2, sum[e]);
        sq[e] = rq_dot2_bf16(w[e], w[e], sq[e]);
    }
}

constexpr int HT_H = 8, HT_W = 32;                 // output tile
constexpr int HP_W = HT_W + 2;                     // halo patch of the plain conv: 34 x 10 = 340 pixels
constexpr int H_BN = 128;
constexpr int HW_BYTES = H_BN * 128;               // one weight tile: 128 rows x 64 k
// RQ_HALO_WDMA (A/B switch, default 1; round 5): the per-tile kernel's weight units go from global memory straight into a ring of FOUR LDS
// slots by LDS-DMA -- no staging registers, no ds_write -- instead of through three register sets into three slots.
#ifndef RQ_HALO_WDMA
#define RQ_HALO_WDMA 1
#endif
// RQ_HALO_PMID / RQ_HALO_RMID (A/B switches of the DMA form): the next chunk's patch pieces / the residual pieces are requested behind the
// tap's weight request (1) or at the top of the tap (0)
#ifndef RQ_HALO_PMID
#define RQ_HALO_PMID 0
#endif
#ifndef RQ_HALO_RMID
#define RQ_HALO_RMID 1
#endif
#ifndef RQ_HALO_RES_F32
#define RQ_HALO_RES_F32 1
#endif
constexpr int H_WSLOTS = RQ_HALO_WDMA ? 4 : 3;
constexpr int H_SMEM_BYTES = 2 * (HT_H + 2) * HP_W * 128 + H_WSLOTS * HW_BYTES;     // two patch buffers + the weight slots = 133 / 149 KB (every variant)
// counted wait of the DMA form: at most nd LDS-DMAs and no ordinary loads of the wavefront stay in flight (both constants once the tap loop is unrolled)
static __device__ __forceinline__ void halo_wait_dma(int nd, int no) {
#define RQ_HC(ND, K) case K: rq_wait_vmcnt_mixed<ND, K>(); break;
#define RQ_HW(ND) switch (no) { RQ_HC(ND, 0) RQ_HC(ND, 1) RQ_HC(ND, 2) RQ_HC(ND, 3) RQ_HC(ND, 4) RQ_HC(ND, 5) RQ_HC(ND, 6) RQ_HC(ND, 7) RQ_HC(ND, 8) \
    RQ_HC(ND, 9) RQ_HC(ND, 10) RQ_HC(ND, 11) RQ_HC(ND, 12) default: rq_wait_vmcnt_mixed<ND, 0>(); break; }      /* (default: a stronger wait) */
    if (nd >= 2) { RQ_HW(2) } else { RQ_HW(0) }
#undef RQ_HW
#undef RQ_HC
}

// LDS byte offset of 16-byte chunk c8 of patch pixel (hy, hx).  The XOR swizzle depends on the patch COLUMN
// only, so a tap's row shift (and the fragment row i, and the double-buffer index) are plain multiples of 128
// added to one per-kx base address -- they fold into the ds_read immediate, and the k-step is an XOR of bits
// 5..6.  (With the swizzle on the linear pixel index every (tap, i, k-step) needed its own address register:
// 72 of them, which pushed the kernel past 256 VGPRs and into scratch.)
template <int PW = HP_W>
static __device__ __forceinline__ unsigned halo_lds_off(int hy, int hx, int c8) {
    return (unsigned)((hy * PW + hx) * 128 + ((c8 ^ ((hx >> 1) & 7)) << 4));
}

// UPS = 1: the conv reads its input through a nearest-neighbour 2x upsample (Upsample.forward, layers.py:31-35): x is
// [B][H/2][W/2][Cin], output pixel (oy, ox) tap (ky, kx) reads source pixel ((oy+ky-1) >> 1, (ox+kx-1) >> 1).  The
// patch of an 8 x 32 output tile is then only (4+2) x (16+2) source pixels; the fragment address of a lane becomes
// row wm + ((i+ky+1) >> 1) (still an immediate) and column (lane + kx + 1) >> 1 (one base register per kx).
//
// Pipeline of one tap (round 2; the barrier timeline that led to it is profiles/r02_conv_halo_barrier_timeline.txt -- a tap
// took ~1580 cycles for 1024 cycles of MFMA per SIMD: after every barrier the pipe waited for the first LDS fragments, and
// after its last MFMA a wavefront still had its staging stores to issue before it could arrive at the next barrier):
//   * the weight tiles go through a ring of THREE LDS slots (unit g in slot g % 3): during tap t slot t % 3 is read, slot
//     (t+1) % 3 is already published, and slot (t+2) % 3 -- last read in tap t-1 -- is written at the START of the tap
//     from the registers that fetched unit t+2 three taps ago (five units are in flight: two in LDS, three in registers);
//   * the fragments of the NEXT tap's first k-step are read BEFORE the barrier that ends the tap (their slot and patch were
//     published one barrier earlier), so the first four MFMAs after a barrier need no LDS access, and the wait for those
//     reads overlaps the tap's last MFMAs;
//   * the normalised patch pieces of the next chunk are stored in taps 2..7, one barrier before the first fragment read
//     (end of tap 8) that touches them.
// Tile heights of 4 and 16 rows (two workgroups per CU / 128 x 64 wave tiles) were measured in rounds 1-2 and removed
// (profiles/r02_conv_halo_tile_height.txt, profiles/r02_conv_halo_variants.txt).
#ifdef RQ_CONV_TRACE
// Diagnostics build only (scripts/conv_trace.sh): shader-clock stamps of one workgroup's barrier arrivals / releases.
__device__ unsigned long long g_conv_trace[8 * 64];
#define RQ_CT(slot) do { if (blockIdx.x == RQ_CONV_TRACE && lane == 0 && (slot) < 64) g_conv_trace[wave * 64 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define RQ_CT(slot) do { } while (0)
#endif
template <int FUSE_GN, int UPS, int RES>
__global__ __launch_bounds__(512, 1) void conv3x3_halo_kernel(ConvHaloArgs p) {
    constexpr bool WDMA = RQ_HALO_WDMA != 0;
    constexpr int TH = HT_H, NTH = 512, RPW = 2, NJ = 2, W_IT = 2, TPX = TH * HT_W, W_SETS = 3, W_SLOTS = H_WSLOTS;
    static_assert(!(FUSE_GN && UPS), "the upsample conv has no Normalize in front of it");
    constexpr int PW = UPS ? HT_W / 2 + 2 : HP_W, PH = UPS ? TH / 2 + 2 : TH + 2;
    constexpr int HP_N = PW * PH;                  // patch pixels: 340, or 108 through the upsample
    constexpr int HALO_BYTES = HP_N * 128;
    constexpr int H_IT = (HP_N * 8 + NTH - 1) / NTH;
    RQ_DYN_SMEM(smem);
    char* sH = (char*)smem;                        // [2][HALO_BYTES]
    char* sW = sH + 2 * HALO_BYTES;                // [3][HW_BYTES]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;       // 4 x 2 wavefronts: row pairs x cout halves
#ifdef RQ_CONV_TRACE
    if (blockIdx.x == RQ_CONV_TRACE && lane == 0) g_conv_trace[wave * 64 + 58] = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- tile decode: contiguous band of tiles per XCD (neighbouring tiles share halo rows in that L2)
    const int tiles_x = p.W / HT_W, tiles_y = p.H / TH, NT = p.Cout / H_BN;
    const int n_mt = p.B * tiles_y * tiles_x;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int per = (n_mt + 7) >> 3;
    const int mtile = xcd * per + slot / NT;
    const int nt = slot - (slot / NT) * NT;
    if (slot >= per * NT || mtile >= n_mt) return;
    const int img = mtile / (tiles_y * tiles_x);
    const int trem = mtile - img * (tiles_y * tiles_x);
    const int ty0 = (trem / tiles_x) * TH, tx0 = (trem - (trem / tiles_x) * tiles_x) * HT_W;
    const int n0 = nt * H_BN;

    // ---- The tile's first five weight units and its first patch are requested BEFORE the rest of the set-up.  The kernel
    // issued its first global load as instruction ~310 (patch descriptors with their divisions, LDS offsets, the bias, fragment
    // addresses all came first), and a workgroup's second four wavefronts -- the losers of the SIMD's issue arbitration -- got
    // through that code ~6000 cycles after the first four: the prologue barrier waited ~9000 cycles of a 40 000-cycle tile for
    // THEIR loads (barrier timeline, profiles/r03_conv_halo_ab2_timeline.txt).  The loads only need the tile coordinates.
    const char* gX = (const char*)p.x;
    const char* gWt = (const char*)p.w;
    // weight staging: 128 rows x 8 chunks = 1024 chunks, W_IT per thread
    const int w_row = tid >> 3, w_c8 = tid & 7;    // rows w_row + (NTH / 8) i
    unsigned w_goff[W_IT];
#pragma unroll
    for (int i = 0; i < W_IT; ++i) w_goff[i] = (unsigned)(((long)(n0 + w_row + (NTH / 8) * i) * 9 * p.Cin + w_c8 * 8) * 2);
    auto load_w = [&](int c, int tap, rq_u128* rw) {
        const unsigned kb = (unsigned)(tap * p.Cin + c * 64) * 2u;
#pragma unroll
        for (int i = 0; i < W_IT; ++i) rw[i] = ld128(gWt + (w_goff[i] + kb));
    };
    rq_u128 rh[H_IT], rw[W_SETS][W_IT], r01[2][W_IT];
    // DMA form: a unit's 16 KB are sixteen 1-KB wave-instructions, two per wavefront: wavefront w fills rows 16 w .. 16 w + 15 of the
    // slot; LDS position (l & 7) of row r holds chunk (l & 7) ^ ((r >> 1) & 7) (the swizzle of w_loff below), so lane l FETCHES that
    // chunk.  Source = wave-uniform base (tap, chunk) + a loop-invariant per-lane offset: no VALU work per unit.
    unsigned w_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wave * 16 + i * 8 + (lane >> 3);
        w_voff[i] = (unsigned)(((long)(n0 + r) * 9 * p.Cin + ((lane & 7) ^ ((r >> 1) & 7)) * 8) * 2);
    }
    const rq_lds_t sW_dma = rq_lds_addr(sW) + (rq_lds_t)rq_uniform(wave) * 2048u;
    // unit (c, tap) into the slot at byte offset woff of the ring; tap may run into the next chunk
    auto dma_w = [&](int c, int tap, unsigned woff) {
        if (tap >= 9) { tap -= 9; ++c; }
        rq_glds16_s2(sW_dma + woff, gWt + (unsigned)rq_uniform((tap * p.Cin + c * 64) * 2), w_voff[0], w_voff[1]);    // (wave-uniform: the asm wants it in SGPRs)
    };
    if (WDMA) {
        // units 0, 1, 2 of the tile into slots 0, 1, 2
        dma_w(0, 0, 0u);
        dma_w(0, 1, (unsigned)HW_BYTES);
        dma_w(0, 2, 2u * HW_BYTES);
    } else {
        // weight units 0..4 of the tile (0, 1 go to LDS before the first barrier; 2, 3, 4 wait in register sets 2, 0, 1)
        load_w(0, 0, r01[0]);
        load_w(0, 1, r01[1]);
        load_w(0, 2, rw[2]);
        load_w(0, 3, rw[0]);
        load_w(0, 4, rw[1]);
    }
    rq_sched_barrier();

    // ---- halo staging bookkeeping (loop-invariant), ONE packed register per piece: source pixel index inside the image
    // (16 bits: H * W <= 65536, clamped into the image so that every load is readable), LDS offset in 16-byte units
    // (13 bits), "inside the image" (bit 29: else the piece is stored as zeros) and "piece exists" (bit 30).  Piece `it` is
    // 16-byte chunk tid & 7 of patch pixel (tid >> 3) + 64 it: one division for piece 0, then steps of 64 pixels.
    unsigned hd[H_IT];
    const int Hs = UPS ? p.H >> 1 : p.H, Ws = UPS ? p.W >> 1 : p.W;                       // source image
    const unsigned x_img = (unsigned)((long)img * Hs * Ws * p.Cin * 2) + (unsigned)((tid & 7) * 16);   // bytes; < 4 GiB (launcher)
    const unsigned cin2 = (unsigned)p.Cin * 2u;
    const int hp0 = tid >> 3, hy0 = hp0 / PW, hx0 = hp0 - hy0 * PW;
    // pass 1: the clamped source pixel of every piece, and its load on the way at once (chunk 0)
    {
        int hy = hy0, hx = hx0;
#pragma unroll
        for (int it = 0; it < H_IT; ++it) {
            const int gy = (UPS ? ty0 >> 1 : ty0) + hy - 1, gx = (UPS ? tx0 >> 1 : tx0) + hx - 1;
            const int cy = gy < 0 ? 0 : (gy >= Hs ? Hs - 1 : gy), cx = gx < 0 ? 0 : (gx >= Ws ? Ws - 1 : gx);
            hd[it] = (unsigned)(cy * Ws + cx);
            rh[it] = ld128_act(gX + (x_img + hd[it] * cin2));
            hy += (NTH / 8) / PW; hx += (NTH / 8) % PW;
            if (hx >= PW) { hx -= PW; ++hy; }
        }
    }
    const float* gn = FUSE_GN ? p.gn + (long)img * p.Cin * 2 : nullptr;
    // (scale, shift) of this thread's 8 channels of the chunk: the same for all of its pieces (c8 = tid & 7)
    // (fetched raw and scaled by log2 e one tap later, scale_gs: multiplied on arrival, the loads' round trip stood at the head of
    // every chunk's first tap, in front of the MFMAs of both wavefronts of the SIMD)
    f32x4 gs[4];
    auto load_gs = [&](int c) {
        if (FUSE_GN) {
#pragma unroll
            for (int e = 0; e < 4; ++e) gs[e] = *(const f32x4*)(gn + (c * 64 + (tid & 7) * 8 + e * 2) * 2);
        }
    };
    auto scale_gs = [&]() {
        if (FUSE_GN) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { rq_opaque_f4(gs[e]); gs[e] = gs[e] * RQ_LOG2E; }
        }
    };
    load_gs(0);
    rq_sched_barrier();
    // pass 2: LDS offset and flags
    {
        int hp = hp0, hy = hy0, hx = hx0;
        const int c8 = tid & 7;
#pragma unroll
        for (int it = 0; it < H_IT; ++it) {
            const bool in = hp < HP_N;
            const int gy = (UPS ? ty0 >> 1 : ty0) + hy - 1, gx = (UPS ? tx0 >> 1 : tx0) + hx - 1;
            const bool ok = in && gy >= 0 && gy < Hs && gx >= 0 && gx < Ws;
            const unsigned loff16 = in ? halo_lds_off<PW>(hy, hx, c8) >> 4 : 0u;
            hd[it] |= (loff16 << 16) | (ok ? 1u << 29 : 0u) | (in ? 1u << 30 : 0u);
            hp += NTH / 8; hy += (NTH / 8) / PW; hx += (NTH / 8) % PW;
            if (hx >= PW) { hx -= PW; ++hy; }
        }
    }
    auto h_in = [&](int it) { return (hd[it] >> 30) & 1u; };
    auto h_ok = [&](int it) { return (hd[it] >> 29) & 1u; };
    auto h_loff = [&](int it) { return ((hd[it] >> 16) & 0x1fffu) << 4; };
    auto h_goff = [&](int it) { return x_img + (hd[it] & 0xffffu) * cin2; };
    auto load_halo_piece = [&](int c, rq_u128* rh, int it) { rh[it] = ld128_act(gX + (h_goff(it) + (unsigned)c * 128u)); };

    unsigned w_loff[W_IT];
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
        const int r = w_row + (NTH / 8) * i;
        w_loff[i] = (unsigned)(r * 128 + ((w_c8 ^ ((r >> 1) & 7)) << 4));
    }
    // one 16-byte piece of the patch: registers -> (GroupNorm + SiLU) -> LDS.  The fused form is spread over
    // the taps of a chunk (one piece per tap) so that its VALU / transcendental work issues under the MFMAs;
    // done in one lump before the barrier it cost +73 % on the 128->128 @256^2 layer.
    auto halo_piece_value = [&](const rq_u128* rh, int it) -> rq_u128 {
        rq_u128 v = rh[it];
        if (FUSE_GN) {
            // 8 channels of this image: y = silu(x * scale + shift), (scale, shift) pre-multiplied by log2 e (rq_silu_l2)
            float f[8];
            rq_unpack2(v.x, f[0], f[1]);
            rq_unpack2(v.y, f[2], f[3]);
            rq_unpack2(v.z, f[4], f[5]);
            rq_unpack2(v.w, f[6], f[7]);
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x4 ss = gs[e >> 1];                              // (scale, shift) x 2 channels
                const float a = fmaf(f[e], ss[0], ss[1]), b = fmaf(f[e + 1], ss[2], ss[3]);
                f[e] = rq_silu_l2(a);
                f[e + 1] = rq_silu_l2(b);
            }
            v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
            v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
        }
        if (!h_ok(it)) v = zero128();              // zero padding of the (normalised) input
        return v;
    };
    // a quarter of that: channels 2e, 2e+1 of piece `it` (one 32-bit word), for the k-step regions of the tap loop
    auto halo_word = [&](rq_u128* rh, int it, int e) -> uint32_t& { return e == 0 ? rh[it].x : e == 1 ? rh[it].y : e == 2 ? rh[it].z : rh[it].w; };
    auto halo_piece_word = [&](rq_u128* rh, int it, int e) {
        uint32_t& w = halo_word(rh, it, e);
        if (FUSE_GN) {
            const f32x4 ss = gs[e];                                       // (scale, shift) x 2 channels
            float wlo, whi;
            rq_unpack2(w, wlo, whi);
            const float a = fmaf(wlo, ss[0], ss[1]), b = fmaf(whi, ss[2], ss[3]);
            w = pack_bf16x2(rq_silu_l2(a), rq_silu_l2(b));
        }
        if (!h_ok(it)) w = 0u;                     // zero padding of the (normalised) input
    };
    auto store_w = [&](int wslot, const rq_u128* rw) {
#pragma unroll
        for (int i = 0; i < W_IT; ++i) st128(sW + wslot * HW_BYTES + w_loff[i], rw[i]);
    };

    // the accumulators start from the bias (transposed tile: register 4q+e of block j is output channel
    // wn*64 + j*32 + 8q + 4*(lane>>5) + e): its loads hide under the patch prologue instead of opening the epilogue
    f32x16 acc[RPW][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bv = *(const f32x4*)(p.bias + n0 + wn * (NJ * 32) + j * 32 + 8 * q + 4 * (lane >> 5));
#pragma unroll
            for (int i = 0; i < RPW; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] = bv[e];
        }

    const int ftx = lane & 31, fk = lane >> 5;
    // fragment base addresses for k-step 0: weights row = wn*64 + ftx (+32 j); patch pixel (wm*RPW, ftx + kx)
    const unsigned rd_w0 = (unsigned)((wn * (NJ * 32) + ftx) * 128 + ((fk ^ ((ftx >> 1) & 7)) << 4));
    unsigned rd_h0[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
        rd_h0[kx] = UPS ? halo_lds_off<PW>(wm * (RPW / 2), (ftx + kx + 1) >> 1, fk) : halo_lds_off<PW>(wm * RPW, ftx + kx, fk);

    auto load_frags = [&](int hbuf, unsigned woff, int ky, int kx, int ks, bf16x8* af, bf16x8* bfr) {
        const char* hb = sH + ((rd_h0[kx] + (unsigned)(hbuf * HALO_BYTES)) ^ (unsigned)(ks << 5));
        const char* wb = sW + ((rd_w0 ^ (unsigned)(ks << 5)) + woff);        // (woff is a multiple of 16 KB: the k-step XOR of bits 5..6 commutes with it)
#pragma unroll
        for (int i = 0; i < RPW; ++i) af[i] = as_bf16x8(ld128(hb + (UPS ? (i + ky + 1) >> 1 : i + ky) * (PW * 128)));
#pragma unroll
        for (int j = 0; j < NJ; ++j) bfr[j] = as_bf16x8(ld128(wb + j * (32 * 128)));
    };
    // Two fragment sets in ping-pong: while the MFMAs of k-step ks run on set ks & 1, the reads of k-step ks + 1 fill the other
    // (left to itself the compiler re-used ONE set -- read, wait, four MFMAs, read, wait ... -- and every k-step exposed the LDS
    // latency: a tap took ~1500 cycles per SIMD for 1024 cycles of MFMA work, profiles/r02_conv_halo_barrier_timeline.txt).
    // The tap's last k-step reads the NEXT tap's first fragments into set 0, before the barrier that ends the tap.
    // RQ_HALO_FDEPTH (A/B switch, default 1) = how many k-steps ahead the fragments are read: FD + 1 sets, k-step g = 4 tap + ks of a
    // chunk (36 per chunk: the rotation is the same in every chunk) uses set g % (FD + 1).
#ifndef RQ_HALO_FDEPTH
#define RQ_HALO_FDEPTH 1
#endif
    constexpr int FD = RQ_HALO_FDEPTH, NSET = FD + 1;
    static_assert(FD >= 1 && FD <= 3 && 36 % NSET == 0, "fragment read-ahead of 1..3 k-steps");
    bf16x8 fa[NSET][RPW], fb[NSET][NJ];
    auto mfma_step = [&](int set) {
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = rq_mfma_32x32x16_bf16(fb[set][j], fa[set][i], acc[i][j]);   // transposed tile
    };

    // ---- main loop over (64-channel chunk, tap).  Weight unit g = c * 9 + tap lives in register set g % 3 from the tap
    // g - 5 that fetches it to the start of tap g - 2 that stores it into LDS slot g % 3 (9 taps per chunk keep both
    // rotations static under the unrolled tap loop): the L2 round trip (~1 us under load) is longer than a tap.
    const int NC = p.Cin / 64, last_c = NC - 1;
    auto load_unit = [&](int c, int tap, rq_u128* r) {      // unit (c, tap), tap may run into the next chunk; false: past the end
        if (tap >= 9) { tap -= 9; ++c; }
        if (c > last_c) return false;
        load_w(c, tap, r);
        return true;
    };
    // prologue: patch of chunk 0 and weight units 0, 1 (requested at the top of the kernel) go to LDS
    {
        RQ_CT(0);
        scale_gs();
#pragma unroll
        for (int it = 0; it < H_IT; ++it)
            if (h_in(it)) st128(sH + h_loff(it), halo_piece_value(rh, it));
        if (WDMA) rq_wait_vmcnt_mixed<0, 0>();     // units 0, 1, 2 have landed
        else {
            store_w(0, r01[0]);
            store_w(1, r01[1]);
        }
    }
    rq_syncthreads();
    RQ_CT(1);
#pragma unroll
    for (int g = 0; g < FD; ++g) load_frags(0, 0u, 0, 0, g, fa[g], fb[g]);           // k-steps 0 .. FD - 1 of tap 0
    // residual tile (epilogue operand): its 8 pieces per thread are fetched one per tap during the LAST chunk, where the
    // patch registers are idle, so the epilogue starts with the data in hand
    constexpr int CPR = H_BN / 8;
    constexpr int R_IT = TPX * CPR / NTH;          // 8
    static_assert(R_IT == 8, "two residual pieces per tap in the first four taps of the last chunk");
    rq_u128 rr[R_IT];
    const bf16_t* rsrc = p.resid;
    // Piece k of thread tid is 16-byte chunk (tid & 15) of pixel (tid >> 4) of tile row k -- of the residual tile and, in the
    // epilogue, of the output tile (same [B][H][W][Cout] layout): one base offset and a row step.  (Written as cidx = tid +
    // NTH k -> (ml, nl) -> (ty, tx) -> pix the compiler derived every piece's address separately: ~120 VALU instructions in one
    // lump at the start of the last chunk and again in the epilogue.)
    static_assert(CPR == 16 && NTH / CPR == HT_W && R_IT == TH, "piece k = tile row k");
    const unsigned io_off0 = (unsigned)((((long)img * p.H + ty0) * p.W + tx0 + (tid >> 4)) * p.Cout + n0 + (tid & 15) * 8) * 2u;   // bytes; < 4 GiB (launcher)
    const unsigned io_step = (unsigned)p.W * (unsigned)p.Cout * 2u;
    // one chunk of the reduction; LAST (compile time): no next patch to stage -- fetch the residual instead
    auto run_chunk = [&](int c, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int hbuf = c & 1;
        // DMA form: unit u = 9 c + tap lives in slot u & 3 = (c + tap) & 3 -- a run-time rotation (9 taps per chunk, four slots), so the
        // slot offset is a scalar that the fragment addresses add; unit u is requested at the start of tap u - 3 (its slot was read in
        // tap u - 4), must have landed in every wavefront before the barrier that ends tap u - 2, and is first read at the end of tap u - 1
        const unsigned cph = (unsigned)c & 3u;
        auto slot_off = [&](int tap) { return WDMA ? ((cph + (unsigned)tap) & 3u) * (unsigned)HW_BYTES : (unsigned)((tap % 3) * HW_BYTES); };
        // next chunk's patch: one piece per tap, normalised and stored in taps FIRST.. (6 pieces: taps 2..7; through the
        // upsample 2 pieces: taps 6, 7), fetched three taps earlier where the chunk is long enough
        constexpr int NPT = H_IT, FIRST = 8 - NPT;
        static_assert(FIRST >= 0, "one patch piece per tap");
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            // ---- stage: weight unit (c, tap + 2) from its registers into the slot tap t-1 just left
            const bool have_w2 = !(LAST && tap + 2 >= 9), have_w3 = !(LAST && tap + 3 >= 9);
#ifndef RQ_CONV_ABLATE_STAGE        // diagnostics build: no weight staging stores (results are wrong)
            if (!WDMA && have_w2) store_w((tap + 2) % 3, rw[(tap + 2) % W_SETS]);
#endif
            // ---- global prefetches, spread over the taps (every workgroup of a launch is in the same phase: eight residual or
            // six patch pieces per thread issued in ONE tap are a 16-MB chip-wide request burst that stalled the issuing
            // wavefronts for ~1800 cycles): weight unit tap + 5 into the registers just stored; patch piece `it` of the next
            // chunk three taps before the tap that normalises it; one residual piece per tap of the last chunk
            if (!WDMA && have_w2) (void)load_unit(c, tap + 5, rw[(tap + 2) % W_SETS]);
            // the ordinary loads of tap t: the next chunk's GroupNorm words and patch pieces / the residual pieces; n_* = how many
            auto n_patch = [&](int t) {
                int n = 0;
                if (!LAST && t >= 0) {
                    if (t == 0 && FUSE_GN) n += 4;
#pragma unroll
                    for (int it = 0; it < H_IT; ++it) n += ((FIRST + it >= 3 ? FIRST + it - 3 : 0) == t) ? 1 : 0;
                }
                return n;
            };
            auto n_resid = [&](int t) { return LAST && RES && t >= 0 && t < R_IT / 2 ? 2 : 0; };
            auto patch_loads = [&]() {
                if (!LAST) {
                    if (tap == 0) load_gs(c + 1);
#pragma unroll
                    for (int it = 0; it < H_IT; ++it) {
                        const int t_use = FIRST + it, t_load = t_use >= 3 ? t_use - 3 : 0;
                        if (t_load == tap) load_halo_piece(c + 1, rh, it);
                    }
                }
            };
            auto resid_loads = [&]() {
                if (LAST && RES && tap < R_IT / 2) {       // (a template parameter, not `if (p.resid)`: behind a run-time branch the
                                                           // compiler's vmcnt for this tap's store_w assumed no residual loads in
                                                           // flight and so also waited for the weight tile fetched two taps ago)
                    // two pieces per tap in the FIRST four taps: the tile is an old activation (HBM, not the L2), and a piece fetched
                    // in tap 8 was still on its way when the epilogue wanted it (~1200 cycles of the epilogue, barrier timeline in
                    // profiles/r03_conv_halo_barrier_timeline.txt)
                    rr[2 * tap] = ld128_act((const char*)rsrc + (io_off0 + (unsigned)(2 * tap) * io_step));
                    rr[2 * tap + 1] = ld128_act((const char*)rsrc + (io_off0 + (unsigned)(2 * tap + 1) * io_step));
                }
            };
            // DMA form: where the tap's ordinary loads stand relative to its weight request (which follows the first k-step).  The
            // vector-memory counter retires in issue order, so the wait for LAST tap's weight request at the end of this tap also
            // waits for every ordinary load issued before that request: loads that come from HBM (the residual tile) go BEHIND it.
            constexpr bool PMID = WDMA && RQ_HALO_PMID, RMID = WDMA && RQ_HALO_RMID;
            if (!PMID) patch_loads();
            if (!RMID) resid_loads();
            // ordinary loads younger than last tap's weight request at the end of this tap
            const int n_vis = n_patch(tap) + n_resid(tap) + (PMID ? n_patch(tap - 1) : 0) + (RMID ? n_resid(tap - 1) : 0);
            if (!LAST && tap == 1) scale_gs();              // (the next chunk's GroupNorm words, fetched a tap ago, first used in tap FIRST >= 2)
            rq_sched_barrier();
            // ---- four k-step regions: reads of the next k-step (the last one: of the coming tap's first k-step -- its slot and
            // patch were published one barrier ago), four MFMAs, and a quarter of the fused GroupNorm+SiLU arithmetic of the tap's
            // patch piece, whose VALU / transcendental instructions issue in the MFMAs' shadow
            const int pit = tap - FIRST;
            const bool ptap = !LAST && pit >= 0 && pit < H_IT;
            const int pi = ptap ? pit : 0;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#ifdef RQ_CONV_ABLATE_LDS          // diagnostics build: no fragment reads (the MFMAs reuse the prologue's fragments; results are wrong)
                if (false) {}
                else
#endif
                // (the accumulators are made opaque at the head of every region for the same reason as the patch words below:
                // an MFMA is register-only, and before instruction selection nothing kept it from rising above the region's
                // fragment reads -- or above the previous sched_barrier: the reads of k-step ks + 1 then came out BEHIND the MFMAs
                // of k-step ks, back to back with those of ks + 2, and the next MFMA waited out a full LDS round trip)
#pragma unroll
                for (int i = 0; i < RPW; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) rq_opaque_acc(acc[i][j]);
                {
                    // read k-step g + FD: this tap's, the next tap's (its weight slot and the patch were published one barrier ago), or
                    // the next chunk's first (other patch buffer, complete since the barrier that ended tap 7)
                    const int g = tap * 4 + ks + FD, t2 = g >> 2, k2 = g & 3, set2 = g % NSET;
                    if (t2 < 9) load_frags(hbuf, slot_off(t2), t2 / 3, t2 % 3, k2, fa[set2], fb[set2]);
                    else if (!LAST) load_frags(hbuf ^ 1, slot_off(t2), (t2 - 9) / 3, (t2 - 9) % 3, k2, fa[set2], fb[set2]);
                }
                mfma_step((tap * 4 + ks) % NSET);
                // The word's arithmetic is pinned INSIDE this region by making its input opaque here and its result opaque before
                // the region closes: left alone, hipcc hoisted the whole tap's GroupNorm + SiLU (pure register arithmetic, which no
                // sched_barrier holds back before instruction selection) to the top of the tap, or sank it below the last MFMA --
                // 70-150 VALU / transcendental instructions in one lump with the matrix pipe idle on both wavefronts of the SIMD
                // (disassembly before / after: profiles/r03_conv_halo_isa_interleave.txt).
                if (FUSE_GN && ptap) rq_opaque_u(halo_word(rh, pi, ks));
                if (ptap) halo_piece_word(rh, pi, ks);
                if (FUSE_GN && ptap) rq_opaque_u(halo_word(rh, pi, ks));
                // issue order inside the region: the fragment reads FIRST (a full k-step of MFMAs covers their latency), then the
                // MFMAs, each carrying a slice of the piece's VALU work (10 VALU + 4 transcendental instructions per word)
                rq_sched_group(0x100, RPW + NJ);
#pragma unroll
                for (int m = 0; m < RPW * NJ; ++m) {
                    rq_sched_group(0x008, 1);
                    if (FUSE_GN && ptap) { rq_sched_group(0x002, 3); rq_sched_group(0x400, 1); }
                }
                rq_sched_barrier();
                // DMA form: unit tap + 3 into the slot tap - 1 just left.  Requested BEHIND the tap's first k-step: the compiler's own
                // wait for the patch piece this tap normalises (vmcnt(k), k = the ordinary loads issued since) stands in that k-step and
                // would otherwise also wait for this request, which is younger than every one of them.
                if (WDMA && ks == 0) {
                    if (have_w3) dma_w(c, tap + 3, slot_off(tap + 3));
                    if (PMID) patch_loads();
                    if (RMID) resid_loads();
                    rq_sched_barrier();
                }
            }
            if (ptap) { if (h_in(pi)) st128(sH + (hbuf ^ 1) * HALO_BYTES + h_loff(pi), rh[pi]); }
            // DMA form: unit tap + 2 (requested one tap ago) has landed before the barrier publishes it; this tap's own requests stay in flight
            if (WDMA && have_w2) halo_wait_dma(have_w3 ? 2 : 0, n_vis);
            RQ_CT(2 + (c * 9 + tap) * 2);
            rq_syncthreads();
            RQ_CT(3 + (c * 9 + tap) * 2);
        }
    };
    for (int c = 0; c + 1 < NC; ++c) run_chunk(c, std::false_type{});
    run_chunk(NC - 1, std::true_type{});

    // ---- epilogue: (+ residual before the single rounding), packed bf16 tile in LDS, 16-byte stores
    constexpr int LDR = H_BN * 2 + 16;
    static_assert(TPX * LDR + 4096 <= 160 * 1024, "epilogue tile must fit the CU's LDS (the launcher allocates max(staging, epilogue))");
    char* sT = (char*)smem;
    // (the loop ended with a barrier: all waves are done with the operand buffers)
    // RQ_HALO_RES_F32 (A/B switch, default 1; round 5): with a residual the accumulators cross the LDS as fp32 and meet the residual piece
    // -- which the streaming thread already holds in registers -- on the way out: 16 ds_write_b128 + 16 ds_read_b128 per lane and one
    // barrier, instead of the residual's round trip through the bf16 tile (8 ds_write_b128, a barrier, 16 + 16 eight-byte accesses, a
    // barrier, 8 ds_read_b128).  Same fp32 sum, same single rounding: bit-identical.  Row = 32 sixteen-byte units (4 channels each);
    // unit u sits in slot (u >> 1) + 16 (u & 1), so that the two units of a thread's 8-channel piece are 256 bytes apart and each
    // ds_read_b128 of 16 lanes covers 256 contiguous bytes.
    constexpr bool RF32 = RES && RQ_HALO_RES_F32;
    constexpr int LDF = H_BN * 4 + 16;
    static_assert(!RF32 || TPX * LDF <= H_SMEM_BYTES, "the fp32 tile overlays the operand buffers");
    if (RF32) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int ml = (wm * RPW + i) * HT_W + (lane & 31);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int u = wn * (NJ * 8) + j * 8 + 2 * q + (lane >> 5);            // 4-channel unit of the row
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
                    *(f32x4*)(sT + ml * LDF + ((u >> 1) + 16 * (u & 1)) * 16) = v;
                }
        }
    }
    if (RES && !RF32) {
        // the residual tile (fetched as row-contiguous 16-byte pieces during the last taps) waits in the LDS tile, where
        // the lane that owns an 8-byte slot adds it in fp32 before the single rounding and overwrites it in place (the
        // per-lane 8-byte global reads of the first version touched 32 cache lines per wavefront load: +46 us on 173)
#pragma unroll
        for (int k = 0; k < R_IT; ++k) st128(sT + ((tid >> 4) + HT_W * k) * LDR + (tid & 15) * 16, rr[k]);
        rq_syncthreads();
    }
    // the lane's residual values, ALL read before the first packed value is written back: each 8-byte slot is read and then
    // overwritten in place, and the compiler will not move a later read above an earlier write to the same tile -- slot by slot
    // that was 32 serialised LDS round trips (~2 800 cycles of a fused + residual tile's epilogue)
    uint32_t rres[RPW][NJ][4][2];
    if (RES && !RF32) {
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ml = (wm * RPW + i) * HT_W + (lane & 31);
                    const int nl = wn * (NJ * 32) + j * 32 + 8 * q + 4 * (lane >> 5);
                    const uint32_t* rp = (const uint32_t*)(sT + ml * LDR + nl * 2);
                    rres[i][j][q][0] = rp[0];
                    rres[i][j][q][1] = rp[1];
                }
        rq_sched_barrier();
    }
    if (!RF32)
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int ty = wm * RPW + i, tx = lane & 31;
        const int ml = ty * HT_W + tx;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = wn * (NJ * 32) + j * 32 + 8 * q + 4 * (lane >> 5);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
                if (RES && !RF32) {
                    const uint32_t r0 = rres[i][j][q][0], r1 = rres[i][j][q][1];
                    { float lo_, hi_; rq_unpack2(r0, lo_, hi_); v[0] += lo_; v[1] += hi_; }
                    { float lo_, hi_; rq_unpack2(r1, lo_, hi_); v[2] += lo_; v[3] += hi_; }
                }
                struct __attribute__((aligned(8))) u64 { uint32_t a, b; } w2;
                w2.a = pack_bf16x2(v[0], v[1]);
                w2.b = pack_bf16x2(v[2], v[3]);
                *(u64*)(sT + ml * LDR + nl * 2) = w2;
            }
        }
    }
    rq_syncthreads();
    RQ_CT(61);
    // Thread tid streams chunk (tid & 15) of pixels (tid >> 4) + 32 k.  The GroupNorm statistics of the NEXT layer are
    // taken here from the rounded bf16 values on their way out (what a separate gn_stats pass would read back from
    // HBM: ~30 us per image over the decoder): per-thread (sum, sumsq) over its 8 pixels, folded to the 1-2 groups
    // its 8 channels belong to, reduced over the 32 threads of the same chunk, one partial per (tile, group).
    float gs_[4], gq_[4];                           // (sum, sum of squares) per channel PAIR of the thread's 8 channels
#pragma unroll
    for (int e = 0; e < 4; ++e) { gs_[e] = 0.f; gq_[e] = 0.f; }
#pragma unroll
    for (int k = 0; k < R_IT; ++k) {
        rq_u128 u;
        if (RF32) {
            const char* src = sT + ((tid >> 4) + HT_W * k) * LDF + (tid & 15) * 16;
            f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 256);
            const rq_u128 r = rr[k];
            { float lo_, hi_; rq_unpack2(r.x, lo_, hi_); a[0] += lo_; a[1] += hi_; }
            { float lo_, hi_; rq_unpack2(r.y, lo_, hi_); a[2] += lo_; a[3] += hi_; }
            { float lo_, hi_; rq_unpack2(r.z, lo_, hi_); b[0] += lo_; b[1] += hi_; }
            { float lo_, hi_; rq_unpack2(r.w, lo_, hi_); b[2] += lo_; b[3] += hi_; }
            u.x = pack_bf16x2(a[0], a[1]); u.y = pack_bf16x2(a[2], a[3]);
            u.z = pack_bf16x2(b[0], b[1]); u.w = pack_bf16x2(b[2], b[3]);
        } else {
            u = ld128(sT + ((tid >> 4) + HT_W * k) * LDR + (tid & 15) * 16);
        }
        st128_act((char*)p.out + (io_off0 + (unsigned)k * io_step), u);
        if (p.stats) rq_stats_piece(u, gs_, gq_);
    }
    if (p.stats) {                                  // uniform
        __shared__ float sred[NTH / 64][16][4];
        const int gsz = p.Cout / 32;                // channels per group: 4, 8 or 16 (Cout = 128, 256, 512)
        float a0, q0, a1, q1;                       // pair 0 = channels 0..3 (gsz 4) or 0..7; pair 1 = channels 4..7 (gsz 4)
        if (gsz == 4) {
            a0 = gs_[0] + gs_[1]; q0 = gq_[0] + gq_[1];
            a1 = gs_[2] + gs_[3]; q1 = gq_[2] + gq_[3];
        } else {
            a0 = (gs_[0] + gs_[1]) + (gs_[2] + gs_[3]);
            q0 = (gq_[0] + gq_[1]) + (gq_[2] + gq_[3]);
            a1 = 0.f; q1 = 0.f;
        }
        // lanes l, l^16, l^32, l^48 hold the same chunk
        a0 += rq_shfl_xor(a0, 16); q0 += rq_shfl_xor(q0, 16); a1 += rq_shfl_xor(a1, 16); q1 += rq_shfl_xor(q1, 16);
        a0 += rq_shfl_xor(a0, 32); q0 += rq_shfl_xor(q0, 32); a1 += rq_shfl_xor(a1, 32); q1 += rq_shfl_xor(q1, 32);
        if (lane < 16) { sred[wave][lane][0] = a0; sred[wave][lane][1] = q0; sred[wave][lane][2] = a1; sred[wave][lane][3] = q1; }
        rq_syncthreads();
        if (wave == 0) {                            // whole wavefront (the shuffle below needs all lanes); lanes >= 32 duplicate
            // lane = (chunk, pair); channels n0 + chunk*8 + pair*4 ... belong to group (n0 + chunk*8 + pair*4) / gsz
            const int chunk = (lane >> 1) & 15, pair = lane & 1;
            float a = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < NTH / 64; ++w) { a += sred[w][chunk][pair * 2]; q += sred[w][chunk][pair * 2 + 1]; }
            if (gsz == 16) {                        // a group spans two chunks: even chunk collects its neighbour
                a += rq_shfl_xor(a, 2);
                q += rq_shfl_xor(q, 2);
            }
            const bool writer = gsz == 4 ? true : (gsz == 8 ? pair == 0 : (pair == 0 && (chunk & 1) == 0));
            if (writer && lane < 32) {
                const int g = (n0 + chunk * 8 + pair * 4) / gsz;
                const int tiles = tiles_x * tiles_y;
                float* o = p.stats + (((long)img * tiles + trem) * 32 + g) * 2;
                o[0] = a;
                o[1] = q;
            }
        }
    }
    RQ_CT(63);
#ifdef RQ_CONV_TRACE
    if (blockIdx.x == RQ_CONV_TRACE && lane == 0) g_conv_trace[wave * 64 + 59] = __builtin_amdgcn_s_memrealtime();
#endif
}

// -------------------------------------------------------------------------------------------------
// Persistent form of the 8-row kernel above (round 2).  A tile is short (20-27 us) and its phases are serial inside the
// one workgroup a CU holds: first loads of the patch / weights (~2.5 us of exposed latency), 18 taps, epilogue (~4.4 us,
// most of it waiting for the stores to be accepted before the workgroup may retire and the next one may start).  Here a
// workgroup stays on its CU and walks every 32nd slot of its XCD's band of tiles; during the LAST chunk of a tile it stages
// the NEXT tile's first patch (into the other patch buffer) and weight tiles exactly as it stages the next chunk inside a
// tile, so that the next tile's first tap follows the epilogue with no load in between, and the epilogue's global stores
// drain while that tile's MFMAs run (nothing waits for them until a younger load is needed, two taps later).
// LDS (all 160 KB): [ W0 W1 | P0 | spare | P1 ].  The epilogue tile (68 KB) must not touch the buffer that already holds the
// next tile's patch: it lies over the OTHER patch buffer and the spare region next to it (P0 + spare, or spare + P1).
constexpr int PK_SMEM = 160 * 1024;
constexpr int PK_PATCH = (HT_H + 2) * HP_W * 128;                  // 43 520
constexpr int PK_P0 = 2 * HW_BYTES;                                // 32 768
constexpr int PK_P1 = PK_SMEM - PK_PATCH;                          // 120 320
constexpr int PK_PSTRIDE = PK_P1 - PK_P0;
constexpr int PK_LDR = H_BN * 2 + 16;
constexpr int PK_TILE = HT_H * HT_W * PK_LDR;                      // 69 632
constexpr int PK_RED = 8 * 16 * 4 * 4;                             // statistics scratch behind the tile
static_assert(PK_P0 + PK_PATCH + PK_TILE + PK_RED <= PK_SMEM, "epilogue tile over spare + P1");
static_assert(PK_P0 + PK_TILE + PK_RED <= PK_P1, "epilogue tile over P0 + spare");

struct HaloTile { int img, ty0, tx0, n0, trem, cls; };      // (UPS = 2: ty0 / tx0 in SOURCE pixels, cls = output parity (oy & 1) * 2 + (ox & 1))

// UPS = 2 (round 5): the nearest-2x upsample conv by SUB-PIXEL DECOMPOSITION.  Output pixel (2y + py, 2x + px) of a 3 x 3 conv over the
// upsampled image reads source rows y + py - 1 and y + py only (taps ky = 0 | 1, 2 for py = 0; 0, 1 | 2 for py = 1) and likewise two
// source columns: for each of the four output parities the layer is a 2 x 2 conv over the SOURCE image with the taps that share a
// source pixel summed -- 4 taps per output pixel instead of 9.  A tile is one parity class of an 8 x 32 block of source pixels (256
// output pixels, the 16 x 64 output block's every-other-pixel lattice): the plain kernel's patch geometry, four taps per chunk at patch
// offsets (py + a, px + b), weights from the pre-summed tensor [class][Cout][2][2][Cin] (ups_subpixel_weights_kernel), outputs and
// statistics written to the class's lattice.
template <int FUSE_GN, int UPS, int RES>
__global__ __launch_bounds__(512, 1) void conv3x3_halo_pk_kernel(ConvHaloArgs p, int wpx) {
    constexpr int TH = HT_H, NTH = 512, RPW = 2, W_IT = 2, TPX = TH * HT_W;
    constexpr bool SUB = UPS == 2;
    constexpr int TAPS = SUB ? 4 : 9, W_SETS = SUB ? 4 : 3;           // (W_SETS divides TAPS: the register-set rotation is static under the unrolled tap loop)
    static_assert(!(FUSE_GN && UPS), "the upsample conv has no Normalize in front of it");
    constexpr int PW = UPS == 1 ? HT_W / 2 + 2 : HP_W, PH = UPS == 1 ? TH / 2 + 2 : TH + 2;
    constexpr int HP_N = PW * PH;
    constexpr int H_IT = (HP_N * 8 + NTH - 1) / NTH;
    RQ_DYN_SMEM(smem);
    char* sW = (char*)smem;                         // [2][HW_BYTES]
    char* sH = (char*)smem + PK_P0;                 // patch buffer b at b * PK_PSTRIDE
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- this workgroup's tiles: slots w, w + wpx, ... of its XCD's contiguous band (neighbours share halo rows in that L2)
    // (SUB: tiles of the SOURCE image x 4 parity classes, classes adjacent: the four tiles of a source block share its patch in the L2)
    const int tiles_x = (SUB ? p.W >> 1 : p.W) / HT_W, tiles_y = (SUB ? p.H >> 1 : p.H) / TH, NT = p.Cout / H_BN;
    constexpr int NCLS = SUB ? 4 : 1;
    const int n_mt = p.B * tiles_y * tiles_x * NCLS;
    const int xcd = blockIdx.x & 7, w0 = blockIdx.x >> 3;
    const int per = (n_mt + 7) >> 3, nslot = per * NT;
    auto slot_ok = [&](int s) { return s < nslot && xcd * per + s / NT < n_mt; };
    auto decode = [&](int s) {
        HaloTile t;
        const int mtile = xcd * per + s / NT;
        t.n0 = (s - (s / NT) * NT) * H_BN;
        const int st = SUB ? mtile >> 2 : mtile;
        t.cls = SUB ? mtile & 3 : 0;
        t.img = st / (tiles_y * tiles_x);
        const int trem = st - t.img * (tiles_y * tiles_x);
        t.trem = trem * NCLS + t.cls;                      // index of the tile's statistics partial inside its image
        t.ty0 = (trem / tiles_x) * TH;
        t.tx0 = (trem - (trem / tiles_x) * tiles_x) * HT_W;
        return t;
    };
    int slot = w0;
    if (!slot_ok(slot)) return;

    // ---- staging state: the tile whose patch is being loaded (the current tile, or the next one during the last chunk)
    unsigned hd[H_IT];
    unsigned x_img = 0;
    const float* gn = nullptr;
    const int Hs = UPS ? p.H >> 1 : p.H, Ws = UPS ? p.W >> 1 : p.W;
    auto set_staging = [&](const HaloTile& t) {
        int tid_s = tid;                            // opaque: keeps the tile-independent half of this out of loop-carried registers
        rq_opaque(tid_s);
#pragma unroll
        for (int it = 0; it < H_IT; ++it) {
            const int q = tid_s + NTH * it;
            const int hp = q >> 3, c8 = q & 7;
            const bool in = hp < HP_N;
            const int hy = hp / PW, hx = hp - hy * PW;
            const int gy = (UPS == 1 ? t.ty0 >> 1 : t.ty0) + hy - 1, gx = (UPS == 1 ? t.tx0 >> 1 : t.tx0) + hx - 1;
            const bool ok = in && gy >= 0 && gy < Hs && gx >= 0 && gx < Ws;
            const int cy = gy < 0 ? 0 : (gy >= Hs ? Hs - 1 : gy), cx = gx < 0 ? 0 : (gx >= Ws ? Ws - 1 : gx);
            const unsigned loff16 = in ? halo_lds_off<PW>(hy, hx, c8) >> 4 : 0u;
            hd[it] = (unsigned)(cy * Ws + cx) | (loff16 << 16) | (ok ? 1u << 29 : 0u) | (in ? 1u << 30 : 0u);
        }
        x_img = (unsigned)((long)t.img * Hs * Ws * p.Cin * 2) + (unsigned)((tid & 7) * 16);
        if (FUSE_GN) gn = p.gn + (long)t.img * p.Cin * 2;
    };
    // the same for a later tile, from the descriptors already in registers: a piece's patch position is recovered from its LDS
    // offset (tile-independent), so the full derivation above is paid once per workgroup, not once per tile (it cost ~1800
    // cycles of VALU per tile when every tile recomputed it, profiles/r02_conv_halo_barrier_timeline.txt)
    auto restage = [&](const HaloTile& t) {
        int tid_s = tid;                            // opaque, as in set_staging
        rq_opaque(tid_s);
        int hp = tid_s >> 3, hy = hp / PW, hx = hp - hy * PW;      // piece `it` = patch pixel (tid >> 3) + 64 it: one division, then steps
#pragma unroll
        for (int it = 0; it < H_IT; ++it) {
            const unsigned in = (hd[it] >> 30) & 1u, lo = hd[it] & 0x1fff0000u;
            const int gy = (UPS == 1 ? t.ty0 >> 1 : t.ty0) + hy - 1, gx = (UPS == 1 ? t.tx0 >> 1 : t.tx0) + hx - 1;
            const bool ok = in && gy >= 0 && gy < Hs && gx >= 0 && gx < Ws;
            const int cy = gy < 0 ? 0 : (gy >= Hs ? Hs - 1 : gy), cx = gx < 0 ? 0 : (gx >= Ws ? Ws - 1 : gx);
            hd[it] = (unsigned)(cy * Ws + cx) | lo | (ok ? 1u << 29 : 0u) | (in << 30);
            hy += (NTH / 8) / PW; hx += (NTH / 8) % PW;
            if (hx >= PW) { hx -= PW; ++hy; }
        }
        x_img = (unsigned)((long)t.img * Hs * Ws * p.Cin * 2) + (unsigned)((tid & 7) * 16);
        if (FUSE_GN) gn = p.gn + (long)t.img * p.Cin * 2;
    };
    const unsigned cin2 = (unsigned)p.Cin * 2u;
    auto h_in = [&](int it) { return (hd[it] >> 30) & 1u; };
    auto h_ok = [&](int it) { return (hd[it] >> 29) & 1u; };
    auto h_loff = [&](int it) { return ((hd[it] >> 16) & 0x1fffu) << 4; };
    auto h_goff = [&](int it) { return x_img + (hd[it] & 0xffffu) * cin2; };
    const int w_row = tid >> 3, w_c8 = tid & 7;
    unsigned w_goff[W_IT], w_loff[W_IT];
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
        const int r = w_row + (NTH / 8) * i;
        w_goff[i] = (unsigned)(((long)r * TAPS * p.Cin + w_c8 * 8) * 2);
        w_loff[i] = (unsigned)(r * 128 + ((w_c8 ^ ((r >> 1) & 7)) << 4));
    }
    const unsigned w_per_n = (unsigned)TAPS * (unsigned)p.Cin * 2u;            // bytes per output channel
    const unsigned w_per_cls = (unsigned)p.Cout * w_per_n;                   // SUB: bytes per parity class of the pre-summed weights
    const char* gX = (const char*)p.x;
    const char* gWt = (const char*)p.w;

    f32x4 gs[4];
    auto load_halo_piece = [&](int c, rq_u128* rh, int it) { rh[it] = ld128_act(gX + (h_goff(it) + (unsigned)c * 128u)); };
    auto load_gs = [&](int c) {
        if (FUSE_GN) {
#pragma unroll
            for (int e = 0; e < 4; ++e) gs[e] = *(const f32x4*)(gn + (c * 64 + (tid & 7) * 8 + e * 2) * 2) * RQ_LOG2E;
        }
    };
    auto load_halo = [&](int c, rq_u128* rh) {
#pragma unroll
        for (int it = 0; it < H_IT; ++it) load_halo_piece(c, rh, it);
        load_gs(c);
    };
    auto halo_piece_value = [&](const rq_u128* rh, int it) -> rq_u128 {
        rq_u128 v = rh[it];
        if (FUSE_GN) {
            float f[8];
            rq_unpack2(v.x, f[0], f[1]);
            rq_unpack2(v.y, f[2], f[3]);
            rq_unpack2(v.z, f[4], f[5]);
            rq_unpack2(v.w, f[6], f[7]);
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x4 ss = gs[e >> 1];
                const float a = fmaf(f[e], ss[0], ss[1]), b = fmaf(f[e + 1], ss[2], ss[3]);
                f[e] = rq_silu_l2(a);
                f[e + 1] = rq_silu_l2(b);
            }
            v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
            v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
        }
        if (!h_ok(it)) v = zero128();
        return v;
    };
    auto load_w = [&](unsigned nbase, int c, int tap, rq_u128* rw) {
        const unsigned kb = nbase + (unsigned)(tap * p.Cin + c * 64) * 2u;
#pragma unroll
        for (int i = 0; i < W_IT; ++i) rw[i] = ld128(gWt + (w_goff[i] + kb));
    };
    auto store_w = [&](int buf, const rq_u128* rw) {
#pragma unroll
        for (int i = 0; i < W_IT; ++i) st128(sW + buf * HW_BYTES + w_loff[i], rw[i]);
    };

    const int ftx = lane & 31, fk = lane >> 5;
    const unsigned rd_w0 = (unsigned)((wn * 64 + ftx) * 128 + ((fk ^ ((ftx >> 1) & 7)) << 4));
    unsigned rd_h0[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
        rd_h0[kx] = UPS == 1 ? halo_lds_off<PW>(wm * (RPW / 2), (ftx + kx + 1) >> 1, fk) : halo_lds_off<PW>(wm * RPW, ftx + kx, fk);
    // SUB: tap (a, b) of parity class (py, px) reads patch pixel (row + py + a, col + px + b): per tile, the two column bases with the
    // class's row offset folded in (a multiple of 128 bytes: the swizzle bits are untouched)
    unsigned rd_hs[2] = {0u, 0u};
    auto set_class = [&](int cls) {
        const unsigned rowoff = (unsigned)((cls >> 1) * (PW * 128));
        rd_hs[0] = ((cls & 1) ? rd_h0[1] : rd_h0[0]) + rowoff;
        rd_hs[1] = ((cls & 1) ? rd_h0[2] : rd_h0[1]) + rowoff;
    };

    f32x16 acc[RPW][2];
    // accumulators start from the bias of the tile's output channels (as in conv3x3_halo_kernel)
    auto acc_from_bias = [&](int n0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *(const f32x4*)(p.bias + n0 + wn * 64 + j * 32 + 8 * q + 4 * (lane >> 5));
#pragma unroll
                for (int i = 0; i < RPW; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] = bv[e];
            }
    };
    // one tap: four k-steps on two fragment sets in ping-pong (the reads of k-step ks + 1 are issued ahead of the MFMAs of k-step
    // ks, held there by opaque accumulators as in conv3x3_halo_kernel; with one set every k-step waited out its own LDS round trip)
    auto compute = [&](int hbuf, int wbuf, int ky, int kx, rq_u128* rhp, int pit) {      // pit >= 0: patch piece normalised under this tap
        const unsigned ha = (SUB ? rd_hs[kx] : rd_h0[kx]) + (unsigned)(hbuf * PK_PSTRIDE);          // (SUB: ky, kx = the tap's (a, b) in 0..1)
        const unsigned wa = rd_w0 + (unsigned)(wbuf * HW_BYTES);
        bf16x8 af[2][RPW], bfr[2][2];
        auto load_frags = [&](int ks, bf16x8* a, bf16x8* b) {
            const char* hb = sH + (ha ^ (unsigned)(ks << 5));
            const char* wb = sW + (wa ^ (unsigned)(ks << 5));
#pragma unroll
            for (int i = 0; i < RPW; ++i) a[i] = as_bf16x8(ld128(hb + (UPS == 1 ? (i + ky + 1) >> 1 : i + ky) * (PW * 128)));
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = as_bf16x8(ld128(wb + j * (32 * 128)));
        };
        load_frags(0, af[0], bfr[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < RPW; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) rq_opaque_acc(acc[i][j]);
            if (ks < 3) load_frags(ks + 1, af[(ks + 1) & 1], bfr[(ks + 1) & 1]);
#pragma unroll
            for (int i = 0; i < RPW; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = rq_mfma_32x32x16_bf16(bfr[ks & 1][j], af[ks & 1][i], acc[i][j]);
            if (FUSE_GN && pit >= 0) {          // one 32-bit word (two channels) of the piece per k-step region, as in conv3x3_halo_kernel
                uint32_t& w = ks == 0 ? rhp[pit].x : ks == 1 ? rhp[pit].y : ks == 2 ? rhp[pit].z : rhp[pit].w;
                rq_opaque_u(w);
                const f32x4 ss = gs[ks];
                float wlo, whi;
                rq_unpack2(w, wlo, whi);
                const float ga = fmaf(wlo, ss[0], ss[1]), gb = fmaf(whi, ss[2], ss[3]);
                w = pack_bf16x2(rq_silu_l2(ga), rq_silu_l2(gb));
                if (!h_ok(pit)) w = 0u;
                rq_opaque_u(w);
            }
            if (ks < 3) rq_sched_group(0x100, RPW + 2);
            if (FUSE_GN && pit >= 0) {
#pragma unroll
                for (int m = 0; m < 2 * RPW; ++m) { rq_sched_group(0x008, 1); rq_sched_group(0x002, 3); rq_sched_group(0x400, 1); }
            } else {
                rq_sched_group(0x008, 2 * RPW);
            }
            rq_sched_barrier();
        }
    };

    const int NC = p.Cin / 64, last_c = NC - 1;
    rq_u128 rh[H_IT], rw[W_SETS][W_IT];
    constexpr int CPR = H_BN / 8;
    constexpr int R_IT = TPX * CPR / NTH;          // 8
    rq_u128 rr[R_IT];

    // ---- first tile: the only exposed prologue
    HaloTile cur = decode(slot), nxt = cur;
    bool has_next = slot_ok(slot + wpx);
    if (has_next) nxt = decode(slot + wpx);
    unsigned wn_cur = (unsigned)cur.n0 * w_per_n + (unsigned)cur.cls * w_per_cls, wn_nxt = (unsigned)nxt.n0 * w_per_n + (unsigned)nxt.cls * w_per_cls;
    set_staging(cur);
    if (SUB) set_class(cur.cls);
    acc_from_bias(cur.n0);
    load_halo(0, rh);
    // weight units run on across tiles: unit (c, tap) of the tile, then (0, 0..) of the next one; always set tap % 3
    auto load_unit = [&](int c, int tap, rq_u128* r) {
        unsigned nb = wn_cur;
        if (tap >= TAPS) { tap -= TAPS; ++c; }
        if (c > last_c) {
            if (has_next) { c = 0; nb = wn_nxt; }
            else { c = last_c; tap = TAPS - 1; }    // past the end of this workgroup's work: harmless reload
        }
        load_w(nb, c, tap, r);
    };
#pragma unroll
    for (int u = 0; u < W_SETS; ++u) load_unit(0, u, rw[u]);
#pragma unroll
    for (int it = 0; it < H_IT; ++it)
        if (h_in(it)) st128(sH + h_loff(it), halo_piece_value(rh, it));
    store_w(0, rw[0]);
    rq_syncthreads();
    int wbuf = 0, hbuf = 0;

    int tid_o = tid;                               // opaque per-tile copy of the thread index (see the tile loop)
    unsigned io_off0 = 0;                          // byte offset of this thread's piece 0 of the current tile's residual / output
    const unsigned io_step = (unsigned)p.W * (unsigned)p.Cout * (SUB ? 4u : 2u);      // (SUB: consecutive tile rows are two output rows apart)
#ifdef RQ_CONV_TRACE
    int trace_tile = 0;
#define RQ_CTP(slot) do { if (blockIdx.x == (RQ_CONV_TRACE & 255) && trace_tile == 2 && lane == 0 && (slot) < 64) g_conv_trace[wave * 64 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define RQ_CTP(slot) do { } while (0)
#endif
    auto run_chunk = [&](int c, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        // the last chunk stages the next tile's first patch; with no next tile the descriptors are cleared ("piece does not
        // exist"), so that the staging code below stays branch-free inside the MFMA scheduling regions
        if (LAST) {
            if (has_next) restage(nxt);
            else {
#pragma unroll
                for (int it = 0; it < H_IT; ++it) hd[it] = 0u;
            }
        }
        // patch pieces of the next chunk: PPT per tap in the chunk's last NPT taps (9 taps: one per tap; SUB, 4 taps: two per tap)
        constexpr int PPT = (H_IT + TAPS - 2) / (TAPS - 1), NPT = (H_IT + PPT - 1) / PPT, FIRST = TAPS - NPT;
        static_assert(R_IT <= 8 && FIRST >= 0, "one residual piece per tap; the pieces fit the chunk's taps");
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int ky = SUB ? tap >> 1 : tap / 3, kx = SUB ? tap & 1 : tap - ky * 3;
            // loads spread over the taps as in conv3x3_halo_kernel: weight tile three units ahead, patch piece `it` three taps
            // before the tap that normalises and stores it, one residual piece per tap of the last chunk
            load_unit(c, tap + W_SETS, rw[tap % W_SETS]);
            if (tap == 0) load_gs(LAST ? 0 : c + 1);
#pragma unroll
            for (int it = 0; it < H_IT; ++it) {
                constexpr int AHEAD = SUB ? 2 : 3;            // taps between a piece's request and its store
                const int t_use = FIRST + it / PPT, t_load = t_use >= AHEAD ? t_use - AHEAD : 0;
                if (t_load == tap) load_halo_piece(LAST ? 0 : c + 1, rh, it);
            }
            if (LAST && RES && tap >= 1) rr[tap - 1] = ld128_act((const char*)p.resid + (io_off0 + (unsigned)(tap - 1) * io_step));
            rq_sched_barrier();
            const bool ptap = tap >= FIRST;
            static_assert(PPT == 1 || !FUSE_GN, "the fused form normalises one patch piece per tap");
            if (FUSE_GN) {
                compute(hbuf, wbuf, ky, kx, rh, ptap && tap - FIRST < H_IT ? tap - FIRST : -1);
            } else {
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const int it = (tap - FIRST) * PPT + k;
                    if (ptap && it < H_IT) rh[it] = halo_piece_value(rh, it);
                }
                compute(hbuf, wbuf, ky, kx, rh, -1);
            }
            store_w(wbuf ^ 1, rw[(tap + 1) % W_SETS]);
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int it = (tap - FIRST) * PPT + k;
                if (!(ptap && it < H_IT)) continue;
                if (h_in(it)) st128(sH + (hbuf ^ 1) * PK_PSTRIDE + h_loff(it), rh[it]);
            }
            RQ_CTP(2 + (c * 9 + tap) * 2);
            rq_syncthreads();
            RQ_CTP(3 + (c * 9 + tap) * 2);
            wbuf ^= 1;
        }
        hbuf ^= 1;
    };

    for (;;) {
        // the epilogue's addressing is the same for every tile; recomputed from an opaque copy of the thread index each
        // round, or the compiler hoists ~100 registers of loop-invariant offsets across the main loop (and spills them)
        rq_opaque(tid_o);
        // piece k of a thread = 16-byte chunk (tid & 15) of pixel (tid >> 4) of tile row k, residual and output alike
        static_assert(CPR == 16 && NTH / CPR == HT_W && R_IT == TH, "piece k = tile row k");
        if (SUB)      // output pixel (2 (ty0 + k) + py, 2 (tx0 + column) + px) of the class's lattice
            io_off0 = (unsigned)((((long)cur.img * p.H + 2 * cur.ty0 + (cur.cls >> 1)) * p.W + 2 * (cur.tx0 + (tid_o >> 4)) + (cur.cls & 1)) * p.Cout + cur.n0 + (tid_o & 15) * 8) * 2u;
        else
            io_off0 = (unsigned)((((long)cur.img * p.H + cur.ty0) * p.W + cur.tx0 + (tid_o >> 4)) * p.Cout + cur.n0 + (tid_o & 15) * 8) * 2u;
        RQ_CTP(1);
        for (int c = 0; c + 1 < NC; ++c) run_chunk(c, std::false_type{});
        run_chunk(NC - 1, std::true_type{});

        // ---- epilogue of `cur` (hbuf now names the buffer holding the next tile's patch: keep clear of it)
        char* sT = (char*)smem + PK_P0 + (hbuf == 0 ? PK_PATCH : 0);
        const int lane_o = tid_o & 63, wave_o = tid_o >> 6, wm_o = wave_o >> 1, wn_o = wave_o & 1;
        float* sred = (float*)(sT + PK_TILE);       // [8][16][4]
        if (RES) {
#pragma unroll
            for (int k = 0; k < R_IT; ++k) st128(sT + ((tid_o >> 4) + HT_W * k) * PK_LDR + (tid_o & 15) * 16, rr[k]);
            rq_syncthreads();
        }
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int ty = wm_o * RPW + i, tx = lane_o & 31;
            const int ml = ty * HT_W + tx;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nl = wn_o * 64 + j * 32 + 8 * q + 4 * (lane_o >> 5);
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
                    if (RES) {
                        const uint32_t* rp = (const uint32_t*)(sT + ml * PK_LDR + nl * 2);
                        const uint32_t r0 = rp[0], r1 = rp[1];
                        { float lo_, hi_; rq_unpack2(r0, lo_, hi_); v[0] += lo_; v[1] += hi_; }
                        { float lo_, hi_; rq_unpack2(r1, lo_, hi_); v[2] += lo_; v[3] += hi_; }
                    }
                    struct __attribute__((aligned(8))) u64 { uint32_t a, b; } w2;
                    w2.a = pack_bf16x2(v[0], v[1]);
                    w2.b = pack_bf16x2(v[2], v[3]);
                    *(u64*)(sT + ml * PK_LDR + nl * 2) = w2;
                }
            }
        }
        // the accumulators are free: fetch the next tile's bias into them now, under the tile's stores
        if (has_next) acc_from_bias(nxt.n0);
        rq_syncthreads();
        RQ_CTP(61);
        float gs_[4], gq_[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { gs_[e] = 0.f; gq_[e] = 0.f; }
#pragma unroll
        for (int k = 0; k < R_IT; ++k) {
            const rq_u128 u = ld128(sT + ((tid_o >> 4) + HT_W * k) * PK_LDR + (tid_o & 15) * 16);
            st128_act((char*)p.out + (io_off0 + (unsigned)k * io_step), u);
            if (p.stats) rq_stats_piece(u, gs_, gq_);
        }
        if (p.stats) {                              // uniform; same reduction as conv3x3_halo_kernel
            const int gsz = p.Cout / 32;
            float a0, q0, a1, q1;
            if (gsz == 4) {
                a0 = gs_[0] + gs_[1]; q0 = gq_[0] + gq_[1];
                a1 = gs_[2] + gs_[3]; q1 = gq_[2] + gq_[3];
            } else {
                a0 = (gs_[0] + gs_[1]) + (gs_[2] + gs_[3]);
                q0 = (gq_[0] + gq_[1]) + (gq_[2] + gq_[3]);
                a1 = 0.f; q1 = 0.f;
            }
            a0 += rq_shfl_xor(a0, 16); q0 += rq_shfl_xor(q0, 16); a1 += rq_shfl_xor(a1, 16); q1 += rq_shfl_xor(q1, 16);
            a0 += rq_shfl_xor(a0, 32); q0 += rq_shfl_xor(q0, 32); a1 += rq_shfl_xor(a1, 32); q1 += rq_shfl_xor(q1, 32);
            if (lane_o < 16) {
                float* o = sred + (wave_o * 16 + lane_o) * 4;
                o[0] = a0; o[1] = q0; o[2] = a1; o[3] = q1;
            }
            rq_syncthreads();
            if (wave_o == 0) {
                const int chunk = (lane_o >> 1) & 15, pair = lane_o & 1;
                float a = 0.f, q = 0.f;
#pragma unroll
                for (int w = 0; w < NTH / 64; ++w) { a += sred[(w * 16 + chunk) * 4 + pair * 2]; q += sred[(w * 16 + chunk) * 4 + pair * 2 + 1]; }
                if (gsz == 16) {
                    a += rq_shfl_xor(a, 2);
                    q += rq_shfl_xor(q, 2);
                }
                const bool writer = gsz == 4 ? true : (gsz == 8 ? pair == 0 : (pair == 0 && (chunk & 1) == 0));
                if (writer && lane_o < 32) {
                    const int g = (cur.n0 + chunk * 8 + pair * 4) / gsz;
                    const int tiles = tiles_x * tiles_y * NCLS;
                    float* o = p.stats + (((long)cur.img * tiles + cur.trem) * 32 + g) * 2;
                    o[0] = a;
                    o[1] = q;
                }
            }
        }
        RQ_CTP(63);
#ifdef RQ_CONV_TRACE
        ++trace_tile;
#endif
        // ---- next tile: its first patch and weight tile are already in LDS, units 1, 2 wait in their register sets
        if (!has_next) break;
        // Loads and stores share one in-flight counter whose returns are ordered only within each kind, so the first use of a
        // pending load after this tile's stores makes the compiler wait for everything outstanding.  Have that wait HERE, where
        // only the stores and long-issued loads (the next tile's weight tiles 1, 2 and bias) are in flight -- at the next tile's
        // first tap it would also cover that tap's fresh prefetches (6400 instead of 1650 cycles for the tap,
        // profiles/r02_conv_halo_barrier_timeline.txt).
        rq_use(rw[1][0].x, rw[1][1].x, rw[2][0].x, rw[2][1].x);
        if (SUB) rq_use(rw[W_SETS - 1][0].x, rw[W_SETS - 1][1].x, rw[W_SETS - 1][0].y, rw[W_SETS - 1][1].y);
        rq_use(acc[0][0][0], acc[RPW - 1][1][15]);
        slot += wpx;
        cur = nxt;
        wn_cur = wn_nxt;
        if (SUB) set_class(cur.cls);
        has_next = slot_ok(slot + wpx);
        if (has_next) { nxt = decode(slot + wpx); wn_nxt = (unsigned)nxt.n0 * w_per_n + (unsigned)nxt.cls * w_per_cls; }
    }
}

// -------------------------------------------------------------------------------------------------
// Decoder.conv_out (modules.py:165-169 of the reference): Cin -> Cout <= 4 at full resolution, writes the NCHW
// fp32 image.  Same halo idea, but the whole problem of a tile lives in LDS at once: a 4 x 32 pixel tile's
// (4+2) x (32+2) patch of ALL input channels (52 KB at Cin = 128) plus every tap of the (bf16) weights, so there
// is ONE barrier and then 9 * Cin/16 back-to-back MFMAs per wavefront (one output row each).  The transposed
// tile D[cout][pixel] = W[cout][k] * X[k][pixel] is used with a 32-row weight operand of which only rows < Cout
// are real -- the other rows alias them and their outputs are discarded (rows of D are independent): 10x the
// necessary MFMA work, still ~2 us per image, against 68 us for the VALU + LDS-broadcast kernel it replaces
// (3456 LDS weight reads per pixel).  norm_out's GroupNorm + SiLU is applied while staging (gn != null).
struct ConvOutArgs {
    const bf16_t* x;        // NHWC [B][H][W][Cin]
    const float* w;         // [Cout][3][3][Cin] fp32 (repacked conv_out weight)
    const float* bias;      // [Cout]
    const float* gn;        // [B][Cin][2] or null
    float* y;               // NCHW fp32 [B][Cout][H][W]
    int B, H, W, Cin, Cout;
};
constexpr int OT_H = 4, OP_N = (OT_H + 2) * HP_W;          // 6 x 34 = 204 patch pixels
constexpr int O_NTH = 256;
constexpr int O_PLANE = OP_N * 128;                        // one 64-channel plane of the patch

// Persistent form: a workgroup walks a contiguous range of tiles (row-major inside an image, so the halo rows a tile shares with the
// tile above come back out of the L2), the weights are repacked into LDS once per workgroup, and the next tile's patch pieces
// (13 per thread at Cin = 128) are requested into registers before the current tile's MFMAs, normalised and stored after them.
// Two workgroups per CU (71 KB of LDS each) interleave one's MFMA phase with the other's GroupNorm + SiLU arithmetic.
// (Measured and not kept, profiles/r02_conv_out_staging.txt: the weight fragments in registers with v_mfma_f32_16x16x32_bf16 --
// both operands come out of LDS for every MFMA here, 576 KB per tile -- needs ~320 registers, i.e. one workgroup per CU: 1.6x slower.)
template <int FUSE_GN>
__global__ __launch_bounds__(O_NTH) void conv_out_halo_kernel(ConvOutArgs p) {
    RQ_DYN_SMEM(smem);
    const int NC = p.Cin / 64;
    char* sH = (char*)smem;                                // [NC][O_PLANE]
    const int wrow = 9 * p.Cin * 2 + 32;                   // padded weight row (bytes): rows land on different banks
    char* sW = sH + NC * O_PLANE;                          // [8][wrow]: rows >= Cout are zero
    const int tid = threadIdx.x, lane = tid & 63, wave = rq_uniform(tid >> 6);

    const int tiles_x = p.W / HT_W, tiles_y = p.H / OT_H, tiles_img = tiles_x * tiles_y;
    const int n_mt = p.B * tiles_img;
    // tile range: the XCD of this workgroup (blockIdx & 7) owns a contiguous eighth, split evenly over its workgroups
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3, S = (int)gridDim.x >> 3;
    const int per = (n_mt + 7) >> 3;
    const int x0 = xcd * per, x1 = x0 + per < n_mt ? x0 + per : n_mt;
    const int sub = (x1 - x0 + S - 1) / (S > 0 ? S : 1);
    const int t0 = x0 + slot * sub, t1 = t0 + sub < x1 ? t0 + sub : x1;
    if (x0 >= x1 || t0 >= t1) return;

    // piece q = (pixel hp of the patch, 16-byte chunk cq of the Cin channels); cq is the same for all of a thread's pieces
    // (256 % (Cin/8) == 0), so its 8 (scale, shift) pairs are one set of registers per tile
    const int cpp_sh = p.Cin == 64 ? 3 : 4;                // chunks per pixel CPP = Cin / 8 = 8 or 16
    const int CPP = 1 << cpp_sh;
    const int cq = tid & (CPP - 1);
    const int n_piece = OP_N << cpp_sh;
    constexpr int PIT = 13;                                // pieces per thread: ceil(204 * 16 / 256)
    unsigned hyx[PIT];                                     // (hy << 8 | hx) of piece k, or 0xffff beyond the patch
#pragma unroll
    for (int k = 0; k < PIT; ++k) {
        const int q = tid + O_NTH * k;
        const int hp = q >> cpp_sh;
        const int hy = hp / HP_W, hx = hp - hy * HP_W;
        hyx[k] = q < n_piece ? (unsigned)(hy << 8 | hx) : 0xffffu;
    }
    const char* gX = (const char*)p.x;
    struct Tile { int img, ty0, tx0; };
    auto decode = [&](int t) {
        Tile d;
        d.img = t / tiles_img;
        const int trem = t - d.img * tiles_img;
        d.ty0 = (trem / tiles_x) * OT_H;
        d.tx0 = (trem - (trem / tiles_x) * tiles_x) * HT_W;
        return d;
    };
    auto request = [&](const Tile& d, rq_u128* r, f32x4* gs) {
        if (FUSE_GN) {
            const float* gn = p.gn + ((long)d.img * p.Cin + cq * 8) * 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) gs[e] = *(const f32x4*)(gn + e * 4) * RQ_LOG2E;      // rq_silu_l2 takes the argument pre-scaled
        }
#pragma unroll
        for (int k = 0; k < PIT; ++k) {
            if (hyx[k] == 0xffffu) continue;
            const int gy = d.ty0 + (int)(hyx[k] >> 8) - 1, gx = d.tx0 + (int)(hyx[k] & 255u) - 1;
            const int cy = gy < 0 ? 0 : (gy >= p.H ? p.H - 1 : gy), cx = gx < 0 ? 0 : (gx >= p.W ? p.W - 1 : gx);
#if RQ_CONVOUT_NT
            r[k] = ld128_nt(gX + ((((long)d.img * p.H + cy) * p.W + cx) * p.Cin + cq * 8) * 2);
#else
            r[k] = ld128(gX + ((((long)d.img * p.H + cy) * p.W + cx) * p.Cin + cq * 8) * 2);      // clamped: always readable
#endif
        }
    };
    auto stage = [&](const Tile& d, const rq_u128* r, const f32x4* gs) {
#pragma unroll
        for (int k = 0; k < PIT; ++k) {
            if (hyx[k] == 0xffffu) continue;
            const int hy = (int)(hyx[k] >> 8), hx = (int)(hyx[k] & 255u);
            const int gy = d.ty0 + hy - 1, gx = d.tx0 + hx - 1;
            rq_u128 v = r[k];
            if (FUSE_GN) {
                float f[8];
                rq_unpack2(v.x, f[0], f[1]);
                rq_unpack2(v.y, f[2], f[3]);
                rq_unpack2(v.z, f[4], f[5]);
                rq_unpack2(v.w, f[6], f[7]);
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const f32x4 ss = gs[e >> 1];
                    const float a = fmaf(f[e], ss[0], ss[1]), b = fmaf(f[e + 1], ss[2], ss[3]);
                    f[e] = rq_silu_l2(a);
                    f[e + 1] = rq_silu_l2(b);
                }
                v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
                v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
            }
            if (gy < 0 || gy >= p.H || gx < 0 || gx >= p.W) v = zero128();      // zero padding of the (normalised) input
            st128(sH + (unsigned)((cq >> 3) * O_PLANE) + halo_lds_off(hy, hx, cq & 7), v);
        }
    };

    // ---- first tile and the weights (fp32 [Cout][9*Cin], 16-byte loads) in one batch of requests
    rq_u128 r[PIT];
    f32x4 gs[4];
    Tile cur = decode(t0);
    request(cur, r, gs);
    const int KT = 9 * p.Cin;
    const int n_w4 = p.Cout * (KT >> 2);                   // 16-byte groups of real weights
    constexpr int WIT = 5;                                 // covers Cout = 4, Cin = 128
    {
        f32x4 wr[WIT];
#pragma unroll
        for (int k = 0; k < WIT; ++k) {
            const int i = tid + O_NTH * k;
            if (i < n_w4) wr[k] = *(const f32x4*)(p.w + (long)i * 4);
        }
        stage(cur, r, gs);
        // weights into LDS: group i = (row, 4 consecutive k) as bf16; rows Cout..7 feed output rows nobody stores: zero
#pragma unroll
        for (int k = 0; k < WIT; ++k) {
            const int i = tid + O_NTH * k;
            if (i < n_w4) {
                const int row = i / (KT >> 2), k4 = (i - row * (KT >> 2)) * 4;
                *(uint64_t*)(sW + row * wrow + k4 * 2) = (uint64_t)pack_bf16x2(wr[k][0], wr[k][1]) | ((uint64_t)pack_bf16x2(wr[k][2], wr[k][3]) << 32);
            }
        }
        for (int i = n_w4 + tid; i < 8 * (KT >> 2); i += O_NTH) {
            const int row = i / (KT >> 2), k4 = (i - row * (KT >> 2)) * 4;
            *(uint64_t*)(sW + row * wrow + k4 * 2) = 0ull;
        }
    }

    const int ftx = lane & 31, fk = lane >> 5;
    const char* wl = sW + (ftx & 7) * wrow + fk * 16;
    float bias_r[4];
#pragma unroll
    for (int co = 0; co < 4; ++co) bias_r[co] = co < p.Cout ? p.bias[co] : 0.f;
    for (int t = t0; t < t1; ++t) {
        const bool more = t + 1 < t1;
        Tile nxt = cur;
        if (more) { nxt = decode(t + 1); request(nxt, r, gs); }      // in flight under this tile's MFMAs
        rq_syncthreads();                                  // this tile's patch (and the weights) are in LDS
        // ---- 9 taps x Cin/16 MFMAs: wave = output row of the tile, lane&31 = pixel (B operand) / weight row (A operand)
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        // The 36 (tap, k-step) MFMAs of a chunk run on ONE accumulator, each with both operands out of LDS: written as read, read,
        // MFMA the stream was `rr_M` 72 times -- every MFMA waited out its own LDS round trip (~100 cycles for a 32-cycle
        // instruction).  The fragments of step s + OD are read ahead of the MFMA of step s (OD + 1 register sets in rotation).
        constexpr int OD = 4;
        for (int c = 0; c < NC; ++c) {
            bf16x8 af[OD + 1], wf[OD + 1];
            auto frag = [&](int s, bf16x8& a, bf16x8& w) {         // s = tap * 4 + ks
                const int tap = s >> 2, ks = s & 3, ky = tap / 3, kx = tap - ky * 3;
                const unsigned ha = (unsigned)(c * O_PLANE) + halo_lds_off(wave + ky, ftx + kx, fk);
                a = as_bf16x8(ld128(sH + (ha ^ (unsigned)(ks << 5))));
                w = as_bf16x8(ld128(wl + (tap * p.Cin + c * 64) * 2 + ks * 32));
            };
#pragma unroll
            for (int s = 0; s < OD; ++s) frag(s, af[s], wf[s]);
#pragma unroll
            for (int s = 0; s < 36; ++s) {
                if (s + OD < 36) frag(s + OD, af[(s + OD) % (OD + 1)], wf[(s + OD) % (OD + 1)]);
                acc = rq_mfma_32x32x16_bf16(wf[s % (OD + 1)], af[s % (OD + 1)], acc);
                rq_sched_barrier();
            }
        }
        // C/D layout: lanes 0..31 hold rows (= cout) 0..3 of column (= pixel) lane in acc[0..3]
        if (lane < 32) {
            const int oy = cur.ty0 + wave, ox = cur.tx0 + lane;
#pragma unroll
            for (int co = 0; co < 4; ++co)
                if (co < p.Cout) p.y[(((long)cur.img * p.Cout + co) * p.H + oy) * p.W + ox] = acc[co] + bias_r[co];
        }
        if (!more) break;
        rq_syncthreads();                                  // every wavefront is done reading the patch
        stage(nxt, r, gs);
        cur = nxt;
    }
}

bool rq_conv_out_halo_supported(int H, int W, int Cin, int Cout) {
    return H % OT_H == 0 && W % HT_W == 0 && (Cin == 64 || Cin == 128) && Cout >= 1 && Cout <= 4;
}

int rq_launch_conv_out_halo(const bf16_t* x, const float* w, const float* bias, const float* gn, float* y, int B, int H, int W,
                            int Cin, int Cout, hipStream_t s) {
    if (!rq_conv_out_halo_supported(H, W, Cin, Cout)) return rq_fail(RQAMD_ERR_UNSUPPORTED, "conv_out_halo: shape %dx%d %d->%d", H, W, Cin, Cout);
    ConvOutArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.gn = gn; a.y = y; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    const size_t smem = (size_t)(Cin / 64) * O_PLANE + 8 * (size_t)(9 * Cin * 2 + 32);
    static RqDeviceOnce attr_once;      // kernel attributes are per device
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)conv_out_halo_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_out_halo_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    const int n_mt = B * (H / OT_H) * (W / HT_W);
    int nblocks = 8 * ((n_mt + 7) / 8);
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    if (nblocks > 2 * cus) nblocks = 2 * cus / 8 * 8;      // two resident workgroups per CU, each walking a range of tiles
    if (const char* e = getenv("RQAMD_CONV_OUT_WGS")) {    // diagnostics / tests: fewer workgroups, longer walks
        const int v = atoi(e) / 8 * 8;
        if (v >= 8 && v < nblocks) nblocks = v;
    }
    if (gn) RQ_LAUNCH(conv_out_halo_kernel<1>, dim3(nblocks), dim3(O_NTH), smem, s, a);
    else RQ_LAUNCH(conv_out_halo_kernel<0>, dim3(nblocks), dim3(O_NTH), smem, s, a);
    return rq_check_launch("conv_out_halo_kernel");
}

// -------------------------------------------------------------------------------------------------
// Encoder.conv_in (modules.py:23-27 of the reference): 3 -> 128 channels at full resolution, NCHW fp32 image in,
// NHWC bf16 out.  K = 27 is padded to 32 = two MFMA k-steps; the A fragment of a pixel is gathered from an fp32
// patch of the image in LDS through a 32-entry offset table (k -> (ci, ky, kx), entries 27..31 point at a zero),
// the weights sit in LDS as bf16 [128][32].  The VALU version (conv_in3_kernel) read one LDS weight per FMA:
// 40 us per image, 12 % of the encoder; this one is bound by writing the 16.8 MB output.
struct ConvInArgs {
    const float* x;         // NCHW [B][3][H][W]
    const float* w;         // [ky][kx][ci][Cout] fp32 (repacked conv_in weight)
    const float* bias;      // [Cout]
    bf16_t* y;              // NHWC [B][H][W][Cout]
    float* stats;           // GroupNorm partials of y, [B][tiles][32][2] (sum, sumsq) as the halo convs' epilogues write them, or null
    int B, H, W;
};
constexpr int CI_COUT = 128, CI_PATCH = 3 * (HT_H + 2) * HP_W;        // 1020 floats (+ zero slot)
constexpr int CI_WROW = 80;                                            // bytes per weight row: 32 bf16 + 16 pad
constexpr int CI_STG = 32 * (CI_COUT * 2 + 16);                        // per-wave output staging: 32 pixels x 272 B

__global__ __launch_bounds__(256) void conv_in_mfma_kernel(ConvInArgs p) {
    RQ_DYN_SMEM(smem);
    float* sP = (float*)smem;                                          // [3][10][34] + zero slot (+ pad to 16 B)
    int* sK = (int*)(smem + 4096);                                     // [32] patch offset of k (row 0, column 0)
    char* sWt = (char*)smem + 4096 + 128;                              // [128][CI_WROW]
    char* sO = sWt + CI_COUT * CI_WROW;                                // [4 waves][CI_STG]
    const int tid = threadIdx.x, lane = tid & 63, wave = rq_uniform(tid >> 6);
    const int tiles_x = p.W / HT_W, tiles_y = p.H / HT_H;
    const int tile = blockIdx.x;
    const int img = tile / (tiles_y * tiles_x), trem = tile - img * (tiles_y * tiles_x);
    const int ty0 = (trem / tiles_x) * HT_H, tx0 = (trem - (trem / tiles_x) * tiles_x) * HT_W;

    for (int i = tid; i < CI_PATCH + 4; i += 256) {
        float v = 0.f;
        if (i < CI_PATCH) {
            const int ci = i / ((HT_H + 2) * HP_W), r = i - ci * ((HT_H + 2) * HP_W);
            const int hy = r / HP_W, hx = r - hy * HP_W;
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) v = p.x[(((long)img * 3 + ci) * p.H + gy) * p.W + gx];
        }
        sP[i] = v;
    }
    if (tid < 32) {
        const int k = tid, tap = k / 3, ci = k - tap * 3, ky = tap / 3, kx = tap - ky * 3;
        sK[k] = k < 27 ? ci * ((HT_H + 2) * HP_W) + ky * HP_W + kx : -1;
    }
    for (int i = tid; i < CI_COUT * 16; i += 256) {                    // two k per thread-iteration
        const int co = i >> 4, k2 = (i & 15) * 2;
        const float a = k2 < 27 ? p.w[k2 * CI_COUT + co] : 0.f, b = k2 + 1 < 27 ? p.w[(k2 + 1) * CI_COUT + co] : 0.f;
        *(uint32_t*)(sWt + co * CI_WROW + k2 * 2) = pack_bf16x2(a, b);
    }
    rq_syncthreads();

    const int px = lane & 31, g = lane >> 5;
    int koff[2][8];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) koff[ks][e] = sK[ks * 16 + g * 8 + e];
    bf16x8 bfr[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[ks][j] = as_bf16x8(ld128(sWt + (j * 32 + px) * CI_WROW + (ks * 16 + g * 8) * 2));
    char* stg = sO + wave * CI_STG;
    // GroupNorm partials of the tile (round 6: the encoder's first ResnetBlock reads them instead of a 0.41-ms statistics pass over the
    // 2.1-GB output): this lane stores 16-byte chunk (lane & 15) of 32 pixels per row -- channels 8 (lane & 15) .. + 7 = groups
    // 2 (lane & 15), 2 (lane & 15) + 1 of the 32 four-channel groups -- and sums what it stores (the bf16-rounded values)
    float gs_[4] = {0.f, 0.f, 0.f, 0.f}, gq_[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int i = 0; i < 2; ++i) {
        const int row = wave * 2 + i;                                  // output row of the tile
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = koff[ks][e] >= 0 ? sP[koff[ks][e] + row * HP_W + px] : 0.f;
            rq_u128 u;
            u.x = pack_bf16x2(f[0], f[1]); u.y = pack_bf16x2(f[2], f[3]);
            u.z = pack_bf16x2(f[4], f[5]); u.w = pack_bf16x2(f[6], f[7]);
            const bf16x8 af = as_bf16x8(u);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = rq_mfma_32x32x16_bf16(bfr[ks][j], af, acc[j]);   // D[cout][pixel]
        }
        // lane (px, g) holds couts j*32 + 8q + 4g .. +3 in acc[j][4q..4q+3]: 8-byte pieces into the wave's staging tile
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = j * 32 + 8 * q + 4 * g;
                const f32x4 bv = *(const f32x4*)(p.bias + co);
                struct __attribute__((aligned(8))) u64 { uint32_t a, b; } w2;
                w2.a = pack_bf16x2(acc[j][4 * q] + bv[0], acc[j][4 * q + 1] + bv[1]);
                w2.b = pack_bf16x2(acc[j][4 * q + 2] + bv[2], acc[j][4 * q + 3] + bv[3]);
                *(u64*)(stg + px * (CI_COUT * 2 + 16) + co * 2) = w2;
            }
        rq_syncthreads();                                              // (a wave-level sync would do; all waves run both rows)
        const long pix0 = ((long)img * p.H + ty0 + row) * p.W + tx0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = lane + 64 * k, opx = c >> 4, part = c & 15;
            const rq_u128 u = ld128(stg + opx * (CI_COUT * 2 + 16) + part * 16);
            st128(p.y + (pix0 + opx) * CI_COUT + part * 8, u);
            if (p.stats) rq_stats_piece(u, gs_, gq_);
        }
        rq_syncthreads();
    }
    if (p.stats) {                                                      // uniform
        // channel pairs -> groups of four channels, the 4 lanes that share a chunk, then the 4 wavefronts through LDS (fixed order)
        float a0 = gs_[0] + gs_[1], q0 = gq_[0] + gq_[1], a1 = gs_[2] + gs_[3], q1 = gq_[2] + gq_[3];
        a0 += rq_shfl_xor(a0, 16); q0 += rq_shfl_xor(q0, 16); a1 += rq_shfl_xor(a1, 16); q1 += rq_shfl_xor(q1, 16);
        a0 += rq_shfl_xor(a0, 32); q0 += rq_shfl_xor(q0, 32); a1 += rq_shfl_xor(a1, 32); q1 += rq_shfl_xor(q1, 32);
        float* sred = (float*)sO;                                      // [4 waves][16 chunks][4] (the staging tiles are free now)
        if (lane < 16) {
            float* o = sred + (wave * 16 + lane) * 4;
            o[0] = a0; o[1] = q0; o[2] = a1; o[3] = q1;
        }
        rq_syncthreads();
        if (tid < 32) {
            const int chunk = tid >> 1, pair = tid & 1;                // group 2 chunk + pair
            float a = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { a += sred[(w * 16 + chunk) * 4 + pair * 2]; q += sred[(w * 16 + chunk) * 4 + pair * 2 + 1]; }
            float* o = p.stats + (((long)img * (tiles_y * tiles_x) + trem) * 32 + tid) * 2;
            o[0] = a;
            o[1] = q;
        }
    }
}

bool rq_conv_in_mfma_supported(int H, int W, int Cin, int Cout) {
    return Cin == 3 && Cout == CI_COUT && H % HT_H == 0 && W % HT_W == 0;
}

int rq_launch_conv_in_mfma(const float* x, const float* w, const float* bias, bf16_t* y, float* stats, int B, int H, int W, hipStream_t s) {
    ConvInArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.stats = stats; a.B = B; a.H = H; a.W = W;
    const size_t smem = 4096 + 128 + (size_t)CI_COUT * CI_WROW + 4 * (size_t)CI_STG;
    static RqDeviceOnce attr_once;      // kernel attributes are per device
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)conv_in_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    }
    RQ_LAUNCH(conv_in_mfma_kernel, dim3((unsigned)(B * (H / HT_H) * (W / HT_W))), dim3(256), smem, s, a);
    return rq_check_launch("conv_in_mfma_kernel");
}

// (scale, shift) per (image, channel) from the GroupNorm partial statistics (gn_stats_kernel or a conv epilogue): one
// workgroup per image; thread (slice = t >> 5, group = t & 31) sums every 8th partial of its group, the slices are
// folded in a fixed order through LDS, then the 256 threads write the per-channel pairs.  (One thread per channel
// walking all partials serially took 45 us per launch with 256 partials per image.)
__global__ __launch_bounds__(256) void gn_params_kernel(const float* part, const float* gamma, const float* beta, float* gn, int B, int HW,
                                                        int C, int nchunk, float eps) {
    __shared__ float sa[8][32], sq[8][32], smean[32], srstd[32];
    const int b = blockIdx.x, tid = threadIdx.x, g = tid & 31, sl = tid >> 5;
    float a = 0.f, q = 0.f;
    for (int c = sl; c < nchunk; c += 8) {
        const float* o = part + (((long)b * nchunk + c) * 32 + g) * 2;
        a += o[0];
        q += o[1];
    }
    sa[sl][g] = a;
    sq[sl][g] = q;
    rq_syncthreads();
    if (tid < 32) {
        float ta = 0.f, tq = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { ta += sa[k][tid]; tq += sq[k][tid]; }
        const float n = (float)HW * (float)(C / 32);
        const float mean = ta / n;
        float var = tq / n - mean * mean;
        if (var < 0.f) var = 0.f;
        smean[tid] = mean;
        srstd[tid] = 1.0f / sqrtf(var + eps);
    }
    rq_syncthreads();
    const int cpg = C / 32;
    for (int ch = tid; ch < C; ch += 256) {
        const int gg = ch / cpg;
        const float sc = srstd[gg] * gamma[ch];
        gn[((long)b * C + ch) * 2] = sc;
        gn[((long)b * C + ch) * 2 + 1] = beta[ch] - smean[gg] * sc;
    }
}

// Pre-summed weights of the sub-pixel form (conv3x3_halo_pk_kernel<0, 2, 0>): w [Cout][3][3][Cin] bf16 -> wsub [4][Cout][2][2][Cin] bf16,
// wsub[py * 2 + px][co][a][b][ci] = sum of w[co][ky][kx][ci] over the taps (ky, kx) that read source pixel (y + py - 1 + a, x + px - 1 + b)
// from output pixel (2y + py, 2x + px): ky in {0} | {1, 2} for py = 0, {0, 1} | {2} for py = 1, likewise kx.  Summed in fp32, rounded once.
__global__ void ups_subpixel_weights_kernel(const bf16_t* w, bf16_t* wsub, int Cout, int Cin) {
    const long n = (long)4 * Cout * 4 * Cin;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n) return;
    const int ci = (int)(gid % Cin);
    long r = gid / Cin;
    const int tap = (int)(r & 3); r >>= 2;
    const int co = (int)(r % Cout);
    const int cls = (int)(r / Cout);
    const int py = cls >> 1, px = cls & 1, a = tap >> 1, b = tap & 1;
    const int ky_lo = py == 0 ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), ky_hi = py == 0 ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
    const int kx_lo = px == 0 ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), kx_hi = px == 0 ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
    float acc = 0.f;
    for (int ky = ky_lo; ky <= ky_hi; ++ky)
        for (int kx = kx_lo; kx <= kx_hi; ++kx) acc += bf16_to_f32(w[(((long)co * 3 + ky) * 3 + kx) * Cin + ci]);
    wsub[gid] = f32_to_bf16(acc);
}
int rq_launch_ups_subpixel_weights(const bf16_t* w, bf16_t* wsub, int Cout, int Cin, hipStream_t s) {
    const long n = (long)16 * Cout * Cin;
    RQ_LAUNCH(ups_subpixel_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, wsub, Cout, Cin);
    return rq_check_launch("ups_subpixel_weights_kernel");
}
// the sub-pixel form needs whole 8 x 32 tiles of the SOURCE image
bool rq_conv_halo_subpixel_supported(int H, int W, int Cin, int Cout) {
    return rq_conv_halo_supported(H, W, Cin, Cout) && (H >> 1) % HT_H == 0 && (W >> 1) % HT_W == 0;
}

bool rq_conv_halo_supported(int H, int W, int Cin, int Cout) {
    return H % HT_H == 0 && W % HT_W == 0 && Cin % 64 == 0 && Cout % H_BN == 0 && H >= 32 && H * W <= 65536;      // (whether it pays below 64^2: engine_vae.hip)
}

int rq_conv_halo_stat_tiles(int H, int W) { return (H / HT_H) * (W / HT_W); }
static int g_conv_halo_dbg_pk = 0;          // diagnostics entry only: +1 / -1 force the persistent / per-tile form of the 8-row kernel
static int g_conv_halo_dbg_wpx = 0;         // diagnostics entry only: workgroups per XCD of the persistent form
// Measured (profiles/r02_conv_halo_variants.txt): the persistent form gains 4-7 % where the tile is short and carries no fused
// GroupNorm (the upsample convs), nothing or -3 % on the fused ResnetBlock convs (their epilogue is bound by store ISSUE, which
// blocks the issuing wavefront whether or not a next tile waits behind it) -- so by default only the upsample convs use it.
// RQAMD_HALO_PERSIST=1 / 0: every / no 8-row conv.
static bool halo_persistent(int ups) {
    static const int env = getenv("RQAMD_HALO_PERSIST") ? atoi(getenv("RQAMD_HALO_PERSIST")) : -1;
    if (g_conv_halo_dbg_pk) return g_conv_halo_dbg_pk > 0;
    return env < 0 ? ups != 0 : env != 0;
}

static int launch_conv_halo_th(const ConvHaloArgs& a, int ups, hipStream_t s) {
    constexpr int TH = HT_H, NTHR = 512;
    constexpr size_t smem = H_SMEM_BYTES;
    static_assert((size_t)TH * HT_W * (H_BN * 2 + 16) + 4096 <= H_SMEM_BYTES, "the epilogue tile overlays the operand buffers");
    static RqDeviceOnce attr_once;      // kernel attributes are per device
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)conv3x3_halo_kernel<0, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute((const void*)conv3x3_halo_kernel<0, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute((const void*)conv3x3_halo_kernel<1, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute((const void*)conv3x3_halo_kernel<1, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute((const void*)conv3x3_halo_kernel<0, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    }
    const int n_mt = a.B * (a.H / TH) * (a.W / HT_W), NT = a.Cout / H_BN;
    const int nblocks = 8 * ((n_mt + 7) / 8) * NT;
    // (the persistent forms of the plain / fused convs are not compiled: 0 ... -3 % in round 2; again in round 3 with the GroupNorm
    // words dealt out under the MFMAs -- bit-identical, +2 % (GN) ... +9 % (GN + residual) slower, 256 registers with spills in
    // its prologue / epilogue: profiles/r03_conv_halo_persist_fused_ab.txt)
    if (ups == 2) {
        // sub-pixel form of the upsample conv (a.w = the pre-summed weights): persistent, tiles = source tiles x 4 parity classes
        static RqDeviceOnce sub_once;
        static int cus_per_xcd_s[16];
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (sub_once.first()) {
            (void)hipFuncSetAttribute((const void*)conv3x3_halo_pk_kernel<0, 2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, PK_SMEM);
            int cus = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
            cus_per_xcd_s[dev & 15] = cus / 8;
        }
        const int n_sub = a.B * ((a.H >> 1) / TH) * ((a.W >> 1) / HT_W) * 4;
        const int slots = ((n_sub + 7) / 8) * NT;
        int wpx = g_conv_halo_dbg_wpx > 0 ? g_conv_halo_dbg_wpx : cus_per_xcd_s[dev & 15];
        if (wpx > slots) wpx = slots;
        RQ_LAUNCH((conv3x3_halo_pk_kernel<0, 2, 0>), dim3(8 * wpx), dim3(512), PK_SMEM, s, a, wpx);
        return rq_check_launch("conv3x3_halo_pk_kernel<subpixel>");
    }
    if (ups && halo_persistent(ups)) {
        // persistent form: one workgroup per CU, each walking every wpx-th slot of its XCD's band
        static RqDeviceOnce pk_once;
        static int cus_per_xcd[16];
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (pk_once.first()) {
            (void)hipFuncSetAttribute((const void*)conv3x3_halo_pk_kernel<0, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, PK_SMEM);
            int cus = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
            cus_per_xcd[dev & 15] = cus / 8;
        }
        const int slots = ((n_mt + 7) / 8) * NT;
        int wpx = g_conv_halo_dbg_wpx > 0 ? g_conv_halo_dbg_wpx : cus_per_xcd[dev & 15];
        if (wpx > slots) wpx = slots;
        RQ_LAUNCH((conv3x3_halo_pk_kernel<0, 1, 0>), dim3(8 * wpx), dim3(512), PK_SMEM, s, a, wpx);
        return rq_check_launch("conv3x3_halo_pk_kernel");
    }
    if (ups) RQ_LAUNCH((conv3x3_halo_kernel<0, 1, 0>), dim3(nblocks), dim3(NTHR), smem, s, a);
    else if (a.gn && a.resid) RQ_LAUNCH((conv3x3_halo_kernel<1, 0, 1>), dim3(nblocks), dim3(NTHR), smem, s, a);
    else if (a.gn) RQ_LAUNCH((conv3x3_halo_kernel<1, 0, 0>), dim3(nblocks), dim3(NTHR), smem, s, a);
    else if (a.resid) RQ_LAUNCH((conv3x3_halo_kernel<0, 0, 1>), dim3(nblocks), dim3(NTHR), smem, s, a);
    else RQ_LAUNCH((conv3x3_halo_kernel<0, 0, 0>), dim3(nblocks), dim3(NTHR), smem, s, a);
    return rq_check_launch("conv3x3_halo_kernel");
}

int rq_launch_conv_halo(const bf16_t* x, const bf16_t* w, const float* bias, const float* gn, const bf16_t* resid, bf16_t* out,
                        float* stats, int B, int H, int W, int Cin, int Cout, int ups, hipStream_t s) {
    if (stats && Cout != 128 && Cout != 256 && Cout != 512) return rq_fail(RQAMD_ERR_UNSUPPORTED, "conv_halo: fused statistics need Cout 128/256/512");
    if (!rq_conv_halo_supported(H, W, Cin, Cout)) return rq_fail(RQAMD_ERR_UNSUPPORTED, "conv_halo: shape %dx%d %d->%d", H, W, Cin, Cout);
    if (ups && (gn || resid)) return rq_fail(RQAMD_ERR_UNSUPPORTED, "conv_halo: the upsample conv takes no GroupNorm / residual");
    if (ups == 2 && !rq_conv_halo_subpixel_supported(H, W, Cin, Cout)) return rq_fail(RQAMD_ERR_UNSUPPORTED, "conv_halo: sub-pixel form at %dx%d", H, W);
    if (2.0 * B * H * W * (Cin > Cout ? Cin : Cout) >= 4294967296.0) return rq_fail(RQAMD_ERR_UNSUPPORTED, "conv_halo: tensor larger than 4 GiB");
    ConvHaloArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.gn = gn; a.resid = resid; a.out = out; a.stats = stats; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    return launch_conv_halo_th(a, ups, s);
}

// nchunk_have > 0: `part` already holds that many partials per image (written by the producing conv's epilogue)
int rq_launch_gn_params(const bf16_t* x, float* part, const float* gamma, const float* beta, float* gn, int B, int HW, int C,
                        int nchunk_have, hipStream_t s) {
    int nchunk = nchunk_have;
    if (nchunk <= 0) RQ_TRY(rq_launch_gn_stats(x, part, B, HW, C, &nchunk, s));
    RQ_LAUNCH(gn_params_kernel, dim3(B), dim3(256), 0, s, part, gamma, beta, gn, B, HW, C, nchunk, 1e-6f);
    return rq_check_launch("gn_params_kernel");
}

// diagnostics entry (include/rqamd.h)
extern "C" int rqamd_dbg_conv_halo_bf16(const void* x, const void* w, const float* bias, const float* gn, const void* resid,
                                        int B, int H, int W, int Cin, int Cout, int ups, void* out, float* stats, void* stream) {
    if (!x || !w || !bias || !out) return rq_fail(RQAMD_ERR_INVALID, "dbg_conv_halo: null argument");
    g_conv_halo_dbg_pk = (ups & 16) ? 1 : (ups & 32) ? -1 : 0;                    // bits 4 / 5: persistent / per-tile form of the 8-row kernel
    g_conv_halo_dbg_wpx = (ups >> 8) & 0xff;                                      // bits 8..15: workgroups per XCD (0 = one per CU)
    // bit 6: the sub-pixel form of the upsample conv (`w` = the pre-summed weights of rqamd_dbg_ups_subpixel_weights)
    const int rc = rq_launch_conv_halo((const bf16_t*)x, (const bf16_t*)w, bias, gn, (const bf16_t*)resid, (bf16_t*)out, stats, B, H, W,
                                       Cin, Cout, (ups & 64) ? 2 : (ups & 1), (hipStream_t)stream);
    g_conv_halo_dbg_pk = 0;
    g_conv_halo_dbg_wpx = 0;
    return rc;
}

extern "C" int rqamd_dbg_ups_subpixel_weights(const void* w, int Cout, int Cin, void* wsub, void* stream) {
    if (!w || !wsub) return rq_fail(RQAMD_ERR_INVALID, "dbg_ups_subpixel_weights: null argument");
    return rq_launch_ups_subpixel_weights((const bf16_t*)w, (bf16_t*)wsub, Cout, Cin, (hipStream_t)stream);
}

extern "C" int rqamd_dbg_conv_out_bf16(const void* x, const float* w, const float* bias, const float* gn, int B, int H, int W,
                                       int Cin, int Cout, float* y, void* stream) {
    if (!x || !w || !bias || !y) return rq_fail(RQAMD_ERR_INVALID, "dbg_conv_out: null argument");
    return rq_launch_conv_out_halo((const bf16_t*)x, w, bias, gn, y, B, H, W, Cin, Cout, (hipStream_t)stream);
}

extern "C" int rqamd_dbg_conv_in_bf16(const float* x, const float* w, const float* bias, int B, int H, int W, void* y, void* stream) {
    if (!x || !w || !bias || !y) return rq_fail(RQAMD_ERR_INVALID, "dbg_conv_in: null argument");
    if (!rq_conv_in_mfma_supported(H, W, 3, CI_COUT)) return rq_fail(RQAMD_ERR_UNSUPPORTED, "dbg_conv_in: shape %dx%d", H, W);
    return rq_launch_conv_in_mfma(x, w, bias, (bf16_t*)y, nullptr, B, H, W, (hipStream_t)stream);
}

#ifdef RQ_CONV_TRACE
extern "C" int rqamd_dbg_conv_trace(unsigned long long* out_host) {
    if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_conv_trace), sizeof(g_conv_trace)) != hipSuccess) return rq_fail(RQAMD_ERR_HIP, "conv_trace: copy failed");
    return RQAMD_OK;
}
#endif
