// gemm.hip -- launcher / tile selection for the bf16 MFMA GEMM (see gemm.h)
#include "gemm.h"
#include <math.h>
#include <stdlib.h>
#include "rq_common.h"

template <int BM, int BN, int MODE, int TR, int WGM = 2, int WGN = 2, int GL = 0, int VS = 0>
static int launch_c(const GemmArgs& a, hipStream_t stream) {
    const size_t smem = (size_t)(BM + BN) * 64 * 2 * (GL ? GL : 2);
    static RqDeviceOnce attr_once;      // kernel attributes are per device
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<BM, BN, MODE, TR, WGM, WGN, GL, VS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    }
    // XCD-aware schedule (see the kernel): pad the grid to 8 x the largest per-XCD slice
    GemmArgs g = a;
    const int MT = (a.M + BM - 1) / BM, NT = (a.N + BN - 1) / BN;
    int nblocks, zdim = a.splitk;
    static const bool env_sched0 = getenv("RQAMD_GEMM_SCHED0") != nullptr, env_sched1 = getenv("RQAMD_GEMM_SCHED1") != nullptr;   // A/B switches
    static const bool env_nosched4 = getenv("RQAMD_GEMM_NO_SCHED4") != nullptr;
    if (env_sched0) {
        g.sched = 0; g.sched_gm = 1; nblocks = MT * NT;
    } else if (NT >= 8 && !a.conv && MT >= 8 && 7.0 * ((double)a.M - a.N) * a.K * 2.0 > 100e6 && !env_sched1) {
        // dense, weights much smaller than activations (fc2 at large batch): give every XCD a band of m-tiles instead.
        // An XCD then streams its own slice of A once and all of W (N*K bytes, 8x in total) rather than all of A
        // (M*K bytes, 8x): PMC showed 446 MB per fc2 launch at M=4096 against 94 MB algorithmic with n-ranges, 226 MB
        // and 99 -> 90 us with m-bands (proj, where the saving is 55 MB, measured 5 % slower: threshold 100 MB).
        g.sched = 2; g.sched_gm = 1;
        nblocks = 8 * ((MT + 7) / 8) * NT;
    } else if (GL && !a.conv && a.epi == EPI_F32_PARTIAL && !a.accum && (a.splitk == 2 || a.splitk == 4 || a.splitk == 8) && MT >= 4 && MT <= 16 &&
               (a.K / 64) % a.splitk == 0 && !env_sched1 && !env_nosched4) {
        // split-K slab GEMMs at mid batch: K slices to XCD groups (see the kernel); one-dimensional grid.  Measured (profiles/r06_sched4_ab.txt,
        // whole step): +0.7 % at 500 images, -0.5 % at 200 (two m-tiles), level at 100 -- taken from four m-tiles; the fabric traffic of fc2 falls
        // from 3.0 x to the algorithmic bytes either way, which is not what bounds these launches.
        const int xper = 8 / a.splitk;
        g.sched = 4; g.sched_gm = 1;
        nblocks = 8 * ((NT + xper - 1) / xper) * MT;
        zdim = 1;
    } else if (NT >= 8) {
        const int ktiles = (a.K / 64 + a.splitk - 1) / a.splitk;
        long panel = (long)BM * ktiles * 64 * 2;                   // bytes of one m-tile's A panel for this K split
        int gm = (int)((3 << 19) / (panel > 0 ? panel : 1));       // keep ~1.5 MiB of A hot in the 4 MiB L2
        if (gm < 1) gm = 1;
        if (gm > MT) gm = MT;
        g.sched = 1; g.sched_gm = gm;
        nblocks = 8 * ((NT + 7) / 8) * MT;
    } else {
        g.sched = 2; g.sched_gm = 1;
        nblocks = 8 * ((MT + 7) / 8) * NT;
    }
    dim3 grid(nblocks, 1, zdim);
    RQ_LAUNCH((gemm_bf16_kernel<BM, BN, MODE, TR, WGM, WGN, GL, VS>), grid, dim3(64 * WGM * WGN), smem, stream, g);
    return rq_check_launch("gemm_bf16_kernel");
}
// register-blocked 256x128 / 4-wave / BK 32 kernel (gemm.h): dense operands, K % 32 == 0
template <int BM, int TR, int NS>
static int launch_rb(const GemmArgs& a, hipStream_t stream) {
    constexpr int BN = 128;
    constexpr size_t stage = (size_t)(BM + BN) * 32 * 2, epi = (size_t)BM * (BN * 2 + 16);
    const size_t smem = NS * stage > epi ? NS * stage : epi;
    static RqDeviceOnce attr_once;      // kernel attributes are per device
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)gemm_rb_kernel<BM, TR, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    }
    GemmArgs g = a;
    const int MT = (a.M + BM - 1) / BM, NT = (a.N + BN - 1) / BN;
    int nblocks;
    if (NT >= 8 && MT >= 8 && 7.0 * ((double)a.M - a.N) * a.K * 2.0 > 100e6) {
        g.sched = 2; g.sched_gm = 1;
        nblocks = 8 * ((MT + 7) / 8) * NT;
    } else if (NT >= 8) {
        const int ktiles = (a.K / 64 + a.splitk - 1) / a.splitk;
        long panel = (long)BM * ktiles * 64 * 2;
        int gm = (int)((3 << 19) / (panel > 0 ? panel : 1));
        if (gm < 1) gm = 1;
        if (gm > MT) gm = MT;
        g.sched = 1; g.sched_gm = gm;
        nblocks = 8 * ((NT + 7) / 8) * MT;
    } else {
        g.sched = 2; g.sched_gm = 1;
        nblocks = 8 * ((MT + 7) / 8) * NT;
    }
    RQ_LAUNCH((gemm_rb_kernel<BM, TR, NS>), dim3(nblocks, 1, a.splitk), dim3(256), smem, stream, g);
    return rq_check_launch("gemm_rb_kernel");
}

// 256 x 256 eight-phase kernel (gemm.h): dense operands, >= 2 K-tiles per split
template <int TR, int PH = 4, int EK = -1>
static int launch_p8(const GemmArgs& a, hipStream_t stream) {
    constexpr int BM = 256, BN = 256;
    const size_t smem = (size_t)BM * (BN * 2 + 16);
    static RqDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)gemm_p8_kernel<TR, PH, EK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    }
    GemmArgs g = a;
    const int MT = (a.M + BM - 1) / BM, NT = (a.N + BN - 1) / BN;
    int nblocks;
    static const bool env_sched0 = getenv("RQAMD_GEMM_SCHED0") != nullptr;
    static const int env_gm = getenv("RQAMD_P8_GM") ? atoi(getenv("RQAMD_P8_GM")) : 0;      // A/B switch
    if (env_sched0) {
        g.sched = 0; g.sched_gm = 1; nblocks = MT * NT;
    } else {
        // balanced contiguous ranges per XCD (sched 3 in rq_gemm_tile_coords); groups of 8 m-tiles: 32 concurrent tiles of
        // an XCD = 8 A panels x 4 W panels
        g.sched = 3;
        // group height: ~8 m-tiles, preferring a divisor of MT so that no ragged last group is left (42 m-tiles: 6 -- the
        // classifier measured 546 -> 525 us, the other shapes are insensitive; profiles/r02_gemm_p8_picker.txt)
        int gm = 8;
        static const int pref[] = {8, 6, 7, 5, 10, 9, 4};
        for (int c : pref)
            if (MT % c == 0) { gm = c; break; }
        g.sched_gm = env_gm > 0 ? env_gm : gm;
        if (g.sched_gm > MT) g.sched_gm = MT;
        nblocks = 8 * ((MT * NT + 7) / 8);
    }
    RQ_LAUNCH((gemm_p8_kernel<TR, PH, EK>), dim3(nblocks, 1, a.splitk), dim3(512), smem, stream, g);
    return rq_check_launch("gemm_p8_kernel");
}

// weight-streaming kernel for small batches (gemm.h): dense operands, 32 weight rows x BM activation rows per workgroup
template <int BM, int BN = 32>
static int launch_stream(const GemmArgs& a, hipStream_t stream) {
#ifdef RQ_STREAM_NS
    constexpr size_t smem = (size_t)4 * RQ_STREAM_NS * ((BM + BN) * 64 * 2);
#else
    constexpr size_t smem = (size_t)4 * ((BM == 64 && BN == 32) ? 3 : 2) * ((BM + BN) * 64 * 2);
#endif
    static RqDeviceOnce attr_once;
    if (attr_once.first())
        (void)hipFuncSetAttribute((const void*)gemm_stream_kernel<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    RQ_LAUNCH((gemm_stream_kernel<BM, BN>), dim3((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, a.splitk), dim3(256), smem, stream, a);
    return rq_check_launch("gemm_stream_kernel");
}

template <int BM, int BN>
static int launch_t(const GemmArgs& a, hipStream_t stream) {
    // transposed accumulators (TR = 1) for everything except wide fp32 rows (logits / fp32 activations)
    const bool tr = a.epi != EPI_F32;
    if (!a.conv) return tr ? launch_c<BM, BN, 0, 1>(a, stream) : launch_c<BM, BN, 0, 0>(a, stream);
    if (a.vsplit > 1) return a.ups ? launch_c<BM, BN, 2, 1, 2, 2, 0, 1>(a, stream) : launch_c<BM, BN, 1, 1, 2, 2, 0, 1>(a, stream);
    if (a.ups) return tr ? launch_c<BM, BN, 2, 1>(a, stream) : launch_c<BM, BN, 2, 0>(a, stream);
    return tr ? launch_c<BM, BN, 1, 1>(a, stream) : launch_c<BM, BN, 1, 0>(a, stream);
}

int rq_gemm_launch(const GemmArgs& a_in, int bm, int bn, hipStream_t stream) {
    GemmArgs a = a_in;
    if (a.splitk < 1) a.splitk = 1;
    if (a.K % 64 != 0 || a.M <= 0 || a.N <= 0)
        return rq_fail(RQAMD_ERR_UNSUPPORTED, "gemm: K=%d must be a positive multiple of 64 (M=%d N=%d)", a.K, a.M, a.N);
    if (a.conv && (a.Cin % 64 != 0 || a.K != a.ksize * a.ksize * a.Cin))
        return rq_fail(RQAMD_ERR_UNSUPPORTED, "conv gemm: Cin=%d must be a multiple of 64", a.Cin);
    a.cin_shift = -1;
    if (a.conv)
        for (int sh = 0; sh < 16; ++sh)
            if ((1 << sh) == a.Cin) a.cin_shift = sh;
    if (a.conv && a.ups && (a.stride != 1 || a.ksize != 3 || a.pad != 1 || (a.Hin & 1) || (a.Win & 1)))
        return rq_fail(RQAMD_ERR_UNSUPPORTED, "conv gemm: the folded upsample needs a 3x3 stride-1 conv on even sizes");
    if (a.conv && a.ksize != 1 && a.ksize != 3) return rq_fail(RQAMD_ERR_UNSUPPORTED, "conv gemm: kernel size %d", a.ksize);
    {   // the kernel addresses both operands with 32-bit byte offsets
        const double a_bytes = a.conv ? 2.0 * ((double)a.M / (a.Hout * a.Wout)) * (a.Hin >> a.ups) * (a.Win >> a.ups) * a.Cin : 2.0 * a.M * a.lda;
        if (a_bytes >= 4294967296.0 || 2.0 * a.N * a.K >= 4294967296.0)
            return rq_fail(RQAMD_ERR_UNSUPPORTED, "gemm: operand larger than 4 GiB (split the batch)");
    }
    if (a.splitk > 1 && a.epi != EPI_F32_PARTIAL)
        return rq_fail(RQAMD_ERR_INVALID, "gemm: split-K needs the partial-slab epilogue");
    if (a.vsplit < 1) a.vsplit = 1;
    if (a.vsplit > 1 && (!a.conv || a.splitk != 1 || (a.epi != EPI_BF16 && a.epi != EPI_BF16_RESID) || (a.K / 64) % (2 * a.vsplit) != 0 ||
                         (bm == 256 && bn != 128) || bm > 256))
        return rq_fail(RQAMD_ERR_INVALID, "gemm: virtual split-K needs a conv with a bf16 epilogue, one real split and an even number of K-tiles per chunk");
    if (a.accum && (a.epi != EPI_F32_PARTIAL || a.splitk != 1 || (a.N & 3) || (a.ldo & 3) || (bm == 64 && bn == 32)))
        return rq_fail(RQAMD_ERR_INVALID, "gemm: in-place accumulation needs the slab epilogue, one K split and N, ldo multiples of 4");
    if (((bm == 66 || bm == 130) && bn == 32) || (bm == 66 && bn == 64)) {
        // tile codes 66x32 / 130x32: the weight-streaming kernel on 64 / 128 activation rows; 66x64: its 64-row weight tiles
        if (a.conv) return rq_fail(RQAMD_ERR_UNSUPPORTED, "gemm stream: dense operands only");
        if (bn == 64) return launch_stream<64, 64>(a, stream);
        return bm == 66 ? launch_stream<64>(a, stream) : launch_stream<128>(a, stream);
    }
    if (bm == 64 && bn == 32) {       // skinny kernel (M <= 64): 32 weight rows per workgroup, in-workgroup split-K over 8 wavefronts
        if (a.conv || a.M > 64 || a.K % a.splitk != 0 || (a.K / a.splitk) % 512 != 0)
            return rq_fail(RQAMD_ERR_UNSUPPORTED, "gemm skinny: dense operands, M <= 64 and (K / splitk) %% 512 == 0 needed");
        const size_t smem = (size_t)8 * 64 * 36 * 4;
        static RqDeviceOnce attr_once;
        if (attr_once.first())
            (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        RQ_LAUNCH(gemm_skinny_kernel<0>, dim3((a.N + 31) / 32, 1, a.splitk), dim3(512), smem, stream, a);
        return rq_check_launch("gemm_skinny_kernel");
    }
    if (bm == 256 && bn == 256) {     // eight-phase kernel
        const int ktiles = (a.K / 64 + a.splitk - 1) / a.splitk;
        if (a.conv || ktiles < 2 || (a.K / 64) % a.splitk != 0)
            return rq_fail(RQAMD_ERR_UNSUPPORTED, "gemm 256x256: dense operands and >= 2 K-tiles per (even) split needed");
        // phases per K-tile: two for the transposed-accumulator kernels (bf16 outputs, split-K slabs: -3.3 ... -4.5 % at M = 10752),
        // four for the fp32-row kernel (classifier: two phases measured +2.4 %); profiles/r02_gemm_p8_two_phase.txt.
        // RQAMD_P8_PH=2|4 forces one schedule (A/B switch)
        static const int ph_env = getenv("RQAMD_P8_PH") ? atoi(getenv("RQAMD_P8_PH")) : 0;
        const bool ph2 = (a.dbg & 64) ? true : (a.dbg & 128) ? false : ph_env == 2 ? true : ph_env == 4 ? false : a.epi != EPI_F32;
        // the shipped combinations carry their epilogue family as a compile-time constant (gemm.h: EK); the A/B schedules and
        // EPI_BF16_RESID keep the run-time form
        if (ph2 && a.epi != EPI_F32) {
            switch (a.epi) {
                case EPI_BF16: return launch_p8<1, 2, EPI_BF16>(a, stream);
                case EPI_BF16_GELU: return a.gelu_v2 ? launch_p8<1, 2, 6>(a, stream) : launch_p8<1, 2, EPI_BF16_GELU>(a, stream);
                case EPI_F32_PARTIAL: return a.accum ? launch_p8<1, 2, 5>(a, stream) : launch_p8<1, 2, EPI_F32_PARTIAL>(a, stream);
                default: return launch_p8<1, 2>(a, stream);
            }
        }
        if (ph2) return launch_p8<0, 2>(a, stream);
        return a.epi != EPI_F32 ? launch_p8<1>(a, stream) : launch_p8<0, 4, EPI_F32>(a, stream);
    }
    {   // LDS-DMA staged variants (dense operands): RQAMD_GEMM_GL = number of LDS stages (experiment switch)
        static const int gl_env = getenv("RQAMD_GEMM_GL") ? atoi(getenv("RQAMD_GEMM_GL")) : 0;
        const int gl = a.glds ? a.glds : gl_env;
        if (!a.conv && gl >= 2) {
            const bool tr = a.epi != EPI_F32;
#define RQ_GL_CASE(BM_, BN_, WM_, WN_)                                                                                   \
            if (bm == BM_ && bn == BN_) {                                                                                  \
                if (gl == 2) return tr ? launch_c<BM_, BN_, 0, 1, WM_, WN_, 2>(a, stream) : launch_c<BM_, BN_, 0, 0, WM_, WN_, 2>(a, stream); \
                return tr ? launch_c<BM_, BN_, 0, 1, WM_, WN_, 3>(a, stream) : launch_c<BM_, BN_, 0, 0, WM_, WN_, 3>(a, stream);               \
            }
#ifdef RQ_GEMM_SWEEP
            // diagnostics build (scripts/gemm_mid_sweep.py): tile code = BM + wave layout (0: 2 x 2, 1: 4 x 1, 4: 4 x 2 wavefronts), 2 .. 6 stages
#define RQ_SW_GL(BM_, BN_, WM_, WN_, G_)                                                                                  \
            if (gl == G_) {                                                                                                \
                if constexpr ((BM_ + BN_) * 128 * G_ <= 160 * 1024)                                                        \
                    return tr ? launch_c<BM_, BN_, 0, 1, WM_, WN_, G_>(a, stream) : launch_c<BM_, BN_, 0, 0, WM_, WN_, G_>(a, stream); \
                else return rq_fail(RQAMD_ERR_INVALID, "gemm sweep: %d stages of a %d x %d tile exceed the LDS", G_, BM_, BN_);         \
            }
#define RQ_SW_CASE(CODE_, BM_, BN_, WM_, WN_)                                                                             \
            if (bm == CODE_ && bn == BN_) {                                                                                \
                RQ_SW_GL(BM_, BN_, WM_, WN_, 2) RQ_SW_GL(BM_, BN_, WM_, WN_, 3) RQ_SW_GL(BM_, BN_, WM_, WN_, 4)             \
                RQ_SW_GL(BM_, BN_, WM_, WN_, 5) RQ_SW_GL(BM_, BN_, WM_, WN_, 6)                                             \
            }
            RQ_SW_CASE(128, 128, 64, 2, 2)
            RQ_SW_CASE(128, 128, 128, 2, 2)
            RQ_SW_CASE(129, 128, 64, 4, 1)
            RQ_SW_CASE(129, 128, 96, 4, 1)
            RQ_SW_CASE(129, 128, 128, 4, 1)
            RQ_SW_CASE(129, 128, 160, 4, 1)
            RQ_SW_CASE(129, 128, 192, 4, 1)
            RQ_SW_CASE(132, 128, 64, 4, 2)
            RQ_SW_CASE(132, 128, 128, 4, 2)
            RQ_SW_CASE(132, 128, 192, 4, 2)
            RQ_SW_CASE(136, 128, 128, 4, 4)
            RQ_SW_CASE(136, 128, 256, 4, 4)
            RQ_SW_CASE(264, 256, 128, 4, 4)
            RQ_SW_CASE(64, 64, 128, 2, 2)
            RQ_SW_CASE(64, 64, 192, 2, 2)
            RQ_SW_CASE(258, 256, 64, 4, 1)
            RQ_SW_CASE(258, 256, 96, 4, 1)
            RQ_SW_CASE(260, 256, 64, 4, 2)
#undef RQ_SW_CASE
#undef RQ_SW_GL
#endif
            // round 5: eight / sixteen wavefronts per tile, three stages (tile code = rows + 4: 4 x 2 wavefronts, + 8: 4 x 4)
#define RQ_G3_CASE(CODE_, BM_, BN_, WM_, WN_)                                                                              \
            if (bm == CODE_ && bn == BN_ && gl == 3)                                                                        \
                return tr ? launch_c<BM_, BN_, 0, 1, WM_, WN_, 3>(a, stream) : launch_c<BM_, BN_, 0, 0, WM_, WN_, 3>(a, stream);
            RQ_G3_CASE(132, 128, 64, 4, 2)
            RQ_G3_CASE(136, 128, 128, 4, 4)
            RQ_G3_CASE(132, 128, 192, 4, 2)
            RQ_G3_CASE(264, 256, 128, 4, 4)
            RQ_G3_CASE(136, 128, 256, 4, 4)
#undef RQ_G3_CASE
            RQ_GL_CASE(128, 64, 2, 2)
            RQ_GL_CASE(128, 128, 2, 2)
            RQ_GL_CASE(256, 128, 4, 2)
            if (bm == 257 && bn == 128) {     // tile code 257x128: the register-blocked kernel (4 waves, 128x64 wave tiles, BK 32)
                if (gl == 2) return tr ? launch_rb<256, 1, 2>(a, stream) : launch_rb<256, 0, 2>(a, stream);
                return tr ? launch_rb<256, 1, 3>(a, stream) : launch_rb<256, 0, 3>(a, stream);
            }
#undef RQ_GL_CASE
        }
    }
    if (bm == 64 && bn == 64) return launch_t<64, 64>(a, stream);
    if (bm == 64 && bn == 128) return launch_t<64, 128>(a, stream);
    if (bm == 128 && bn == 64) return launch_t<128, 64>(a, stream);
    if (bm == 128 && bn == 128) return launch_t<128, 128>(a, stream);
    if (bm == 256 && bn == 128) {     // 8 wavefronts (4x2), 96 KiB LDS: half the weight staging per FLOP
        const bool tr = a.epi != EPI_F32;
        if (!a.conv) return tr ? launch_c<256, 128, 0, 1, 4, 2>(a, stream) : launch_c<256, 128, 0, 0, 4, 2>(a, stream);
        if (a.vsplit > 1) return a.ups ? launch_c<256, 128, 2, 1, 4, 2, 0, 1>(a, stream) : launch_c<256, 128, 1, 1, 4, 2, 0, 1>(a, stream);
        if (a.ups) return tr ? launch_c<256, 128, 2, 1, 4, 2>(a, stream) : launch_c<256, 128, 2, 0, 4, 2>(a, stream);
        return tr ? launch_c<256, 128, 1, 1, 4, 2>(a, stream) : launch_c<256, 128, 1, 0, 4, 2>(a, stream);
    }
    return rq_fail(RQAMD_ERR_INVALID, "gemm: no tile %dx%d", bm, bn);
}

void rq_gemm_pick_tile(int M_rows, int N, int K, bool allow_splitk, int* bm, int* bn, int* splitk, int* glds) {
    const int M = (long)M_rows * g_rq_row_scale > 1 << 30 ? 1 << 30 : M_rows * g_rq_row_scale;      // (diagnostics factor, normally 1)
    // LDS-DMA staged operands win or tie from M = 512 up (profiles/r01_gemm_bench_lds_dma.txt, MI355X): 8-12 % at
    // M = 4096.  Tiles per shape class, from the same sweep:
    static const bool no_glds = getenv("RQAMD_NO_GLDS") != nullptr;
    static const bool no_p8 = getenv("RQAMD_NO_P8") != nullptr;        // A/B switch
    if (glds) *glds = 0;
    // skinny kernel (gemm_skinny_kernel): one pass over W with every load in flight at once.  Measured against the tiled kernels
    // with rotating weights (profiles/r02_gemm_skinny_ab.txt): it wins only for the bf16-output GEMMs at <= 32 rows (M = 8: qkv
    // 7.8 vs 10.2 us, fc1 7.9 vs 13.6; M = 32: 8.8 vs 10.5, 9.4 vs 12.6) and ties or loses elsewhere (M = 64: qkv 11.2 vs 11.0;
    // split-K slab GEMMs 11.1 vs 7.8) -- at this size a launch is ~7 us of fixed cost around 2-3 us of streaming, whatever the
    // kernel, so the small-batch regime stays launch-latency bound.
    // Weight-streaming kernel (gemm_stream_kernel) for <= 128 rows: 32 weight rows per workgroup, four barrier-free wavefronts with
    // private LDS-DMA rings.  Interleaved A/B with rotating weights (profiles/r03_gemm_stream_ab.txt, MI355X): 64 rows: qkv 10.6 ->
    // 7.4 us, fc1 11.8 -> 7.6, proj 7.0 -> 6.8, fc2 8.5 -> 7.9 (one layer 37.9 -> 29.7 us); 8 rows: level with the round-2 skinny
    // kernel it replaces (30.6 -> 29.2); 65..128 rows (128-row form): qkv 10.6 -> 9.5, fc1 12.2 -> 10.5, proj / fc2 +0.7 / +1.5 (one
    // layer 40.6 -> 40.0) -- taken for ALL four there so that a row's result does not depend on which side of 64 its batch falls (the
    // two forms agree bit for bit).  Not for the classifier (N = 16384: every workgroup re-reads the activations, 16.6 -> 19.7 us) and
    // not beyond 128 rows (the activation tile outgrows the per-CU fill rate: 256 rows 52 -> 64 us per layer).
    // Split-K (slab GEMMs): as many slices as keep (N / 32) x slices within one round of 256 CUs -- the kernel holds 144-160 KB of
    // LDS, one workgroup per CU (fc2: 4 slices 7.9 us, 8 slices 11.3) -- a function of (N, K) only for 32-row weight tiles; the
    // 64-row tiles below (<= 64 activation rows only) change the slice count of some shapes, so THOSE shapes' bits depend on which
    // side of 64 rows the batch falls: fc2 and proj at E = 2560 (four slices of 64-row tiles against two of 32-row tiles) and the
    // N = 16384 classifier (this kernel at <= 64 rows, the tiled kernel above) -- DESIGN.md section 6.
    static const bool no_stream = getenv("RQAMD_NO_STREAM") != nullptr;      // A/B switch
    // (the classifier's N = 16384 only at <= 64 rows, where the 64-row weight tiles below cover it in one round: 15.0 vs 16.9 us at
    // E = 1536, 21.0 vs 25.0 at E = 2560, profiles/r04_gemm_stream_bn64_e2560.txt / r04_stream_k_rotation_ab.txt; at 65 .. 128 rows the
    // tiled kernel stays ahead, 20.4 vs 23.0)
    if (!no_stream && M <= 128 && K % 64 == 0 && (N < 16384 || (M <= 64 && N == 16384)) && N >= 64) {
        *bm = M <= 64 ? 66 : 130; *bn = 32;
        int sk = 1;
        if (allow_splitk) {
            const int nt = (N + 31) / 32, kt = K / 64;
            while (sk < 8 && nt * (sk * 2) <= 256 && kt % (sk * 2) == 0 && kt / (sk * 2) >= 4) sk *= 2;
        }
        // 64-row GEMMs whose 32-row weight tiles do not fit ONE round of the 256 CUs take 64-row tiles (round 4; E = 2560: fc1 has
        // 320 tiles, 23.0 us = two rounds; fc2's 80 tiles allow only two K slices of 80 K-tiles each, 21.8 us): half the workgroups,
        // the same arithmetic per output element.  (65 .. 128 rows keep 32-row tiles: a 128 x 64 slot pair does not fit the LDS.)
        static const bool no_bn64 = getenv("RQAMD_NO_STREAM_BN64") != nullptr;      // A/B switch
        if (M <= 64 && !no_bn64) {
            const int nt32 = (N + 31) / 32, nt64 = (N + 63) / 64, kt = K / 64;
            if (nt32 > 256) *bn = 64;
            // (slab GEMMs of the wide models: four K slices of 64-row tiles move fewer bytes per workgroup than two of 32-row tiles --
            // fc2 21.8 -> 18.0 us, proj 7.9 -> 7.2 at E = 2560)
            else if (allow_splitk && kt >= 32 && nt32 * 4 > 256 && nt64 * 4 <= 256 && kt % 4 == 0) { *bn = 64; sk = 4; }
        }
        *splitk = sk;
        return;
    }
    static const bool skinny = getenv("RQAMD_SKINNY") != nullptr;      // the round-2 skinny kernel, kept selectable for A/B runs
    if (skinny && !allow_splitk && (long)M_rows * g_rq_row_scale <= 32 && K % 512 == 0 && N >= 256 && N < 16384) {
        *bm = 64; *bn = 32; *splitk = 1;
        return;
    }
    // 129 .. 2047 rows (round 5; the reference's own metric batches 200 / 500 are here): eight- and sixteen-wavefront tiles of the
    // LDS-DMA kernel -- tile codes 132 x {64, 192} = 128 rows as 4 x 2 wavefronts, 136 x {128, 256} = 128 rows and 264 x 128 = 256 rows
    // as 4 x 4 -- three ring stages.  What the sweep showed (profiles/r05_gemm_mid_sweep*.txt, in-graph, weights from HBM): at these row
    // counts a launch is bound by how fast a CU takes its operands in, not by MFMAs (the loop without MFMAs takes as long as the whole
    // kernel; ~70 GB/s per CU and ~11 TB/s over the chip, whatever the ring depth or the wavefront count), the four-wavefront tiles
    // lose to their own fragment-read latency (one wavefront per SIMD: 23 us of compute where eight wavefronts take 14), and an XCD
    // that is dealt more than 32 workgroups runs a second round while others idle.  So: among the tiles and K splits, take the one
    // with the smallest modelled time  c0 + max(stream, mma) + 0.4 min(stream, mma) + epilogue (+ the consumer's slab reads), with
    // stream = max(rounds x bytes per workgroup / 110 GB/s, all bytes / 17 TB/s), mma at 5 TFLOP/s per CU, rounds counted per XCD
    // (constants fitted to 370 measured (shape, tile, split) points, rms error 8 %; the pick is within 2 % of the measured best on
    // the 50 shapes of the sweep at E = 1024 / 1536 / 2560, 200 .. 1536 rows: 1599 -> 1334 us summed, best possible 1291).
    static const bool no_mid = getenv("RQAMD_NO_MID") != nullptr;        // A/B switch: the round-1..4 tile rules below
    if (glds && !no_glds && !no_mid && M > 128 && M < 2048 && K % 64 == 0 && N >= 64) {
        const bool wide_f32 = !allow_splitk && N >= 8192;                  // classifier: fp32 rows
        if (wide_f32 && M > 256) {
            // (measured: 128 x 256 sixteen-wavefront tiles 30.8 us at 300 rows against 36.2 for 128 x 128, 56.5 at 768; from 1024 rows
            // the 256 x 128 three-stage tile of round 1 is level or ahead and stays)
            if (M < 1024) { *bm = 136; *bn = 256; *splitk = 1; *glds = 3; return; }
        } else {
            static const int cand[5][3] = {{132, 128, 64}, {136, 128, 128}, {132, 128, 192}, {264, 256, 128}, {136, 128, 256}};      // (128 x 128: sixteen wavefronts measured 1-2 % ahead of eight at every row count)
            static const int sks[6] = {1, 2, 3, 4, 6, 8};
            const int kt = K / 64;
            double best = 1e30;
            for (const auto& c : cand) {
                const int BM = c[1], BN = c[2];
                const int MT = (M + BM - 1) / BM, NT = (N + BN - 1) / BN;
                for (int si = 0; si < (allow_splitk ? 6 : 1); ++si) {
                    const int sk = sks[si];
                    if (kt % sk != 0 || (sk > 1 && kt / sk < 4)) continue;
                    const double W = (double)MT * NT * sk;
                    const double bw = (double)(BM + BN) * 128.0 * (kt / sk);
                    const int wx = NT >= 8 ? ((NT + 7) / 8) * MT * sk : ((MT + 7) / 8) * NT * sk;      // workgroups of the busiest XCD
                    const int rounds = (wx + 31) / 32;
                    const double ts = fmax(rounds * bw / 110e3, W * bw / 17e6);
                    const double tm = rounds * 2.0 * BM * BN * (double)K / sk / 5e6;
                    double t = 2.0 + fmax(ts, tm) + 0.4 * fmin(ts, tm) + 2.5 + (double)BM * BN * (allow_splitk || wide_f32 ? 4 : 2) / 32e3;
                    if (allow_splitk && sk > 1) t += sk * (double)M * N * 4.0 / 4.5e6;      // sk fp32 slabs read again by the LayerNorm that follows (resid_ln at 500 x 1536: 4.9 us with 4 slabs, 7.5 with 8)
                    if (t < best) { best = t; *bm = c[0]; *bn = BN; *splitk = sk; }
                }
            }
            if (best < 1e29) { *glds = 3; return; }
        }
    }
    // 256 x 256 kernel (gemm_p8_kernel): one workgroup per CU, so what decides is how the tile count fills rounds of 256 CUs.  A
    // small cost model fitted to the interleaved A/B runs on MI355X (profiles/r02_gemm_p8_ab.txt, r02_gemm_p8_picker.txt; within
    // ~10 % of the measurements): a round costs nk * t_k + t_epilogue with t_k = 1.05 us per K-tile on a nearly empty chip to
    // 1.5 us with every CU busy (the operand stream through the L2s), t_epilogue 6 us (bf16 tile) / 8 us (fp32 slab), + 4 us per
    // launch; against the other tiles at ~750 TF (bf16 outputs) / ~650 TF (split-K slab GEMMs) at these row counts.
    // Residual-producing GEMMs may split K 2 or 4 ways with >= 12 K-tiles per split.
    if (glds && !no_glds && !no_p8 && M >= 2048 && K % 64 == 0 && K / 64 >= 2) {
        const int MT = (M + 255) / 256, NT = (N + 255) / 256, nk = K / 64;
        const long tiles = (long)MT * NT;
        const double t_epi = allow_splitk ? 8.0 : 6.0;
        double best = 1e30;
        int best_sk = 1;
        for (int sk = 1; sk <= (allow_splitk ? 4 : 1); sk *= 2) {
            if (sk > 1 && (nk % sk != 0 || nk / sk < 12)) continue;
            const long wgs = tiles * sk, full = wgs / 256, rem = wgs % 256;
            const int nkp = nk / sk;
            double t = 4.0 + (double)full * (nkp * 1.5 + t_epi);
            if (rem) t += nkp * (1.05 + 0.45 * (double)rem / 256.0) + t_epi;
            if (sk > 1) t += 6.0 * (sk - 1);                  // sk fp32 slabs written here and read again by the consumer
            if (t < best) { best = t; best_sk = sk; }
        }
        const double other = 2.0 * M * (double)N * K / 1e6 / (allow_splitk ? 650.0 : 750.0);      // us
        if (best < other) {
            *bm = 256; *bn = 256; *splitk = best_sk; *glds = 2;
            return;
        }
    }
    if (glds && !no_glds && M >= 512 && K % 64 == 0) {
        *glds = 2;
        if (N >= 16384 && M >= 512) {                       // classifier: wide N, fp32 rows
            *bm = 256; *bn = 128; *splitk = 1; *glds = M >= 4096 || M < 2048 ? 3 : 2;
            if (M >= 4096) { *bm = 257; *glds = 2; }        // register-blocked kernel: 267 vs 281 us at M = 4096
            return;
        }
        if (M >= 8192) {                                    // M = 8192 sweep: rb kernel for the widest N, 128x128 otherwise
            *splitk = 1;
            if (N >= 6144) { *bm = 257; *bn = 128; } else { *bm = 128; *bn = 128; }
            return;
        }
        if (M >= 4096 && N >= 4096) { *bm = 128; *bn = 128; *splitk = 1; return; }                       // qkv, fc1
        if (M >= 2048 && allow_splitk && K >= 4096 && N <= 2048) {                                       // fc2
            *bm = 256; *bn = 128; *splitk = M >= 4096 ? 1 : 2; *glds = 3;
            return;
        }
        // everything else: 128x64 with the split-K rule below
        *bm = 128; *bn = 64;
        int maxsplit = 1;
        if (allow_splitk) {
            maxsplit = (K / 64) / 8;
            if (maxsplit > 8) maxsplit = 8;
            if (maxsplit < 1) maxsplit = 1;
        }
        const int tiles = ((M + 127) / 128) * ((N + 63) / 64);
        int sp = (768 + tiles - 1) / tiles;
        if (sp > maxsplit) sp = maxsplit;
        if (sp < 1) sp = 1;
        *splitk = sp;
        return;
    }
    // Rule distilled from scripts/gemm_bench.py on MI355X (profiles/r01_gemm_bench.md), M = 64..2048 batch
    // rows against the 1.4B layer shapes: take the LARGEST tile (most MFMAs per barrier) that still yields
    // >= 512 workgroups (2 per CU) once split-K is allowed to multiply the count (split target: 768); residual-producing GEMMs
    // (fp32 partial slabs, reduced by resid_ln) may split K up to 8 ways with >= 8 K-tiles per split.
    auto cdiv = [](int a, int b) { return (a + b - 1) / b; };
    static const int cand[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
    int maxsplit = 1;
    if (allow_splitk) {
        maxsplit = (K / 64) / 8;
        if (maxsplit > 8) maxsplit = 8;
        if (maxsplit < 1) maxsplit = 1;
    }
    // the 8-wave 256x128 tile (half the weight staging per FLOP) pays off only with plenty of tiles and no split
    if (M >= 2048 && cdiv(N, 128) % 8 == 0 && cdiv(M, 256) * cdiv(N, 128) >= 192) {
        *bm = 256; *bn = 128; *splitk = 1;
        return;
    }
    int pick = 3;
    for (int c = 0; c < 4; ++c) {
        if (M <= 64 && cand[c][0] > 64) continue;
        const int nt = cdiv(N, cand[c][1]);
        if (nt >= 8 && (nt & 7) != 0 && cdiv(N, 64) % 8 == 0) continue;   // XCD schedule pads NT to a multiple of 8: avoid idle slices
        if (cdiv(M, cand[c][0]) * nt * maxsplit >= 512) { pick = c; break; }
    }
    *bm = cand[pick][0];
    *bn = cand[pick][1];
    const int tiles = cdiv(M, *bm) * cdiv(N, *bn);
    int s = cdiv(768, tiles);
    if (s > maxsplit) s = maxsplit;
    if (s < 1) s = 1;
    *splitk = s;
}

// diagnostics entry (include/rqamd.h): one raw launch of the decode-step GEMM, for microbenchmarks
// and kernel-level parity tests.  epi: 0 bf16, 1 bf16+GELU, 3 fp32, 4 fp32 split-K slabs.
extern "C" int rqamd_dbg_gemm_bf16(const void* A, const void* W, int M, int N, int K, const float* bias, int epi,
                                   void* out, int bm, int bn, int splitk, void* stream) {
    if (!A || !W || !out) return rq_fail(RQAMD_ERR_INVALID, "dbg_gemm: null argument");
    int flags = 0, glds = 0, accum = 0;
    if (epi >= 8192) { flags |= 4; epi -= 8192; }             // epi + 8192: LDS-DMA kernels without fragment reads / MFMAs (ablation)
    if (epi >= 4096) { flags |= 2; epi -= 4096; }             // epi + 4096: LDS-DMA kernels without operand staging (ablation)
    if (epi >= 2048) { accum = 1; epi -= 2048; }              // 4 + 2048: in-place residual accumulation (splitk 1)
    if (epi >= 1024) { flags |= 128; epi -= 1024; }           // epi + 1024: 256x256 kernel with four phases per K-tile (A/B)
    if (epi >= 512) { flags |= 64; epi -= 512; }              // epi + 512: 256x256 kernel with two phases per K-tile (A/B)
    if (epi >= 256) { flags |= 32; epi -= 256; }              // epi + 256: accepted and ignored (was: 256x256 kernel without s_setprio, measured null in round 2)
    if (epi >= 64) { glds = epi / 32; epi -= glds * 32; }      // epi + 32 * stages: LDS-DMA operand staging (2 or 3 stages)
    if (epi >= 16) { flags |= 1; epi -= 16; }      // epi + 16: skip the epilogue (ablation)
    GemmArgs a{};
    a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.M = M; a.N = N; a.K = K; a.lda = K; a.epi = epi;
    a.bias = bias; a.out = out; a.ldo = N; a.splitk = splitk; a.dbg = flags; a.glds = glds; a.accum = accum;
    if (bm <= 0 || bn <= 0) {
        int sk, gl = 0;
        rq_gemm_pick_tile(M, N, K, epi == EPI_F32_PARTIAL, &bm, &bn, &sk, &gl);
        if (splitk <= 0) a.splitk = sk;
        else if (gl && splitk != sk) { gl = 0; rq_gemm_pick_tile(M, N, K, epi == EPI_F32_PARTIAL, &bm, &bn, &sk, nullptr); }
        a.glds = gl;
    }
    if (a.splitk <= 0) a.splitk = 1;
    return rq_gemm_launch(a, bm, bn, (hipStream_t)stream);
}

// diagnostics: one implicit-GEMM convolution launch.  x NHWC bf16 [B][H>>ups][W>>ups][Cin], w [Cout][k][k][Cin] bf16,
// out NHWC bf16 [B][Ho][Wo][Cout] (+bias, +resid when given).  flags bit0: skip the epilogue (ablation).
extern "C" int rqamd_dbg_conv_bf16(const void* x, const void* w, const float* bias, const void* resid, int B, int H, int W,
                                   int Cin, int Cout, int ksize, int stride, int ups, void* out, int bm, int bn, int flags,
                                   void* stream) {
    if (!x || !w || !out) return rq_fail(RQAMD_ERR_INVALID, "dbg_conv: null argument");
    GemmArgs a{};
    int Ho = H, Wo = W, pad = ksize / 2;
    if (stride == 2) { Ho = H / 2; Wo = W / 2; pad = 0; }
    a.A = (const bf16_t*)x; a.W = (const bf16_t*)w; a.M = B * Ho * Wo; a.N = Cout; a.K = ksize * ksize * Cin; a.lda = Cin;
    a.conv = 1; a.Hin = H; a.Win = W; a.Cin = Cin; a.Hout = Ho; a.Wout = Wo; a.ksize = ksize; a.stride = stride; a.pad = pad; a.ups = ups;
    a.epi = resid ? EPI_BF16_RESID : EPI_BF16; a.bias = bias; a.out = out; a.ldo = Cout; a.resid = (const bf16_t*)resid; a.ldr = Cout;
    a.splitk = 1; a.dbg = flags;
    if (bm <= 0) bm = a.M >= 128 ? 128 : 64;
    if (bn <= 0) bn = (Cout % 128 == 0) ? 128 : 64;
    return rq_gemm_launch(a, bm, bn, (hipStream_t)stream);
}

#ifdef RQ_STREAM_TRACE
// diagnostics build only (scripts/stream_trace.py)
extern "C" int rqamd_dbg_stream_trace(unsigned long long* out_host, int clear) {
    if (clear) {
        static unsigned long long zeros[1024 * 2 * 24];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stream_trace), zeros, sizeof(zeros)) != hipSuccess) return rq_fail(RQAMD_ERR_HIP, "stream_trace: clear failed");
        return RQAMD_OK;
    }
    if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_stream_trace), sizeof(g_stream_trace)) != hipSuccess) return rq_fail(RQAMD_ERR_HIP, "stream_trace: copy failed");
    return RQAMD_OK;
}
#endif
#ifdef RQ_GL_TRACE
// diagnostics build only (scripts/gl_trace.py)
extern "C" int rqamd_dbg_gl_trace(unsigned long long* out_host) {
    if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_gl_trace), sizeof(g_gl_trace)) != hipSuccess) return rq_fail(RQAMD_ERR_HIP, "gl_trace: copy failed");
    return RQAMD_OK;
}
#endif
#ifdef RQ_GEMM_TRACE
// diagnostics build only (scripts/gemm_trace.sh)
extern "C" int rqamd_dbg_gemm_trace(unsigned long long* out_host) {
    if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_gemm_trace), sizeof(g_gemm_trace)) != hipSuccess) return rq_fail(RQAMD_ERR_HIP, "gemm_trace: copy failed");
    return RQAMD_OK;
}
#endif
