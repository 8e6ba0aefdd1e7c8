// gemm.hip -- launcher / tile selection for the bf16 MFMA GEMM (see gemm.h)
#include "gemm.h"
#include "rq_common.h"

template <int BM, int BN>
static int launch_t(const GemmArgs& a, hipStream_t stream) {
    const size_t smem = (size_t)(BM + BN) * 64 * 2 * 2;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_done = true;
    }
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, a.splitk);
    RQ_LAUNCH((gemm_bf16_kernel<BM, BN>), grid, dim3(256), smem, stream, a);
    return rq_check_launch("gemm_bf16_kernel");
}

int rq_gemm_launch(const GemmArgs& a_in, int bm, int bn, hipStream_t stream) {
    GemmArgs a = a_in;
    if (a.splitk < 1) a.splitk = 1;
    if (a.K % 64 != 0 || a.M <= 0 || a.N <= 0)
        return rq_fail(RQAMD_ERR_UNSUPPORTED, "gemm: K=%d must be a positive multiple of 64 (M=%d N=%d)", a.K, a.M, a.N);
    if (a.conv && (a.Cin % 64 != 0 || a.K != a.ksize * a.ksize * a.Cin))
        return rq_fail(RQAMD_ERR_UNSUPPORTED, "conv gemm: Cin=%d must be a multiple of 64", a.Cin);
    if (a.splitk > 1 && a.epi != EPI_F32_PARTIAL)
        return rq_fail(RQAMD_ERR_INVALID, "gemm: split-K needs the partial-slab epilogue");
    if (bm == 64 && bn == 64) return launch_t<64, 64>(a, stream);
    if (bm == 64 && bn == 128) return launch_t<64, 128>(a, stream);
    if (bm == 128 && bn == 64) return launch_t<128, 64>(a, stream);
    if (bm == 128 && bn == 128) return launch_t<128, 128>(a, stream);
    return rq_fail(RQAMD_ERR_INVALID, "gemm: no tile %dx%d", bm, bn);
}

void rq_gemm_pick_tile(int M, int N, int K, bool allow_splitk, int* bm, int* bn, int* splitk) {
    auto cdiv = [](int a, int b) { return (a + b - 1) / b; };
    *bm = M <= 64 ? 64 : 128;
    *bn = 128;
    int tiles = cdiv(M, *bm) * cdiv(N, 128);
    if (tiles < 160 || N % 128 != 0) {
        *bn = 64;
        tiles = cdiv(M, *bm) * cdiv(N, 64);
    }
    *splitk = 1;
    if (allow_splitk && tiles < 192) {
        int s = 256 / tiles;
        int max_by_k = (K / 64) / 4;
        if (s > max_by_k) s = max_by_k;
        if (s > 8) s = 8;
        if (s < 1) s = 1;
        *splitk = s;
    }
}
