// api.hip -- ABI-level entry points that do not belong to one engine (see include/rqamd.h)
#include "rq_common.h"

thread_local char rq_err_buf[512] = {0};

extern "C" int rqamd_abi_version(void) { return RQAMD_ABI_VERSION; }
extern "C" const char* rqamd_last_error(void) { return rq_err_buf; }

// Diagnostics: kernel variants are chosen by the row count (batch); tests on a handful of rows set a factor so that the
// SELECTION logic sees rows * factor and the large-batch variants (LDS-DMA / register-blocked GEMM tiles, two-head attention
// wavefronts, wave-per-row resid_ln) run on small inputs -- in the host emulator too.  Results must not change.
int g_rq_row_scale = 1;
extern "C" int rqamd_dbg_set_row_scale(int factor) {
    if (factor < 1) return rq_fail(RQAMD_ERR_INVALID, "dbg_set_row_scale: factor %d < 1", factor);
    g_rq_row_scale = factor;
    return RQAMD_OK;
}

// Diagnostics: what dense bf16 MFMA rate does this board SUSTAIN?  One 512-thread workgroup per CU, every wavefront issuing
// v_mfma_f32_32x32x16_bf16 back to back on eight accumulators from registers alone (no LDS, no memory traffic).  mode 0: constant
// operands (the instruction-rate peak the data sheet quotes: 2.5 PFLOP/s); mode 1: every MFMA sees operands that differ from the previous
// one's (eight rotating register sets of ~N(0,1) bf16 values), which is what a GEMM on real activations feeds the pipes -- and what the
// board's power / current limits let through is lower (profiles/r05_clock_under_load.txt: 1.70 PFLOP/s at 1.78 GHz).  bench.py times `launches`
// launches of n_per_wave MFMAs per wavefront and reports the rate beside `roofline.peak`.
#ifndef RQ_EMU
static __device__ __forceinline__ float rq_mfma_rate_rnd(unsigned& s) {
    float a = 0.f;
    for (int i = 0; i < 4; ++i) { s = s * 1664525u + 1013904223u; a += (float)(s >> 8) * (1.0f / 16777216.0f); }
    return (a - 2.0f) * 1.7320508f;
}
__global__ __launch_bounds__(512, 1) void mfma_rate_kernel(float* out, int n, int mode) {
    unsigned s = 12345u + threadIdx.x * 7919u + blockIdx.x * 104729u;
    bf16x8 x[8], y[8];
    for (int k = 0; k < 8; ++k)
        for (int e = 0; e < 8; ++e) {
            const float a = rq_mfma_rate_rnd(s), b = rq_mfma_rate_rnd(s);
            x[k][e] = (short)(__float_as_uint(mode == 0 ? 1.0f : a) >> 16);         // (bf16x8 holds raw bf16 bits)
            y[k][e] = (short)(__float_as_uint(mode == 0 ? 0.5f : b) >> 16);
        }
    f32x16 acc[8];
    for (int a = 0; a < 8; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int a = 0; a < 8; ++a) acc[a] = rq_mfma_32x32x16_bf16(x[a], y[(a * 3 + 1) & 7], acc[a]);
        const bf16x8 t = x[0];
#pragma unroll
        for (int a = 0; a < 7; ++a) x[a] = x[a + 1];
        x[7] = t;
    }
    float sum = 0.f;
    for (int a = 0; a < 8; ++a)
        for (int r = 0; r < 16; ++r) sum += acc[a][r];
    if (sum == 12345.678f) out[0] = sum;
}
#endif
extern "C" int rqamd_dbg_mfma_rate(int mode, int n_per_wave, int launches, float* scratch, double* flop_out, void* stream) {
#ifdef RQ_EMU
    (void)mode; (void)n_per_wave; (void)launches; (void)scratch; (void)flop_out; (void)stream;
    return rq_fail(RQAMD_ERR_INVALID, "dbg_mfma_rate: not available in the host emulator");
#else
    if (!scratch || !flop_out || n_per_wave < 8 || launches < 1 || mode < 0 || mode > 1) return rq_fail(RQAMD_ERR_INVALID, "dbg_mfma_rate: bad argument");
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(mfma_rate_kernel, dim3(cus), dim3(512), 0, (hipStream_t)stream, scratch, n_per_wave & ~7, mode);
    *flop_out = (double)launches * cus * 8.0 * (double)(n_per_wave & ~7) * 2.0 * 32 * 32 * 16;
    return rq_check_launch("mfma_rate_kernel");
#endif
}
