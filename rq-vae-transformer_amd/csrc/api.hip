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
