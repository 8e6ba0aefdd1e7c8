// api.hip -- ABI-level entry points that do not belong to one engine (see include/rqamd.h)
#include "rq_common.h"

thread_local char rq_err_buf[512] = {0};

extern "C" int rqamd_abi_version(void) { return RQAMD_ABI_VERSION; }
extern "C" const char* rqamd_last_error(void) { return rq_err_buf; }
