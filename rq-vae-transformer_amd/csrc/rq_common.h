// rq_common.h -- host-side helpers shared by the librqamd translation units.
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <atomic>
#include <string>
#include "rq_hip.h"
#include "../../include/rqamd.h"

extern thread_local char rq_err_buf[512];

static inline int rq_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(rq_err_buf, sizeof(rq_err_buf), fmt, ap);
    va_end(ap);
    return code;
}

static inline int rq_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rq_fail(RQAMD_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return RQAMD_OK;
}

#define RQ_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) return rq_fail(RQAMD_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

#define RQ_TRY(call)                \
    do {                            \
        int s_ = (call);            \
        if (s_ != RQAMD_OK) return s_; \
    } while (0)

// true exactly once per (call site, device): kernel attributes such as the dynamic-LDS limit are per device, and a process
// may drive several (model.to('cuda:1') with current device 0, one engine per device)
struct RqDeviceOnce {
    std::atomic<unsigned long long> seen{0};
    bool first() {
        int d = 0;
        (void)hipGetDevice(&d);
        const unsigned long long bit = 1ull << (d & 63);
        return !(seen.fetch_or(bit) & bit);
    }
};

// device buffer with RAII (engine-owned workspace)
extern int g_rq_row_scale;     // diagnostics (api.hip): variant selection sees rows * this factor

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int reserve(size_t n, unsigned ext_flags = 0) {      // ext_flags != 0: hipExtMallocWithFlags (diagnostics: RQAMD_KV_UNCACHED)
        if (n <= bytes) return RQAMD_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = ext_flags ? hipExtMallocWithFlags(&p, n, ext_flags) : hipMalloc(&p, n);
        if (ext_flags && e != hipSuccess && e != hipErrorOutOfMemory) {      // a runtime without that memory type: the ordinary kind
            (void)hipGetLastError();
            p = nullptr;
            e = hipMalloc(&p, n);
        }
        if (e != hipSuccess) {
            // the runtime keeps a failed call as its "last error" until somebody reads it: clear it here, or the next
            // rq_check_launch (e.g. of the retried call, after the caller released cached memory) reports this stale failure
            (void)hipGetLastError();
            p = nullptr;
            return rq_fail(e == hipErrorOutOfMemory ? RQAMD_ERR_NOMEM : RQAMD_ERR_HIP, "hipMalloc(%zu): %s", n, hipGetErrorString(e));
        }
        bytes = n;
        return RQAMD_OK;
    }
    template <typename T> T* as() const { return (T*)p; }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    ~DevBuf() { if (p) (void)hipFree(p); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};
