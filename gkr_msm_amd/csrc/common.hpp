// Shared host-side plumbing for libgkrmsm_hip.so: error reporting and launch helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/gkrmsm.h"

namespace gm {

inline char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

inline int32_t set_err(int32_t code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define GM_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return gm::set_err(GM_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call,                \
                               hipGetErrorString(e__));                                                \
    } while (0)

#define GM_REQUIRE(cond, ...)                                                                          \
    do {                                                                                               \
        if (!(cond)) return gm::set_err(GM_ERR_INVALID, __VA_ARGS__);                                  \
    } while (0)

#define GM_LAUNCH_CHECK() GM_HIP(hipGetLastError())

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline unsigned ceil_div(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace gm
