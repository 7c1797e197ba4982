// Shared host-side helpers of libhx_mi355.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/hirl4ucav.h"

namespace hx {

char* error_buffer();  // thread-local, 512 bytes

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define HX_REQUIRE(cond, ...)                                    \
    do {                                                         \
        if (!(cond)) return ::hx::fail(HX_ERR_ARG, __VA_ARGS__); \
    } while (0)

#define HX_CHECK_LAUNCH(what)                                                                    \
    do {                                                                                         \
        hipError_t e_ = hipGetLastError();                                                       \
        if (e_ != hipSuccess) return ::hx::fail(HX_ERR_HIP, "%s: %s", what, hipGetErrorString(e_)); \
    } while (0)

#define HX_CHECK_HIP(expr)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return ::hx::fail(HX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

}  // namespace hx
