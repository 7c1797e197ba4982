// Shared host-side helpers of libhx_mi355.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/hirl4ucav.h"
#include "../../include/hirl4ucav_debug.h"

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

#ifdef HX_HOST_DRYRUN
// make asan-host (csrc/Makefile): the host side alone, compiled with --offload-host-only under AddressSanitizer and driven on a box WITHOUT a GPU
// (tools/asan_host.sh).  Every launch fails there by construction; the dry run drops that error so that the host code BEHIND a first launch — the
// later stages of hx_hirl_learn*, hx_sac_learn, the front launch's back half — runs under the sanitizer too.  Never defined in the shipped build.
#define HX_CHECK_LAUNCH(what) \
    do {                      \
        (void)hipGetLastError(); \
    } while (0)
#else
#define HX_CHECK_LAUNCH(what)                                                                    \
    do {                                                                                         \
        hipError_t e_ = hipGetLastError();                                                       \
        if (e_ != hipSuccess) return ::hx::fail(HX_ERR_HIP, "%s: %s", what, hipGetErrorString(e_)); \
    } while (0)
#endif

#define HX_CHECK_HIP(expr)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return ::hx::fail(HX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// diagnostic build (make stamps) only: each kernel file hands its share of the phase stamps / workgroup spans to hx_core.hip (hx_update.h)
int dbg_stamps_fwdbwd(float* host80);
int dbg_stamps_wgrad(float* host80);
int dbg_stamps_act(float* host80);
int dbg_stamps_actp(float* host80);
int dbg_spans_fwdbwd(unsigned long long* spans, unsigned* tags, unsigned* n, unsigned cap);
int dbg_spans_wgrad(unsigned long long* spans, unsigned* tags, unsigned* n, unsigned cap);
int dbg_spans_act(unsigned long long* spans, unsigned* tags, unsigned* n, unsigned cap);
int dbg_spans_front(unsigned long long* spans, unsigned* tags, unsigned* n, unsigned cap);
int dbg_stamps_front(float* host80);

}  // namespace hx
