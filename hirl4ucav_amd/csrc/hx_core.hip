// hx_core.hip — error string and version of libhx_mi355.so
#include "hx_common.h"

namespace hx {
char* error_buffer() {
    static thread_local char buf[512] = "";
    return buf;
}
}  // namespace hx

// PMC calibration (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are calibrated per access width): a copy with the env
// kernel's access shape — one dword per lane, 256 contiguous bytes per wave-instruction — over a known byte count
__global__ void calib_copy_dword(const float* __restrict__ src, float* __restrict__ dst, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

extern "C" {
int hx_debug_copy_dword(const float* src, float* dst, int64_t n, void* stream) {
    HX_REQUIRE(src && dst && n > 0, "hx_debug_copy_dword: bad arguments");
    hipLaunchKernelGGL(calib_copy_dword, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, (long long)n);
    HX_CHECK_LAUNCH("hx_debug_copy_dword");
    return 0;
}
const char* hx_last_error(void) { return hx::error_buffer(); }
int hx_version(void) { return 100; }
}
