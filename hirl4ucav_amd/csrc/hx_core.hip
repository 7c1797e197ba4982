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
/* diagnostic builds only (make stamps): the 80 phase-stamp floats of the kernel files -> host memory; returns -1 in the shipped build */
int hx_debug_stamps(float* host_out) {
#ifdef HX_STAMPS
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    for (int i = 0; i < 80; ++i) host_out[i] = 0.0f;
    if (int rc = hx::dbg_stamps_fwdbwd(host_out)) return rc;
    if (int rc = hx::dbg_stamps_wgrad(host_out)) return rc;
    return hx::dbg_stamps_act(host_out);
#else
    (void)host_out;
    return -1;
#endif
}
/* diagnostic builds only: the 80 stamp words of the persistent acting kernel (hx_actp.hip keeps its own) */
int hx_debug_stamps_actp(float* host_out) {
#ifdef HX_STAMPS
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    for (int i = 0; i < 80; ++i) host_out[i] = 0.0f;
    return hx::dbg_stamps_actp(host_out);
#else
    (void)host_out;
    return -1;
#endif
}
/* diagnostic builds only: the 80 stamp words of the front launch (hx_front.hip: [0..7] a launch-B workgroup, [8..15] a launch-A workgroup, [56..] an acting one) */
int hx_debug_stamps_front(float* host_out) {
#ifdef HX_STAMPS
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    for (int i = 0; i < 80; ++i) host_out[i] = 0.0f;
    return hx::dbg_stamps_front(host_out);
#else
    (void)host_out;
    return -1;
#endif
}
/* diagnostic builds only: the workgroup life-span logs (start, end in 10 ns ticks; tag = HX_SPAN_* kernel id) -> host, then cleared */
int hx_debug_spans(unsigned long long* host_spans /* [8192][2] */, unsigned* host_tags /* [8192] */, unsigned* host_n) {
#ifdef HX_STAMPS
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    *host_n = 0;
    if (int rc = hx::dbg_spans_fwdbwd(host_spans, host_tags, host_n, 8192u)) return rc;
    if (int rc = hx::dbg_spans_wgrad(host_spans, host_tags, host_n, 8192u)) return rc;
    if (int rc = hx::dbg_spans_front(host_spans, host_tags, host_n, 8192u)) return rc;
    return hx::dbg_spans_act(host_spans, host_tags, host_n, 8192u);
#else
    (void)host_spans; (void)host_tags; (void)host_n;
    return -1;
#endif
}
int hx_debug_copy_dword(const float* src, float* dst, int64_t n, void* stream) {
    HX_REQUIRE(src && dst && n > 0, "hx_debug_copy_dword: bad arguments");
    hipLaunchKernelGGL(calib_copy_dword, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, (long long)n);
    HX_CHECK_LAUNCH("hx_debug_copy_dword");
    return 0;
}
void* hx_event_create(void) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) {
        hx::fail(HX_ERR_ARG, "hx_event_create: hipEventCreate failed");
        return nullptr;
    }
    return (void*)e;
}
int hx_event_destroy(void* ev) {
    HX_REQUIRE(ev, "hx_event_destroy: null event");
    HX_REQUIRE(hipEventDestroy((hipEvent_t)ev) == hipSuccess, "hx_event_destroy: hipEventDestroy failed");
    return 0;
}
int hx_event_elapsed_us(void* start, void* stop, float* us) {
    HX_REQUIRE(start && stop && us, "hx_event_elapsed_us: bad arguments");
    float ms = 0.0f;
    HX_REQUIRE(hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop) == hipSuccess, "hx_event_elapsed_us: events not complete");
    *us = ms * 1000.0f;
    return 0;
}
const char* hx_last_error(void) { return hx::error_buffer(); }
int hx_version(void) { return HX_ABI_VERSION; }
int hx_abi_sizes(int32_t* sizes8) {
    HX_REQUIRE(sizes8, "hx_abi_sizes: null");
    const int32_t v[8] = {(int32_t)sizeof(HxStepOpts), (int32_t)sizeof(HxNets), (int32_t)sizeof(HxHyper), (int32_t)sizeof(HxBatch), (int32_t)sizeof(HxSample),
                          (int32_t)sizeof(HxSacNets), (int32_t)sizeof(HxSacBatch), HX_STAT_WAYS * HX_STAT_PITCH};
    for (int i = 0; i < 8; ++i) sizes8[i] = v[i];
    return 0;
}
}
