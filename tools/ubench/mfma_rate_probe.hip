// mfma_rate_probe.hip — what does the acting kernel's fp32 MFMA work take with operands already in registers?
// 256 workgroups x 16 waves; every wave issues N v_mfma_f32_16x16x4_f32 in two independent accumulator chains (the acting kernel: 128 per
// wave and 16-row tile = 2,048 per workgroup).  Also the bf16 form (v_mfma_f32_16x16x32_bf16, 16 per wave for the same tile).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
template <int N>
__global__ __launch_bounds__(1024) void k32(float* out, float a, float b) {
    v4f c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    float x = a + threadIdx.x, y = b + threadIdx.x;
#pragma unroll 16
    for (int i = 0; i < N / 2; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, c1, 0, 0, 0);
    }
    out[blockIdx.x * 1024 + threadIdx.x] = c0[0] + c1[1] + c0[2] + c1[3];
}
template <int N>
__global__ __launch_bounds__(1024) void k16(float* out, float a) {
    v4f c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    v8bf x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(a + j); y[j] = (__bf16)(a - j); }
#pragma unroll 8
    for (int i = 0; i < N / 2; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, c1, 0, 0, 0);
    }
    out[blockIdx.x * 1024 + threadIdx.x] = c0[0] + c1[1] + c0[2] + c1[3];
}
template <typename F>
static double timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(e0);
    for (int i = 0; i < 500; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / 500;
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    const double e = timeit([&] { hipLaunchKernelGGL(k32<2>, dim3(256), dim3(1024), 0, 0, out, 1.f, 2.f); });
    const double t128 = timeit([&] { hipLaunchKernelGGL(k32<128>, dim3(256), dim3(1024), 0, 0, out, 1.f, 2.f); });
    const double t1024 = timeit([&] { hipLaunchKernelGGL(k32<1024>, dim3(256), dim3(1024), 0, 0, out, 1.f, 2.f); });
    const double b16 = timeit([&] { hipLaunchKernelGGL(k16<16>, dim3(256), dim3(1024), 0, 0, out, 1.f); });
    const double b1024 = timeit([&] { hipLaunchKernelGGL(k16<1024>, dim3(256), dim3(1024), 0, 0, out, 1.f); });
    printf("launch floor (2 MFMAs per wave)           %.2f us\n", e);
    printf("fp32 16x16x4 : 128 per wave (one act tile) %.2f us  -> %.2f us of MFMA work; 1024 per wave %.2f us = %.1f TFLOP/s\n", t128, t128 - e, t1024,
           256.0 * 16 * 1024 * 2048 / (t1024 - e) * 1e-6);
    printf("bf16 16x16x32:  16 per wave (one act tile) %.2f us; 1024 per wave %.2f us = %.1f TFLOP/s\n", b16, b1024, 256.0 * 16 * 1024 * 16384 / (b1024 - e) * 1e-6);
    return 0;
}
