// permlane_rate.hip — what a cross-row sum costs on gfx950: v_permlane16_swap / v_permlane32_swap (hx_act.h sum_rows4) against the DPP butterfly inside a 16-lane
// row (sum16u), ds_swizzle and ds_bpermute, as dependent chains with 1 / 4 / 16 waves per CU.   hipcc --offload-arch=gfx950 -O2 permlane_rate.hip -o permlane_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) { return v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true)); }
__device__ __forceinline__ float sum16u(float v) { v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); v = dpp_add<0x141>(v); v = dpp_add<0x140>(v); return v; }
__device__ __forceinline__ float sum_rows4(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float sum_rows4_bperm(float v) {
    const int l = threadIdx.x & 63;
    v += __int_as_float(__builtin_amdgcn_ds_bpermute((l ^ 16) << 2, __float_as_int(v)));
    v += __int_as_float(__builtin_amdgcn_ds_bpermute((l ^ 32) << 2, __float_as_int(v)));
    return v;
}
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    float v = 1.0f + threadIdx.x * 1e-3f, w = 0.5f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) v = sum16u(v) * 0.0625f;        // 4 DPP adds + 1 mul
        if (MODE == 1) v = sum_rows4(v) * 0.25f;        // 2 swaps + 2 adds + 1 mul
        if (MODE == 2) v = sum_rows4_bperm(v) * 0.25f;  // 2 ds_bpermute + 2 adds + 1 mul
        if (MODE == 3) { v = __builtin_fmaf(v, 0.999f, w); v = __builtin_fmaf(v, 1.001f, -w); v = __builtin_fmaf(v, 0.999f, w); v = __builtin_fmaf(v, 1.001f, -w); v *= 1.0f; }  // 5 plain dependent VALU
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
template <int MODE>
static void run(const char* name, int threads) {
    float* d;
    (void)hipMalloc(&d, 256 * 1024 * 4);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 100);
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    printf("%-34s %4d threads per CU: %7.1f ns per chain step (5 dependent instructions) = %5.1f clocks at 2.4 GHz\n", name, threads, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
    (void)hipFree(d);
}
int main() {
    for (int t : {64, 256, 1024}) {
        run<3>("5 plain dependent v_fma", t);
        run<0>("sum16u (4 DPP adds)", t);
        run<1>("sum_rows4 (2 permlane swaps)", t);
        run<2>("sum_rows4 via 2 ds_bpermute", t);
    }
    return 0;
}
