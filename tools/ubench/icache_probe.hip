// icache_probe.hip — what does COLD code cost on MI355X?  Every launch of a step runs a different kernel, so every kernel starts with an
// instruction cache that holds somebody else's code.  One wave per workgroup executes a straight line of N dependent v_fma_f32 (8 bytes
// each, N = 2,048: 16 KB of code) three times inside ONE launch: pass 0 runs cold, passes 1-2 from the instruction cache.  Variant B: 16
// waves per workgroup run the same line (the first wave takes the misses for all).  Before every timed launch another kernel with its own
// 48 KB line runs on the same CUs, as the previous launch of a step would.
// build: hipcc --offload-arch=gfx950 -O3 icache_probe.hip -o icache_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define F1(v) v = __builtin_fmaf(v, a, b);
#define F8(v) F1(v) F1(v) F1(v) F1(v) F1(v) F1(v) F1(v) F1(v)
#define F64(v) F8(v) F8(v) F8(v) F8(v) F8(v) F8(v) F8(v) F8(v)
#define F512(v) F64(v) F64(v) F64(v) F64(v) F64(v) F64(v) F64(v) F64(v)
#define F2048(v) F512(v) F512(v) F512(v) F512(v)
__global__ __launch_bounds__(1024) void line(float* out, unsigned long long* ts, float a, float b, int passes) {
    float v = (float)threadIdx.x;
    unsigned long long t[5];
    t[0] = __builtin_amdgcn_s_memrealtime();
    for (int p = 0; p < passes; ++p) {
        F2048(v)
        asm volatile("" : "+v"(v));
        t[p + 1] = __builtin_amdgcn_s_memrealtime();
    }
    if (threadIdx.x == 0) for (int p = 0; p <= passes; ++p) ts[blockIdx.x * 5 + p] = t[p];
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
__global__ __launch_bounds__(1024) void evict(float* out, float a, float b) {  // 3 x 16 KB of other code through the same caches
    float v = (float)threadIdx.x, w = v + 1.0f, x = v + 2.0f;
    F2048(v) F2048(w) F2048(x)
    out[blockIdx.x * blockDim.x + threadIdx.x] = v + w + x;
}
int main() {
    float* out; unsigned long long* ts;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&ts, 256 * 5 * 8);
    std::vector<unsigned long long> h(256 * 5);
    for (int threads : {64, 1024}) {
        for (int warm : {0, 1}) {
            double acc[3] = {0, 0, 0};
            const int reps = 20;
            for (int rep = 0; rep < reps; ++rep) {
                if (!warm) hipLaunchKernelGGL(evict, dim3(256), dim3(1024), 0, 0, out, 0.999f, 0.001f);
                else hipLaunchKernelGGL(line, dim3(256), dim3(threads), 0, 0, out, ts, 0.999f, 0.001f, 3);
                hipLaunchKernelGGL(line, dim3(256), dim3(threads), 0, 0, out, ts, 0.999f, 0.001f, 3);
                hipDeviceSynchronize();
                hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost);
                for (int p = 0; p < 3; ++p) {
                    std::vector<double> d(256);
                    for (int b = 0; b < 256; ++b) d[b] = (double)(h[b * 5 + p + 1] - h[b * 5 + p]) * 0.01;
                    std::sort(d.begin(), d.end());
                    acc[p] += d[128];
                }
            }
            printf("%4d threads/workgroup, previous launch = %s: 2,048 dependent v_fma (16 KB) median over workgroups: pass0 %.2f us  pass1 %.2f us  pass2 %.2f us\n",
                   threads, warm ? "the same kernel" : "another kernel (48 KB of code)", acc[0] / reps, acc[1] / reps, acc[2] / reps);
        }
    }
    return 0;
}
