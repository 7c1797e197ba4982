// kernarg_probe.hip — what does the first scalar load of a kernel cost?  Chains of 2,000 dependent tiny launches (256 workgroups x 1024
// threads, like the update kernels), each reading ONE 64-byte job line and then one vector load + store:
//   A  job lines inside the kernel argument block (by value, indexed by blockIdx.y): cold every launch (the host writes it)
//   B  job lines in a device table that stays resident (read every launch), table pointer a plain kernel argument
//   C  as B with the pointer preloaded into SGPRs by the command processor (-mllvm -amdgpu-kernarg-preload-count=8)
// build: hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-kernarg-preload-count=8 -DPRELOAD] kernarg_probe.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Line { float* p[6]; unsigned cfg; int rows; float a, b; };
struct Args { Line job[6]; };
__global__ __launch_bounds__(1024) void byval(Args A, int it) {
    const Line L = A.job[blockIdx.y];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    L.p[1][i] = L.p[0][i] + L.a + (float)(L.cfg + it);
}
__global__ __launch_bounds__(1024) void bytable(const Line* __restrict__ table, int it) {
    const Line L = table[blockIdx.y];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    L.p[1][i] = L.p[0][i] + L.a + (float)(L.cfg + it);
}
int main() {
    float *x, *y; Line* tab;
    hipMalloc(&x, 4 << 20); hipMalloc(&y, 4 << 20); hipMalloc(&tab, sizeof(Args));
    hipMemset(x, 0, 4 << 20);
    Args A{};
    for (int j = 0; j < 6; ++j) { A.job[j].p[0] = (j & 1) ? y : x; A.job[j].p[1] = (j & 1) ? x : y; A.job[j].a = 1.0f; A.job[j].cfg = j; }
    hipMemcpy(tab, &A, sizeof A, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 2000;
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        hipEventRecord(e0);
        for (int k = 0; k < N; ++k) hipLaunchKernelGGL(byval, dim3(64, 4), dim3(1024), 0, 0, A, k);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("by value   %.3f us per launch\n", ms * 1e3 / N);
        hipEventRecord(e0);
        for (int k = 0; k < N; ++k) hipLaunchKernelGGL(bytable, dim3(64, 4), dim3(1024), 0, 0, tab, k);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
#ifdef PRELOAD
        printf("table+preload %.3f us per launch\n", ms * 1e3 / N);
#else
        printf("table      %.3f us per launch\n", ms * 1e3 / N);
#endif
    }
    return 0;
}
