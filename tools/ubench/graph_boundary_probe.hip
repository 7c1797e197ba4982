// graph_boundary_probe.hip — does the way a chain of DEPENDENT launches is submitted change the boundary between them on MI355X?
// A chain of N kernels (256 x 1024 threads, each busy for ~3 us) logs per kernel min(first instruction) and max(exit) on s_memrealtime
// (100 MHz).  gap(i) = first(i+1) - exit(i).  Submission modes:
//   stream        hipLaunchKernelGGL x N on one stream (what the product does)
//   graph         the same chain captured once and replayed with hipGraphLaunch
//   anyorder      hipExtLaunchKernelGGL(..., hipExtAnyOrderLaunch): no barrier bit — NOT a legal way to run dependent work, shown only
//                 as the floor of the command processor's dispatch latency (negative gap = the kernels overlapped)
// build: hipcc --offload-arch=gfx950 -O3 graph_boundary_probe.hip -o graph_boundary_probe.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <chrono>
__global__ __launch_bounds__(1024) void kc(unsigned long long* ts, int idx, int spin) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t = t0;
    while (t - t0 < (unsigned long long)spin) t = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(&ts[2 * idx], t0);
        atomicMax(&ts[2 * idx + 1], __builtin_amdgcn_s_memrealtime());
    }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const int N = 8, reps = 200;
    unsigned long long* ts;
    CK(hipMalloc((void**)&ts, 2 * N * 8));
    std::vector<unsigned long long> init(2 * N), h(2 * N);
    for (int i = 0; i < N; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(kc, dim3(256), dim3(1024), 0, st, ts, i, 300);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    const char* names[] = {"stream", "graph", "anyorder"};
    for (int mode = 0; mode < 3; ++mode) {
        std::vector<double> gaps, spans;
        double host = 0;
        for (int rep = 0; rep < reps + 5; ++rep) {
            CK(hipMemcpyAsync(ts, init.data(), 2 * N * 8, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            const auto h0 = std::chrono::steady_clock::now();
            if (mode == 0) for (int i = 0; i < N; ++i) hipLaunchKernelGGL(kc, dim3(256), dim3(1024), 0, st, ts, i, 300);
            if (mode == 1) CK(hipGraphLaunch(exec, st));
            if (mode == 2) for (int i = 0; i < N; ++i) hipExtLaunchKernelGGL(kc, dim3(256), dim3(1024), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, ts, i, 300);
            const auto h1 = std::chrono::steady_clock::now();
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(h.data(), ts, 2 * N * 8, hipMemcpyDeviceToHost));
            if (rep < 5) continue;
            host += std::chrono::duration<double, std::micro>(h1 - h0).count();
            for (int i = 0; i + 1 < N; ++i) gaps.push_back(((double)h[2 * (i + 1)] - (double)h[2 * i + 1]) * 0.01);
            spans.push_back(((double)h[2 * (N - 1) + 1] - (double)h[0]) * 0.01);
        }
        std::sort(gaps.begin(), gaps.end()); std::sort(spans.begin(), spans.end());
        printf("%-9s: gap last exit -> first instruction median %.2f us (p10 %.2f, p90 %.2f) | chain of %d: %.2f us | host time to submit: %.2f us\n", names[mode],
               gaps[gaps.size() / 2], gaps[gaps.size() / 10], gaps[gaps.size() * 9 / 10], N, spans[spans.size() / 2], host / reps);
    }
    return 0;
}
