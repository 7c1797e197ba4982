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
__global__ __launch_bounds__(1024) void kc(unsigned long long* ts, int idx, int spin, float* buf, int n4_per_wg, int nt) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t = t0;
    while (t - t0 < (unsigned long long)spin) t = __builtin_amdgcn_s_memrealtime();
    if (n4_per_wg) {  // dirty n4_per_wg float4 per workgroup (the same lines in every kernel of the chain: like the update's slots)
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4* p = reinterpret_cast<f4*>(buf) + (size_t)blockIdx.x * n4_per_wg;
        const f4 v = {(float)idx, 1.f, 2.f, 3.f};
        if (nt) for (int i = threadIdx.x; i < n4_per_wg; i += 1024) __builtin_nontemporal_store(v, p + i);
        else for (int i = threadIdx.x; i < n4_per_wg; i += 1024) p[i] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // (one slot per workgroup: atomics on one address would add their own drain to the boundary)
        ts[(2 * idx) * 256 + blockIdx.x] = t0;
        ts[(2 * idx + 1) * 256 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const int N = 8, reps = 200;
    unsigned long long* ts;
    CK(hipMalloc((void**)&ts, 2 * N * 256 * 8));
    std::vector<unsigned long long> raw(2 * N * 256), h(2 * N);
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(kc, dim3(256), dim3(1024), 0, st, ts, i, 300, (float*)nullptr, 0, 0);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    const char* names[] = {"stream", "graph", "anyorder"};
    for (int mode = 0; mode < 3; ++mode) {
        std::vector<double> gaps, spans;
        double host = 0;
        for (int rep = 0; rep < reps + 5; ++rep) {
            CK(hipStreamSynchronize(st));
            const auto h0 = std::chrono::steady_clock::now();
            if (mode == 0) for (int i = 0; i < N; ++i) hipLaunchKernelGGL(kc, dim3(256), dim3(1024), 0, st, ts, i, 300, (float*)nullptr, 0, 0);
            if (mode == 1) CK(hipGraphLaunch(exec, st));
            if (mode == 2) for (int i = 0; i < N; ++i) hipExtLaunchKernelGGL(kc, dim3(256), dim3(1024), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, ts, i, 300, (float*)nullptr, 0, 0);
            const auto h1 = std::chrono::steady_clock::now();
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(raw.data(), ts, 2 * N * 256 * 8, hipMemcpyDeviceToHost));
            for (int i = 0; i < N; ++i) {
                h[2 * i] = *std::min_element(raw.begin() + (2 * i) * 256, raw.begin() + (2 * i + 1) * 256);
                h[2 * i + 1] = *std::max_element(raw.begin() + (2 * i + 1) * 256, raw.begin() + (2 * i + 2) * 256);
            }
            if (rep < 5) continue;
            host += std::chrono::duration<double, std::micro>(h1 - h0).count();
            for (int i = 0; i + 1 < N; ++i) gaps.push_back(((double)h[2 * (i + 1)] - (double)h[2 * i + 1]) * 0.01);
            spans.push_back(((double)h[2 * (N - 1) + 1] - (double)h[0]) * 0.01);
        }
        std::sort(gaps.begin(), gaps.end()); std::sort(spans.begin(), spans.end());
        printf("%-9s: gap last exit -> first instruction median %.2f us (p10 %.2f, p90 %.2f) | chain of %d: %.2f us | host time to submit: %.2f us\n", names[mode],
               gaps[gaps.size() / 2], gaps[gaps.size() / 10], gaps[gaps.size() * 9 / 10], N, spans[spans.size() / 2], host / reps);
    }
    // the same chain on the stream, every kernel dirtying `mb` MB (plain / non-temporal stores): what does the end-of-kernel release cost?
    float* buf; CK(hipMalloc((void**)&buf, 64ll << 20));
    for (int nt = 0; nt < 2; ++nt)
        for (int mb : {0, 1, 2, 4, 8, 16, 32}) {
            std::vector<double> gaps, spans;
            const int n4 = mb * (1 << 20) / 16 / 256;
            for (int rep = 0; rep < reps + 5; ++rep) {
                CK(hipStreamSynchronize(st));
                for (int i = 0; i < N; ++i) hipLaunchKernelGGL(kc, dim3(256), dim3(1024), 0, st, ts, i, 300, buf, n4, nt);
                CK(hipStreamSynchronize(st));
                CK(hipMemcpy(raw.data(), ts, 2 * N * 256 * 8, hipMemcpyDeviceToHost));
                for (int i = 0; i < N; ++i) {
                    h[2 * i] = *std::min_element(raw.begin() + (2 * i) * 256, raw.begin() + (2 * i + 1) * 256);
                    h[2 * i + 1] = *std::max_element(raw.begin() + (2 * i + 1) * 256, raw.begin() + (2 * i + 2) * 256);
                }
                if (rep < 5) continue;
                for (int i = 2; i + 1 < N; ++i) gaps.push_back(((double)h[2 * (i + 1)] - (double)h[2 * i + 1]) * 0.01);  // (from the 3rd kernel on: the host is ahead by then)
                spans.push_back(((double)h[2 * (N - 1) + 1] - (double)h[0]) * 0.01);
            }
            std::sort(gaps.begin(), gaps.end()); std::sort(spans.begin(), spans.end());
            printf("each kernel dirties %2d MB with %s stores: gap last exit (stores issued) -> first instruction of the next median %.2f us (p10 %.2f, p90 %.2f) | chain of %d: %.2f us\n", mb,
                   nt ? "non-temporal" : "plain       ", gaps[gaps.size() / 2], gaps[gaps.size() / 10], gaps[gaps.size() * 9 / 10], N, spans[spans.size() / 2]);
        }
    return 0;
}
