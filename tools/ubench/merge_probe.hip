// merge_probe.hip — what does riding one launch's workgroups inside another launch buy on MI355X, and what does a second stream cost?
// (Round-3 verdict item 6: launch A of learn() as extra workgroups of the act + env launch.)
// Workgroups of 1,024 threads + 100 KB of LDS (one per CU, like every kernel of the step) spin for a fixed time on s_memrealtime (100 MHz).
// One "step" = the headline's critic-only step in the kernels' measured in-workgroup times:
//   act+env 256 wg x 19.0 us | A 192 x 8.0 | B 128 x 6.5 | C 256 x 6.5 | D 256 x 6.5
//   serial     five dependent launches on one stream                                  (what the product does)
//   A-first    [A | act+env] as ONE launch of 448 workgroups, A's first               (the verdict's sketch)
//   act-first  [act+env | A] as one launch
//   D-merge    [D of the previous step | act+env] as one launch, then A, B, C          (no change of the draw's meaning; critic-only steps only)
//   2-stream   A on a second stream beside act+env, joined with events (hipEventDisableTiming) before B
//   A-32       [act+env as 128 wg x 25.5 us | A] — 32-row acting workgroups leave half the CUs to A
// build: hipcc --offload-arch=gfx950 -O3 merge_probe.hip -o merge_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(1024) void spin(int n0, int t0_ticks, int t1_ticks, int* sink) {
    extern __shared__ int lds[];
    const unsigned long long s = __builtin_amdgcn_s_memrealtime();
    const unsigned long long want = (int)blockIdx.x < n0 ? t0_ticks : t1_ticks;
    while (__builtin_amdgcn_s_memrealtime() - s < want) {}
    if (threadIdx.x == 0) lds[0] = 1;
    __syncthreads();
    if (sink && threadIdx.x == 0 && lds[0] == 2) *sink = 1;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static const int kLds = 100 * 1024;
static void L(hipStream_t st, int n0, int t0, int n1, int t1) { hipLaunchKernelGGL(spin, dim3(n0 + n1), dim3(1024), kLds, st, n0, t0, t1, (int*)nullptr); }
int main() {
    CK(hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t eA, eD;
    CK(hipEventCreateWithFlags(&eA, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eD, hipEventDisableTiming));
    const int ACT = 1900, A = 800, B = 650, C = 650, D = 650, ACT32 = 2550;
    const char* names[] = {"serial", "A-first", "act-first", "D-merge", "2-stream", "A-32"};
    const int steps = 400;
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 6; ++mode) {
            CK(hipDeviceSynchronize());
            const auto h0 = std::chrono::steady_clock::now();
            for (int k = 0; k < steps; ++k) {
                switch (mode) {
                case 0: L(s1, 256, ACT, 0, 0); L(s1, 192, A, 0, 0); L(s1, 128, B, 0, 0); L(s1, 256, C, 0, 0); L(s1, 256, D, 0, 0); break;
                case 1: L(s1, 192, A, 256, ACT); L(s1, 128, B, 0, 0); L(s1, 256, C, 0, 0); L(s1, 256, D, 0, 0); break;
                case 2: L(s1, 256, ACT, 192, A); L(s1, 128, B, 0, 0); L(s1, 256, C, 0, 0); L(s1, 256, D, 0, 0); break;
                case 3: L(s1, 256, D, 256, ACT); L(s1, 192, A, 0, 0); L(s1, 128, B, 0, 0); L(s1, 256, C, 0, 0); break;
                case 4:
                    if (k) CK(hipStreamWaitEvent(s2, eD, 0));
                    L(s2, 192, A, 0, 0); CK(hipEventRecord(eA, s2));
                    L(s1, 256, ACT, 0, 0); CK(hipStreamWaitEvent(s1, eA, 0));
                    L(s1, 128, B, 0, 0); L(s1, 256, C, 0, 0); L(s1, 256, D, 0, 0); CK(hipEventRecord(eD, s1));
                    break;
                case 5: L(s1, 128, ACT32, 192, A); L(s1, 128, B, 0, 0); L(s1, 256, C, 0, 0); L(s1, 256, D, 0, 0); break;
                }
            }
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count() / steps;
            if (rep) printf("%-10s %7.2f us per step\n", names[mode], us);
        }
    return 0;
}
