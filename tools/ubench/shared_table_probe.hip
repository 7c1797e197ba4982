// shared_table_probe.hip — how fast can EVERY workgroup stream the same 512 KB table (W2 of the acting kernel) out of L2?
// 256 workgroups x 1024 threads; the table is [512 columns][256 floats] (1 KB per column); per "chunk" of 16 k-values a workgroup reads
// 64 B of every column.  Patterns:
//   0  half lines: thread -> (column tid >> 2, 16-B piece tid & 3), +256 columns for the second load; chunk c = bytes [64 c, 64 c + 64)
//   1  the same, two chunks (both halves of each 128-B line) requested back to back
//   2  full lines: thread -> (column tid >> 3, piece tid & 7): 128 B per column and instruction, 4 loads cover 512 columns x 2 chunks
// Loads are summed into a register (no LDS, no MFMA): this is the pure fetch rate.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int PAT>
__global__ __launch_bounds__(1024) void k(const float* __restrict__ w, float* out) {
    const int tid = threadIdx.x;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add = [&](const float4 v) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; };
    if (PAT == 0) {
        const float* p = w + (size_t)(tid >> 2) * 256 + (tid & 3) * 4;
#pragma unroll 4
        for (int c = 0; c < 16; ++c) {
            add(*reinterpret_cast<const float4*>(p + c * 16));
            add(*reinterpret_cast<const float4*>(p + (size_t)256 * 256 + c * 16));
        }
    } else if (PAT == 1) {
        const float* p = w + (size_t)(tid >> 2) * 256 + (tid & 3) * 4;
#pragma unroll 2
        for (int c = 0; c < 16; c += 2) {
            const float4 a = *reinterpret_cast<const float4*>(p + c * 16), b = *reinterpret_cast<const float4*>(p + c * 16 + 16);
            const float4 d = *reinterpret_cast<const float4*>(p + (size_t)256 * 256 + c * 16), e = *reinterpret_cast<const float4*>(p + (size_t)256 * 256 + c * 16 + 16);
            add(a); add(b); add(d); add(e);
        }
    } else {
        const float* p = w + (size_t)(tid >> 3) * 256 + (tid & 7) * 4;
#pragma unroll 2
        for (int d = 0; d < 8; ++d) {
#pragma unroll
            for (int q = 0; q < 4; ++q) add(*reinterpret_cast<const float4*>(p + (size_t)(128 * q) * 256 + d * 32));
        }
    }
    out[(size_t)blockIdx.x * 1024 + tid] = acc.x + acc.y + acc.z + acc.w;
}
int main() {
    float *w, *out;
    hipMalloc(&w, 512 * 256 * 4); hipMalloc(&out, 256 * 1024 * 4);
    hipMemset(w, 0, 512 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 500;
    for (int rep = 0; rep < 2; ++rep)
        for (int pat = 0; pat < 3; ++pat) {
            float ms;
            hipEventRecord(e0);
            for (int i = 0; i < N; ++i) {
                if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, w, out);
                else if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, w, out);
                else hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 0, 0, w, out);
            }
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / N;
            printf("pattern %d: %.2f us per launch, %.1f B/clk/CU at 2.4 GHz (512 KB per workgroup)\n", pat, us, 524288.0 / (us * 2400.0));
        }
    return 0;
}
