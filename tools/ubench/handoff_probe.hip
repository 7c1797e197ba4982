// handoff_probe.hip — what does an in-launch hand-off between the 4 workgroups of a group cost on MI355X?
// 256 workgroups x 1024 threads (one per CU); group = 4 workgroups (a) consecutive block ids (4 different XCDs) or (b) block ids 8 apart
// (same XCD under round-robin placement).  Each workgroup: write 64 x 8 floats of partials, fence, bump the group's counter, spin
// (bounded) until the 4 peers arrived, fence, read the 4 partial sets; twice per launch (two hand-offs), counters monotonic over launches.
// build: hipcc --offload-arch=gfx950 -O3 handoff_probe.hip -o handoff_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(1024) void k(float* part, unsigned* cnt, unsigned* status, float* out, unsigned gen, int same_xcd, int handoffs, int variant) {
    const int b = blockIdx.x, tid = threadIdx.x;
    int grp, sl;
    if (same_xcd) { const int x = b & 7, y = b >> 3; grp = x + 8 * (y >> 2); sl = y & 3; }   // b = x + 8 (4 q + s)
    else { grp = b >> 2; sl = b & 3; }
    float acc = (float)tid;
    for (int h = 0; h < handoffs; ++h) {
        float* mine = part + ((size_t)(grp * 2 + h) * 4 + sl) * 512;
        const float* g0 = part + ((size_t)(grp * 2 + h) * 4) * 512;
        const unsigned target = 4u * (gen + 1u);
        if (variant == 0) {            // fences by every thread
            if (tid < 512) mine[tid] = acc + (float)sl;
            __threadfence();
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(&cnt[grp * 2 + h], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(&cnt[grp * 2 + h], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (++spins > 2000000) { atomicOr(status, 1u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            __threadfence();
            if (tid < 512) acc = (g0[tid] + g0[512 + tid]) + (g0[1024 + tid] + g0[1536 + tid]);
        } else if (variant == 1) {     // release by thread 0 only, acquire fence by everyone
            if (tid < 512) mine[tid] = acc + (float)sl;
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(&cnt[grp * 2 + h], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(&cnt[grp * 2 + h], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (++spins > 2000000) { atomicOr(status, 1u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (tid < 512) acc = (g0[tid] + g0[512 + tid]) + (g0[1024 + tid] + g0[1536 + tid]);
        } else if (variant == 3) {     // release fence by ONE thread per workgroup; data read with agent-scope relaxed atomic loads
            if (tid < 512) mine[tid] = acc + (float)sl;
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(&cnt[grp * 2 + h], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(&cnt[grp * 2 + h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (++spins > 2000000) { atomicOr(status, 1u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (tid < 512) {
                const float a0 = __hip_atomic_load(&g0[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a1 = __hip_atomic_load(&g0[512 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float a2 = __hip_atomic_load(&g0[1024 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a3 = __hip_atomic_load(&g0[1536 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc = (a0 + a1) + (a2 + a3);
            }
        } else if (variant == 4) {     // data stored with agent-scope relaxed atomics; ACQUIRE fence by every thread before plain loads
            if (tid < 512) __hip_atomic_store(&mine[tid], acc + (float)sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(&cnt[grp * 2 + h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(&cnt[grp * 2 + h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (++spins > 2000000) { atomicOr(status, 1u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (tid < 512) acc = (g0[tid] + g0[512 + tid]) + (g0[1024 + tid] + g0[1536 + tid]);
        } else {                       // no fences: agent-scope relaxed atomic stores / loads for the data, flag after the stores completed
            if (tid < 512) __hip_atomic_store(&mine[tid], acc + (float)sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(&cnt[grp * 2 + h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(&cnt[grp * 2 + h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (++spins > 2000000) { atomicOr(status, 1u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (tid < 512) {
                const float a0 = __hip_atomic_load(&g0[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a1 = __hip_atomic_load(&g0[512 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float a2 = __hip_atomic_load(&g0[1024 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a3 = __hip_atomic_load(&g0[1536 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc = (a0 + a1) + (a2 + a3);
            }
        }
    }
    if (tid < 512) out[(size_t)b * 512 + tid] = acc;
}
int main() {
    float *part, *out; unsigned *cnt, *status;
    hipMalloc(&part, 64 * 2 * 4 * 512 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cnt, 64 * 2 * 4 * 4); hipMalloc(&status, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 2000;
    for (int variant = 0; variant < 5; ++variant)
    for (int handoffs = (variant ? 1 : 0); handoffs <= 2; ++handoffs)
        for (int same = 0; same < 2; ++same) {
            hipMemset(cnt, 0, 64 * 2 * 4 * 4); hipMemset(status, 0, 4);
            unsigned gen = 0;
            for (int k2 = 0; k2 < 50; ++k2) hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, part, cnt, status, out, gen++, same, handoffs, variant);
            hipDeviceSynchronize();
            float ms;
            hipEventRecord(e0);
            for (int k2 = 0; k2 < N; ++k2) hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, part, cnt, status, out, gen++, same, handoffs, variant);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            unsigned st = 0; hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost);
            std::vector<float> h(512); hipMemcpy(h.data(), out + 5 * 512, 2048, hipMemcpyDeviceToHost);
            printf("variant %d  handoffs %d  %s  %.3f us per launch  status %u  out[5][3] = %.1f\n", variant, handoffs, same ? "same XCD (ids 8 apart)" : "4 XCDs (consecutive ids)", ms * 1e3 / N, st, h[3]);
        }
    return 0;
}
