// boundary_probe.hip — what does the boundary between two dependent launches on one stream cost on MI355X, and which part of it?
// K1 (256 x 1024 threads) dirties `mb` MB and logs every workgroup's exit time; K2 logs (t0) the time of its first instruction, (t1) the
// time its first kernel argument has arrived in an SGPR, both with s_memrealtime (100 MHz, one clock for the device).
//   drain + dispatch = min t0 (K2) - max exit (K1)        kernarg = t1 - t0 (median)       K2 variants: LDS 0 / 64 KB / 160 KB, 64 / 1024 threads
// build: hipcc --offload-arch=gfx950 -O3 boundary_probe.hip -o boundary_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(1024) void k1(float* buf, long long n_per_wg, unsigned long long* ts) {
    float* p = buf + (size_t)blockIdx.x * n_per_wg;
    if (n_per_wg == 0 && threadIdx.x < 64) buf[(size_t)blockIdx.x * 64 + threadIdx.x] = 1.0f;  // (the line K2's modes 0 / 1 read)
    for (long long i = threadIdx.x; i < n_per_wg; i += blockDim.x) p[i] = (float)i;
    __syncthreads();
    if (threadIdx.x == 0) ts[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}
template <int LDSF>
__global__ __launch_bounds__(1024) void k2(unsigned long long* ts, const float* buf, int probe) {
    __shared__ float lds[LDSF > 0 ? LDSF : 1];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_nop 0" ::"s"(probe));  // the first use of a kernel argument
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    // the first global operand.  probe >> 16 selects where it comes from: 0 = a line K1's workgroup of the SAME index (same XCD) wrote,
    // 1 = a line K1's NEXT workgroup (another XCD) wrote, 2 = a read-only line this workgroup read in the previous launch of K2,
    // 3 = a read-only line nobody has touched yet (moves on every launch)
    const int mode = probe >> 16, rep = probe & 0xFFFF;
    const size_t ro = (size_t)(48 << 20) / 4;  // read-only region: K1 never writes past 32 MB
    const float* src = mode == 0 ? buf + (size_t)blockIdx.x * 64 : mode == 1 ? buf + (size_t)((blockIdx.x + 1) & 255) * 64
                     : mode == 2 ? buf + ro + (size_t)blockIdx.x * 64 : buf + ro + (size_t)(1 << 20) + ((size_t)rep * 256 + blockIdx.x) * 64;
    const float v = src[threadIdx.x & 63];
    asm volatile("" ::"v"(v));
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    if (LDSF > 0) lds[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        ts[256 + blockIdx.x * 3] = t0;
        ts[256 + blockIdx.x * 3 + 1] = t1;
        ts[256 + blockIdx.x * 3 + 2] = t2 + (LDSF > 0 ? (unsigned long long)(lds[0] == 12345.0f) : 0ull);
    }
}

// K3: what a real prologue does — every wave of the workgroup requests NL dwords per lane from NL different arrays (3 MB + 4 KB apart),
// lines that (fresh = 1) K1's workgroup of ANOTHER XCD has just written, or (fresh = 0) that this workgroup read one launch ago.
template <int NL>
__global__ __launch_bounds__(1024) void k3(unsigned long long* ts, const float* buf, int fresh) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const size_t arr = ((size_t)(3 << 20) + 4096) / 4;
    const float* base = buf + (size_t)((blockIdx.x + (fresh ? 0 : 0)) & 255) * 1024 + threadIdx.x;
    float v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) v[i] = base[(size_t)i * arr];
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < NL; ++i) acc += v[i];
    asm volatile("" ::"v"(acc));
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { ts[256 + blockIdx.x * 3] = t0; ts[256 + blockIdx.x * 3 + 1] = t1; }
    if (acc == 12345.678f) ts[0] = 0;
}
__global__ __launch_bounds__(1024) void k1w(float* buf, unsigned long long* ts, int nl) {  // writes the lines K3's workgroup (index + 1: another XCD) will read
    const size_t arr = ((size_t)(3 << 20) + 4096) / 4;
    float* base = buf + (size_t)((blockIdx.x + 1) & 255) * 1024 + threadIdx.x;
    for (int i = 0; i < nl; ++i) base[(size_t)i * arr] = (float)i;
    __syncthreads();
    if (threadIdx.x == 0) ts[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}
int main() {
    float* buf; unsigned long long* ts;
    hipMalloc((void**)&buf, 128ll << 20); hipMalloc((void**)&ts, (256 + 256 * 3) * 8);
    std::vector<unsigned long long> h(256 + 256 * 3);
    const char* names[] = {"1024 thr, LDS 0", "1024 thr, LDS 64 KB", "1024 thr, LDS 156 KB", "64 thr, LDS 0"};
    for (int mode = 1; mode < 4; ++mode) {
        double ld = 0;
        const int reps = 30;
        for (int rep = 0; rep < reps + 2; ++rep) {
            hipLaunchKernelGGL(k1, dim3(256), dim3(1024), 0, 0, buf, 0ll, ts);
            hipLaunchKernelGGL(k2<0>, dim3(256), dim3(1024), 0, 0, ts, buf, (mode << 16) | rep);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> l(256);
            for (int b = 0; b < 256; ++b) l[b] = (double)(h[256 + b * 3 + 2] - h[256 + b * 3 + 1]) * 0.01;
            std::sort(l.begin(), l.end());
            if (rep >= 2) ld += l[128];
        }
        printf("first global load of K2, source %s: %.2f us\n", mode == 1 ? "written by ANOTHER XCD's workgroup in K1" : mode == 2 ? "read-only, read by this workgroup one launch ago" : "read-only, untouched (HBM)", ld / reps);
    }

    for (int fresh = 0; fresh < 2; ++fresh)
        for (int nl : {1, 4, 16}) {
            double lat = 0;
            const int reps = 30;
            for (int rep = 0; rep < reps + 2; ++rep) {
                if (fresh) hipLaunchKernelGGL(k1w, dim3(256), dim3(1024), 0, 0, buf, ts, nl);
                else hipLaunchKernelGGL(k1, dim3(256), dim3(1024), 0, 0, buf + (100 << 20) / 4, 0ll, ts);
                if (nl == 1) hipLaunchKernelGGL(k3<1>, dim3(256), dim3(1024), 0, 0, ts, buf, fresh);
                if (nl == 4) hipLaunchKernelGGL(k3<4>, dim3(256), dim3(1024), 0, 0, ts, buf, fresh);
                if (nl == 16) hipLaunchKernelGGL(k3<16>, dim3(256), dim3(1024), 0, 0, ts, buf, fresh);
                hipDeviceSynchronize();
                hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost);
                std::vector<double> l(256);
                for (int b = 0; b < 256; ++b) l[b] = (double)(h[256 + b * 3 + 1] - h[256 + b * 3]) * 0.01;
                std::sort(l.begin(), l.end());
                if (rep >= 2) lat += l[128];
            }
            printf("prologue-like batch: 16 waves x %2d dword loads per lane from %2d arrays, lines %s: all data back after %.2f us (median workgroup)\n", nl, nl,
                   fresh ? "just written by another XCD's workgroup" : "read by this workgroup one launch ago", lat / reps);
        }
    for (int mb : {0, 2, 8, 32})
        for (int var = 0; var < 4; ++var) {
            double g = 0, ka = 0, ld = 0, span1 = 0;
            const int reps = 30;
            for (int rep = 0; rep < reps; ++rep) {
                const long long n = (long long)mb * (1 << 20) / 4 / 256;
                hipLaunchKernelGGL(k1, dim3(256), dim3(1024), 0, 0, buf, n, ts);
                if (var == 0) hipLaunchKernelGGL(k2<0>, dim3(256), dim3(1024), 0, 0, ts, buf, rep);
                if (var == 1) hipLaunchKernelGGL(k2<16384>, dim3(256), dim3(1024), 0, 0, ts, buf, rep);
                if (var == 2) hipLaunchKernelGGL(k2<39936>, dim3(256), dim3(1024), 0, 0, ts, buf, rep);
                if (var == 3) hipLaunchKernelGGL(k2<0>, dim3(256), dim3(64), 0, 0, ts, buf, rep);
                hipDeviceSynchronize();
                hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost);
                unsigned long long last = 0, first = ~0ull, lastst = 0;
                std::vector<double> a(256), l(256);
                for (int b = 0; b < 256; ++b) {
                    last = std::max(last, h[b]);
                    first = std::min(first, h[256 + b * 3]);
                    lastst = std::max(lastst, h[256 + b * 3]);
                    a[b] = (double)(h[256 + b * 3 + 1] - h[256 + b * 3]) * 0.01;
                    l[b] = (double)(h[256 + b * 3 + 2] - h[256 + b * 3 + 1]) * 0.01;
                }
                std::sort(a.begin(), a.end()); std::sort(l.begin(), l.end());
                g += ((double)first - (double)last) * 0.01; ka += a[128]; ld += l[128]; span1 += (double)(lastst - first) * 0.01;
            }
            printf("K1 dirties %2d MB | K2 %-22s: last exit -> first instruction %.2f us | first -> last workgroup start %.2f us | kernarg %.2f us | first global load %.2f us\n",
                   mb, names[var], g / reps, span1 / reps, ka / reps, ld / reps);
        }
    return 0;
}
