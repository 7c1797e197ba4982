// permlane_probe.hip — what hx_act.h's sum_rows4 relies on (gfx950): v_permlane16_swap_b32 exchanges the ODD 16-lane rows of its first operand with
// the EVEN rows of its second, v_permlane32_swap_b32 the upper half of the first with the lower half of the second.  With both operands = v, one swap
// + one add leaves (row 0 + row 1) in rows 0, 1 and (row 2 + row 3) in rows 2, 3; the second swap + add the wave's four-row total in every lane —
// the same bits in all four lanes of a column.   hipcc --offload-arch=gfx950 -O2 permlane_probe.hip -o permlane_probe.bin && ./permlane_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__device__ __forceinline__ float sum_rows4(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__global__ void k(const float* in, float* out, unsigned* raw) {
    const float v = in[threadIdx.x];
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    raw[threadIdx.x] = a[0];
    raw[64 + threadIdx.x] = a[1];
    out[threadIdx.x] = sum_rows4(v);
}
int main() {
    float h[64], o[64], *di, *dout;
    unsigned raw[128], *dr;
    for (int i = 0; i < 64; ++i) h[i] = 1.0f + 0.37f * i + 1e-3f * (float)((i * 7919) % 13);
    (void)hipMalloc(&di, 256); (void)hipMalloc(&dout, 256); (void)hipMalloc(&dr, 512);
    (void)hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout, dr);
    (void)hipMemcpy(o, dout, 256, hipMemcpyDeviceToHost);
    (void)hipMemcpy(raw, dr, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int c = 0; c < 16; ++c) {
        const float want = (h[c] + h[16 + c]) + (h[32 + c] + h[48 + c]);
        for (int r = 0; r < 4; ++r)
            if (memcmp(&o[16 * r + c], &want, 4)) { ++bad; printf("lane %d: %.9g, want %.9g\n", 16 * r + c, o[16 * r + c], want); }
    }
    for (int l = 0; l < 64; ++l) {  // first result: rows (0, 0, 2, 2); second: rows (1, 1, 3, 3)
        const int row = l >> 4, c = l & 15;
        float f0, f1;
        memcpy(&f0, &raw[l], 4); memcpy(&f1, &raw[64 + l], 4);
        if (f0 != h[16 * (row & ~1) + c] || f1 != h[16 * (row | 1) + c]) { ++bad; printf("swap16 lane %d: (%g, %g)\n", l, f0, f1); }
    }
    if (bad) printf("permlane_probe: FAILED (%d)\n", bad);
    else printf("permlane_probe: ok\n");
    return bad != 0;
}
