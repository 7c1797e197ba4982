/*
 * hirl4ucav_debug.h — measurement-only entry points of libhx_mi355.so (tools/, never the product path).  No reference counterpart.
 */
#ifndef HIRL4UCAV_DEBUG_H
#define HIRL4UCAV_DEBUG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* PMC calibration helper: dst[i] = src[i], one dword per lane (the env kernel's access shape), n floats */
int hx_debug_copy_dword(const float* src, float* dst, int64_t n, void* stream);
/* tests: of the update's forward launches that follow this call, numbers [skip, skip + count) keep 64-column workgroups whatever their job count — the
 * tiling hx_hirl_front gives launch B, whose K-split (hence the last bits of the target critics' z2) differs from the 32-column workgroups a two-net
 * launch of its own takes.  nt = 0: back to the default */
int hx_debug_set_fwd_nt(int32_t nt, int32_t skip, int32_t count);
/* diagnostic builds only (make -C hirl4ucav_amd/csrc stamps); all return -1 in the shipped build */
int hx_debug_stamps(float* host_out /* host, 80 floats: in-kernel phase stamps, 10 ns ticks */);
int hx_debug_stamps_actp(float* host_out /* host, 80 floats: the persistent acting kernel's stamps (csrc/hx_actp.hip) */);
int hx_debug_stamps_front(float* host_out /* host, 80 floats: the front launch's stamps (csrc/hx_front.hip): [0..7] launch B, [8..15] launch A, [56..] acting */);
int hx_debug_spans(unsigned long long* host_spans /* host [8192][2] */, unsigned* host_tags /* host [8192]: 1 fwd_l2, 2 act_fused, 3 bwd_l2, 4 wgrad, 5 / 6 launch A / B inside the front launch */,
                   unsigned* host_n /* host */); /* life span of every workgroup since the last call */

#ifdef __cplusplus
}
#endif
#endif /* HIRL4UCAV_DEBUG_H */
