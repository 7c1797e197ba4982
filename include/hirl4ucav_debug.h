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
/* diagnostic builds only (make -C hirl4ucav_amd/csrc stamps); all return -1 in the shipped build */
int hx_debug_stamps(float* host_out /* host, 80 floats: in-kernel phase stamps, 10 ns ticks */);
int hx_debug_stamps_actp(float* host_out /* host, 80 floats: the persistent acting kernel's stamps (csrc/hx_actp.hip) */);
int hx_debug_spans(unsigned long long* host_spans /* host [8192][2] */, unsigned* host_tags /* host [8192]: 1 fwd_l2, 2 act_fused, 3 bwd_l2, 4 wgrad */,
                   unsigned* host_n /* host */); /* life span of every workgroup since the last call */

#ifdef __cplusplus
}
#endif
#endif /* HIRL4UCAV_DEBUG_H */
