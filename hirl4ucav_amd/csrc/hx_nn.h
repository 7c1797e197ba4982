// hx_nn.h — device building blocks of the actor/critic kernels (gfx950): parameter-block layout, wave reductions,
// LayerNorm rows, fp32 MFMA tiles.
//
// Network shape is the reference's (train_all.py:190-208: hidden 256 / 512, LayerNorm after each hidden Linear):
//   Actor  (hirl/agents/HIRL.py:105-146)  13 -> 256 -> LN -> act -> 512 -> LN -> act -> 4 -> tanh
//   Q head (hirl/agents/HIRL.py:19-103)   17 -> 256 -> LN -> act -> 512 -> LN -> act -> 1      (Critic = 2 heads)
// One "MLP block" = one such 3-layer net.  Parameters live in ONE flat fp32 buffer in the reference's state_dict
// order, so an actor is 138,756 contiguous floats and a critic 2 x 138,241 = 276,482; gradients and
// Adam moments use the same layout, which makes Adam/Polyak elementwise and the gradient all-reduce one message.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hxnn {

constexpr int H1 = 256;
constexpr int H2 = 512;
constexpr int XP = 20;       // padded input row (13 or 17 used)
constexpr int RT = 16;       // rows per tile (MFMA M)
// LDS pitch of a [16][256] / [16][512] tile: = 8 mod 64 dwords.  A ds_read_b128 is served in four fixed 16-lane groups
// ({0-3,12-15,20-27}, ...) over 64 banks; with lane -> (row = l & 15, 16-B piece = l >> 4) a pitch of 8 mod 16 dwords puts the 16
// lanes of every group on 16 different 16-B slots (a pitch of 4 mod 64 leaves one 2-way conflict per group).
constexpr int LDA1 = H1 + 8;
constexpr int LDA2 = H2 + 8;
constexpr float LN_EPS = 1e-5f;

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// offsets (in floats) inside one MLP block, reference state_dict order:
// full{1,3}.weight, .bias, layernorm{1,3}.weight, .bias, full{2,4}.weight, .bias, layernorm{2,4}.weight, .bias, final.weight, .bias
struct Mlp {
    int in, out;
    int no_ln;  // 1: plain Linear-ReLU stack (SAC nets, rltorch builder): the LayerNorm slots hold (1, 0) and statistics are (0, 1)
    __host__ __device__ int W1() const { return 0; }
    __host__ __device__ int b1() const { return H1 * in; }
    __host__ __device__ int g1() const { return b1() + H1; }
    __host__ __device__ int be1() const { return g1() + H1; }
    __host__ __device__ int W2() const { return be1() + H1; }
    __host__ __device__ int b2() const { return W2() + H2 * H1; }
    __host__ __device__ int g2() const { return b2() + H2; }
    __host__ __device__ int be2() const { return g2() + H2; }
    __host__ __device__ int W3() const { return be2() + H2; }
    __host__ __device__ int b3() const { return W3() + out * H2; }
    __host__ __device__ int size() const { return b3() + out; }
    // blocks are laid out at multiples of 4 floats so that every row inside a block stays 16-byte aligned
    __host__ __device__ int padded() const { return (size() + 3) & ~3; }
};

// RELU = the activation is known to be a plain ReLU at compile time (slope == 0: HIRL, SAC): one v_max instead of compare + multiply +
// select, and the backward factor is a select instead of a select and a multiply.  The update kernels issue an instruction nearly every
// cycle of their life (16 waves per CU), so their run time follows the instruction count; the launchers pick the instantiation from the slope.
template <bool RELU>
__device__ __forceinline__ float act_f(float y, float slope) { return RELU ? fmaxf(y, 0.0f) : (y > 0.0f ? y : y * slope); }
// v * act'(y)
template <bool RELU>
__device__ __forceinline__ float act_bwd(float v, float y, float slope) { return RELU ? (y > 0.0f ? v : 0.0f) : v * (y > 0.0f ? 1.0f : slope); }

// tanh on the hardware exponential: (1 - e) / (1 + e), e = exp(-2 |x|), sign restored — 7 instructions against the library's ~40, executed
// by whole waves for four numbers per row in the heads' prologues.  Absolute error below 1e-7 over the whole range (the cancellation in
// 1 - e near 0 costs relative, not absolute, accuracy: |x| = 1e-3 -> 3e-8); every kernel that needs the policy's tanh uses THIS one, so
// the acting kernels, the forward heads and the backward pass see the same action bit for bit.
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = __expf(-2.0f * fabsf(x));
    const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    return copysignf(t, x);
}

// ---- cross-lane sums on DPP (VALU latency) instead of ds_bpermute (LDS latency) ---------------------------------
// The update kernels run ONE wave per SIMD (B = 128 fills < 256 CUs), so every dependent reduction is exposed; a
// 6-step __shfl_xor tree costs ~6 LDS round trips, the DPP form 4 VALU ops + 4 v_readlane.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    // mov_dpp (no `old` operand: every lane has a source under these permutations) folds into ONE v_add_f32_dpp; where the compiler packs two
    // reductions into v_pk_add_f32 it stays a bare v_mov_b32_dpp (update_dpp(0, ...) cost an extra v_mov 0 per value and step there)
    return v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
// every lane of each aligned 16-lane row ends up with that row's sum
__device__ __forceinline__ float sum16(float v) {
    v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x124>(v);  // row_ror:4
    v = dpp_add<0x128>(v);  // row_ror:8
    return v;
}
// full 64-lane sum, result uniform across the wave
__device__ __forceinline__ float wave_sum(float v) {
    v = sum16(v);
    const int iv = __float_as_int(v);  // the builtin is typed int: pass BITS, not a value conversion
    const float a = __int_as_float(__builtin_amdgcn_readlane(iv, 0)), b = __int_as_float(__builtin_amdgcn_readlane(iv, 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(iv, 32)), d = __int_as_float(__builtin_amdgcn_readlane(iv, 48));
    return (a + b) + (c + d);
}

__device__ __forceinline__ v4f mfma16(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ---- fp32 MFMA tiles ----------------------------------------------------------------------------------------
// Lane l of a wave: r = l & 15, g = l >> 4.  v_mfma_f32_16x16x4_f32 takes A[i = r][k = g], B[k = g][j = r] and
// returns D[row = 4 g + reg][col = r].  A k-step of 16 is 4 MFMAs; MFMA j consumes k = kk + 4 g + j, the same
// k on the A and B side, so each lane's A (and where possible B) fragment is ONE 16-byte load along k.

// D[16 x 16] += A[16 x K] * Bt^T where A rows are in LDS (pitch lda) and Bt is [n][K] row-major in global memory
// (torch Linear weight): Bt row = output column.
template <int K>
__device__ __forceinline__ v4f tile_a_lds_bt_global(const float* __restrict__ a_lds, int lda, const float* __restrict__ bt_row, v4f acc) {
    const int l = threadIdx.x & 63, r = l & 15, g = l >> 4;
    const float* ap = a_lds + r * lda + 4 * g;
    const float* bp = bt_row + 4 * g;  // bt_row already points at Bt[n0 + r][0]
#pragma unroll 4
    for (int kk = 0; kk < K; kk += 16) {
        const float4 a4 = *reinterpret_cast<const float4*>(ap + kk);
        const float4 b4 = *reinterpret_cast<const float4*>(bp + kk);
        acc = mfma16(a4.x, b4.x, acc);
        acc = mfma16(a4.y, b4.y, acc);
        acc = mfma16(a4.z, b4.z, acc);
        acc = mfma16(a4.w, b4.w, acc);
    }
    return acc;
}

// ---- register-resident B fragments: issued at kernel entry so that their L2/HBM latency overlaps the prologue ------
template <int K>
struct BtFrag {  // for tile_a_lds_bt_global's operand: K/16 float4 per lane
    float4 v[K / 16];
    __device__ __forceinline__ void load(const float* __restrict__ bt_row) {
        const int g = (threadIdx.x & 63) >> 4;
#pragma unroll
        for (int i = 0; i < K / 16; ++i) v[i] = *reinterpret_cast<const float4*>(bt_row + 4 * g + 16 * i);
    }
};
template <int K>
__device__ __forceinline__ v4f tile_a_lds_bt_frag(const float* __restrict__ a_lds, int lda, const BtFrag<K>& f, v4f acc) {
    const int l = threadIdx.x & 63, r = l & 15, g = l >> 4;
    const float* ap = a_lds + r * lda + 4 * g;
#pragma unroll
    for (int i = 0; i < K / 16; ++i) {
        const float4 a4 = *reinterpret_cast<const float4*>(ap + 16 * i);
        acc = mfma16(a4.x, f.v[i].x, acc);
        acc = mfma16(a4.y, f.v[i].y, acc);
        acc = mfma16(a4.z, f.v[i].z, acc);
        acc = mfma16(a4.w, f.v[i].w, acc);
    }
    return acc;
}
template <int K>
struct BFrag {  // B is [K][ldb] row-major in global memory, b_col points at B[0][n0 + r]: one dword per MFMA
    float v[K / 4];
    __device__ __forceinline__ void load(const float* __restrict__ b_col, int ldb) {
        const int g = (threadIdx.x & 63) >> 4;
        const float* bp = b_col + (size_t)(4 * g) * ldb;
#pragma unroll
        for (int i = 0; i < K / 16; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 * i + j] = bp[(size_t)(16 * i + j) * ldb];
    }
};
template <int K>
__device__ __forceinline__ v4f tile_a_lds_b_frag(const float* __restrict__ a_lds, int lda, const BFrag<K>& f, v4f acc) {
    const int l = threadIdx.x & 63, r = l & 15, g = l >> 4;
    const float* ap = a_lds + r * lda + 4 * g;
#pragma unroll
    for (int i = 0; i < K / 16; ++i) {
        const float4 a4 = *reinterpret_cast<const float4*>(ap + 16 * i);
        acc = mfma16(a4.x, f.v[4 * i], acc);
        acc = mfma16(a4.y, f.v[4 * i + 1], acc);
        acc = mfma16(a4.z, f.v[4 * i + 2], acc);
        acc = mfma16(a4.w, f.v[4 * i + 3], acc);
    }
    return acc;
}

// two-pass LayerNorm statistics of one row spread over a wave: each lane holds PER values
template <int PER>
__device__ __forceinline__ void row_stats(const float (&v)[PER], int n, float& mean, float& rstd) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < PER; ++i) s += v[i];
    mean = wave_sum(s) / (float)n;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const float d = v[i] - mean;
        q += d * d;
    }
    // v_rsq_f32 (1 ulp) instead of an IEEE square root and an IEEE division (~25 instructions on every wave's critical path, three times
    // per bwd_l2 prologue): 6e-8 relative on the normalised activations, against a parity tolerance of 1e-5
    rstd = __builtin_amdgcn_rsqf(wave_sum(q) / (float)n + LN_EPS);
}

}  // namespace hxnn
