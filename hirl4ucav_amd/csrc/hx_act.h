// hx_act.h — what the two acting kernels share (gfx950): the launch description and the exploration-noise draw.
//   hx_act.hip    act_fused_kernel: one row tile (16 / 32 rows) per workgroup — up to 8,192 rows (one round of workgroups)
//   hx_actp.hip   act_persist_kernel: persistent workgroups that keep W2 and loop over their row tiles — beyond 8,192 rows
// Agent.chooseAction* hirl/agents/HIRL.py:192-212; SacAgent.explore / exploit hirl/agents/SAC/agent.py:183-196.
#pragma once
#include "hx_update.h"

// The "x9" acting format: both operands of the fp32 256 -> 512 product as EXACT three-way bf16 splits (x = hi + mid + lo, 3 x 8 significand bits = 24: nothing
// is lost), their partial products — each exact in fp32 — accumulated in fp32 on the bf16 matrix cores.  [r5] SIX of the nine partial products are formed:
// lo x lo, lo x mid and mid x lo are at most 2^-24 of the product they belong to — below the fp32 resolution of the sum they would join — and cost a third of
// the matrix-core work.  Measured (tools/ubench/x9_terms_ab.sh, profiles/r05_x9_terms_ab.txt): error against an fp64 evaluation over 16,384 rows 2.68e-7 max /
// 4.59e-8 mean with six terms, 2.72e-7 / 4.60e-8 with nine, 3.78e-7 / 7.30e-8 for the fp32-MFMA kernel; 65,536 circular envs 384 -> 439 M env steps/s, SAC at
// 16,384 envs 165 -> 174 M, the headline's front launch 24.4 -> 23.0 us.  -DHX_X9_TERMS=9 builds the nine-term form (the A/B's other side).
#ifndef HX_X9_TERMS
#define HX_X9_TERMS 6
#endif
namespace hxact {
using namespace hxnn;
using namespace hxu;

struct ActFusedArgs {
    const float* net;
    Mlp m;
    float* obs;        // [rows][13]; written only by the ENV instantiations (next observation)
    int rows;
    float slope;
    float* actions;      // [rows][4]
    const float* noise;  // deterministic head: nullptr, [4] (shared) or [rows][4] additive noise; Gaussian head: eps [rows][4] or nullptr
    int noise_per_row;
    float sigma;
    int mode;            // Gaussian head: 0 exploit tanh(mean), 1 sample with eps, 2 sample with Philox
    uint64_t seed;
    uint32_t row0, call;
    // ENV instantiations: HarfangEnv.step for the same rows in the tail of this launch (obs is then in/out)
    float* state;
    int64_t stride;
    float* reward;
    uint8_t* done;
    int8_t* success;
    HxStepOpts o;
    double inv_cap;  // 1 / o.cap
    const uint16_t* w2b;  // BF16 instantiations: bf16 image of W2 [512][256] (hx_pack_w2_bf16 / the actor's Adam step keep it current)
    const float* w2f;     // F32I instantiations: fp32 image of W2 (hx_pack_w2_f32i)
    int x9;               // X3 instantiations: w2b is the first of THREE images hi | mid | lo (hx_pack_w2_x9): the exact bf16 split of W2
};

// one standard-normal draw per (row, component j) of the acting kernels: Philox4x32-10(seed; row, call, tag) + Box-Muller
__device__ __forceinline__ float philox_normal(uint32_t row, uint32_t call, uint32_t tag, uint64_t seed, int j) {
    uint32_t u[4];
    philox4x32_10(row, call, tag, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), u);
    const float ua = u01(u[j & 2]), ub = u01(u[(j & 2) + 1]);
    const float rad = sqrtf(-2.0f * logf(ua)), ang = 6.28318530717958647692f * ub;
    return (j & 1) ? rad * sinf(ang) : rad * cosf(ang);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Row arithmetic of the acting kernels: SIXTEEN LANES PER ROW, four rows per wave.
// Lane (q = lane >> 4, c = lane & 15) of a wave works on row q of the wave's four rows and owns that row's columns 64 i + 4 c .. + 3
// (i < N / 64): one 16-byte LDS read per i, and a row sum is the lane's sequential sum followed by FOUR DPP adds inside the 16-lane row
// (sum16u) — every lane of the row then holds it, bit for bit.  With a whole wave per row (round 1-3) each of the six sums of a head row cost 4 DPP adds
// + 4 v_readlane + 3 adds + the hazard nops: 280 of the ~850 instructions two head rows took, on a phase that is VALU-issue bound
// (131,072 rows x ~350 instructions = 75 us of the whole chip's vector issue).  Here the reductions of FOUR rows are 4 instructions.
// Both acting kernels use these functions, with contraction off and every fused multiply-add spelled out, so a row's action does not depend
// on the kernel, the row tiling or the wave that computed it (tests/test_actp_gpu.py, test_act_row_tilings_agree_bit_for_bit).
// ---------------------------------------------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
// Sum over the 16 lanes of a row that leaves the SAME BITS in every lane: two quad permutes, then half-mirror and mirror — a butterfly whose
// partners add the same two numbers (a + b and b + a).  hx_nn.h's sum16 ends with two rotations instead: its quads 0 / 2 and 1 / 3 hold
// (B0 + B1) + (B2 + B3) and (B1 + B2) + (B3 + B0) — one rounding apart now and then, which is fine for wave_sum (it reads quad 0) but not
// for lanes that each go on with their own copy.  The value equals sum16's quad-0 value.
__device__ __forceinline__ float sum16u(float v) {
    v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);  // row_half_mirror
    v = dpp_add<0x140>(v);  // row_mirror
    return v;
}
// The element arithmetic runs on PACKED fp32 (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two IEEE operations per lane and instruction —
// the phase is bound by vector issue): every running sum is kept as an (even, odd) pair — elements 0, 1 of a 16-byte piece open / extend
// the pair, elements 2, 3 extend it — and the two halves meet in one add before the cross-lane sum.
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
// two-pass LayerNorm statistics of this lane's row from its PER elements (nn.LayerNorm: biased variance, eps inside the root)
template <int PER>
__device__ __forceinline__ void row_stats16(const float (&v)[PER], int n, float& mean, float& rstd) {
    v2f s2 = v2f{v[0], v[1]} + v2f{v[2], v[3]};
#pragma unroll
    for (int i = 4; i < PER; i += 4) {
        s2 = s2 + v2f{v[i], v[i + 1]};
        s2 = s2 + v2f{v[i + 2], v[i + 3]};
    }
    mean = sum16u(s2.x + s2.y) / (float)n;
    const v2f m2 = {mean, mean};
    v2f q2 = {0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < PER; i += 2) {
        const v2f d = v2f{v[i], v[i + 1]} - m2;
        q2 = fma2(d, d, q2);
    }
    rstd = __builtin_amdgcn_rsqf(sum16u(q2.x + q2.y) / (float)n + LN_EPS);  // v_rsq_f32, as row_stats
}
// [r5] The same statistics with THIRTY-TWO lanes per row, two rows per wave (the bf16 acting kernels' LayerNorm 1: every one of the 16 waves takes two of a
// 32-row tile's rows — with 16 lanes per row eight waves carried the whole phase while the other eight waited, profiles/r05_actp_bf16_wave_stamps_v2.txt).
// Lane (h = lane >> 5, c = lane & 31) holds 8 elements of row h: columns 4 c .. + 3 and 128 + 4 c .. + 3.  The butterfly leaves the same bits in all 32 lanes.
__device__ __forceinline__ float sum32u(float v) {
    v = sum16u(v);
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);  // rows (0, 0, 2, 2) | (1, 1, 3, 3): see sum_rows4
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
__device__ __forceinline__ void row_stats32(const v4f& x0, const v4f& x1, int n, float& mean, float& rstd) {
    const v2f s2 = (v2f{x0[0], x0[1]} + v2f{x0[2], x0[3]}) + (v2f{x1[0], x1[1]} + v2f{x1[2], x1[3]});
    mean = sum32u(s2.x + s2.y) / (float)n;
    const v2f m2 = {mean, mean};
    v2f d = v2f{x0[0], x0[1]} - m2;
    v2f q2 = d * d;
    d = v2f{x0[2], x0[3]} - m2;
    q2 = fma2(d, d, q2);
    d = v2f{x1[0], x1[1]} - m2;
    q2 = fma2(d, d, q2);
    d = v2f{x1[2], x1[3]} - m2;
    q2 = fma2(d, d, q2);
    rstd = __builtin_amdgcn_rsqf(sum32u(q2.x + q2.y) / (float)n + LN_EPS);
}
// this lane's 16 (N = 256) or 32 (N = 512) elements of the LDS row `row`
template <int N>
__device__ __forceinline__ void load_row16(const float* row, int c, float (&v)[N / 16]) {
#pragma unroll
    for (int i = 0; i < N / 64; ++i) {
        const float4 t = *reinterpret_cast<const float4*>(row + 64 * i + 4 * c);
        v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
    }
}
// LN1 + activation, elementwise (any layout): h1 = act(g1 (z1 - mean) rstd + be1)   Actor.forward, HIRL.py:128-131
template <bool RELU>
__device__ __forceinline__ float ln_act(float z, float mean, float rstd, float g, float be, float slope) {
    return act_f<RELU>(__builtin_fmaf(g, (z - mean) * rstd, be), slope);
}
template <bool RELU>
__device__ __forceinline__ v2f ln_act2(v2f z, v2f mean, v2f rstd, v2f g, v2f be, float slope) {  // the same, two elements
    const v2f y = fma2(g, (z - mean) * rstd, be);
    return v2f{act_f<RELU>(y.x, slope), act_f<RELU>(y.y, slope)};
}
// Head of this lane's row from the LDS copy of z2 and the LDS head image (HeadImage<IMG>: g2 | be2 | W3 rows | b3): LN2, activation, final
// layer.  Every lane of the row's 16 ends up with all OUT pre-tanh outputs.  The row's 32 elements per lane are read from LDS in each of the
// three passes (statistics twice, projection) instead of living in 32 registers: the reads of a later pass do not depend on the earlier
// pass's result, so they cost LDS bandwidth (which this phase does not lack), not latency.
template <int OUT, int IMG, bool RELU, bool ZREG = false>
__device__ __forceinline__ void head16(const float* zrow, const float* hp, int c, float slope, int no_ln, float (&o)[OUT]) {
    float mean, rstd;
    v2f m2;
    if constexpr (ZREG) {  // both statistics passes from ONE batch of reads (32 registers): the same values, the same operations
        float z[32];
        load_row16<H2>(zrow, c, z);
        row_stats16<32>(z, H2, mean, rstd);
    } else {
        v2f s2 = {0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const v4f t = *reinterpret_cast<const v4f*>(zrow + 64 * i + 4 * c);
            // (the association of row_stats16: the first piece opens the pair)
            s2 = i == 0 ? v2f{t[0], t[1]} + v2f{t[2], t[3]} : (s2 + v2f{t[0], t[1]}) + v2f{t[2], t[3]};
        }
        mean = sum16u(s2.x + s2.y) / (float)H2;
        m2 = v2f{mean, mean};
        v2f q2 = {0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const v4f t = *reinterpret_cast<const v4f*>(zrow + 64 * i + 4 * c);
            v2f d = v2f{t[0], t[1]} - m2;
            q2 = fma2(d, d, q2);
            d = v2f{t[2], t[3]} - m2;
            q2 = fma2(d, d, q2);
        }
        rstd = __builtin_amdgcn_rsqf(sum16u(q2.x + q2.y) / (float)H2 + LN_EPS);
    }
    if (no_ln) { mean = 0.0f; rstd = 1.0f; }
    m2 = v2f{mean, mean};
    const v2f r2 = {rstd, rstd};
    v2f acc[OUT];
#pragma unroll
    for (int j = 0; j < OUT; ++j) acc[j] = v2f{0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const v4f z = *reinterpret_cast<const v4f*>(zrow + 64 * i + 4 * c);
        const v4f g = *reinterpret_cast<const v4f*>(hp + 64 * i + 4 * c);
        const v4f be = *reinterpret_cast<const v4f*>(hp + H2 + 64 * i + 4 * c);
        const v2f h01 = ln_act2<RELU>(v2f{z[0], z[1]}, m2, r2, v2f{g[0], g[1]}, v2f{be[0], be[1]}, slope);
        const v2f h23 = ln_act2<RELU>(v2f{z[2], z[3]}, m2, r2, v2f{g[2], g[3]}, v2f{be[2], be[3]}, slope);
        v4f w[OUT];
#pragma unroll
        for (int j = 0; j < OUT; ++j) w[j] = *reinterpret_cast<const v4f*>(hp + (2 + j) * H2 + 64 * i + 4 * c);
#pragma unroll
        for (int j = 0; j < OUT; ++j) acc[j] = fma2(h01, v2f{w[j][0], w[j][1]}, acc[j]);  // (OUT independent chains between a sum's two steps)
#pragma unroll
        for (int j = 0; j < OUT; ++j) acc[j] = fma2(h23, v2f{w[j][2], w[j][3]}, acc[j]);
    }
#pragma unroll
    for (int j = 0; j < OUT; ++j) o[j] = sum16u(acc[j].x + acc[j].y) + hp[(2 + IMG) * H2 + j];
}
// ---------------------------------------------------------------------------------------------------------------------------------------
// [r5] LayerNorm 2 + final layer STRAIGHT FROM THE ACCUMULATORS of the 256 -> 512 product (the bf16 acting kernels: act_fused_body<.., BF16>,
// act_persist_bf16_body).  Actor.forward hirl/agents/HIRL.py:132-140.
// Until round 4 the z2 tile went to LDS (64 KB per 32 rows) and a 16-lane group per row read it three times (sum, centred squares, projection) beside
// the whole head image (g2, be2, four W3 rows: 12 KB per ROW) — 650 KB of LDS reads per 32-row tile, an LDS pipe 62 % busy, and a ~2 us dependent chain
// per head phase (tools/pmc_actp_passes.sh, profiles/r05_pmc_sq_actp_per_wave.txt).  Now, with the product's operands swapped (weights as A), lane
// (lr, lg) of the wave that owns column group cw holds EIGHT values of row lr — columns 16 cw + 4 lg .. + 3 and 256 + 16 cw + 4 lg .. + 3 — and:
//   1. the wave's 32 columns of a row give a PARTIAL (mean_w, M2_w = sum of squares about mean_w): 8 values per lane, then lanes l, l ^ 16, l ^ 32,
//      l ^ 48 (v_permlane16_swap / v_permlane32_swap: two instructions per sum); one float2 per (row, column group) to LDS — 4 KB per 32 rows;
//   2. behind ONE barrier the 16 partials of a row combine (Chan et al.'s pairwise update for equal counts: mean = sum mean_w / 16,
//      M2 = sum M2_w + 32 sum (mean_w - mean)^2) IN COLUMN-GROUP ORDER — the statistics do not depend on which wave owned which group;
//   3. h2 = act(LN2(z2)) of the lane's eight values, rounded to bf16, IS the B operand of one more v_mfma_f32_16x16x32_bf16 (K = the wave's 32
//      columns; A = W3's four rows at those columns, bf16, twelve zero rows): its fp32 result, in the lg = 0 lanes, is the wave's share of the
//      four pre-tanh outputs of row lr; 16 shares per row meet in LDS (16 bytes each) and are summed in column-group order + b3.
// Numerics: the statistics agree with the two-pass form to fp32 rounding (M2 is a sum of centred squares at every level — no E[z^2] - mean^2
// cancellation); the final layer's operands are bf16 like the 256 -> 512 product's (products exact in fp32, fp32 accumulation) — tests/test_bf16_gpu.py
// states the rounded-operand reference accordingly.  Every kernel that uses these functions produces the same bits for a row.
// ---------------------------------------------------------------------------------------------------------------------------------------
// sum over lanes l, l ^ 16, l ^ 32, l ^ 48 (the four 16-lane rows of the wave), the same bits in all four: (row 0 + row 1) + (row 2 + row 3)
__device__ __forceinline__ float sum_rows4(float v) {
    // v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second: with both = v, the first comes back holding
    // rows (0, 0, 2, 2) and the second rows (1, 1, 3, 3); v_permlane32_swap likewise for the wave's halves
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// floats per row of the partial-statistics tile: 16 column groups x (mean_w, M2_w) + 4 of padding — with a pitch of 32 the sixteen float2 stores of a wave fell on
// ONE bank pair (16-way) and the 16-byte reads of step 2 on two (4-way); 36 puts the stores on 16 different pairs and the reads on disjoint quads
constexpr int kPartPitch = 36;
// step 1 for one row tile: z0 / z1 = the lane's four values of column tile cw / 16 + cw (bias added).  Lanes lg == 0 store the pair.
// (Packed fp32 throughout — the vector units are what bounds the tile loop, tools/ubench/actp_variants.sh: every running sum is an (even, odd) pair of
// columns, as in row_stats16.)
__device__ __forceinline__ void row_partial32(const v4f& z0, const v4f& z1, int lg, float* prow_cw) {
    const v2f a = (v2f{z0[0], z0[1]} + v2f{z0[2], z0[3]}) + (v2f{z1[0], z1[1]} + v2f{z1[2], z1[3]});
    const float mw = sum_rows4(a.x + a.y) * (1.0f / 32.0f);
    const v2f m2 = {mw, mw};
    v2f d = v2f{z0[0], z0[1]} - m2;
    v2f q2 = d * d;
    d = v2f{z0[2], z0[3]} - m2;
    q2 = fma2(d, d, q2);
    d = v2f{z1[0], z1[1]} - m2;
    q2 = fma2(d, d, q2);
    d = v2f{z1[2], z1[3]} - m2;
    q2 = fma2(d, d, q2);
    const float q = sum_rows4(q2.x + q2.y);
    if (lg == 0) *reinterpret_cast<v2f*>(prow_cw) = v2f{mw, q};
}
// step 2: lane (lr, lg) reads the partials of column groups 4 lg .. 4 lg + 3 of its row (32 bytes); the four lanes of the row end with the same bits
__device__ __forceinline__ void row_combine16(const float* prow, int lg, int no_ln, float& mean, float& rstd) {
    const v4f p0 = *reinterpret_cast<const v4f*>(prow + 8 * lg), p1 = *reinterpret_cast<const v4f*>(prow + 8 * lg + 4);  // (m, M2, m, M2) x 2
    const v2f t = (v2f{p0[0], p0[1]} + v2f{p0[2], p0[3]}) + (v2f{p1[0], p1[1]} + v2f{p1[2], p1[3]});  // (sum of the four means, sum of the four M2)
    mean = sum_rows4(t.x) * (1.0f / 16.0f);
    const v2f mm = {mean, mean};
    const v2f e01 = v2f{p0[0], p0[2]} - mm, e23 = v2f{p1[0], p1[2]} - mm;
    const v2f de = fma2(e23, e23, e01 * e01);
    const float m2 = sum_rows4(__builtin_fmaf(32.0f, de.x + de.y, t.y));
    rstd = __builtin_amdgcn_rsqf(m2 * (1.0f / (float)H2) + LN_EPS);  // nn.LayerNorm: biased variance, eps inside the root (v_rsq_f32, as row_stats16)
    if (no_ln) { mean = 0.0f; rstd = 1.0f; }
}
// step 3, the operand: h2 of the lane's eight values -> bf16 (round to nearest even) in the order of the final MFMA's k = 8 lg + e:
// e < 4 column 16 cw + 4 lg + e, e >= 4 column 256 + 16 cw + 4 lg + e - 4.  hp: the LDS head image (g2 | be2 | ...), col0 = 16 cw + 4 lg.
// Elementwise ln_act2 (the same bits as ln_act per element).
template <bool RELU>
__device__ __forceinline__ uint4 ln2_operand(const v4f& z0, const v4f& z1, float mean, float rstd, const float* hp, int col0, float slope) {
    const v2f m2 = {mean, mean}, r2 = {rstd, rstd};
    float h[8];
    {
        const v4f g0 = *reinterpret_cast<const v4f*>(hp + col0), b0 = *reinterpret_cast<const v4f*>(hp + H2 + col0);
        const v2f a = ln_act2<RELU>(v2f{z0[0], z0[1]}, m2, r2, v2f{g0[0], g0[1]}, v2f{b0[0], b0[1]}, slope);
        const v2f b = ln_act2<RELU>(v2f{z0[2], z0[3]}, m2, r2, v2f{g0[2], g0[3]}, v2f{b0[2], b0[3]}, slope);
        h[0] = a.x; h[1] = a.y; h[2] = b.x; h[3] = b.y;
    }
    __builtin_amdgcn_sched_barrier(0);  // (one column tile's eight LayerNorm parameters at a time: with all sixteen requested up front the persistent kernel spills)
    {
        const v4f g1 = *reinterpret_cast<const v4f*>(hp + 256 + col0), b1 = *reinterpret_cast<const v4f*>(hp + H2 + 256 + col0);
        const v2f a = ln_act2<RELU>(v2f{z1[0], z1[1]}, m2, r2, v2f{g1[0], g1[1]}, v2f{b1[0], b1[1]}, slope);
        const v2f b = ln_act2<RELU>(v2f{z1[2], z1[3]}, m2, r2, v2f{g1[2], g1[3]}, v2f{b1[2], b1[3]}, slope);
        h[4] = a.x; h[5] = a.y; h[6] = b.x; h[7] = b.y;
    }
    return pack8_bf16(h);
}
// ... and the other operand: W3's rows at the same columns, for lane (i = lane & 15, g = lane >> 4) of the wave that owns column group cw —
// output i < OUT, k = 8 g + e; zero rows beyond OUT (the MFMA tile is 16 wide).  Loaded once per workgroup from the fp32 parameters.
__device__ __forceinline__ uint4 w3_fragment(const float* __restrict__ net, const Mlp& m, int cw, int lane) {
    const int i = lane & 15, g = lane >> 4;
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = 0.0f;
    if (i < m.out) {
        const float* row = net + m.W3() + (size_t)i * H2 + 16 * cw + 4 * g;
        const v4f a = *reinterpret_cast<const v4f*>(row), b = *reinterpret_cast<const v4f*>(row + 256);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            w[e] = a[e];
            w[4 + e] = b[e];
        }
    }
    return pack8_bf16(w);
}
// the last step for (row, component c): the 16 column groups' shares in group order + b3.  outp: [16 groups][rows][4]
__device__ __forceinline__ float head_sum16(const float* outp, int rows, int row, int c, float b3) {
    float p[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) p[k] = outp[((size_t)k * rows + row) * 4 + c];
#pragma unroll
    for (int w = 1; w < 16; w *= 2)
#pragma unroll
        for (int k = 0; k < 16; k += 2 * w) p[k] += p[k + w];
    return p[0] + b3;
}
#pragma clang fp contract(fast)

// What follows the head for the row of lane (q, c), c < 4 = the action component: tanh, exploration noise, clamp (chooseAction*,
// HIRL.py:192-212) or the tanh-Gaussian sample (GaussianPolicy.sample, SAC/model.py:63-82).  s_noise: the row's four standard-normal draws.
template <bool GAUSS, int OUT>
__device__ __forceinline__ float action_of(const ActFusedArgs& A, const float (&o)[OUT], int c, int r, const float* s_noise) {
    if constexpr (!GAUSS) {
        float a = fast_tanh(c == 0 ? o[0] : c == 1 ? o[1] : c == 2 ? o[2] : o[3]);  // no dynamic register index
        if (A.noise) a = fminf(fmaxf(a + A.noise[(A.noise_per_row ? (size_t)r * 4 : 0) + c], -1.0f), 1.0f);
        else if (A.sigma > 0.0f) a = fminf(fmaxf(a + A.sigma * s_noise[c], -1.0f), 1.0f);
        return a;
    } else {
        // (selected from VALUES: a select chain over the elements of the array itself is folded back into a run-time index, which puts the
        //  array in scratch or — promoted — in 32 KB of LDS)
        const float o0 = o[0], o1 = o[1], o2 = o[2], o3 = o[3], o4 = o[4], o5 = o[5], o6 = o[6], o7 = o[7];
        const float mu = c == 0 ? o0 : c == 1 ? o1 : c == 2 ? o2 : o3;
        float a = mu;
        if (A.mode != 0) {
            const float ls = fminf(fmaxf(c == 0 ? o4 : c == 1 ? o5 : c == 2 ? o6 : o7, -20.0f), 2.0f);  // model.py:65-66
            const float e = A.mode == 1 ? A.noise[(size_t)r * 4 + c] : s_noise[c];
            a = mu + expf(ls) * e;
        }
        return tanhf(a);
    }
}

// the same for ONE pre-tanh output o of component c (deterministic head; the [r5] bf16 kernels hold one output per lane)
__device__ __forceinline__ float action_of1(const ActFusedArgs& A, float o, int c, int r, const float* s_noise) {
    float a = fast_tanh(o);
    if (A.noise) a = fminf(fmaxf(a + A.noise[(A.noise_per_row ? (size_t)r * 4 : 0) + c], -1.0f), 1.0f);
    else if (A.sigma > 0.0f) a = fminf(fmaxf(a + A.sigma * s_noise[c], -1.0f), 1.0f);
    return a;
}

// Rows beyond which the persistent kernel takes over (hx_actp.hip).  Up to here ONE round of 16- / 32-row workgroups covers the rows and
// the env step rides in wave 0 of each; beyond, every further round of workgroups would fetch the whole W2 image again.
constexpr int64_t kFuseEnvMax = 8192;

// hx_actp.hip.  mode: 0 fp32 from the fp32 image (H.w2f), 1 bf16 (H.w2b), 2 the exact three-way bf16 split, six partial products (H.w2b = hi | mid | lo, H.x9).
// Returns false when no persistent instantiation covers the request (the caller falls back to act_fused_kernel).
bool launch_act_persist(const ActFusedArgs& H, bool gauss, hipStream_t st);

// the buffers and options of an env step that rides in an acting launch (hx_*_act_step*, hx_hirl_front)
inline int check_step_args(const float* state, int64_t n, int64_t stride, const float* obs_io, const float* actions, const float* reward,
                           const uint8_t* done, const int8_t* success, const HxStepOpts& o, const char* who) {
    HX_REQUIRE(state && obs_io && actions && reward && done && success && n > 0 && stride >= n, "%s: bad buffers", who);
    HX_REQUIRE(n < (int64_t)1 << 31, "%s: at most 2^31 - 1 envs per launch", who);
    HX_REQUIRE(!o.auto_reset || o.episode_ctr, "%s: auto_reset needs episode_ctr", who);
    HX_REQUIRE(stride < ((int64_t)1 << 25), "%s: stride must be below 2^25 envs", who);
    HX_REQUIRE(!o.ring || (o.cap >= 512 && o.cap < ((int64_t)1 << 31) && o.total && (reinterpret_cast<uintptr_t>(o.ring) & 15u) == 0),
               "%s: ring needs 512 <= cap < 2^31, total and 16-byte alignment", who);
    return 0;
}

}  // namespace hxact

namespace hxu {
// hx_front.hip: the SAC policy's act + env + insert launch (persistent streaming kernel) with the first forward launch of learn() behind it
int launch_front_sac(const hxact::ActFusedArgs& H, const FwdArgs& FA, hipStream_t st);
}  // namespace hxu
