// hx_actp_body.h — the persistent bf16 acting workgroup (act_persist_bf16_kernel of hx_actp.hip) as a device function over an explicit LDS block and an
// explicit workgroup index / count, + the helpers both persistent kernels share; hx_front.hip runs the same workgroups beside learn()'s first launches.
#pragma once
#include "hx_act.h"
#include "hx_env_block.h"

namespace hxact {
using namespace hxnn;
using namespace hxu;

// -DHX_PX=mask builds a timing experiment (tools/ubench/actp_variants.sh; wrong results): 1 no LayerNorm 2 / final layer / last step (the product then
// is dead code and goes too), 2 no LayerNorm 1, 4 no product MFMAs, 8 no layer 1, 32 no noise draw, 128 every h1 fragment from slab 0
#ifndef HX_PX
#define HX_PX 0
#endif
// per-wave stamps of ONE steady-state iteration of the bf16 tile loop (diagnostic build only: tools/ubench/stamps_actp_waves.py)
#ifdef HX_STAMPS
#define WSTAMP(k) do { if (i == 4) ws_[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WSTAMP(k)
#endif
constexpr int kEnvPass = 512;  // envs per pass of the tail: pair layout = 1,024 lanes = the workgroup

// Layer 1 of NRT row tiles on the fp32 matrix cores: the products and the k order of act_fused_kernel's layer 1, with the two MFMA operands
// SWAPPED — A = W1 (wave w: hidden units 16 w .. + 15), B = the observation rows — so that lane (lr, lg) ends up with FOUR CONSECUTIVE units
// 16 w + 4 lg .. + 3 of row lr: one 16-byte LDS store per row tile instead of four conflicting dword stores.  The W1 fragments (w1f) and
// the biases of the lane's four units (b1f) are loop invariants held in registers.
template <int NRT>
__device__ __forceinline__ void layer1_tiles(const float* xs, const float (&w1f)[4], v4f b1f, int wave, int lr, int lg, float* h1s) {
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
        v4f acc = b1f;
        const float* xrow = xs + (t * RT + lr) * XP + lg;  // (columns 13.. of xs are zero, and so are the W1 fragments there)
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) acc = mfma16(w1f[mm], xrow[4 * mm], acc);  // K = 16 covers the 13 inputs
        *reinterpret_cast<v4f*>(h1s + (t * RT + lr) * LDA1 + wave * 16 + 4 * lg) = acc;
    }
}

// the four standard-normal draws of a row, two per lane: lane (row, pair p) -> components 2 p (cos) and 2 p + 1 (sin) of philox_normal's
// Box-Muller pairs — one Philox evaluation, logarithm and root per PAIR instead of per component, the same bits
__device__ __forceinline__ void philox_normal_pair(uint32_t row, uint32_t call, uint32_t tag, uint64_t seed, int p, float& n_cos, float& n_sin) {
    uint32_t u[4];
    philox4x32_10(row, call, tag, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), u);
    const float ua = u01(p ? u[2] : u[0]), ub = u01(p ? u[3] : u[1]);
    const float rad = sqrtf(-2.0f * logf(ua)), ang = 6.28318530717958647692f * ub;
    n_cos = rad * cosf(ang);
    n_sin = rad * sinf(ang);
}

// The env step of rows [row_begin, row_end) — whose actions this workgroup has written — kEnvPass at a time (HarfangEnv.step,
// train_all.py:345).  It reads the launch description from the kernel-argument SEGMENT (uniform scalar loads, here, after the tile loop)
// instead of from the kernel's parameter: as a parameter the env step's ~30 argument words stayed live in SGPRs across the tile loop, next to
// the loop's own ~60, spilled into VGPR lanes, and pushed the loop's vector registers (64 of them resident weights) into scratch.
typedef const __attribute__((address_space(4))) ActFusedArgs* KernArgs;
__device__ __forceinline__ void env_tail(KernArgs Ap, int row_begin, int row_end, float* elds, unsigned* s_slot0, int* s_wcount, int bid, int nwg) {
    using namespace hxenv;
    StepArgs S;
    S.state = Ap->state; S.n = (int64_t)Ap->rows; S.stride = Ap->stride; S.actions = Ap->actions; S.obs_io = Ap->obs; S.reward = Ap->reward;
    S.done = Ap->done; S.success = Ap->success; S.inv_cap = Ap->inv_cap;
    S.o.max_step = Ap->o.max_step; S.o.auto_reset = Ap->o.auto_reset; S.o.randomize = Ap->o.randomize; S.o.env_id0 = Ap->o.env_id0; S.o.seed = Ap->o.seed;
    S.o.episode_ctr = Ap->o.episode_ctr; S.o.ring = Ap->o.ring; S.o.ring_success = Ap->o.ring_success; S.o.cap = Ap->o.cap; S.o.total = Ap->o.total;
    S.o.stats = Ap->o.stats; S.o.ev_start = nullptr; S.o.ev_stop = nullptr; S.o.layout = 0;
    unsigned way = (unsigned)bid;
    for (int i0 = row_begin; i0 < row_end; i0 += kEnvPass, way += (unsigned)nwg) {
        // (behind opaque copies of the base pointers nothing of a pass is invariant across passes: hoisted out of this loop, the 37 state
        //  words' 64-bit addresses alone are 74 VGPRs and the pass spills ~170 bytes per lane)
        asm volatile("" : "+s"(S.state), "+s"(S.obs_io), "+s"(S.actions), "+s"(S.stride));
        env_block_step<true, true, kEnvPass>(S, i0, row_end, elds, *s_slot0, s_wcount, way);  // (launches without a replay ring take two launches)
        __syncthreads();  // the pass's LDS tiles and slot words are free again
    }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16, weight-stationary (BASELINE.json configs[4]: bf16 actor, fp32 dynamics)
// ---------------------------------------------------------------------------------------------------------------
// the workgroup's LDS as one object (hx_front.hip overlays it with the forward workgroups' block)
template <bool ENV>
struct ActpLds {
    static constexpr int NRT = 2, TR = NRT * RT;
    typedef HeadImage<4> Img;
    // the loop's tiles: h1s (fp32 pre-activations of layer 1), h1b (bf16 h1, TWO tiles: the product of tile i - 1 reads one while LayerNorm 1 of tile i
    // fills the other), the partial LayerNorm-2 statistics [TR][16 groups][2] and the column groups' shares of the outputs [2 tile parities][16 groups][TR][4]
    // ([r5]: no z2 tile — hx_act.h "straight from the accumulators")
    static constexpr int kLoop = TR * LDA1 + 2 * (TR * LDB1 / 2) + TR * kPartPitch + 2 * 16 * TR * 4;
    static constexpr int kTail = ENV ? hxenv::kEnvBlockLds<true, kEnvPass> : 0;
    static constexpr int kUnion = kLoop > kTail ? kLoop : kTail;
    __attribute__((aligned(16))) float lds[Img::kStride + TR * XP + 3 * H1 + H2 + 2 * kWide * 4 + 2 * TR * 4 + kUnion];
    unsigned s_slot0;
    int s_wcount[kWide / 64];
};
// bid / nwg: this workgroup's index among the acting workgroups and their number (the kernel's grid, or the acting role's share of a front launch)
template <bool ENV, bool RELU>
__device__ __forceinline__ void act_persist_bf16_body(const ActFusedArgs& A, const int tiles_per_wg, const int bid, const int nwg, ActpLds<ENV>& SL) {
    constexpr int NRT = 2, TR = NRT * RT;
    typedef HeadImage<4> Img;
    float* const lds = SL.lds;
    unsigned& s_slot0 = SL.s_slot0;
    int* const s_wcount = SL.s_wcount;
    float* hps = lds;                   // g2 | be2 | (W3 rows: unused here, the final layer's operand lives in w3t) | b3
    float* xs = hps + Img::kStride;
    float* g1s = xs + TR * XP;          // LayerNorm 1 weight | bias, read four columns at a time
    float* b1s = g1s + 2 * H1;          // full1.bias
    float* b2s = b1s + H1;              // full2.bias
    float* w1t = b2s + H2;              // [1024][4]: every lane's four layer-1 A fragments of W1 (zero beyond the 13 inputs): one 16-byte read per tile
    float* w3t = w1t + kWide * 4;       // [1024][4]: every lane's A fragment of the final layer (bf16 x 8: W3 at the wave's 32 columns, hx_act.h w3_fragment)
    float* s_noise = w3t + kWide * 4;   // [2][TR][4]: the draws of tile t live in half t & 1
    float* h1s = s_noise + 2 * TR * 4;
    __bf16* h1b = reinterpret_cast<__bf16*>(h1s + TR * LDA1);  // [2][TR][LDB1]
    float* part = h1s + TR * LDA1 + 2 * (TR * LDB1 / 2);  // [TR][16][2]
    float* outp = part + TR * kPartPitch;           // [2][16][TR][4]
    const int tid0 = threadIdx.x;
    const int row_begin = bid * tiles_per_wg * TR;
    if (row_begin >= A.rows) return;
    const int row_end = min(A.rows, row_begin + tiles_per_wg * TR);
    const int ntile = (row_end - row_begin + TR - 1) / TR;
    const float* net = A.net;
    const Mlp m = A.m;
    const float slope = A.slope;
    STAMP_DECL;
    STAMP();
    // ---- once per workgroup: the small operands first, then this wave's share of the W2 image --------------------------------------
    auto obs_of = [&](int tile, int tid) -> float {  // element `tid` of the tile's [TR][13] observation block (0 beyond the rows)
        const int r0 = row_begin + tile * TR;
        return (tile < ntile && tid < TR * 13 && r0 + tid / 13 < row_end) ? A.obs[(size_t)r0 * 13 + tid] : 0.0f;
    };
    float xv;
    uint4 bq[2][8];  // B fragments of this wave's two column tiles, all of K: resident for every tile
    {
        const int tid = tid0, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lg = lane >> 4;
        xv = obs_of(0, tid);
        float w1f[4];  // layer 1: this lane's A fragments of W1 (unit 16 wave + lr, inputs 4 mm + lg)
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) w1f[mm] = net[m.W1() + (wave * 16 + lr) * 13 + min(4 * mm + lg, 12)];
        const float b1v = tid < H1 ? net[m.b1() + tid] : 0.0f;
        const float gb = tid < 2 * H1 ? net[m.g1() + tid] : 0.0f;  // g1 | be1 are adjacent in the parameter block
        const float b2v = tid < H2 ? net[m.b2() + tid] : 0.0f;
        Img himg;
        himg.fetch(net, m, tid);
        // which 32 columns this wave owns rotates with the workgroup: the workgroups of a launch do not all ask L2 for the same lines at once
        const int cw = (wave + bid) & 15;
        const uint4 w3q = w3_fragment(net, m, cw, lane);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint16_t* blk = A.w2b + (size_t)((t * 16 + cw) * 8) * 512 + lane * 8;  // (lane = 16 g + r: w2_image_index's block order)
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) bq[t][sl] = *reinterpret_cast<const uint4*>(blk + sl * 512);
        }
        *reinterpret_cast<float4*>(w1t + tid * 4) = make_float4(w1f[0], w1f[1], w1f[2], 12 + lg < 13 ? w1f[3] : 0.0f);  // (only 4 mm + lg = 13..15 are beyond)
        *reinterpret_cast<uint4*>(w3t + tid * 4) = w3q;
        if (tid < H1) b1s[tid] = b1v;
        if (tid < 2 * H1) g1s[tid] = gb;
        if (tid < H2) b2s[tid] = b2v;
        if (tid < TR * XP) xs[tid] = 0.0f;
        himg.store(hps, net, m, tid);
        __syncthreads();
        if (tid < TR * 13) xs[(tid / 13) * XP + tid % 13] = xv;
        xv = obs_of(1, tid);
        __syncthreads();
    }
    const bool draw_noise = !A.noise && A.sigma > 0.0f;
    STAMP();
    // ---- the tile loop, two barriers per tile, four tiles in flight.  Each phase pairs MATRIX work of one tile with VECTOR work of another, so that neither
    //      pipe idles while the other runs (rounds 3-4 alternated an MFMA-only phase with a VALU-only one: 27 % matrix cores, 43 % VALU busy):
    //        X(i):  every wave: LayerNorm 1 + activation of two rows of tile i -> bf16 h1, then the product of tile i - 1 (bf16 matrix cores) -> accumulators,
    //               bias, partial LayerNorm-2 statistics | wave 0: the exploration noise of tile i - 1 | waves 2, 3: the last sum + tanh + noise + clamp of
    //               tile i - 2 | the observation tile i + 1 -> LDS
    //        Y(i):  LayerNorm 2 + activation of tile i - 1 from the accumulators, the final layer's shares (hx_act.h), with layer 1 of tile i + 1 (fp32 matrix
    //               cores) dealt between its steps
    {   // layer 1 of tile 0 (the loop computes tile i + 1's in Y(i))
        const int tid = tid0, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lg = lane >> 4;
        if (!(HX_PX & 8)) {
            const v4f w1v = *reinterpret_cast<const v4f*>(w1t + tid * 4);
            const float w1f[4] = {w1v[0], w1v[1], w1v[2], w1v[3]};
            layer1_tiles<NRT>(xs, w1f, *reinterpret_cast<const v4f*>(b1s + wave * 16 + 4 * lg), wave, lr, lg, h1s);
        }
        __syncthreads();
    }
    v4f acc[NRT][2];  // z2 of tile i - 1: this lane's 4 + 4 columns of rows lr and 16 + lr — alive from the product (X) across barrier A into Y
#pragma unroll
    for (int t = 0; t < NRT; ++t) acc[t][0] = acc[t][1] = v4f{0.f, 0.f, 0.f, 0.f};
#ifdef HX_STAMPS
    unsigned long long ws_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int i = 0; i <= ntile + 1; ++i) {
        // The lane's LDS addresses are loop invariants, and with 64 registers of weights resident the allocator spills them; behind this
        // opaque copy of the thread id they are recomputed per tile (a few VALU instructions) instead of living across the loop.
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        const int lr = lane & 15, lg = lane >> 4;
        const int cw = (wave + bid) & 15;
        const bool mid = i >= 1 && i <= ntile;  // tile i - 1 exists
        WSTAMP(0);
        // ---- X: vector work first (its waves' matrix work follows while the other waves' runs) ----
        {
            // LayerNorm 1 + activation of tile i -> bf16 h1: EVERY wave takes two rows, 32 lanes per row (hx_act.h row_stats32; lane (h, c): row 2 wave + h,
            // columns 4 c .. + 3 and 128 + 4 c .. + 3).  With 16 lanes per row on waves 8-15 this was the phase's long pole: 1.3-1.8 us of its 2.9
            // (profiles/r05_actp_bf16_wave_stamps_v2.txt)
            if (i < ntile && !(HX_PX & 2)) {
                const int row = 2 * wave + (lane >> 5), c = lane & 31;
                __bf16* const hb_row = h1b + (i & 1) * TR * LDB1 + row * LDB1;
                const v4f x0 = *reinterpret_cast<const v4f*>(h1s + row * LDA1 + 4 * c), x1 = *reinterpret_cast<const v4f*>(h1s + row * LDA1 + 128 + 4 * c);
                float mean, rstd;
                row_stats32(x0, x1, H1, mean, rstd);
                if (m.no_ln) { mean = 0.0f; rstd = 1.0f; }
                typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const v4f x = k ? x1 : x0;
                    const v4f g = *reinterpret_cast<const v4f*>(g1s + 128 * k + 4 * c);
                    const v4f be = *reinterpret_cast<const v4f*>(g1s + H1 + 128 * k + 4 * c);
                    v4bf hb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hb[e] = (__bf16)ln_act<RELU>(x[e], mean, rstd, g[e], be[e], slope);  // round to nearest even
                    *reinterpret_cast<v4bf*>(hb_row + 128 * k + 4 * c) = hb;
                }
            }
        }
        if (wave == 0) {
            // the exploration noise of tile i - 1 (its last step runs in X(i + 1), reading the other half): a lane per (row, Box-Muller pair)
            if (draw_noise && mid && !(HX_PX & 32)) {
                float nc, ns;
                philox_normal_pair(A.row0 + (uint32_t)(row_begin + (i - 1) * TR + (lane >> 1)), A.call, 0x61637421u, A.seed, lane & 1, nc, ns);
                *reinterpret_cast<float2*>(s_noise + ((i - 1) & 1) * TR * 4 + lane * 2) = make_float2(nc, ns);
            }
        } else if (wave == 2 || wave == 3) {
            // the last step of tile i - 2 — its 16 shares per output in column-group order + b3, tanh, exploration noise, clamp (a lane per (row, component):
            // 256 contiguous bytes of actions per wave); the shares were written in Y(i - 1)
            const int lrow = (wave - 2) * 16 + (lane >> 2), c = lane & 3, r = row_begin + (i - 2) * TR + lrow;
            if (i >= 2 && r < row_end && !(HX_PX & 1)) {
                const float o = head_sum16(outp + (size_t)((i & 1) * 16) * TR * 4, TR, lrow, c, hps[(2 + 4) * H2 + c]);
                A.actions[(size_t)r * 4 + c] = action_of1(A, o, c, r, s_noise + (i & 1) * TR * 4 + lrow * 4);
            }
        }
        if (i + 1 < ntile) {  // (layer 1 of tile i read xs in Y(i - 1))
            if (tid < TR * 13) xs[(tid / 13) * XP + tid % 13] = xv;
            xv = obs_of(i + 2, tid);
        }
        WSTAMP(1);
        // z2(i - 1) = h1(i - 1) W2^T (weights from registers).  The MFMA operands are swapped (weights as A, rows as B): the same products in the same
        // k order, but lane (lr, lg) then holds FOUR CONSECUTIVE columns of row lr.
        if (mid) {
#pragma unroll
            for (int t = 0; t < NRT; ++t) acc[t][0] = acc[t][1] = v4f{0.f, 0.f, 0.f, 0.f};
            // K = 256 in 8 slabs of 32: lane (r, g) holds h1[row r][32 sl + 8 g ..+7] and W2[col r][32 sl + 8 g ..+7].  (One register set for
            // the h1 fragments: the other three waves of the SIMD cover a wave's LDS round trip, and a second set costs spills of the weights.)
            const __bf16* ap = h1b + ((i - 1) & 1) * TR * LDB1 + lr * LDB1 + 8 * lg;
#pragma unroll
            for (int sl = 0; sl < ((HX_PX & 4) ? 0 : 8); ++sl) {
                uint4 aq[NRT];
#pragma unroll
                for (int t = 0; t < NRT; ++t) aq[t] = *reinterpret_cast<const uint4*>(ap + t * RT * LDB1 + 32 * ((HX_PX & 128) ? 0 : sl));
#pragma unroll
                for (int t = 0; t < NRT; ++t) {
                    acc[t][0] = mfma16_bf16(bq[0][sl], aq[t], acc[t][0]);
                    acc[t][1] = mfma16_bf16(bq[1][sl], aq[t], acc[t][1]);
                }
            }
            WSTAMP(2);
            // bias, then the wave's partial LayerNorm-2 statistics of its 32 columns of rows lr / 16 + lr (hx_act.h: step 1)
            const v4f bb0 = *reinterpret_cast<const v4f*>(b2s + cw * 16 + 4 * lg), bb1 = *reinterpret_cast<const v4f*>(b2s + 256 + cw * 16 + 4 * lg);
#pragma unroll
            for (int t = 0; t < NRT; ++t) {
                acc[t][0] = acc[t][0] + bb0;
                acc[t][1] = acc[t][1] + bb1;
                if (!(HX_PX & 1)) row_partial32(acc[t][0], acc[t][1], lg, part + (t * RT + lr) * kPartPitch + 2 * cw);
            }
        }
        WSTAMP(3);
        __syncthreads();  // A: the partial statistics of tile i - 1, h1 of tile i (bf16) and the observation tile i + 1 are in LDS; the pre-activations are free
        WSTAMP(4);
        // ---- Y: layer 1 of tile i + 1 on the fp32 matrix cores, its eight instructions (two dependent chains of four) DEALT between the steps of LayerNorm 2 +
        //      activation of tile i - 1 from the accumulators and the final layer's shares (hx_act.h: steps 2, 3 -> outp[(i - 1) & 1]): a wave that issues
        //      its chain in one go waits at every link for the link before, with its vector work queued behind ----
        v4f acc1[NRT];
        const bool l1 = i + 1 < ntile && !(HX_PX & 8);
        const bool ln2 = mid && !(HX_PX & 1);
        const v4f w1f = *reinterpret_cast<const v4f*>(w1t + tid * 4);
        float mean = 0.0f, rstd = 1.0f;
        uint4 hq[NRT];
#pragma unroll
        for (int t = 0; t < NRT; ++t) {
            acc1[t] = *reinterpret_cast<const v4f*>(b1s + wave * 16 + 4 * lg);
            hq[t] = uint4{0u, 0u, 0u, 0u};
        }
        WSTAMP(5);
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            if (l1) {
#pragma unroll
                for (int t = 0; t < NRT; ++t) acc1[t] = mfma16(w1f[mm], xs[(t * RT + lr) * XP + lg + 4 * mm], acc1[t]);  // K = 16 covers the 13 inputs; k ascending
            }
            if (ln2) {  // row tile mm >> 1: statistics, then the operand
                if (!(mm & 1)) row_combine16(part + ((mm >> 1) * RT + lr) * kPartPitch, lg, m.no_ln, mean, rstd);
                else hq[mm >> 1] = ln2_operand<RELU>(acc[mm >> 1][0], acc[mm >> 1][1], mean, rstd, hps, cw * 16 + 4 * lg, slope);
            }
            __builtin_amdgcn_sched_barrier(0);  // (the deal stays as written)
        }
        if (ln2) {
            const uint4 w3q = *reinterpret_cast<const uint4*>(w3t + tid * 4);
            float* const op = outp + (size_t)(((i - 1) & 1) * 16 + cw) * TR * 4;
#pragma unroll
            for (int t = 0; t < NRT; ++t) {
                const v4f o = mfma16_bf16(w3q, hq[t], v4f{0.f, 0.f, 0.f, 0.f});  // D[i = 4 lg + q][j = lr]: outputs 0..3 of row lr in the lg = 0 lanes
                if (lg == 0) *reinterpret_cast<v4f*>(op + (t * RT + lr) * 4) = o;
            }
        }
        WSTAMP(6);
        if (l1) {
#pragma unroll
            for (int t = 0; t < NRT; ++t) *reinterpret_cast<v4f*>(h1s + (t * RT + lr) * LDA1 + wave * 16 + 4 * lg) = acc1[t];
        }
        WSTAMP(7);
        __syncthreads();  // B: the pre-activations of tile i + 1 and the output shares of tile i - 1 are in LDS; the partial statistics and xs are free
        WSTAMP(8);
    }
#ifdef HX_STAMPS
    {   // waves 0, 5, 9, 15 of workgroup 3: eight phase lengths each -> hx_dbg[16 + 8 slot ..]
        const int wv = tid0 >> 6, slot = wv == 0 ? 0 : wv == 5 ? 1 : wv == 9 ? 2 : wv == 15 ? 3 : -1;
        if (bid == 3 && (tid0 & 63) == 0 && slot >= 0)
            for (int k = 0; k < 8; ++k) hx_dbg[16 + 8 * slot + k] = (float)(ws_[k + 1] - ws_[k]);
    }
#endif
    STAMP();
    if (ENV) env_tail((KernArgs)__builtin_amdgcn_kernarg_segment_ptr(), row_begin, row_end, h1s, &s_slot0, s_wcount, bid, nwg);  // (barrier B: every action of the block is written; the launch description is the kernel's FIRST argument)
    STAMP();
    STAMP_FLUSH(0, (bid == 0 || bid == 200) && tid0 == 0);
    SPAN_LOG(HX_SPAN_ACT);
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 (MODE 0: the fp32 image, fp32 MFMA) and fp32 through the exact three-way bf16 split (MODE 1: hi | mid | lo images, bf16 MFMA; six partial products, hx_act.h HX_X9_TERMS): W2 does
// not fit the register file (512 / 768 KB), so each wave STREAMS its column slices from L2 once per pass over 64 rows — four row tiles per B
// fragment, a quarter of act_fused_kernel<2>'s L2 traffic per row — and z2 leaves the accumulators in two halves of 32 rows through the
// LDS that held h1.  Per pass: layer 1 | LayerNorm 1 (16 lanes per row, every wave four rows) | the product | 2 x { z2 half -> LDS, head }.
// The k order of every accumulator, and with MODE 1 the order of the partial products, are act_fused_kernel's: the same bits.
// ---------------------------------------------------------------------------------------------------------------
template <int MODE, bool GAUSS, bool ENV>
struct ActpsLds {
    static constexpr bool X9 = MODE == 1;
    static constexpr int NRT = 4, TR = NRT * RT, HR = 32;  // rows per pass; rows per z2 half
    typedef HeadImage<GAUSS ? 8 : 4> Img;
    static constexpr int kH1 = X9 ? 3 * TR * LDB1 / 2 : TR * LDA1;  // h1: three bf16 tiles (hi | mid | lo) or one fp32 tile; before that the fp32 pre-activations
    static constexpr int kPre = TR * LDA1;
    static constexpr int kZ = HR * LDA2;
    static constexpr int kTail = ENV ? hxenv::kEnvBlockLds<true, kEnvPass> : 0;
    static constexpr int kUnion = (kH1 > kPre ? kH1 : kPre) > (kZ > kTail ? kZ : kTail) ? (kH1 > kPre ? kH1 : kPre) : (kZ > kTail ? kZ : kTail);
    __attribute__((aligned(16))) float lds[Img::kStride + TR * XP + 3 * H1 + H2 + kWide * 4 + 2 * TR * 4 + kUnion];
    unsigned s_slot0;
    int s_wcount[kWide / 64];
};
template <int MODE, bool GAUSS, bool ENV, bool RELU>
__device__ __forceinline__ void act_persist_stream_body(const ActFusedArgs& A, const int tiles_per_wg, const int bid, const int nwg, ActpsLds<MODE, GAUSS, ENV>& SL) {
    typedef ActpsLds<MODE, GAUSS, ENV> Lds;
    constexpr bool X9 = MODE == 1;
    constexpr int NRT = 4, TR = NRT * RT, HR = 32;
    constexpr int OUT = GAUSS ? 8 : 4;
    typedef HeadImage<OUT> Img;
    constexpr int kH1 = Lds::kH1, kPre = Lds::kPre, kZ = Lds::kZ;
    (void)kH1; (void)kPre; (void)kZ;
    float* const lds = SL.lds;
    unsigned& s_slot0 = SL.s_slot0;
    int* const s_wcount = SL.s_wcount;
    float* hps = lds;
    float* xs = hps + Img::kStride;
    float* g1s = xs + TR * XP;          // LayerNorm 1 weight | bias
    float* b1s = g1s + 2 * H1;          // full1.bias
    float* b2s = b1s + H1;              // full2.bias
    float* w1t = b2s + H2;              // [1024][4]: every lane's four layer-1 A fragments of W1 (zero beyond the 13 inputs)
    float* s_noise = w1t + kWide * 4;   // [2][TR][4]: the draws of pass p live in half p & 1
    float* h1s = s_noise + 2 * TR * 4;  // the pass's region: pre-activations -> h1 -> z2 halves (-> the env tail's tiles)
    float* z2s = h1s;
    uint16_t* h1x = reinterpret_cast<uint16_t*>(h1s);
    const int tid0 = threadIdx.x;
    const int row_begin = bid * tiles_per_wg * TR;
    if (row_begin >= A.rows) return;
    const int row_end = min(A.rows, row_begin + tiles_per_wg * TR);
    const int npass = (row_end - row_begin + TR - 1) / TR;
    const float* net = A.net;
    const Mlp m = A.m;
    const float slope = A.slope;
    const bool draw_noise = GAUSS ? (A.mode != 0 && A.mode != 1) : (!A.noise && A.sigma > 0.0f);
    const uint32_t noise_tag = GAUSS ? 0x53414331u : 0x61637421u;
    STAMP_DECL;
    STAMP();
    auto obs_of = [&](int pass, int tid) -> float {  // element `tid` of the pass's [TR][13] observation block (0 beyond the rows)
        const int r0 = row_begin + pass * TR;
        return (pass < npass && tid < TR * 13 && r0 + tid / 13 < row_end) ? A.obs[(size_t)r0 * 13 + tid] : 0.0f;
    };
    // the exploration noise of a pass: waves 8 and 9, a lane per (row, Box-Muller pair)
    auto draw = [&](int pass, int wave, int lane) {
        if (draw_noise && (wave == 8 || wave == 9) && pass < npass) {
            const int lrow = (wave - 8) * 32 + (lane >> 1);
            float nc, ns;
            philox_normal_pair(A.row0 + (uint32_t)(row_begin + pass * TR + lrow), A.call, noise_tag, A.seed, lane & 1, nc, ns);
            *reinterpret_cast<float2*>(s_noise + (pass & 1) * TR * 4 + lrow * 4 + (lane & 1) * 2) = make_float2(nc, ns);
        }
    };
    float xv;
    {
        const int tid = tid0, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lg = lane >> 4;
        xv = obs_of(0, tid);
        float w1f[4];  // layer 1: this lane's A fragments of W1 (unit 16 wave + lr, inputs 4 mm + lg)
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) w1f[mm] = net[m.W1() + (wave * 16 + lr) * 13 + min(4 * mm + lg, 12)];
        const float b1v = tid < H1 ? net[m.b1() + tid] : 0.0f;
        const float gb = tid < 2 * H1 ? net[m.g1() + tid] : 0.0f;  // g1 | be1 are adjacent in the parameter block
        const float b2v = tid < H2 ? net[m.b2() + tid] : 0.0f;
        Img himg;
        himg.fetch(net, m, tid);
        draw(0, wave, lane);
        *reinterpret_cast<float4*>(w1t + tid * 4) = make_float4(w1f[0], w1f[1], w1f[2], 12 + lg < 13 ? w1f[3] : 0.0f);
        if (tid < H1) b1s[tid] = b1v;
        if (tid < 2 * H1) g1s[tid] = gb;
        if (tid < H2) b2s[tid] = b2v;
        for (int k = tid; k < TR * XP; k += kWide) xs[k] = 0.0f;
        himg.store(hps, net, m, tid);
        __syncthreads();
        if (tid < TR * 13) xs[(tid / 13) * XP + tid % 13] = xv;
        xv = obs_of(1, tid);
        __syncthreads();
    }
    STAMP();
    for (int p = 0; p < npass; ++p) {
        // (an opaque copy of the thread id per pass: the lane's LDS / image addresses are recomputed instead of being hoisted out of the pass
        //  loop as dozens of 64-bit invariants that spill)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 15, lg = lane >> 4, gq = lg, gc = lr;
        const int r0 = row_begin + p * TR;
        // ---- layer 1 of the pass's four row tiles (fp32 matrix cores) -> pre-activations -------------------------------------------
        {
            const v4f w1v = *reinterpret_cast<const v4f*>(w1t + tid * 4);
            const float w1f[4] = {w1v[0], w1v[1], w1v[2], w1v[3]};
            layer1_tiles<NRT>(xs, w1f, *reinterpret_cast<const v4f*>(b1s + wave * 16 + 4 * lg), wave, lr, lg, h1s);
        }
        __syncthreads();
        // ---- LayerNorm 1 + activation, 16 lanes per row: wave w rows w, w + 16, w + 32, w + 48 (pitch 8 mod 64 dwords: disjoint banks) ---
        {
            const int row = wave + 16 * gq;
            float v[16];
            load_row16<H1>(h1s + row * LDA1, gc, v);
            float mean, rstd;
            row_stats16<16>(v, H1, mean, rstd);
            if (m.no_ln) { mean = 0.0f; rstd = 1.0f; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const v4f g = *reinterpret_cast<const v4f*>(g1s + 64 * k + 4 * gc);
                const v4f be = *reinterpret_cast<const v4f*>(g1s + H1 + 64 * k + 4 * gc);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * k + e] = ln_act<RELU>(v[4 * k + e], mean, rstd, g[e], be[e], slope);
            }
            if (X9) __syncthreads();  // the three bf16 tiles lie over the pre-activations: every row has been read
            // the next pass's observations (xs was last read by layer 1, two barriers ago in either mode)
            if (tid < TR * 13) xs[(tid / 13) * XP + tid % 13] = xv;
            xv = obs_of(p + 2, tid);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if constexpr (X9) {
                    uint16_t hi[4], mid[4], lo[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) split3_bf16(v[4 * k + e], hi[e], mid[e], lo[e]);
                    uint16_t* dst = h1x + row * LDB1 + 64 * k + 4 * gc;
                    *reinterpret_cast<uint2*>(dst) = make_uint2(hi[0] | ((unsigned)hi[1] << 16), hi[2] | ((unsigned)hi[3] << 16));
                    *reinterpret_cast<uint2*>(dst + TR * LDB1) = make_uint2(mid[0] | ((unsigned)mid[1] << 16), mid[2] | ((unsigned)mid[3] << 16));
                    *reinterpret_cast<uint2*>(dst + 2 * TR * LDB1) = make_uint2(lo[0] | ((unsigned)lo[1] << 16), lo[2] | ((unsigned)lo[3] << 16));
                } else {
                    *reinterpret_cast<v4f*>(h1s + row * LDA1 + 64 * k + 4 * gc) = v4f{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
                }
            }
        }
        __syncthreads();
        // ---- z2 = h1 W2^T for 64 rows x this wave's 32 columns; MFMA operands swapped (weights as A): lane (lr, lg) ends up with four
        //      consecutive columns of row lr -----------------------------------------------------------------------------------------
        v4f acc[NRT][2];
#pragma unroll
        for (int t = 0; t < NRT; ++t) acc[t][0] = acc[t][1] = v4f{0.f, 0.f, 0.f, 0.f};
        const int cw = X9 ? ((wave + bid) & 15) : wave;  // which 32 columns (x9: rotated with the workgroup, as act_fused_kernel)
        if constexpr (X9) {
            v4f rest[NRT][2];
#pragma unroll
            for (int t = 0; t < NRT; ++t) rest[t][0] = rest[t][1] = v4f{0.f, 0.f, 0.f, 0.f};
            const uint16_t* img = A.w2b + (size_t)(cw * 8) * 512 + lane * 8;  // image s at + s kImgElems, column tile 16 + cw at + 16 * 8 * 512, slab sl at + 512 sl
            uint4 bb[2][3];  // the three parts (hi | mid | lo) of ONE column tile's slab; set ct holds column tile ct, requested one step ahead
            auto request = [&](uint4(&dst)[3], int sl, int ct) {
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) dst[sx] = *reinterpret_cast<const uint4*>(img + (size_t)sx * kImgElems + (size_t)ct * (16 * 8 * 512) + sl * 512);
            };
            auto multiply = [&](const uint4(&b)[3], int sl, int ct) {
#pragma unroll
                for (int t = 0; t < NRT; ++t) {
                    uint4 a3[3];
#pragma unroll
                    for (int sx = 0; sx < 3; ++sx) a3[sx] = *reinterpret_cast<const uint4*>(h1x + ((sx * TR) + t * RT + lr) * LDB1 + 32 * sl + 8 * lg);
                    // smallest first, as act_fused_kernel: lo lo, lo mid, mid lo | lo hi, hi lo, mid mid | mid hi, hi mid -> rest; hi hi -> acc
                    v4f r = rest[t][ct];
                    if (HX_X9_TERMS == 9) {
                        r = mfma16_bf16(b[2], a3[2], r);
                        r = mfma16_bf16(b[1], a3[2], r);
                        r = mfma16_bf16(b[2], a3[1], r);
                    }
                    r = mfma16_bf16(b[0], a3[2], r);
                    r = mfma16_bf16(b[2], a3[0], r);
                    r = mfma16_bf16(b[1], a3[1], r);
                    r = mfma16_bf16(b[0], a3[1], r);
                    r = mfma16_bf16(b[1], a3[0], r);
                    rest[t][ct] = r;
                    acc[t][ct] = mfma16_bf16(b[0], a3[0], acc[t][ct]);
                    __builtin_amdgcn_sched_barrier(0);  // (one row tile's fragments at a time: hoisted together, the twelve loads of a step spill)
                }
            };
            request(bb[0], 0, 0);
#pragma unroll 1
            for (int sl = 0; sl < 8; ++sl) {
                request(bb[1], sl, 1);
                __builtin_amdgcn_sched_barrier(0);  // the requests stay ahead of the multiply
                multiply(bb[0], sl, 0);
                if (sl + 1 < 8) request(bb[0], sl + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                multiply(bb[1], sl, 1);
            }
#pragma unroll
            for (int t = 0; t < NRT; ++t)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[t][ct] = acc[t][ct] + rest[t][ct];
        } else {
            // 16 chunks of 16 k: B fragments (one contiguous kilobyte per load from the fp32 image) PF chunks ahead; the h1 fragments are
            // read per chunk (32 MFMAs = 1,024 matrix-core cycles per wave and chunk, and three more waves on the SIMD, cover the round trip)
            constexpr int NCH = H1 / 16, PF = 2;
            const float* img0 = A.w2f + (size_t)wave * (16 * 256) + lane * 4;  // column tile `wave`, chunk c at + 256 c; column tile 16 + wave 65,536 floats on
            const float* ap = h1s + lr * LDA1 + 4 * lg;
            float4 pb[NCH], qb[NCH];
#pragma unroll
            for (int c = 0; c < PF; ++c) {
                pb[c] = *reinterpret_cast<const float4*>(img0 + c * 256);
                qb[c] = *reinterpret_cast<const float4*>(img0 + (size_t)16 * 16 * 256 + c * 256);
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                float4 a4[NRT];
#pragma unroll
                for (int t = 0; t < NRT; ++t) a4[t] = *reinterpret_cast<const float4*>(ap + t * RT * LDA1 + c * 16);
                if (c + PF < NCH) {
                    pb[c + PF] = *reinterpret_cast<const float4*>(img0 + (c + PF) * 256);
                    qb[c + PF] = *reinterpret_cast<const float4*>(img0 + (size_t)16 * 16 * 256 + (c + PF) * 256);
                }
                __builtin_amdgcn_sched_barrier(0);  // the requests stay ahead of the multiply (hx_act.hip)
                const float4 p4 = pb[c], q4 = qb[c];
#pragma unroll
                for (int t = 0; t < NRT; ++t) {
                    acc[t][0] = mfma16(p4.x, a4[t].x, acc[t][0]); acc[t][1] = mfma16(q4.x, a4[t].x, acc[t][1]);
                    acc[t][0] = mfma16(p4.y, a4[t].y, acc[t][0]); acc[t][1] = mfma16(q4.y, a4[t].y, acc[t][1]);
                    acc[t][0] = mfma16(p4.z, a4[t].z, acc[t][0]); acc[t][1] = mfma16(q4.z, a4[t].z, acc[t][1]);
                    acc[t][0] = mfma16(p4.w, a4[t].w, acc[t][0]); acc[t][1] = mfma16(q4.w, a4[t].w, acc[t][1]);
                }
            }
        }
        const v4f bb0 = *reinterpret_cast<const v4f*>(b2s + cw * 16 + 4 * lg), bb1 = *reinterpret_cast<const v4f*>(b2s + 256 + cw * 16 + 4 * lg);
        __syncthreads();  // every wave has read its last h1 fragment: the region takes z2
        // ---- two halves of 32 rows: z2 -> LDS, head (waves 0-7, four rows each, eight apart); waves 8, 9 draw the NEXT pass's noise ------
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                *reinterpret_cast<v4f*>(z2s + (t * RT + lr) * LDA2 + cw * 16 + 4 * lg) = acc[2 * half + t][0] + bb0;
                *reinterpret_cast<v4f*>(z2s + (t * RT + lr) * LDA2 + 256 + cw * 16 + 4 * lg) = acc[2 * half + t][1] + bb1;
            }
            __syncthreads();
            if (wave < 8) {
                const int hrow = wave + 8 * gq, lrow = HR * half + hrow;
                if (r0 + lrow < row_end) {
                    float o[OUT];
                    head16<OUT, OUT, RELU, true>(z2s + hrow * LDA2, hps, gc, slope, m.no_ln, o);
                    if (gc < 4) A.actions[(size_t)(r0 + lrow) * 4 + gc] = action_of<GAUSS>(A, o, gc, r0 + lrow, s_noise + (p & 1) * TR * 4 + lrow * 4);
                }
            } else if (half == 0) {
                draw(p + 1, wave, lane);
            }
            __syncthreads();
        }
    }
    STAMP();
    if (ENV) env_tail((KernArgs)__builtin_amdgcn_kernarg_segment_ptr(), row_begin, row_end, h1s, &s_slot0, s_wcount, bid, nwg);  // (every action of the block is written)
    STAMP();
    STAMP_FLUSH(0, (bid == 0 || bid == 200) && tid0 == 0);
    SPAN_LOG(HX_SPAN_ACT);
}

}  // namespace hxact
