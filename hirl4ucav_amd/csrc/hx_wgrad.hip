// hx_wgrad.hip — parameter gradients and optimizer steps (gfx950).
//   wgrad    all parameter gradients of one MLP block into the flat gradient buffer; on one GPU the same threads apply Adam, the Polyak step of
//            the target and refresh the W2 images                                   (loss.backward() + optimizer.step(), HIRL.py:284-288,322-330)
//   adam     torch.optim.Adam defaults over a flat buffer (sharded path, SAC, BC)   (HIRL.py:50,123; SAC/agent.py:310-325)
//   polyak   soft_update                                                             (HIRL.py:11-13)
#include <cmath>

#include "hx_update.h"

using namespace hxnn;
using namespace hxu;

namespace {

// compact kernel argument (see FwdJobC): 128 bytes per job; everything a workgroup needs before its first vector load is in its own job
struct WgJobC {
    const float* net; float* grad;
    float* ws0; float* ws1;
    float* mom; float* var; float* target; uint16_t* w2b; float* w2f;
    uint32_t cfg;  // m:10 | nslots:2 | wmode0:2 | wmode1:2 | w_kind:2 | adam.finish_actor:1 | adam.use_bc:1 | w2b_x9:1 | sac_alpha:1
                   // (sac_alpha: a SAC policy step — wstate holds alpha_state, w_given the target entropy, warm the log-alpha step size)
    int32_t rows0, rows1;
    float slope, w_given, warm, inv_batch;
    float b1, b2, eps, step_size, bc2_sqrt, tau;
    uint32_t pad_;
    const int* soft_count; float* wstate; float* losses;
    uint16_t* w2tb; uint16_t* tgt_w2b;  // bf16 update path: the transposed image of W2 and the image of the target's W2
    float* count_out;                    // job 0 only: (float)*soft_count goes here (the merged actor message of a sharded run)
};
static_assert(sizeof(WgJobC) == 176, "WgJobC layout");
struct WgArgsC {
    WgJobC job[2];
    // hx_hirl_learn_back: workgroup (kWgPerJob, 0) — one beyond the jobs' own, on a CU this launch leaves idle — assembles the next front launch's minibatch
    int has_pre, pre_batch;
    SampleDev pre;
};

// One parameter of the fused step: Adam with the gradient just produced, then (optionally) Polyak of the target and the bf16 image.
// p/m/v/t are this element's values requested at kernel entry (their latency hides under the gradient's own operand loads).
struct AdamElem {
    float p, m, v, t;
    __device__ __forceinline__ void fetch(const WgJob& J, unsigned idx) {
        p = J.p[idx];
        m = J.mom[idx];
        v = J.var[idx];
        t = J.target ? J.target[idx] : 0.0f;
    }
    __device__ __forceinline__ void apply(const WgJob& J, const WgAdam& a, unsigned idx, float g) {
        adam_update(p, m, v, g, a.b1, a.b2, a.eps, a.step_size, a.bc2_sqrt);
        J.p[idx] = p;
        J.mom[idx] = m;
        J.var[idx] = v;
        if (J.target) {
            t = polyak_update(t, p, a.tau);
            J.target[idx] = t;
        }
    }
};

// 112 workgroups per job: the critic launch (two jobs) is 224 of the 256 CUs, ONE round.  Half the rows per thread of the 56-workgroup
// partition before it: a workgroup's time is its operands' round trip plus rows-per-thread of arithmetic, and the launch is as long as
// its slowest workgroup.
constexpr int kWgTilesPerBlock = 2 * (H2 / 16);          // 64 workgroups: 16 (n) x 128 (k) of dW2; a 16 x 16 tile per wave PAIR, each half of the rows
constexpr int kWgCols = 16;                              // columns per vector / layer-1 workgroup (a quarter wave)
constexpr int kWgVecWgs = H2 / kWgCols;                  // 32 workgroups: 16 columns x 64 row groups, 512-wide vector gradients
constexpr int kWgL1Wgs = H1 / kWgCols;                   // 16 workgroups: layer-1 gradients, 16 units x 64 row groups
constexpr int kWgRG = kWide / kWgCols;                   // row groups = quarter waves: the batch rows of a column are split 64 ways
constexpr int kWgPerJob = kWgTilesPerBlock + kWgVecWgs + kWgL1Wgs;
constexpr int kWgRowChunk = 256;                         // rows whose per-row scalars are staged in LDS at a time

// fixed-order sum of the 16 row groups' partial results of one (column, item): red[group][64][kRedP]; the odd pitch keeps the 64 lanes of a
// wave on 64 different banks (pitch 20: 16 banks, every read and write of the reduction four-way conflicted)
constexpr int kRedP = 21;
// two threads per (column, item): each sums 32 of the 64 row groups (tree) — the even groups / the odd groups: one group apart is 16 banks
// apart, so the pair's reads never meet on a bank (groups 0..31 / 32..63 would: 32 groups are a multiple of 32 banks) — the even lane
// adds its neighbour's sum (one DPP move)
__device__ __forceinline__ float sum_groups(const float* p, int half) {
    constexpr int N = kWgRG / 2;
    float v[N];
#pragma unroll
    for (int g = 0; g < N; ++g) v[g] = p[(2 * g + half) * kWgCols * kRedP];
#pragma unroll
    for (int w = 1; w < N; w *= 2)
#pragma unroll
        for (int g = 0; g < N; g += 2 * w) v[g] += v[g + w];
    const float other = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v[0]), 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
    return half ? other + v[0] : v[0] + other;  // (even groups) + (odd groups) on both lanes
}

// ADAM: each thread applies the optimizer step to the gradient elements it has just produced (every parameter's gradient is
// produced by exactly one thread), so a single-GPU learn() has no separate Adam / Polyak launch.  Same adam_update on the same
// gradient values as adam_kernel: the one-call and the staged (sharded) paths stay bit-identical.
__device__ __forceinline__ void expand_wg(const WgJobC& c, WgJob& J, WgArgs& A) {
    J.net = c.net; J.grad = c.grad; J.m = mlp_of(c.cfg & 1023u);
    J.nslots = (int)((c.cfg >> 10) & 3u);
    J.wmode[0] = (int)((c.cfg >> 12) & 3u); J.wmode[1] = (int)((c.cfg >> 14) & 3u);
    J.ws[0] = carve_slot(c.ws0, c.rows0); J.ws[1] = carve_slot(c.ws1, c.rows1);
    J.rows[0] = c.rows0; J.rows[1] = c.rows1;
    J.p = const_cast<float*>(c.net); J.mom = c.mom; J.var = c.var; J.target = c.target; J.w2b = c.w2b; J.w2f = c.w2f;
    J.w2tb = c.w2tb; J.tgt_w2b = c.tgt_w2b; J.w2b_x9 = (int)((c.cfg >> 20) & 1u);
    A.slope = c.slope; A.w_kind = (int)((c.cfg >> 16) & 3u); A.w_given = c.w_given; A.warm = c.warm; A.inv_batch = c.inv_batch;
    A.soft_count = c.soft_count; A.wstate = c.wstate;
    A.ad.b1 = c.b1; A.ad.b2 = c.b2; A.ad.eps = c.eps; A.ad.step_size = c.step_size; A.ad.bc2_sqrt = c.bc2_sqrt; A.ad.tau = c.tau;
    A.ad.finish_actor = (int)((c.cfg >> 18) & 1u); A.ad.use_bc = (int)((c.cfg >> 19) & 1u);
    A.ad.losses = c.losses; A.ad.wstate = c.wstate;
    A.count_out = c.count_out;
    A.ad.alpha_state = nullptr; A.ad.target_entropy = 0.0f; A.ad.alpha_step_size = 0.0f;
    if ((c.cfg >> 21) & 1u) {  // the three fields a SAC policy step does not use otherwise carry its log-alpha step
        A.ad.alpha_state = c.wstate; A.ad.target_entropy = c.w_given; A.ad.alpha_step_size = c.warm;
        A.ad.wstate = nullptr; A.wstate = nullptr; A.w_given = 0.0f; A.warm = 0.0f;
    }
}
inline WgJobC pack_wg(const WgJob& J, const WgArgs& A) {
    WgJobC c{};
    c.net = J.net; c.grad = J.grad; c.ws0 = J.ws[0].x; c.ws1 = J.nslots > 1 ? J.ws[1].x : J.ws[0].x;
    c.mom = J.mom; c.var = J.var; c.target = J.target; c.w2b = J.w2b; c.w2f = J.w2f; c.w2tb = J.w2tb; c.tgt_w2b = J.tgt_w2b;
    c.cfg = mlp_bits(J.m) | ((uint32_t)J.nslots << 10) | ((uint32_t)J.wmode[0] << 12) | ((uint32_t)J.wmode[1] << 14) | ((uint32_t)A.w_kind << 16) |
            ((uint32_t)(A.ad.finish_actor ? 1 : 0) << 18) | ((uint32_t)(A.ad.use_bc ? 1 : 0) << 19) | ((uint32_t)(J.w2b_x9 ? 1 : 0) << 20) |
            ((uint32_t)(A.ad.alpha_state ? 1 : 0) << 21);
    c.rows0 = J.rows[0]; c.rows1 = J.nslots > 1 ? J.rows[1] : J.rows[0];
    c.slope = A.slope; c.w_given = A.w_given; c.warm = A.warm; c.inv_batch = A.inv_batch;
    c.b1 = A.ad.b1; c.b2 = A.ad.b2; c.eps = A.ad.eps; c.step_size = A.ad.step_size; c.bc2_sqrt = A.ad.bc2_sqrt; c.tau = A.ad.tau;
    c.soft_count = A.soft_count; c.wstate = const_cast<float*>(A.wstate); c.losses = A.ad.losses; c.count_out = A.count_out;
    if (A.ad.alpha_state) { c.wstate = A.ad.alpha_state; c.w_given = A.ad.target_entropy; c.warm = A.ad.alpha_step_size; }
    return c;
}

// BF16 (the bf16 update path): dW2 = dz2^T h1 on v_mfma_f32_16x16x32_bf16 — both operands rounded to bf16 as they are loaded (32 batch rows
// per MFMA instead of 4), fp32 accumulation per slot, the slot's BC weight applied to the fp32 sum; every other gradient (biases, LayerNorm,
// layer 1, the head) is fp32 arithmetic on fp32 values.  With ADAM the step also refreshes the bf16 images of W2 (forward, transposed,
// target).
// the layer-1 workgroups' inner stamps (a dozen, each an s_waitcnt lgkmcnt(0)) lengthen that path: only with -DHX_STAMPS_L1 (tools/ubench/stamps.py's
// layer-1 line); the default stamps build keeps the workgroup's life span alone, so that tools/ubench/wgrad_blocks.py compares like with like
#ifdef HX_STAMPS_L1
#define STAMP_L1() STAMP()
#else
#define STAMP_L1()
#endif
template <bool ADAM, bool RELU, bool BF16 = false>
__global__ __launch_bounds__(kWide) void wgrad_kernel(WgArgsC AC) {
    __shared__ __attribute__((aligned(16))) float lds[kWgRowChunk * XP + kWgRowChunk * 12 + kWgRG * kWgCols * kRedP];
    float* xs = lds;                          // [chunk][XP]   inputs (layer-1 job)
    float* rinfo = lds + kWgRowChunk * XP;    // [chunk][<=12] per-row scalars
    float* red = rinfo + kWgRowChunk * 12;    // [16][64][kRedP]  cross-row-group reduction

    {
        static_assert(sizeof(lds) >= (4 * kFusedSlots + 2 * kFusedBatchMax) * 4, "predraw_wg's tables fit");
        if (AC.has_pre && blockIdx.x >= kWgPerJob) {  // (eight columns beyond the jobs' own: the XCD of every other workgroup stays what it was)
            if (blockIdx.x == kWgPerJob && blockIdx.y == 0) predraw_wg(AC.pre, AC.pre_batch, lds);
            return;
        }
    }
    const int j = blockIdx.y, b = blockIdx.x;
    WgJob J;
    WgArgs A;
    expand_wg(AC.job[j], J, A);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float slope = A.slope;
    STAMP_DECL;
    STAMP();
    const float w = effective_w(A.w_kind, A.w_given, A.warm, A.inv_batch, A.soft_count, A.wstate);
    float scale[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) scale[s] = J.wmode[s] == 0 ? 1.0f : (J.wmode[s] == 1 ? 1.0f - w : w);
#ifdef HX_STAMPS
    asm volatile("" ::"v"(w));
    STAMP();
#endif
    if (A.count_out && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *A.count_out = A.soft_count ? (float)*A.soft_count : 0.0f;
    if (ADAM && A.ad.finish_actor && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {  // what adam_kernel's thread 0 does on an actor step
        if (A.ad.use_bc) {
            A.ad.losses[1] = A.ad.losses[2] * w + A.ad.losses[3] * (1.0f - w);  // HIRL.py:321
            A.ad.losses[5] = w;
            *A.ad.wstate = w;
        } else {
            A.ad.losses[1] = A.ad.losses[3];  // TD3.py:236
        }
    }
    if (ADAM && A.ad.alpha_state && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {  // what adam_kernel's thread 0 does on a SAC policy step
        const float mean_h = A.ad.losses[4];
        float la = A.ad.alpha_state[0], m = A.ad.alpha_state[1], v = A.ad.alpha_state[2];
        A.ad.losses[3] = -(la * (A.ad.target_entropy - mean_h));  // entropy_loss with the log_alpha BEFORE its step
        adam_update(la, m, v, mean_h - A.ad.target_entropy, A.ad.b1, A.ad.b2, A.ad.eps, A.ad.alpha_step_size, A.ad.bc2_sqrt);
        A.ad.alpha_state[0] = la; A.ad.alpha_state[1] = m; A.ad.alpha_state[2] = v;
        A.ad.alpha_state[3] = expf(la);  // self.alpha = self.log_alpha.exp()
        A.ad.losses[5] = A.ad.alpha_state[3];
    }

    if (b < kWgTilesPerBlock) {
        // dW2[n][k] = sum_r scale dz2[r][n] h1[r][k].  Tile 16 (n) x 16 (k) per wave PAIR (w, w + 8): each wave reduces over half of the
        // batch rows, 16 per MFMA group; all operands of a 64-row chunk (16 + 16 dwords per lane) are requested before the first MFMA.
        // The pair then swaps half of its accumulator through LDS: wave w finishes (and steps) elements 0, 1 of every lane, wave w + 8
        // elements 2, 3 — two parameters per lane.
        const int n0 = (b >> 1) * 16, k0 = (b & 1) * (H1 / 2) + (wave & 7) * 16, half = wave >> 3;
        const int r = lane & 15, g = lane >> 4;
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        #pragma unroll
        for (int s = 0; s < 2; ++s) if (s < J.nslots) {  // compile-time slot index: J lives in registers, not in scratch
            const float sc = scale[s];
            const float* dz = J.ws[s].dz2;  // uniform bases + 32-bit lane offsets: no 64-bit address arithmetic per load
            const float* h1 = J.ws[s].h1;
            const unsigned dzo = (unsigned)(n0 + r), h1o = (unsigned)(k0 + r);
            const int rows = J.rows[s];
            const int hr = ((rows + 7) >> 3) << 2;  // rows per half, a multiple of the MFMA's 4
            const int rbeg = half * hr, rend = min(rows, rbeg + hr);
            if constexpr (BF16) {
                // MFMA m reduces over rows c0 + 32 m .. + 31: lane (r, g) feeds rows + 8 g .. + 7 of column n0 + r (A) / k0 + r (B)
                v4f accs = {0.f, 0.f, 0.f, 0.f};
                for (int c0 = rbeg; c0 < rend; c0 += 64) {
                    float av[2][8], hv[2][8];
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int row = c0 + 32 * m + 8 * g + j;
                            const unsigned rc = (unsigned)(row < rows ? row : rows - 1);  // unconditional loads (clamped)
                            av[m][j] = dz[rc * (unsigned)H2 + dzo];
                            hv[m][j] = h1[rc * (unsigned)H1 + h1o];
                        }
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (c0 + 32 * m + 8 * g + j >= rend) av[m][j] = 0.0f;  // rows past this half's end contribute nothing
                        accs = mfma16_bf16(pack8_bf16(av[m]), pack8_bf16(hv[m]), accs);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_fmaf(sc, accs[q], acc[q]);
            } else {
                for (int c0 = rbeg; c0 < rend; c0 += 64) {
                    float av[16], hv[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int row = c0 + 4 * i + g;  // MFMA i reduces over rows c0+4i .. c0+4i+3 (one per lane group)
                        const unsigned rc = (unsigned)(row < rows ? row : rows - 1);  // unconditional loads (clamped); rows past the end get scale 0
                        av[i] = dz[rc * (unsigned)H2 + dzo];
                        hv[i] = h1[rc * (unsigned)H1 + h1o];
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc = mfma16(av[i] * (c0 + 4 * i + g < rend ? sc : 0.0f), hv[i], acc);
                }
            }
        }
        STAMP();
        // ADAM: this lane's two parameters are requested only now — the operand registers of the reduction above are dead; the round trip
        // (L2-resident: touched once per learn()) hides under the pair's exchange
        const int q0 = 2 * half;  // this wave finishes elements q0, q0 + 1
        AdamElem ae[2];
        if (ADAM) {
#pragma unroll
            for (int q = 0; q < 2; ++q) ae[q].fetch(J, J.m.W2() + (n0 + 4 * g + q0 + q) * H1 + k0 + r);
        }
        float2* xch = reinterpret_cast<float2*>(red);  // [8 pairs][2 halves][64 lanes]
        xch[((wave & 7) * 2 + half) * 64 + lane] = half ? make_float2(acc[0], acc[1]) : make_float2(acc[2], acc[3]);
        __syncthreads();
        const float2 got = xch[((wave & 7) * 2 + (half ^ 1)) * 64 + lane];
        // (first half of the rows) + (second half), whoever adds them
        const float fin[2] = {half ? got.x + acc[2] : acc[0] + got.x, half ? got.y + acc[3] : acc[1] + got.y};
        float* out = J.grad + J.m.W2();
#pragma unroll
        for (int q = 0; q < 2; ++q) out[(unsigned)((n0 + 4 * g + q0 + q) * H1 + k0 + r)] = fin[q];
        if (ADAM) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int idx = J.m.W2() + (n0 + 4 * g + q0 + q) * H1 + k0 + r;
                ae[q].apply(J, A.ad, idx, fin[q]);
                if (J.w2b) {
                    const uint32_t ix = w2_image_index((uint32_t)(n0 + 4 * g + q0 + q), (uint32_t)(k0 + r));
                    uint16_t hi, mid, lo;
                    split3_bf16(ae[q].p, hi, mid, lo);  // hi = the weight rounded to bf16 (the plain image); mid | lo only for the x9 images
                    J.w2b[ix] = hi;
                    if (J.w2b_x9) { J.w2b[kImgElems + ix] = mid; J.w2b[2 * kImgElems + ix] = lo; }
                }
                if (J.w2f) J.w2f[w2f_image_index((uint32_t)(n0 + 4 * g + q0 + q), (uint32_t)(k0 + r))] = ae[q].p;
                if (J.w2tb) {
                    const __bf16 bv = (__bf16)ae[q].p;
                    J.w2tb[w2t_image_index((uint32_t)(k0 + r), (uint32_t)(n0 + 4 * g + q0 + q))] = __builtin_bit_cast(uint16_t, bv);
                }
                if (J.target && J.tgt_w2b) {
                    const __bf16 bv = (__bf16)ae[q].t;
                    J.tgt_w2b[w2_image_index((uint32_t)(n0 + 4 * g + q0 + q), (uint32_t)(k0 + r))] = __builtin_bit_cast(uint16_t, bv);
                }
            }
        }
        STAMP();
        STAMP_FLUSH(32, blockIdx.x == 0 && blockIdx.y == 0 && tid == 0);
        SPAN_LOG(HX_SPAN_WGRAD);
        return;
    }
    // Vector and layer-1 workgroups: 16 columns x 64 row groups (a quarter wave per row group): two batch rows per thread at B = 128.
    // (64 columns x 16 row groups put eight rows on every thread, 32 x 32 four: these workgroups, not the MFMA tiles, set the launch's duration.)
    const int cl = lane & (kWgCols - 1);           // column inside the workgroup
    const int rg = tid / kWgCols;                  // row group: rows rg, rg + 64, ...
    // after the reduction: thread pair -> (item, column); each thread of the pair sums half of the row groups
    const int ohalf = tid & 1, oitem = (tid >> 1) / kWgCols, ocol = (tid >> 1) % kWgCols;
    if (b < kWgTilesPerBlock + kWgVecWgs) {
        // column n: db2, dg2, dbe2, dW3[j][n] (+ db3 by the first workgroup)
        constexpr int RP = 12;  // rinfo pitch: mean, rstd, dout[0..7], pad
        const int vb = b - kWgTilesPerBlock;
        const int n = vb * kWgCols + cl;
        const float g2 = J.net[J.m.g2() + n], be2 = J.net[J.m.be2() + n];
        float w3[OW];
#pragma unroll
        for (int jj = 0; jj < OW; ++jj) w3[jj] = jj < J.m.out ? J.net[J.m.W3() + jj * H2 + n] : 0.0f;
        float db2 = 0.f, dg = 0.f, dbe = 0.f, dw3[OW] = {}, db3 = 0.f;
        // ADAM: the parameter this thread will step after the reduction (item oitem of column ocol; the last items: b3), requested now
        const int on = vb * kWgCols + ocol;
        const unsigned vidx = (unsigned)(oitem == 0 ? J.m.b2() + on : oitem == 1 ? J.m.g2() + on : oitem == 2 ? J.m.be2() + on : J.m.W3() + (oitem - 3) * H2 + on);
        const bool vlive = oitem < 3 + J.m.out;
        const bool b3live = vb == 0 && oitem == 3 + OW && ocol < J.m.out;  // (item 3 + OW of the reduction tile carries db3)
        const unsigned b3idx = (unsigned)(J.m.b3() + (ocol < J.m.out ? ocol : 0));
        AdamElem vae;
        bool vae_pending = ADAM && (vlive || b3live) && ohalf == 0;  // its operands are needed last: requested behind the first chunk's loads
        #pragma unroll
        for (int s = 0; s < 2; ++s) if (s < J.nslots) {  // compile-time slot index: J lives in registers, not in scratch
            const Slot& S = J.ws[s];
            const float sc = scale[s];
            for (int c0 = 0; c0 < J.rows[s]; c0 += kWgRowChunk) {
                const int nr = min(kWgRowChunk, J.rows[s] - c0);
                // this thread's first 4 rows are requested before the per-row scalars are staged: one round trip, not two
                float zv[4], dv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) if (kWgRG * i < nr) {  // (scalar: at B = 128 row blocks 2, 3 are past the end for every thread)
                    const int r = rg + kWgRG * i;
                    const unsigned o = (unsigned)(c0 + (r < nr ? r : nr - 1)) * (unsigned)H2 + (unsigned)n;  // unconditional inside a block, clamped
                    zv[i] = S.z2[o];
                    dv[i] = S.dz2[o];
                }
                // the per-row scalars are requested BEFORE the barrier that frees the LDS tile: one round trip with the loads above
                static_assert(kWgRowChunk <= kWide, "staging: one row per thread");
                // (whole waves without a row skip theirs behind a scalar branch: 2 of the 16 waves have rows at B = 128, and the address
                //  pipeline takes 16 lanes per clock whatever they ask for)
                const int er = tid < nr ? tid : 0;
                v2f st2v;  // (unset in a skipping wave, which never stores it: see HeadImage::fetch)
                v4f d4, d5;
                if (__builtin_amdgcn_readfirstlane(tid) < nr) {
                    st2v = *reinterpret_cast<const v2f*>(S.st2 + (size_t)(c0 + er) * 2);
                    d4 = *reinterpret_cast<const v4f*>(S.dout + (size_t)(c0 + er) * OW);
                    d5 = *reinterpret_cast<const v4f*>(S.dout + (size_t)(c0 + er) * OW + 4);
                }
                if (vae_pending) {
                    vae.fetch(J, b3live ? b3idx : vidx);
                    vae_pending = false;
                }
                __syncthreads();
                if (tid < nr) {
                    float4* r4 = reinterpret_cast<float4*>(rinfo + tid * RP);
                    r4[0] = make_float4(st2v[0], st2v[1], d4[0], d4[1]);
                    r4[1] = make_float4(d4[2], d4[3], d5[0], d5[1]);
                    r4[2] = make_float4(d5[2], d5[3], 0.0f, 0.0f);
                }
                __syncthreads();
                for (int rb0 = 0; rb0 < nr; rb0 += kWgRG * 4) {  // 4 rows per thread per block, all loads in flight together
                    const int rb = rg + rb0;
                    if (rb0 != 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) if (rb0 + kWgRG * i < nr) {
                            const int r = rb + kWgRG * i;
                            const unsigned o = (unsigned)(c0 + (r < nr ? r : nr - 1)) * (unsigned)H2 + (unsigned)n;
                            zv[i] = S.z2[o];
                            dv[i] = S.dz2[o];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = rb + kWgRG * i;
                        if (r < nr) {
                            // the row's scalars in three 16-byte LDS reads (one address per half wave: broadcast)
                            float ri[RP];
                            {
                                const float4* r4 = reinterpret_cast<const float4*>(rinfo + r * RP);
                                const float4 a = r4[0], b4 = r4[1], c4 = r4[2];
                                ri[0] = a.x; ri[1] = a.y; ri[2] = a.z; ri[3] = a.w; ri[4] = b4.x; ri[5] = b4.y; ri[6] = b4.z; ri[7] = b4.w;
                                ri[8] = c4.x; ri[9] = c4.y; ri[10] = c4.z; ri[11] = c4.w;
                            }
                            const float xh = (zv[i] - ri[0]) * ri[1];
                            const float y = g2 * xh + be2;
                            float dh2 = (ri[2] * w3[0] + ri[3] * w3[1]) + (ri[4] * w3[2] + ri[5] * w3[3]);
                            if (J.m.out > 4) dh2 += (ri[6] * w3[4] + ri[7] * w3[5]) + (ri[8] * w3[6] + ri[9] * w3[7]);
                            const float dy = act_bwd<RELU>(dh2, y, slope);
                            const float h2 = act_f<RELU>(y, slope);
                            db2 += sc * dv[i];
                            dbe += sc * dy;
                            dg += sc * dy * xh;
#pragma unroll
                            for (int jj = 0; jj < OW; ++jj) dw3[jj] += sc * ri[2 + jj] * h2;
                            // db3[j] = sum_r dout[r][j]: columns 0..out-1 of the first column block, over this row group's rows
                            if (vb == 0 && cl < J.m.out) db3 += sc * rinfo[r * RP + 2 + cl];
                        }
                    }
                }
            }
        }
        STAMP();
        float* my = red + (rg * kWgCols + cl) * kRedP;
        my[0] = db2; my[1] = dg; my[2] = dbe;
#pragma unroll
        for (int jj = 0; jj < OW; ++jj) my[3 + jj] = dw3[jj];
        my[3 + OW] = db3;
        __syncthreads();
        if (vlive || b3live) {  // thread pair -> (item, column): 32 partial sums each, the even thread finishes
            float v = sum_groups(red + ocol * kRedP + oitem, ohalf);
            if ((oitem == 1 || oitem == 2) && J.m.no_ln) v = 0.0f;
            const unsigned idx = b3live ? b3idx : vidx;
            if (ohalf == 0) {
                J.grad[idx] = v;
                if (ADAM) vae.apply(J, A.ad, idx, v);
            }
        }
        STAMP();
        STAMP_FLUSH(40, b == kWgTilesPerBlock && j == 0 && tid == 0);
        SPAN_LOG(HX_SPAN_WGRAD);
        return;
    }
    // layer 1: hidden unit k; dz1 = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat)) with the row means taken from the
    // per-workgroup partial sums bwd_l2 left in lnp -> no cross-column work here.
    {
        const int kb = (b - kWgTilesPerBlock - kWgVecWgs) * kWgCols;
        const int k = kb + cl;
        const int in = J.m.in;
        const float g1 = J.net[J.m.g1() + k], be1 = J.net[J.m.be1() + k];
        STAMP_L1();
        float db1 = 0.f, dg = 0.f, dbe = 0.f, dw1[17];
#pragma unroll
        for (int i = 0; i < 17; ++i) dw1[i] = 0.f;
        // ADAM: the parameter this thread will step (item oitem of unit kb + ocol; 3 + in <= 20 items x 16 units, two threads each), requested now
        const int ok = kb + ocol;
        const unsigned lidx = (unsigned)(oitem == 0 ? J.m.b1() + ok : oitem == 1 ? J.m.g1() + ok : oitem == 2 ? J.m.be1() + ok : J.m.W1() + ok * in + (oitem - 3));
        const bool llive = oitem < 3 + in;
        AdamElem lae;
        bool lae_pending = ADAM && llive && ohalf == 0;  // needed last: requested behind the first chunk's loads
        #pragma unroll
        for (int s = 0; s < 2; ++s) if (s < J.nslots) {  // compile-time slot index: J lives in registers, not in scratch
            const Slot& S = J.ws[s];
            const float sc = scale[s];
            for (int c0 = 0; c0 < J.rows[s]; c0 += kWgRowChunk) {
                const int nr = min(kWgRowChunk, J.rows[s] - c0);
                float zv[4], dv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) if (kWgRG * i < nr) {  // (scalar: at B = 128 row blocks 2, 3 are past the end for every thread)
                    const int r = rg + kWgRG * i;
                    const unsigned o = (unsigned)(c0 + (r < nr ? r : nr - 1)) * (unsigned)H1 + (unsigned)k;  // unconditional inside a block, clamped
                    zv[i] = S.z1[o];
                    dv[i] = S.dh1[o];
                }
                STAMP_L1();
                // the chunk's shared operands are requested BEFORE the barrier that frees the LDS tiles: one round trip with the loads above
                static_assert(kWgRowChunk * XP <= 5 * kWide && kWgRowChunk <= kWide && kColWgB == 8, "staging: five words + one row per thread");
                // (whole waves past the end of a tile skip their loads behind a scalar branch: at B = 128 the input tile is 2.5 of the 5
                //  passes and 2 of the 16 waves have a row; the address pipeline takes 16 lanes per clock whatever they ask for)
                const int w0 = __builtin_amdgcn_readfirstlane(tid);
                float xst[5];
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const int e = tid + q * kWide;
                    if (w0 + q * kWide < nr * XP) xst[q] = S.x[(size_t)c0 * XP + (e < nr * XP ? e : 0)];
                }
                STAMP_L1();
                const int er = tid < nr ? tid : 0;
                v2f st1v;  // (unset in a skipping wave, which never stores it)
                v4f l0, l1, l2, l3;
                if (w0 < nr) {
                    st1v = *reinterpret_cast<const v2f*>(S.st1 + (size_t)(c0 + er) * 2);
                    const v4f* lp4 = reinterpret_cast<const v4f*>(S.lnp + (size_t)(c0 + er) * (2 * kColWgB));  // [8 column workgroups][2]
                    l0 = lp4[0]; l1 = lp4[1]; l2 = lp4[2]; l3 = lp4[3];
                }
                STAMP_L1();
                if (lae_pending) {
                    lae.fetch(J, lidx);
                    lae_pending = false;
                }
                STAMP_L1();
                __syncthreads();
                STAMP_L1();
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const int e = tid + q * kWide;
                    if (e < nr * XP) xs[e] = xst[q];
                }
                if (tid < nr) {  // lnp_sum's fixed-order tree over the eight partials of each of the two row sums
                    const float s1 = ((l0[0] + l0[2]) + (l1[0] + l1[2])) + ((l2[0] + l2[2]) + (l3[0] + l3[2]));
                    const float s2 = ((l0[1] + l0[3]) + (l1[1] + l1[3])) + ((l2[1] + l2[3]) + (l3[1] + l3[3]));
                    *reinterpret_cast<float4*>(rinfo + tid * 8) = make_float4(st1v[0], st1v[1], s1 * (1.0f / H1), s2 * (1.0f / H1));
                }
                __syncthreads();
                STAMP_L1();
                for (int rb0 = 0; rb0 < nr; rb0 += kWgRG * 4) {
                    const int rb = rg + rb0;
                    if (rb0 != 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) if (rb0 + kWgRG * i < nr) {
                            const int r = rb + kWgRG * i;
                            const unsigned o = (unsigned)(c0 + (r < nr ? r : nr - 1)) * (unsigned)H1 + (unsigned)k;
                            zv[i] = S.z1[o];
                            dv[i] = S.dh1[o];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = rb + kWgRG * i;
                        if (r < nr) {
                            const float4 ri = *reinterpret_cast<const float4*>(rinfo + r * 8);  // mean, rstd, the two LN1-backward row means
                            const float xh = (zv[i] - ri.x) * ri.y;
                            const float dy = act_bwd<RELU>(dv[i], g1 * xh + be1, slope);
                            const float dz = sc * (ri.y * (dy * g1 - ri.z - xh * ri.w));
                            db1 += dz;
                            dbe += sc * dy;
                            dg += sc * dy * xh;
                            // the input row in five 16-byte LDS reads (one address per half wave: broadcast), not seventeen 4-byte ones
                            const float4* xr4 = reinterpret_cast<const float4*>(xs + r * XP);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float4 x4 = xr4[q];
                                dw1[4 * q] += dz * x4.x; dw1[4 * q + 1] += dz * x4.y; dw1[4 * q + 2] += dz * x4.z; dw1[4 * q + 3] += dz * x4.w;
                            }
                            dw1[16] += dz * xs[r * XP + 16];
                        }
                    }
                }
            }
        }
        STAMP_L1();
        float* my = red + (rg * kWgCols + cl) * kRedP;
        my[0] = db1; my[1] = dg; my[2] = dbe;
#pragma unroll
        for (int i = 0; i < 17; ++i) my[3 + i] = dw1[i];
        STAMP_L1();
        __syncthreads();
        STAMP_L1();
        if (llive) {
            float v = sum_groups(red + ocol * kRedP + oitem, ohalf);
            if ((oitem == 1 || oitem == 2) && J.m.no_ln) v = 0.0f;
#ifdef HX_STAMPS
            asm volatile("" ::"v"(v));
            STAMP_L1();
#endif
            if (ohalf == 0) {
                J.grad[lidx] = v;
                if (ADAM) lae.apply(J, A.ad, lidx, v);
            }
        }
        STAMP_L1();
        STAMP_FLUSH(48, b == kWgTilesPerBlock + kWgVecWgs && j == 0 && tid == 0);
        SPAN_LOG(HX_SPAN_WGRAD);
    }
}

// the BC weight of this call (HIRL.py:299-308); countf: the soft count as a float word of the all-reduced message
__device__ __forceinline__ float adam_w(const AdamArgs& A) {
    if (A.countf && A.w_kind == 1) {
        const float w = *A.countf * A.inv_batch + A.warm;
        return w > 1.0f ? 1.0f : w;
    }
    return effective_w(A.w_kind, A.w_given, A.warm, A.inv_batch, A.soft_count, A.wstate);
}

__global__ __launch_bounds__(kThreads) void adam_kernel(AdamArgs A) {
    if (A.guard && __hip_atomic_load(A.guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return;  // failed exchange: fail-stop
    if (A.alpha_state && blockIdx.x == 0 && threadIdx.x == 0) {
        const float mean_h = A.losses[4];
        float la = A.alpha_state[0], m = A.alpha_state[1], v = A.alpha_state[2];
        A.losses[3] = -(la * (A.target_entropy - mean_h));  // entropy_loss with the log_alpha BEFORE its step
        adam_update(la, m, v, mean_h - A.target_entropy, A.b1, A.b2, A.eps, A.alpha_step_size, A.bc2_sqrt);
        A.alpha_state[0] = la; A.alpha_state[1] = m; A.alpha_state[2] = v;
        A.alpha_state[3] = expf(la);  // self.alpha = self.log_alpha.exp()
        A.losses[5] = A.alpha_state[3];
    }
    const float wmix = A.g2 ? adam_w(A) : 0.0f;  // (read before thread 0 of block 0 may store the new weight: same value either way)
    if (A.finish_actor && blockIdx.x == 0 && threadIdx.x == 0) {
        if (A.use_bc) {
            const float w = adam_w(A);
            A.losses[1] = A.losses[2] * w + A.losses[3] * (1.0f - w);  // HIRL.py:321
            A.losses[5] = w;
            *A.wstate = w;
        } else {
            A.losses[1] = A.losses[3];  // TD3.py:236
        }
    }
    const int i = (blockIdx.x * kThreads + threadIdx.x) * 4;
    if (i >= A.n) return;
    if (i + 4 <= A.n) {  // the flat buffers are 16-B aligned and a multiple of 4 floats long: one 16-B access per array
        float4 p4 = *reinterpret_cast<const float4*>(A.p + i), g4 = *reinterpret_cast<const float4*>(A.g + i);
        float4 m4 = *reinterpret_cast<const float4*>(A.m + i), v4 = *reinterpret_cast<const float4*>(A.v + i);
        if (A.g2) {  // g = w dL_bc + (1 - w) dL_rl  (HIRL.py:321), combined AFTER the exchange
            const float4 b4 = *reinterpret_cast<const float4*>(A.g2 + i);
            g4.x = wmix * b4.x + (1.0f - wmix) * g4.x;
            g4.y = wmix * b4.y + (1.0f - wmix) * g4.y;
            g4.z = wmix * b4.z + (1.0f - wmix) * g4.z;
            g4.w = wmix * b4.w + (1.0f - wmix) * g4.w;
        }
        adam_update(p4.x, m4.x, v4.x, g4.x * A.gscale, A.b1, A.b2, A.eps, A.step_size, A.bc2_sqrt);
        adam_update(p4.y, m4.y, v4.y, g4.y * A.gscale, A.b1, A.b2, A.eps, A.step_size, A.bc2_sqrt);
        adam_update(p4.z, m4.z, v4.z, g4.z * A.gscale, A.b1, A.b2, A.eps, A.step_size, A.bc2_sqrt);
        adam_update(p4.w, m4.w, v4.w, g4.w * A.gscale, A.b1, A.b2, A.eps, A.step_size, A.bc2_sqrt);
        *reinterpret_cast<float4*>(A.p + i) = p4;
        *reinterpret_cast<float4*>(A.m + i) = m4;
        *reinterpret_cast<float4*>(A.v + i) = v4;
        if (A.w2b && i >= A.w2_lo && i < A.w2_lo + H2 * H1) {  // W2 starts at a multiple of 4 floats: the float4 is inside or outside
            typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
            const v4bf r = {(__bf16)p4.x, (__bf16)p4.y, (__bf16)p4.z, (__bf16)p4.w};
            const uint32_t e = (uint32_t)(i - A.w2_lo);  // four consecutive k of one column: adjacent in the image too
            const uint32_t ix = w2_image_index(e / H1, e % H1);
            *reinterpret_cast<uint2*>(A.w2b + ix) = __builtin_bit_cast(uint2, r);
            if (A.w2b_x9) {  // the other two parts of the exact split (split3_bf16)
                const float pv[4] = {p4.x, p4.y, p4.z, p4.w};
                uint16_t md[4], lw[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { uint16_t h; split3_bf16(pv[q], h, md[q], lw[q]); }
                *reinterpret_cast<uint2*>(A.w2b + kImgElems + ix) = make_uint2((uint32_t)md[0] | ((uint32_t)md[1] << 16), (uint32_t)md[2] | ((uint32_t)md[3] << 16));
                *reinterpret_cast<uint2*>(A.w2b + 2 * kImgElems + ix) = make_uint2((uint32_t)lw[0] | ((uint32_t)lw[1] << 16), (uint32_t)lw[2] | ((uint32_t)lw[3] << 16));
            }
        }
        if (A.w2f && i >= A.w2_lo && i < A.w2_lo + H2 * H1) {
            const uint32_t e = (uint32_t)(i - A.w2_lo);
            *reinterpret_cast<float4*>(A.w2f + w2f_image_index(e / H1, e % H1)) = p4;
        }
        float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (A.target) {
            t4 = *reinterpret_cast<const float4*>(A.target + i);
            t4.x = polyak_update(t4.x, p4.x, A.tau);
            t4.y = polyak_update(t4.y, p4.y, A.tau);
            t4.z = polyak_update(t4.z, p4.z, A.tau);
            t4.w = polyak_update(t4.w, p4.w, A.tau);
            *reinterpret_cast<float4*>(A.target + i) = t4;
        }
        // bf16 update path: the images of every W2 this step changes follow it (forward, transposed, and the target's when it moved)
        for (int sg = 0; sg < A.nseg; ++sg) {
            if (i < A.seg_lo[sg] || i >= A.seg_lo[sg] + H2 * H1) continue;
            typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
            const uint32_t e = (uint32_t)(i - A.seg_lo[sg]), col = e / H1, k = e % H1;  // four consecutive k of one column
            const v4bf r = {(__bf16)p4.x, (__bf16)p4.y, (__bf16)p4.z, (__bf16)p4.w};
            if (A.seg_w2b[sg]) *reinterpret_cast<uint2*>(A.seg_w2b[sg] + w2_image_index(col, k)) = __builtin_bit_cast(uint2, r);
            if (A.seg_w2tb[sg]) {
                const uint2 q = __builtin_bit_cast(uint2, r);
                uint16_t* tb = A.seg_w2tb[sg];
                tb[w2t_image_index(k, col)] = (uint16_t)(q.x & 0xFFFFu);
                tb[w2t_image_index(k + 1, col)] = (uint16_t)(q.x >> 16);
                tb[w2t_image_index(k + 2, col)] = (uint16_t)(q.y & 0xFFFFu);
                tb[w2t_image_index(k + 3, col)] = (uint16_t)(q.y >> 16);
            }
            if (A.target && A.seg_tgt_w2b[sg]) {
                const v4bf rt = {(__bf16)t4.x, (__bf16)t4.y, (__bf16)t4.z, (__bf16)t4.w};
                *reinterpret_cast<uint2*>(A.seg_tgt_w2b[sg] + w2_image_index(col, k)) = __builtin_bit_cast(uint2, rt);
            }
        }
        return;
    }
    for (int c = 0; c < A.n - i; ++c) {  // ragged tail
        float pv = A.p[i + c], mv = A.m[i + c], vv = A.v[i + c];
        const float gv = A.g2 ? wmix * A.g2[i + c] + (1.0f - wmix) * A.g[i + c] : A.g[i + c];
        adam_update(pv, mv, vv, gv * A.gscale, A.b1, A.b2, A.eps, A.step_size, A.bc2_sqrt);
        A.p[i + c] = pv; A.m[i + c] = mv; A.v[i + c] = vv;
        if (A.target) A.target[i + c] = polyak_update(A.target[i + c], pv, A.tau);
    }
}

// up to two (target, source) segments in one launch: blocks [0, nb1) walk the first, the rest the second
// guard: nullptr, or a word that must be 0 for the step to happen (HxNets.xchg_status: after a failed exchange no target moves either)
__global__ __launch_bounds__(kThreads) void polyak_kernel(float* target, const float* source, int n, float tau, float* target2 = nullptr,
                                                          const float* source2 = nullptr, int n2 = 0, const uint32_t* guard = nullptr) {
    if (guard && *guard != 0u) return;
    const int nb1 = (n / 4 + kThreads) / kThreads;
    int blk = blockIdx.x;
    if (blk >= nb1) {
        blk -= nb1; target = target2; source = source2; n = n2;
    }
    const int i = (blk * kThreads + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n) {
        float4 t4 = *reinterpret_cast<const float4*>(target + i);
        const float4 s4 = *reinterpret_cast<const float4*>(source + i);
        t4.x = polyak_update(t4.x, s4.x, tau);
        t4.y = polyak_update(t4.y, s4.y, tau);
        t4.z = polyak_update(t4.z, s4.z, tau);
        t4.w = polyak_update(t4.w, s4.w, tau);
        *reinterpret_cast<float4*>(target + i) = t4;
        return;
    }
    for (int c = 0; c < n - i; ++c) target[i + c] = polyak_update(target[i + c], source[i + c], tau);
}

}  // namespace

namespace hxu {

void launch_wg(const WgArgs& W, bool adam, hipStream_t st) {
    WgArgsC C{};
    for (int j = 0; j < W.njobs; ++j) C.job[j] = pack_wg(W.job[j], W);
    const bool pre = W.predraw != nullptr;
    C.has_pre = pre ? 1 : 0;
    if (pre) { C.pre = *W.predraw; C.pre_batch = W.predraw_batch; }
    const dim3 grid(kWgPerJob + (pre ? 8 : 0), W.njobs);
    const bool relu = W.slope == 0.0f;
#define HX_WG(ADAM_, RELU_, BF16_) hipLaunchKernelGGL((wgrad_kernel<ADAM_, RELU_, BF16_>), grid, dim3(kWide), 0, st, C)
    static const int dbg_off = getenv("HX_DBG_BF16_OFF") ? atoi(getenv("HX_DBG_BF16_OFF")) : 0;
    if (W.bf16 && !(dbg_off & 4)) {
        if (adam) { if (relu) HX_WG(true, true, true); else HX_WG(true, false, true); }
        else { if (relu) HX_WG(false, true, true); else HX_WG(false, false, true); }
    } else {
        if (adam) { if (relu) HX_WG(true, true, false); else HX_WG(true, false, false); }
        else { if (relu) HX_WG(false, true, false); else HX_WG(false, false, false); }
    }
#undef HX_WG
}
void launch_adam(const AdamArgs& A, hipStream_t st) {
    hipLaunchKernelGGL(adam_kernel, dim3((A.n / 4 + kThreads) / kThreads), dim3(kThreads), 0, st, A);
}
void launch_polyak(float* target, const float* source, int n, float tau, float* target2, const float* source2, int n2, hipStream_t st, const uint32_t* guard) {
    const int nb = (n / 4 + kThreads) / kThreads + (target2 ? (n2 / 4 + kThreads) / kThreads : 0);
    hipLaunchKernelGGL(polyak_kernel, dim3(nb), dim3(kThreads), 0, st, target, source, n, tau, target2, source2, n2, guard);
}

}  // namespace hxu

extern "C" {

/* Adam step over a flat buffer (torch.optim.Adam defaults; step = 1-based step count; grad is multiplied by
 * grad_scale first — 1/world_size after a SUM all-reduce).  which: 0 critic, 1 actor (also finishes actor_loss /
 * bc_weight bookkeeping: w_kind 0 given, 1 estimate from soft_count, 2 reuse stored). */
static int adam_impl(const HxNets* N, const HxHyper* Hy, int32_t which, int32_t step, float grad_scale, int32_t w_kind, float w_given,
                     float warm, int32_t batch, const float* msg, void* stream) {
    const bool polyak = (which & 16) != 0;  // + 16: soft_update of this network's target in the same launch
    which &= 15;
    HX_REQUIRE(N && Hy && step >= 1 && which >= 0 && which <= 2, "hx_adam: bad arguments");
    const double b1 = 0.9, b2 = 0.999;
    const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
    AdamArgs A{};
    A.n = which == 0 ? 2 * kQ.padded() : kActor.size();
    A.p = which == 0 ? N->critic : N->actor;
    A.g = which == 0 ? N->grad_critic : N->grad_actor;
    A.m = which == 0 ? N->m_critic : N->m_actor;
    A.v = which == 0 ? N->v_critic : N->v_actor;
    A.b1 = (float)b1; A.b2 = (float)b2; A.eps = 1e-8f;
    A.step_size = (float)((which == 0 ? Hy->lr_critic : Hy->lr_actor) / bc1);
    A.bc2_sqrt = (float)sqrt(bc2);
    A.gscale = grad_scale;
    A.finish_actor = which == 1;  // which == 2: the actor's Adam step alone (BC pre-training)
    A.w_kind = w_kind; A.w_given = w_given; A.warm = warm; A.inv_batch = 1.0f / (batch > 0 ? batch : 1);
    A.soft_count = N->soft_count; A.wstate = N->wstate; A.losses = N->losses; A.use_bc = Hy->use_bc;
    if (polyak) {
        A.target = which == 0 ? N->target_critic : N->target_actor;
        A.tau = Hy->tau;
    }
    if (which != 0 && N->actor_w2_f32i) {
        A.w2f = N->actor_w2_f32i;
        A.w2_lo = kActor.W2();
    }
    if (which != 0 && N->actor_w2_bf16 && !N->w2_bf16_all) {
        A.w2b = N->actor_w2_bf16;
        A.w2_lo = kActor.W2();
    }
    if (which != 0 && N->actor_w2_x9) {
        HX_REQUIRE(!N->actor_w2_bf16 && !N->w2_bf16_all, "hx_adam: actor_w2_x9 excludes the plain bf16 images (one acting format at a time)");
        A.w2b = N->actor_w2_x9;
        A.w2b_x9 = 1;
        A.w2_lo = kActor.W2();
    }
    if (uint16_t* im = N->w2_bf16_all) {  // bf16 update path: the images of every W2 this step changes (and of the target it moves) follow it
        HX_REQUIRE(!N->actor_w2_bf16 || N->actor_w2_bf16 == im + IM_ACTOR * kImgElems, "hx_adam: with w2_bf16_all set, actor_w2_bf16 must be NULL or its first image");
        if (which == 0) {
            A.nseg = 2;
            for (int h = 0; h < 2; ++h) {
                A.seg_lo[h] = h * kQ.padded() + kQ.W2();
                A.seg_w2b[h] = im + (IM_C1 + h) * kImgElems;
                A.seg_w2tb[h] = im + (IM_C1_T + h) * kImgElems;
                A.seg_tgt_w2b[h] = im + (IM_TC1 + h) * kImgElems;
            }
        } else {
            A.nseg = 1;
            A.seg_lo[0] = kActor.W2();
            A.seg_w2b[0] = im + IM_ACTOR * kImgElems;
            A.seg_w2tb[0] = im + IM_ACTOR_T * kImgElems;
            A.seg_tgt_w2b[0] = im + IM_TA * kImgElems;
        }
    }
    A.guard = N->xchg_status;
    if (msg) {  // merged actor message: [dL_rl | dL_bc | count ...]
        HX_REQUIRE(which == 1 && (reinterpret_cast<uintptr_t>(msg) & 15u) == 0, "hx_adam_mixed: actor step only, 16-byte aligned message");
        A.g = msg;
        if (Hy->use_bc) {
            A.g2 = msg + kActor.padded();
            A.countf = msg + 2 * kActor.padded();
        }
    }
    HX_REQUIRE((((uintptr_t)A.p | (uintptr_t)A.g | (uintptr_t)A.m | (uintptr_t)A.v) & 15u) == 0, "hx_adam: buffers must be 16-byte aligned");
    launch_adam(A, (hipStream_t)stream);
    HX_CHECK_LAUNCH("hx_adam");
    return 0;
}
int hx_adam(const HxNets* N, const HxHyper* Hy, int32_t which, int32_t step, float grad_scale, int32_t w_kind, float w_given,
            float warm, int32_t batch, void* stream) {
    return adam_impl(N, Hy, which, step, grad_scale, w_kind, w_given, warm, batch, nullptr, stream);
}
/* The actor's optimizer step from the MERGED message of a sharded run (one collective for the whole actor phase):
 * msg = [dL_rl (hx_actor_param_count() floats, padded to 4) | dL_bc (same) | soft count as a float | ...], already summed over the
 * ranks; w = count / batch + warm (w_kind 1), the given or the stored weight otherwise; g = w dL_bc + (1 - w) dL_rl (HIRL.py:321). */
int hx_adam_mixed(const HxNets* N, const HxHyper* Hy, int32_t polyak, int32_t step, float grad_scale, int32_t w_kind, float w_given,
                  float warm, int32_t batch, const float* msg, void* stream) {
    HX_REQUIRE(msg, "hx_adam_mixed: null message");
    return adam_impl(N, Hy, 1 | (polyak ? 16 : 0), step, grad_scale, w_kind, w_given, warm, batch, msg, stream);
}

/* soft_update of both targets (HIRL.py:327-330) */
int hx_polyak(const HxNets* N, const HxHyper* Hy, void* stream) {
    HX_REQUIRE(N && Hy, "hx_polyak: bad arguments");
    const int nc = 2 * kQ.padded(), na = kActor.size();
    // (HxNets.xchg_status non-zero — a failed one-shot exchange: the targets stay where they are, like everything hx_adam* guards; the image
    //  refresh below then re-derives the same images from the unchanged targets)
    launch_polyak(N->target_critic, N->critic, nc, Hy->tau, N->target_actor, N->actor, na, (hipStream_t)stream, N->xchg_status);
    if (uint16_t* im = N->w2_bf16_all) {  // bf16 update path: the targets' images follow
        launch_pack_bf16(N->target_actor + kActor.W2(), im + IM_TA * kImgElems, false, (hipStream_t)stream);
        for (int h = 0; h < 2; ++h)
            launch_pack_bf16(N->target_critic + h * kQ.padded() + kQ.W2(), im + (IM_TC1 + h) * kImgElems, false, (hipStream_t)stream);
    }
    HX_CHECK_LAUNCH("hx_polyak");
    return 0;
}

/* Adam (torch defaults) for SAC.  which 0: q1_optim + q2_optim over the flat critic (SAC/agent.py:310-313); which 1:
 * policy_optim, followed in the same launch by the log-alpha step of alpha_optim with the mean entropy in losses[4]
 * (SAC/agent.py:318-325).  step: 1-based (all four optimisers step once per learn()).  grad_scale: 1/world after a SUM. */
int hx_sac_adam(const HxSacNets* N, const HxHyper* Hy, int32_t which, int32_t step, float grad_scale, float target_entropy, void* stream) {
    HX_REQUIRE(N && Hy && step >= 1 && (which == 0 || which == 1), "hx_sac_adam: bad arguments");
    const double b1 = 0.9, b2 = 0.999;
    const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
    AdamArgs A{};
    A.n = which == 0 ? 2 * kQs.padded() : kPolicy.padded();
    A.p = which == 0 ? N->critic : N->policy;
    A.g = which == 0 ? N->grad_critic : N->grad_policy;
    A.m = which == 0 ? N->m_critic : N->m_policy;
    A.v = which == 0 ? N->v_critic : N->v_policy;
    A.b1 = (float)b1; A.b2 = (float)b2; A.eps = 1e-8f;
    A.step_size = (float)((which == 0 ? Hy->lr_critic : Hy->lr_actor) / bc1);
    A.bc2_sqrt = (float)sqrt(bc2);
    A.gscale = grad_scale;
    A.losses = N->losses;
    if (which == 1 && N->policy_w2_f32i) {  // the acting kernel's image of the policy's W2 follows its optimizer step
        A.w2f = N->policy_w2_f32i;
        A.w2_lo = kPolicy.W2();
    }
    if (which == 1 && N->policy_w2_x9) {  // ... and the hi | mid | lo images of the large-population format
        A.w2b = N->policy_w2_x9;
        A.w2b_x9 = 1;
        A.w2_lo = kPolicy.W2();
    }
    if (which == 1) {
        A.alpha_state = N->alpha_state;
        A.target_entropy = target_entropy;
        A.alpha_step_size = (float)(Hy->lr_actor / bc1);
    }
    launch_adam(A, (hipStream_t)stream);
    HX_CHECK_LAUNCH("hx_sac_adam");
    return 0;
}

}  // extern "C"

HX_DEFINE_DEBUG_COLLECTORS(wgrad, 32, 56)
