// hx_actp.hip — policy inference for MANY rows (gfx950): persistent workgroups that fetch the 256 -> 512 weights ONCE and loop over
// their row tiles, with the env step of the same rows in the launch's tail.
//   Agent.chooseAction / chooseActionSmallNoise / chooseActionNoNoise   hirl/agents/HIRL.py:192-212   (U5)
//   chooseAction + HarfangEnv.step for every env                        hirl/train_all.py:343-345
//   SacAgent.explore / exploit (+ step)                                 hirl/agents/SAC/agent.py:183-196, hirl/train_sac.py:238-241
//
// Why a second kernel.  act_fused_kernel (hx_act.hip) gives every 16 / 32 rows a workgroup of their own, and every workgroup pulls the
// whole W2 image (256 KB bf16, 512 KB fp32) through its L2 again: at 131,072 rows that is 4,096 workgroups x 256 KB = 1 GB per launch and
// 4.5 us of every 6.5 us workgroup (profiles/r03e_stamps.txt).  Here the grid is one workgroup per CU (<= 256) and each owns a CONTIGUOUS
// block of rows:
//   bf16   the wave's B fragments (its 32 columns x all of K: 64 VGPRs) are loaded once and stay in registers for every tile
//          (weight-stationary); the tile loop is software-pipelined over three barriers per 32 rows:
//            P1  z2(t-1) = h1(t-1) W2^T on the bf16 matrix cores        | layer 1 of tile t on the fp32 matrix cores  | noise draw (t-1)
//            P2  LN2 + final layer + tanh + noise of tile t-1 (a wave per row) | LN1 statistics of tile t | the next observation tile -> LDS
//            P3  LN1 + activation of tile t -> bf16 h1
//   fp32 / x9  W2 does not fit the register file (512 / 768 KB): the wave streams its column slices ONCE per pass over 64 rows (four row
//          tiles per B fragment: a quarter of act_fused_kernel<2>'s L2 traffic per row), z2 leaves the accumulators in two halves.
// Per-row arithmetic is the arithmetic of act_fused_kernel (same layer-1 MFMA sequence, same k order, same LayerNorm / head functions): the
// two kernels agree bit for bit (tests/test_hirl_gpu.py::test_act_row_tilings_agree_bit_for_bit, tests/test_actp_gpu.py).
// ENV: once a workgroup's actions are written it steps the SAME rows' envs, 512 at a time on all 16 waves (hx_env_block.h: the body of
// env_step_kernel, two lanes per env, fused replay insert) — act + env + insert is one launch at every size.
#include <hip/hip_ext.h>

#include "hx_act.h"
#include "hx_env_block.h"

using namespace hxnn;
using namespace hxu;
using namespace hxact;

namespace {

constexpr int kEnvPass = 512;  // envs per pass of the tail: pair layout = 1,024 lanes = the workgroup

// layer 1 of NRT row tiles on the fp32 matrix cores, exactly as act_fused_kernel does it: wave w owns hidden units 16 w .. 16 w + 15,
// lane (lr, lg) ends up with rows 4 lg .. 4 lg + 3 of unit u; the pre-activations also go to h1s for the LayerNorm statistics
template <int NRT>
__device__ __forceinline__ void layer1_tiles(const float* xs, const float* w1s, float bias1, int u, int lr, int lg, float (&z1)[NRT][4], float* h1s) {
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
        v4f acc = {bias1, bias1, bias1, bias1};
        const float* wrow = w1s + u * 13 + lg;  // columns 13.. of xs are zero; W1 is masked (the LDS words behind a row are not zeros)
        const float* xrow = xs + (t * RT + lr) * XP + lg;
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {  // K = 16 covers the 13 inputs
            const float wv = wrow[4 * mm];
            acc = mfma16(xrow[4 * mm], 4 * mm + lg < 13 ? wv : 0.0f, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            z1[t][r] = acc[r];
            h1s[(t * RT + 4 * lg + r) * LDA1 + u] = z1[t][r];
        }
    }
}

// LN1 statistics: wave w owns rows w, 16 + w, ...
template <int NRT>
__device__ __forceinline__ void ln1_stats(const float* h1s, float* sts, int wave, int lane, int no_ln) {
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
        const int row = t * RT + wave;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = h1s[row * LDA1 + i * 64 + lane];
        float mean, rstd;
        row_stats<4>(v, H1, mean, rstd);
        if (no_ln) { mean = 0.0f; rstd = 1.0f; }
        if (lane == 0) {
            sts[row * 2] = mean;
            sts[row * 2 + 1] = rstd;
        }
    }
}

// head of the rows of NRT tiles (one wave per row: LN2, final layer, tanh, exploration noise, clamp) from the LDS copy of z2
template <int NRT, bool GAUSS, bool RELU>
__device__ __forceinline__ void head_tiles(const ActFusedArgs& A, const float* z2s, const float* hps, const float* s_noise, int r0, int nrow, int wave, int lane,
                                           int t_lo = 0, int t_hi = NRT) {
    const Mlp m = A.m;
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
        if (t < t_lo || t >= t_hi) continue;
        const int lrow = t * RT + wave;
        if (lrow >= nrow) continue;
        const int r = r0 + lrow;
        RowReg<H2> xh, y, z;
        float mean, rstd;
        z.load(z2s + (lrow - t_lo * RT) * LDA2);
        if (!GAUSS) {
            float o[4];
            head_regs<4, 4, RELU>(z, hps, m.out, A.slope, xh, y, mean, rstd, o, m.no_ln);
            if (lane < 4) {
                float a = fast_tanh(lane == 0 ? o[0] : lane == 1 ? o[1] : lane == 2 ? o[2] : o[3]);  // no dynamic register index
                if (A.noise) {
                    a = fminf(fmaxf(a + A.noise[(A.noise_per_row ? (size_t)r * 4 : 0) + lane], -1.0f), 1.0f);
                } else if (A.sigma > 0.0f) {
                    a = fminf(fmaxf(a + A.sigma * s_noise[lrow * 4 + lane], -1.0f), 1.0f);
                }
                A.actions[(size_t)r * 4 + lane] = a;
            }
        } else {
            float o[8];
            head_regs<8, 8, RELU>(z, hps, m.out, A.slope, xh, y, mean, rstd, o, m.no_ln);
            if (lane < 4) {
                const float o0 = o[0], o1 = o[1], o2 = o[2], o3 = o[3], o4 = o[4], o5 = o[5], o6 = o[6], o7 = o[7];  // (values, not a run-time index: hx_act.hip)
                const float mu = lane == 0 ? o0 : lane == 1 ? o1 : lane == 2 ? o2 : o3;
                float a = mu;
                if (A.mode != 0) {
                    const float ls = fminf(fmaxf(lane == 0 ? o4 : lane == 1 ? o5 : lane == 2 ? o6 : o7, -20.0f), 2.0f);  // model.py:65-66
                    const float e = A.mode == 1 ? A.noise[(size_t)r * 4 + lane] : s_noise[lrow * 4 + lane];
                    a = mu + expf(ls) * e;
                }
                a = tanhf(a);
                A.actions[(size_t)r * 4 + lane] = a;
            }
        }
    }
}

// the env step of rows [row_begin, row_end) — whose actions this workgroup has written — kEnvPass at a time (HarfangEnv.step, train_all.py:345)
__device__ __forceinline__ void env_tail(const ActFusedArgs& A, int row_begin, int row_end, float* elds, unsigned& s_slot0, int* s_wcount) {
    using namespace hxenv;
    const StepArgs S{A.state, (int64_t)A.rows, A.stride, A.actions, A.obs, A.reward, A.done, A.success, A.o, A.inv_cap};
    unsigned way = blockIdx.x;
    for (int i0 = row_begin; i0 < row_end; i0 += kEnvPass, way += gridDim.x) {
        if (A.o.ring) env_block_step<true, true, kEnvPass>(S, i0, row_end, elds, s_slot0, s_wcount, way);
        else env_block_step<true, false, kEnvPass>(S, i0, row_end, elds, s_slot0, s_wcount, way);
        __syncthreads();  // the pass's LDS tiles and slot words are free again
    }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16, weight-stationary (BASELINE.json configs[4]: bf16 actor, fp32 dynamics)
// ---------------------------------------------------------------------------------------------------------------
template <bool ENV, bool RELU>
__global__ __launch_bounds__(kWide) void act_persist_bf16_kernel(ActFusedArgs A, int tiles_per_wg) {
    constexpr int NRT = 2, TR = NRT * RT;
    typedef HeadImage<4> Img;
    constexpr int kW1 = H1 * 13;
    constexpr int kLoop = TR * LDA1 + TR * LDA2 + TR * LDB1 / 2;  // h1s (fp32 pre-activations), z2s, h1b (bf16)
    constexpr int kTail = ENV ? hxenv::kEnvBlockLds<true, kEnvPass> : 0;
    constexpr int kUnion = kLoop > kTail ? kLoop : kTail;
    __shared__ __attribute__((aligned(16))) float lds[kW1 + Img::kStride + TR * XP + TR * 2 + TR * 4 + kUnion];
    __shared__ unsigned s_slot0;
    __shared__ int s_wcount[kWide / 64];
    float* w1s = lds;
    float* hps = w1s + kW1;
    float* xs = hps + Img::kStride;
    float* sts = xs + TR * XP;
    float* s_noise = sts + TR * 2;
    float* h1s = s_noise + TR * 4;
    float* z2s = h1s + TR * LDA1;
    __bf16* h1b = reinterpret_cast<__bf16*>(z2s + TR * LDA2);
    const int tid0 = threadIdx.x;
    const int row_begin = (int)blockIdx.x * tiles_per_wg * TR;
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
    float xv, bias1, g1v, be1v, bb0, bb1;
    uint4 bq[2][8];  // B fragments of this wave's two column tiles, all of K: resident for every tile
    {
    const int tid = tid0, wave = tid >> 6, lane = tid & 63, lr = lane & 15, u = wave * 16 + lr;
    xv = obs_of(0, tid);
    float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < kW1 / 4) wv = reinterpret_cast<const float4*>(net + m.W1())[tid];
    bias1 = net[m.b1() + u]; g1v = net[m.g1() + u]; be1v = net[m.be1() + u];
    Img himg;
    himg.fetch(net, m, tid);
    // which 32 columns this wave owns rotates with the workgroup: the workgroups of a launch do not all ask L2 for the same lines at once
    const int cw = (wave + (int)blockIdx.x) & 15;
    bb0 = net[m.b2() + cw * 16 + lr]; bb1 = net[m.b2() + 256 + cw * 16 + lr];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const uint16_t* blk = A.w2b + (size_t)((t * 16 + cw) * 8) * 512 + lane * 8;  // (lane = 16 g + r: w2_image_index's block order)
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) bq[t][sl] = *reinterpret_cast<const uint4*>(blk + sl * 512);
    }
    if (tid < kW1 / 4) reinterpret_cast<float4*>(w1s)[tid] = wv;
    if (tid < TR * XP) xs[tid] = 0.0f;
    himg.store(hps, net, m, tid);
    __syncthreads();
    if (tid < TR * 13) xs[(tid / 13) * XP + tid % 13] = xv;
    xv = obs_of(1, tid);
    __syncthreads();
    }
    const bool draw_noise = !A.noise && A.sigma > 0.0f;
    STAMP();
    // ---- the tile loop: iteration i runs tile i's layer 1 / LN1 and tile i - 1's product and head ----------------------------------
    for (int i = 0; i <= ntile; ++i) {
        if (i == 2) STAMP();
        // The lane's LDS addresses are loop invariants, and with 64 registers of weights resident the allocator spills them; behind this
        // opaque copy of the thread id they are recomputed per tile (a few VALU instructions) instead of living across the loop.
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        const int lr = lane & 15, lg = lane >> 4, u = wave * 16 + lr;
        const int r0p = row_begin + (i - 1) * TR;  // first row of tile i - 1
        const int cw = (wave + (int)blockIdx.x) & 15;
        float z1[NRT][4];
        // P1
        if (i < ntile) layer1_tiles<NRT>(xs, w1s, bias1, u, lr, lg, z1, h1s);
        if (i == 2) STAMP();
        if (i >= 1) {
            if (draw_noise && wave >= kWide / 64 - NRT) {
                const int lrow = (kWide / 64 - 1 - wave) * RT + (lane >> 2);
                s_noise[lrow * 4 + (lane & 3)] = philox_normal(A.row0 + (uint32_t)(r0p + lrow), A.call, 0x61637421u, A.seed, lane & 3);
            }
            v4f acc[NRT][2];
#pragma unroll
            for (int t = 0; t < NRT; ++t) acc[t][0] = acc[t][1] = v4f{0.f, 0.f, 0.f, 0.f};
            // K = 256 in 8 slabs of 32: lane (r, g) holds A[row r][32 sl + 8 g ..+7] and B[32 sl + 8 g ..+7][col r]
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {
#pragma unroll
                for (int t = 0; t < NRT; ++t) {
                    const uint4 aq = *reinterpret_cast<const uint4*>(h1b + (t * RT + lr) * LDB1 + 32 * sl + 8 * lg);
                    acc[t][0] = mfma16_bf16(aq, bq[0][sl], acc[t][0]);
                    acc[t][1] = mfma16_bf16(aq, bq[1][sl], acc[t][1]);
                }
            }
            if (i == 2) STAMP();
#pragma unroll
            for (int t = 0; t < NRT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    z2s[(t * RT + 4 * lg + q) * LDA2 + cw * 16 + lr] = acc[t][0][q] + bb0;
                    z2s[(t * RT + 4 * lg + q) * LDA2 + 256 + cw * 16 + lr] = acc[t][1][q] + bb1;
                }
        }
        if (i == 2) STAMP();
        __syncthreads();  // A: z2 of tile i - 1 and the pre-activations of tile i are in LDS; xs is free
        if (i == 2) STAMP();
        // P2
        if (i >= 1) head_tiles<NRT, false, RELU>(A, z2s, hps, s_noise, r0p, min(TR, row_end - r0p), wave, lane);
        if (i == 2) STAMP();
        if (i < ntile) ln1_stats<NRT>(h1s, sts, wave, lane, m.no_ln);
        if (i + 1 < ntile) {
            if (tid < TR * 13) xs[(tid / 13) * XP + tid % 13] = xv;
            xv = obs_of(i + 2, tid);
        }
        if (i == 2) STAMP();
        __syncthreads();  // B: LN1 statistics of tile i; every read of z2 is done
        if (i == 2) STAMP();
        // P3
        if (i < ntile) {
#pragma unroll
            for (int t = 0; t < NRT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = t * RT + 4 * lg + r;
                    const float hv = act_f<RELU>(g1v * ((z1[t][r] - sts[row * 2]) * sts[row * 2 + 1]) + be1v, slope);
                    h1b[row * LDB1 + u] = (__bf16)hv;  // v_cvt_pk_bf16_f32: round to nearest even
                }
        }
        if (i == 2) STAMP();
        __syncthreads();  // C: h1 of tile i (bf16) is in LDS
        if (i == 2) STAMP();
    }
    STAMP();
    if (ENV) env_tail(A, row_begin, row_end, h1s, s_slot0, s_wcount);  // (barrier C: every action of the block is written)
    STAMP();
    STAMP_FLUSH(0, (blockIdx.x == 0 || blockIdx.x == 200) && tid0 == 0);
    SPAN_LOG(HX_SPAN_ACT);
}

template <typename K>
static void launch_k(K kernel, dim3 grid, const ActFusedArgs& H, int tiles_per_wg, hipStream_t st) {
    if (H.state && H.o.ev_start && H.o.ev_stop)
        hipExtLaunchKernelGGL(kernel, grid, dim3(kWide), 0, st, (hipEvent_t)H.o.ev_start, (hipEvent_t)H.o.ev_stop, 0, H, tiles_per_wg);
    else
        hipLaunchKernelGGL(kernel, grid, dim3(kWide), 0, st, H, tiles_per_wg);
}

// one workgroup per CU: the grid that keeps every weight fetch to ONE per CU (a second resident workgroup would not fit beside 128 VGPRs x
// 1,024 threads anyway)
static int persistent_cus() {
    static const int n = [] {
        if (const char* e = getenv("HX_ACT_PERSIST_WGS")) return atoi(e) > 0 ? atoi(e) : 256;  // tuning knob
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) return cus;
        return 256;
    }();
    return n;
}

}  // namespace

namespace hxact {

bool launch_act_persist(const ActFusedArgs& H, bool gauss, hipStream_t st) {
    const bool env = H.state != nullptr;
    const bool relu = gauss || H.slope == 0.0f;
    if (!gauss && H.w2b && !H.x9) {
        constexpr int TR = 32;
        const int ntiles = (H.rows + TR - 1) / TR;
        const int per = (ntiles + persistent_cus() - 1) / persistent_cus();
        const dim3 grid((unsigned)((ntiles + per - 1) / per));
        if (env) {
            if (relu) launch_k(act_persist_bf16_kernel<true, true>, grid, H, per, st);
            else launch_k(act_persist_bf16_kernel<true, false>, grid, H, per, st);
        } else {
            if (relu) launch_k(act_persist_bf16_kernel<false, true>, grid, H, per, st);
            else launch_k(act_persist_bf16_kernel<false, false>, grid, H, per, st);
        }
        return true;
    }
    return false;
}

}  // namespace hxact

HX_DEFINE_DEBUG_COLLECTORS(actp, 0, 80)
