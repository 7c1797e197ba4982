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

#include "hx_actp_body.h"

using namespace hxnn;
using namespace hxu;
using namespace hxact;

namespace {

// the persistent bf16 acting workgroup lives in hx_actp_body.h (hx_front.hip runs it beside learn()'s first launches)
template <bool ENV, bool RELU>
__global__ __launch_bounds__(kWide) void act_persist_bf16_kernel(ActFusedArgs A, int tiles_per_wg) {
    __shared__ ActpLds<ENV> SL;
    act_persist_bf16_body<ENV, RELU>(A, tiles_per_wg, (int)blockIdx.x, (int)gridDim.x, SL);
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 (MODE 0: the fp32 image, fp32 MFMA) and fp32 through the exact 9-term bf16 split (MODE 1: hi | mid | lo images, bf16 MFMA): W2 does
// not fit the register file (512 / 768 KB), so each wave STREAMS its column slices from L2 once per pass over 64 rows — four row tiles per B
// fragment, a quarter of act_fused_kernel<2>'s L2 traffic per row — and z2 leaves the accumulators in two halves of 32 rows through the
// LDS that held h1.  Per pass: layer 1 | LayerNorm 1 (16 lanes per row, every wave four rows) | the product | 2 x { z2 half -> LDS, head }.
// The k order of every accumulator, and with MODE 1 the order of the nine partial products, are act_fused_kernel's: the same bits.
// ---------------------------------------------------------------------------------------------------------------
template <int MODE, bool GAUSS, bool ENV, bool RELU>
__global__ __launch_bounds__(kWide) void act_persist_stream_kernel(ActFusedArgs A, int tiles_per_wg) {
    constexpr bool X9 = MODE == 1;
    constexpr int NRT = 4, TR = NRT * RT, HR = 32;  // rows per pass; rows per z2 half
    constexpr int OUT = GAUSS ? 8 : 4;
    typedef HeadImage<OUT> Img;
    constexpr int kH1 = X9 ? 3 * TR * LDB1 / 2 : TR * LDA1;  // h1: three bf16 tiles (hi | mid | lo) or one fp32 tile; before that the fp32 pre-activations
    constexpr int kPre = TR * LDA1;
    constexpr int kZ = HR * LDA2;
    constexpr int kTail = ENV ? hxenv::kEnvBlockLds<true, kEnvPass> : 0;
    constexpr int kUnion = (kH1 > kPre ? kH1 : kPre) > (kZ > kTail ? kZ : kTail) ? (kH1 > kPre ? kH1 : kPre) : (kZ > kTail ? kZ : kTail);
    __shared__ __attribute__((aligned(16))) float lds[Img::kStride + TR * XP + 3 * H1 + H2 + kWide * 4 + 2 * TR * 4 + kUnion];
    __shared__ unsigned s_slot0;
    __shared__ int s_wcount[kWide / 64];
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
    const int row_begin = (int)blockIdx.x * tiles_per_wg * TR;
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
        const int cw = X9 ? ((wave + (int)blockIdx.x) & 15) : wave;  // which 32 columns (x9: rotated with the workgroup, as act_fused_kernel)
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
                    r = mfma16_bf16(b[2], a3[2], r);
                    r = mfma16_bf16(b[1], a3[2], r);
                    r = mfma16_bf16(b[2], a3[1], r);
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
    if (ENV) env_tail((KernArgs)__builtin_amdgcn_kernarg_segment_ptr(), row_begin, row_end, h1s, &s_slot0, s_wcount, (int)blockIdx.x, (int)gridDim.x);  // (every action of the block is written)
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
    if (env && !H.o.ring) return false;  // the env tail is built with the fused replay insert only; without a ring: act (persistent) + hx_env_step
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
    const bool x9 = H.w2b && H.x9;
    if (x9 || H.w2f) {  // fp32 from an image: 64 rows per pass
        constexpr int TR = 64;
        const int ntiles = (H.rows + TR - 1) / TR;
        const int per = (ntiles + persistent_cus() - 1) / persistent_cus();
        const dim3 grid((unsigned)((ntiles + per - 1) / per));
#define HX_STREAM(MODE, G, R) \
        { if (env) launch_k(act_persist_stream_kernel<MODE, G, true, R>, grid, H, per, st); else launch_k(act_persist_stream_kernel<MODE, G, false, R>, grid, H, per, st); }
        if (x9 && gauss) HX_STREAM(1, true, true)
        else if (x9) { if (relu) HX_STREAM(1, false, true) else HX_STREAM(1, false, false) }
        else if (gauss) HX_STREAM(0, true, true)
        else if (relu) HX_STREAM(0, false, true)
        else HX_STREAM(0, false, false)
#undef HX_STREAM
        return true;
    }
    return false;
}

}  // namespace hxact

HX_DEFINE_DEBUG_COLLECTORS(actp, 0, 80)
