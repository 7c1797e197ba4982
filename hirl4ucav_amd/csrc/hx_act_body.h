// hx_act_body.h — the per-tile acting workgroup (act_fused_kernel of hx_act.hip) as a device function over an explicit LDS block and an explicit
// workgroup index, so that hx_front.hip can run the same workgroups beside the first launches of learn() inside ONE launch.
#pragma once
#include "hx_act.h"
#include "hx_env_dev.h"

namespace hxact {
using namespace hxnn;
using namespace hxu;

// ---------------------------------------------------------------------------------------------------------------
// act_fused: the whole policy for 16 observation rows in ONE workgroup — layer 1 + LN1 (VALU), z2 = h1 W2^T for all 512
// columns (two 16-column MFMA tiles per wave, K = 256; W2 streams through two LDS buffers in 16-wide k-chunks, register-
// prefetched two chunks ahead), then LN2 + final layer + tanh + exploration noise + clamp with one wave per row straight from the LDS copy
// of z2.  No z2 round trip through HBM, no second launch.
// chooseAction / chooseActionSmallNoise / chooseActionNoNoise, HIRL.py:192-212.
// ---------------------------------------------------------------------------------------------------------------
constexpr int ACT_KC = 16;            // k-chunk of W2 staged through LDS (64 B per column), double-buffered
constexpr int ACT_LDW = ACT_KC + 8;   // pitch = 8 mod 16 dwords: conflict-free ds_read_b128 (see LDA1)
constexpr int ACT_NCH = H1 / ACT_KC;  // 16 chunks

// NRT = 16-row tiles per workgroup: 1 keeps 256 workgroups busy at 4,096 rows; 2 (from 8,192 rows on) multiplies every W2
// chunk against two row tiles, halving W2's L2 traffic and the barriers per MFMA.
// GAUSS = the SAC policy: plain Linear-ReLU stack (m.no_ln), 8-wide head = mean ++ log_std, tanh-Gaussian sample
// (SacAgent.explore / exploit, SAC/agent.py:183-196, GaussianPolicy.sample, SAC/model.py:63-82).
// ENV   = the env step of the same rows runs in the tail: the 16 (32) actions meet in LDS and the lanes of wave 0 each step
//         one env (hx_env_dev.h: the code of env_step_kernel, contraction off), with the fused replay insert — no second
//         launch, and the env's ~2,500-instruction chain runs on every CU at once instead of on 16 of them.
// BF16 = the policy's 256 -> 512 layer on v_mfma_f32_16x16x32_bf16 (BASELINE.json configs[4]: bf16 actor, fp32 dynamics): h1 is rounded to
//         bf16 once, W2 comes from a bf16 image; accumulation, both LayerNorms, layer 1 and the head stay fp32.  Every wave owns 32 of
//         the 512 columns and nobody else reads them, so its B fragments (16 x 16 B per lane = the 256 KB image once per
//         workgroup) go from L2 straight into registers at kernel entry — no LDS staging, no chunk barriers; the 16 (32) rows
//         of h1 are the only shared operand.
// X3   = fp32 policy, the 256 -> 512 product on v_mfma_f32_16x16x32_bf16 EXACTLY: h1 and W2 as hi + mid + lo bf16 parts (split3_bf16: nothing
//         is lost), the partial products (each exact in fp32; [r5] six of the nine: hx_act.h HX_X9_TERMS) accumulated in fp32 — the small ones in their own accumulator, joined with
//         hi x hi at the end.  144 matrix-core cycles per 32 k and column tile against 256 for fp32 MFMA; the B fragments stream from the three
//         images (768 KB per workgroup) one slab ahead of the multiply.
// The workgroup's LDS as ONE object: the kernel declares it (hx_act.hip), hx_front.hip overlays it with the forward workgroups' block in a union.
template <int NRT, bool ENV, bool BF16, bool F32I, bool X3>
struct ActLds {
    static constexpr int ROWS = NRT * RT;
    static_assert(2 * H2 * ACT_LDW >= ROWS * LDA2, "the z2 tile reuses the W2 chunk buffers");
    // fp32: two W2 chunk buffers (reused for z2 and, in the env tail, the replay rows / next observations)
    // bf16: the z2 tile, then the replay rows / next observations, and the bf16 h1 tile
    static constexpr bool IMG = BF16 || F32I || X3;  // W2 comes from an image straight into registers: no chunk buffers in LDS
    static constexpr int kTileA = IMG ? ROWS * LDA2 : H2 * ACT_LDW;
    static constexpr int kTileB = IMG ? (ENV ? ROWS * (hxenv::kRowPitch + HX_OBS_DIM) : 4) : H2 * ACT_LDW;
    static constexpr int kPrologue = ROWS * LDA1 + ROWS * XP + ROWS * 2 + H1 * 13;
    __attribute__((aligned(16))) float lds[kPrologue + kTileA + kTileB];
    __attribute__((aligned(16))) __bf16 h1b[BF16 ? ROWS * LDB1 : 8];
    // X3: the hi | mid | lo tiles of h1.  With 32 rows they do not fit beside the z2 tile (52 + 67 KB + the prologue's 49): they LIE IN the z2 tile's
    // LDS, which is written only once the product is through (one more barrier)
    static constexpr bool kX3InZ2 = X3 && NRT == 2;
    __attribute__((aligned(16))) uint16_t h1x[(X3 && !kX3InZ2) ? 3 * ROWS * LDB1 : 8];
    static_assert(!kX3InZ2 || 3 * ROWS * LDB1 * 2 <= kTileA * 4, "the three bf16 tiles of h1 fit in the z2 tile");
    float s_act[ENV ? ROWS * 4 : 4];
    float s_noise[ROWS * 4];  // exploration noise of the workgroup's rows, drawn by the last wave(s) under the prologue's loads
    unsigned s_base;          // ring slot of the workgroup's first row
    int s_nstore;
};

template <int NRT, bool GAUSS, bool ENV, bool BF16, bool RELU, bool F32I = false, bool X3 = false>
__device__ __forceinline__ void act_fused_body(const ActFusedArgs& A, const int bid, ActLds<NRT, ENV, BF16, F32I, X3>& SL) {
    static_assert(!(BF16 && F32I) && !(X3 && (BF16 || F32I)), "one image format at a time");
    typedef ActLds<NRT, ENV, BF16, F32I, X3> Lds;
    constexpr int ROWS = NRT * RT;
    constexpr bool IMG = Lds::IMG;
    constexpr int kTileA = Lds::kTileA;
    float* const lds = SL.lds;
    __bf16* const h1b = SL.h1b;
    uint16_t* const h1x = Lds::kX3InZ2 ? reinterpret_cast<uint16_t*>(SL.lds + Lds::kPrologue) : SL.h1x;
    float* const s_act = SL.s_act;
    float* const s_noise = SL.s_noise;
    unsigned& s_base = SL.s_base;
    int& s_nstore = SL.s_nstore;
    float* h1s = lds;
    float* xs = h1s + ROWS * LDA1;
    float* sts = xs + ROWS * XP;
    float* w1s = sts + ROWS * 2;
    float* wb0 = w1s + H1 * 13;        // [H2][ACT_LDW]: even k-chunks of W2, every column
    float* wb1 = wb0 + kTileA;         // odd k-chunks
    float* z2s = wb0;                  // [ROWS][LDA2] once the last chunk has been multiplied
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r0 = bid * ROWS;
    const int nrow = min(ROWS, A.rows - r0);
    const float* net = A.net;
    const Mlp m = A.m;
    const float slope = A.slope;
    // layer 1 runs on MFMA (as in fwd_l2): wave w owns hidden units 16 w .. 16 w + 15; lane (lr, lg) ends up with rows 4 lg .. 4 lg + 3 of unit u
    const int lr = lane & 15, lg = lane >> 4, u = wave * 16 + lr;
    STAMP_DECL;
    STAMP();
    // W2 chunk loader: 4 lanes cover one column's 64 B, the workgroup 256 columns per pass, 2 passes.  Every byte of W2 enters
    // this CU once and is shared by all 16 waves from LDS.  Two register sets run two chunks ahead of the multiply, two LDS
    // buffers one chunk ahead: per chunk one barrier, and the LDS stores of chunk c+1 sit under the MFMAs of chunk c.
    const int piece = tid & 3, colb = tid >> 2;
    const float* w2g = net + m.W2() + (size_t)colb * H1 + piece * 4;
    const int w2w = colb * ACT_LDW + piece * 4;
#define ACT_LOAD(ra, rb, c) { ra = *reinterpret_cast<const float4*>(w2g + (c) * ACT_KC); rb = *reinterpret_cast<const float4*>(w2g + (size_t)256 * H1 + (c) * ACT_KC); }
#define ACT_STORE(buf, ra, rb) { *reinterpret_cast<float4*>((buf) + w2w) = ra; *reinterpret_cast<float4*>((buf) + w2w + 256 * ACT_LDW) = rb; }
    // four register sets: chunk c travels in set c % 4 and is requested FOUR multiply phases before it is stored to LDS — with two sets
    // (64 KB in flight per CU) the loop ran at the L2 round trip, not at the MFMA rate
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    uint4 bq[BF16 ? 2 : 1][BF16 ? 8 : 1];  // BF16: B fragments of this wave's two column tiles, all of K (requested below)
    // F32I: every wave owns 32 of the 512 columns and nobody else reads them, so its fp32 B fragments go from L2 straight into registers
    // in MFMA operand order — one contiguous kilobyte per load from the image — ACT_PF chunks ahead of the multiply: no LDS staging (80 KB of
    // LDS traffic per chunk with it), no barrier per chunk; the waves stream independently.  Same k order as the staged loop: same bits.
    constexpr int ACT_PF = 3;
    float4 pb[F32I ? ACT_NCH : 1], qb[F32I ? ACT_NCH : 1];
    const float* img0 = F32I ? A.w2f + (size_t)wave * (16 * 256) + lane * 4 : nullptr;  // 1 KB block (column tile `wave`, chunk c) at + 256 c floats; column tile 16 + wave 65,536 floats on
    if constexpr (!IMG) {
        ACT_LOAD(ra0, rb0, 0);
        ACT_LOAD(ra1, rb1, 1);
    }
    // head parameters (g2, be2, W3, b3): requested now, parked in 4-8 registers, laid out in LDS once h1 is dead
    typedef HeadImage<GAUSS ? 8 : 4> Img;
    static_assert(Img::kStride <= Lds::kPrologue, "the head image reuses the prologue's LDS");
    Img himg;
    float* hps = lds;
    // all independent operands first
    float xv = 0.0f;
    if (tid < ROWS * 13) {
        const int r = tid / 13;
        if (r < nrow) xv = A.obs[(size_t)r0 * 13 + tid];  // the ROWS x 13 tile is contiguous
    }
    float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < H1 * 13 / 4) wv = reinterpret_cast<const float4*>(net + m.W1())[tid];
    const float bias1 = net[m.b1() + u], g1v = net[m.g1() + u], be1v = net[m.be1() + u];
    himg.fetch(net, m, tid);  // (needed last: behind the prologue's own operands)
    if constexpr (!IMG) {  // behind the prologue's own operands
        ACT_LOAD(ra2, rb2, 2);
        ACT_LOAD(ra3, rb3, 3);
    }
    // The standard-normal draws of the rows' exploration noise depend on (row, call, seed) only.  In the head they cost every wave ~320
    // instructions for 4 useful lanes (Philox + Box-Muller with the library's log / sin / cos), 16 waves deep on an issue-bound phase;
    // here ONE wave draws all 64 (row, component) values of a row tile while its own loads are in flight.  Same function, same bits.
    const bool draw_noise = GAUSS ? (A.mode != 0 && A.mode != 1) : (!A.noise && A.sigma > 0.0f);
    if (draw_noise && wave >= kWide / 64 - NRT) {
        const int lrow = (kWide / 64 - 1 - wave) * RT + (lane >> 2);
        s_noise[lrow * 4 + (lane & 3)] = philox_normal(A.row0 + (uint32_t)(r0 + lrow), A.call, GAUSS ? 0x53414331u : 0x61637421u, A.seed, lane & 3);
    }
    if (tid < H1 * 13 / 4) reinterpret_cast<float4*>(w1s)[tid] = wv;
    if (tid < ROWS * XP) xs[tid] = 0.0f;
    __syncthreads();
    // BF16: which 32 columns this wave owns rotates with the workgroup, so that the 256 workgroups of a launch do not all ask L2 for
    // the same lines of the W2 image at the same moment
    const int cw = (BF16 || X3) ? ((wave + bid) & 15) : wave;
#ifdef HX_DBG_ACT_HOT
    // TIMING EXPERIMENT ONLY (tools/ubench/act_l2_stream_ab.sh builds a second library with it; wrong results by construction): the exact-split product's
    // B fragments all come from the FIRST k-slab's addresses (level 1: 6 KB per wave, L2-hot) or from ONE wave's first slab (level 2: 6 KB per workgroup,
    // L1-hot) — the same load instructions and bytes per lane, none of the 768 KB image stream: what the product phase costs when the stream costs nothing
    const int cwi = HX_DBG_ACT_HOT >= 2 ? 0 : cw;
    constexpr int kHotSlab = 0;
#else
    const int cwi = cw;
    constexpr int kHotSlab = 1;
#endif
    uint4 w3q = {0u, 0u, 0u, 0u};  // BF16: this lane's A fragment of the final layer (hx_act.h w3_fragment)
    if constexpr (BF16) w3q = w3_fragment(net, m, cw, lane);
    if constexpr (BF16) {
        // requested only now, behind the prologue's own operands: every workgroup pulls the whole 256 KB image through L2 (64 MB per
        // launch at 4,096 rows, ~6 us of L2 service); issued at kernel entry those requests queue up in front of OTHER workgroups'
        // small operands and stall every prologue for that long.  From here they overlap layer 1 and LayerNorm 1.
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint16_t* blk = A.w2b + (size_t)((t * 16 + cw) * 8) * 512 + lane * 8;  // (lane = 16 g + r: w2_image_index's block order)
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) bq[t][sl] = *reinterpret_cast<const uint4*>(blk + sl * 512);
        }
    }
    // X3: the first slab of this wave's B fragments (three images x two column tiles) is requested HERE, under layer 1 and LayerNorm 1, instead of in front of
    // the product loop (where its L2 round trip was the loop's first wait)
    uint4 bb0[X3 ? 3 : 1][2];
    if constexpr (X3) {
        const uint16_t* img = A.w2b + (size_t)(cwi * 8) * 512 + lane * 8;
#pragma unroll
        for (int sx = 0; sx < 3; ++sx)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) bb0[sx][ct] = *reinterpret_cast<const uint4*>(img + (size_t)sx * kImgElems + (size_t)ct * (16 * 8 * 512));
    }
    if (tid < ROWS * 13) xs[(tid / 13) * XP + tid % 13] = xv;
    __syncthreads();
    float z1[NRT][4];
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
    __syncthreads();
    // LN1 statistics, 16 lanes per row (hx_act.h): waves 0 .. 4 NRT - 1 take four rows each — wave w rows w, w + 4 NRT, ... (eight rows
    // apart at NRT = 2: the pitch is 8 mod 64 dwords, so the four rows of a wave's 16-byte reads fall on disjoint banks)
    constexpr int kRowWaves = 4 * NRT;
    const int gq = lane >> 4, gc = lane & 15;
    if constexpr (BF16) {
        // [r5] 32 lanes per row, two rows per wave (hx_act.h row_stats32): the statistics act_persist_bf16_body computes on all of its 16 waves — same bits
        if (wave < ROWS / 2) {
            const int row = 2 * wave + (lane >> 5), c = lane & 31;
            const v4f x0 = *reinterpret_cast<const v4f*>(h1s + row * LDA1 + 4 * c), x1 = *reinterpret_cast<const v4f*>(h1s + row * LDA1 + 128 + 4 * c);
            float mean, rstd;
            row_stats32(x0, x1, H1, mean, rstd);
            if (m.no_ln) { mean = 0.0f; rstd = 1.0f; }
            if (c == 0) {
                sts[row * 2] = mean;
                sts[row * 2 + 1] = rstd;
            }
        }
    } else if (wave < kRowWaves) {
        const int row = wave + kRowWaves * gq;
        float v[16];
        load_row16<H1>(h1s + row * LDA1, gc, v);
        float mean, rstd;
        row_stats16<16>(v, H1, mean, rstd);
        if (m.no_ln) { mean = 0.0f; rstd = 1.0f; }
        if (gc == 0) {
            sts[row * 2] = mean;
            sts[row * 2 + 1] = rstd;
        }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NRT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = t * RT + 4 * lg + r;
            const float hv = ln_act<RELU>(z1[t][r], sts[row * 2], sts[row * 2 + 1], g1v, be1v, slope);
            if (BF16) h1b[row * LDB1 + u] = (__bf16)hv;  // v_cvt_pk_bf16_f32: round to nearest even
            else if (X3) split3_bf16(hv, h1x[row * LDB1 + u], h1x[(ROWS + row) * LDB1 + u], h1x[(2 * ROWS + row) * LDB1 + u]);
            else h1s[row * LDA1 + u] = hv;
        }
    if (!IMG) {
        ACT_STORE(wb0, ra0, rb0);
        ACT_LOAD(ra0, rb0, 4);
    }
    if constexpr (F32I) {  // the first chunks of this wave's columns (behind the prologue's own traffic)
#pragma unroll
        for (int c = 0; c < ACT_PF; ++c) {
            pb[c] = *reinterpret_cast<const float4*>(img0 + c * 256);
            qb[c] = *reinterpret_cast<const float4*>(img0 + (size_t)16 * 16 * 256 + c * 256);
        }
    }
    __syncthreads();
    STAMP();
    // ENV: the env lanes (wave 0, two lanes per env: hx_env_dev.h "Pair") request their state words and the current observation while the
    // policy's last phases run
    hxenv::Stepper<true> envT;
    float envPrev[HX_OBS_DIM];
    const int env_e = lane >> 1;            // env of this lane inside the workgroup's rows
    const bool env_opp = (lane & 1) != 0;   // this lane owns the opponent aircraft
    auto env_load = [&]() {
        if (ENV && wave == 0 && env_e < nrow) {
            envT.load(A.state, A.stride, r0, (uint32_t)env_e, env_opp);
#pragma unroll
            for (int j = 0; j < HX_OBS_DIM; ++j) envPrev[j] = (A.o.ring && env_opp) ? A.obs[((size_t)r0 + env_e) * HX_OBS_DIM + j] : 0.0f;
        }
    };
    {   // z2 tiles: columns 16*wave .. and 256 + 16*wave .. of every row tile; k ascending, chunk by chunk
        const int r = lane & 15, g = lane >> 4;
        v4f acc[NRT][2];
#pragma unroll
        for (int t = 0; t < NRT; ++t) acc[t][0] = acc[t][1] = v4f{0.f, 0.f, 0.f, 0.f};
        if constexpr (BF16) {
            // K = 256 in 8 slabs of 32: lane (r, g) holds A[row r][32 sl + 8 g ..+7] and B[32 sl + 8 g ..+7][col r]
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {
#pragma unroll
                for (int t = 0; t < NRT; ++t) {
                    // [r5] operands swapped (weights as A, rows as B) as in act_persist_bf16_body: the same products in the same k order, and lane
                    // (r, g) holds FOUR CONSECUTIVE columns 4 g .. + 3 of row r — what hx_act.h's "straight from the accumulators" steps take
                    const uint4 aq = *reinterpret_cast<const uint4*>(h1b + (t * RT + r) * LDB1 + 32 * sl + 8 * g);
                    acc[t][0] = mfma16_bf16(bq[0][sl], aq, acc[t][0]);
                    acc[t][1] = mfma16_bf16(bq[1][sl], aq, acc[t][1]);
                }
            }
        }
        if constexpr (X3) {
            // K = 256 in 8 slabs of 32; per slab 3 A fragments (LDS) and 2 x 3 B fragments (one contiguous kilobyte each, from the images)
            v4f rest[NRT][2];
#pragma unroll
            for (int t = 0; t < NRT; ++t) rest[t][0] = rest[t][1] = v4f{0.f, 0.f, 0.f, 0.f};
            const uint16_t* img = A.w2b + (size_t)(cwi * 8) * 512 + lane * 8;  // image s at + s kImgElems, column tile 16 + cw at + 16 * 8 * 512
            uint4 bb[2][3][2];
#pragma unroll
            for (int sx = 0; sx < 3; ++sx)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) bb[0][sx][ct] = bb0[sx][ct];
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {
                if (sl + 1 < 8) {
#pragma unroll
                    for (int sx = 0; sx < 3; ++sx)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
                            bb[(sl + 1) & 1][sx][ct] = *reinterpret_cast<const uint4*>(img + (size_t)sx * kImgElems + (size_t)ct * (16 * 8 * 512) + kHotSlab * (sl + 1) * 512);
                }
                __builtin_amdgcn_sched_barrier(0);  // the requests stay ahead of the multiply (see the F32I loop)
#pragma unroll
                for (int t = 0; t < NRT; ++t) {
                    uint4 a3[3];
#pragma unroll
                    for (int sx = 0; sx < 3; ++sx) a3[sx] = *reinterpret_cast<const uint4*>(h1x + ((sx * ROWS) + t * RT + r) * LDB1 + 32 * sl + 8 * g);
                    const uint4(&b)[3][2] = bb[sl & 1];
                    // smallest first: lo lo, lo mid, mid lo | lo hi, hi lo, mid mid | mid hi, hi mid -> rest; hi hi -> acc
                    if (HX_X9_TERMS == 9) {
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) rest[t][ct] = mfma16_bf16(a3[2], b[2][ct], rest[t][ct]);
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) rest[t][ct] = mfma16_bf16(a3[2], b[1][ct], rest[t][ct]);
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) rest[t][ct] = mfma16_bf16(a3[1], b[2][ct], rest[t][ct]);
                    }
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) rest[t][ct] = mfma16_bf16(a3[2], b[0][ct], rest[t][ct]);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) rest[t][ct] = mfma16_bf16(a3[0], b[2][ct], rest[t][ct]);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) rest[t][ct] = mfma16_bf16(a3[1], b[1][ct], rest[t][ct]);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) rest[t][ct] = mfma16_bf16(a3[1], b[0][ct], rest[t][ct]);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) rest[t][ct] = mfma16_bf16(a3[0], b[1][ct], rest[t][ct]);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) acc[t][ct] = mfma16_bf16(a3[0], b[0][ct], acc[t][ct]);
                }
            }
#pragma unroll
            for (int t = 0; t < NRT; ++t)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[t][ct] = acc[t][ct] + rest[t][ct];
            if constexpr (Lds::kX3InZ2) __syncthreads();  // every wave has read its last h1 fragments: their LDS takes z2 now
        }
        const float* ap = h1s + r * LDA1 + 4 * g;
        const int boff = (wave * 16 + r) * ACT_LDW + 4 * g;
#define ACT_MUL(buf, c) { \
            float4 a4[NRT]; \
            _Pragma("unroll") for (int t = 0; t < NRT; ++t) a4[t] = *reinterpret_cast<const float4*>(ap + t * RT * LDA1 + (c) * ACT_KC); \
            const float4 p4 = *reinterpret_cast<const float4*>((buf) + boff); \
            const float4 q4 = *reinterpret_cast<const float4*>((buf) + boff + 256 * ACT_LDW); \
            _Pragma("unroll") for (int t = 0; t < NRT; ++t) { \
                acc[t][0] = mfma16(a4[t].x, p4.x, acc[t][0]); acc[t][1] = mfma16(a4[t].x, q4.x, acc[t][1]); \
                acc[t][0] = mfma16(a4[t].y, p4.y, acc[t][0]); acc[t][1] = mfma16(a4[t].y, q4.y, acc[t][1]); \
                acc[t][0] = mfma16(a4[t].z, p4.z, acc[t][0]); acc[t][1] = mfma16(a4[t].z, q4.z, acc[t][1]); \
                acc[t][0] = mfma16(a4[t].w, p4.w, acc[t][0]); acc[t][1] = mfma16(a4[t].w, q4.w, acc[t][1]); } }
        static_assert(ACT_NCH % 4 == 0, "the chunk loop is unrolled by the four register sets");
        if constexpr (F32I) {
            float4 an[NRT];  // the h1 fragment of the NEXT chunk: its LDS round trip runs under this chunk's MFMAs
#pragma unroll
            for (int t = 0; t < NRT; ++t) an[t] = *reinterpret_cast<const float4*>(ap + t * RT * LDA1);
#pragma unroll
            for (int c = 0; c < ACT_NCH; ++c) {
                float4 a4[NRT];
#pragma unroll
                for (int t = 0; t < NRT; ++t) a4[t] = an[t];
                if (c + ACT_PF < ACT_NCH) {
                    pb[c + ACT_PF] = *reinterpret_cast<const float4*>(img0 + (c + ACT_PF) * 256);
                    qb[c + ACT_PF] = *reinterpret_cast<const float4*>(img0 + (size_t)16 * 16 * 256 + (c + ACT_PF) * 256);
                }
                if (c + 1 < ACT_NCH) {
#pragma unroll
                    for (int t = 0; t < NRT; ++t) an[t] = *reinterpret_cast<const float4*>(ap + t * RT * LDA1 + (c + 1) * ACT_KC);
                }
                // the requests stay HERE, ahead of the multiply: the scheduler otherwise sinks them to just before their use (fewer live
                // registers) and every chunk then waits a full L2 / LDS round trip
                __builtin_amdgcn_sched_barrier(0);
                const float4 p4 = pb[c], q4 = qb[c];
#pragma unroll
                for (int t = 0; t < NRT; ++t) {
                    acc[t][0] = mfma16(a4[t].x, p4.x, acc[t][0]); acc[t][1] = mfma16(a4[t].x, q4.x, acc[t][1]);
                    acc[t][0] = mfma16(a4[t].y, p4.y, acc[t][0]); acc[t][1] = mfma16(a4[t].y, q4.y, acc[t][1]);
                    acc[t][0] = mfma16(a4[t].z, p4.z, acc[t][0]); acc[t][1] = mfma16(a4[t].z, q4.z, acc[t][1]);
                    acc[t][0] = mfma16(a4[t].w, p4.w, acc[t][0]); acc[t][1] = mfma16(a4[t].w, q4.w, acc[t][1]);
                }
            }
            __syncthreads();  // every wave has read its last h1 fragment: the tile's LDS may now take z2 and the head image
        }
        for (int c = 0; c < (IMG ? 0 : ACT_NCH); c += 4) {
            // chunk c is in wb0; set 1 holds chunk c+1, sets 2, 3, 0 hold c+2, c+3, c+4 (in flight)
            ACT_STORE(wb1, ra1, rb1);
            if (c + 5 < ACT_NCH) ACT_LOAD(ra1, rb1, c + 5);
            ACT_MUL(wb0, c);
            __syncthreads();
            ACT_STORE(wb0, ra2, rb2);
            if (c + 6 < ACT_NCH) ACT_LOAD(ra2, rb2, c + 6);
            ACT_MUL(wb1, c + 1);
            __syncthreads();
            ACT_STORE(wb1, ra3, rb3);
            if (c + 7 < ACT_NCH) ACT_LOAD(ra3, rb3, c + 7);
            ACT_MUL(wb0, c + 2);
            __syncthreads();
            if (c + 4 < ACT_NCH) {
                ACT_STORE(wb0, ra0, rb0);
                if (c + 8 < ACT_NCH) ACT_LOAD(ra0, rb0, c + 8);
            }
            ACT_MUL(wb1, c + 3);
            __syncthreads();
        }
#undef ACT_MUL
#undef ACT_LOAD
#undef ACT_STORE
        if constexpr (BF16) {
            // [r5] no z2 tile: bias, then the wave's partial LayerNorm-2 statistics of its 32 columns of rows r / 16 + r (hx_act.h, step 1) -> part
            const v4f bb0 = *reinterpret_cast<const v4f*>(net + m.b2() + cw * 16 + 4 * g), bb1 = *reinterpret_cast<const v4f*>(net + m.b2() + 256 + cw * 16 + 4 * g);
#pragma unroll
            for (int t = 0; t < NRT; ++t) {
                acc[t][0] = acc[t][0] + bb0;
                acc[t][1] = acc[t][1] + bb1;
                row_partial32(acc[t][0], acc[t][1], g, z2s + (t * RT + r) * kPartPitch + 2 * cw);
            }
            himg.store(hps, net, m, tid);
            env_load();
            __syncthreads();
            // steps 2, 3: LayerNorm 2 + activation from the accumulators, the final layer's share on the bf16 matrix cores -> outp[16 groups][ROWS][4]
            float* const outp = z2s + ROWS * kPartPitch;
#pragma unroll
            for (int t = 0; t < NRT; ++t) {
                float mean, rstd;
                row_combine16(z2s + (t * RT + r) * kPartPitch, g, m.no_ln, mean, rstd);
                const uint4 hq = ln2_operand<RELU>(acc[t][0], acc[t][1], mean, rstd, hps, cw * 16 + 4 * g, slope);
                const v4f o = mfma16_bf16(w3q, hq, v4f{0.f, 0.f, 0.f, 0.f});  // outputs 0..3 of row r in the g = 0 lanes
                if (g == 0) *reinterpret_cast<v4f*>(outp + ((size_t)cw * ROWS + t * RT + r) * 4) = o;
            }
        } else {
        const float bb0 = net[m.b2() + cw * 16 + r], bb1 = net[m.b2() + 256 + cw * 16 + r];
#pragma unroll
        for (int t = 0; t < NRT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // every wave is past the last barrier: the chunk buffers are free for z2
                z2s[(t * RT + 4 * g + q) * LDA2 + cw * 16 + r] = acc[t][0][q] + bb0;
                z2s[(t * RT + 4 * g + q) * LDA2 + 256 + cw * 16 + r] = acc[t][1][q] + bb1;
            }
        himg.store(hps, net, m, tid);  // ... and h1 / x / W1 are dead: their LDS takes the head image
        }
    }
    __syncthreads();
    STAMP();
    if (!BF16) env_load();  // (BF16: requested in front of the LayerNorm-2 steps above — the short last step would not hide the round trip)
    // head, 16 lanes per row (hx_act.h): waves 0 .. 4 NRT - 1, four rows each, the same rows as in the LN1 statistics
    if constexpr (BF16) {
        // [r5] the last step: waves 0 .. NRT - 1, a lane per (row, component) — the 16 column groups' shares in group order + b3, tanh, noise, clamp
        static_assert(!(BF16 && GAUSS), "the bf16 acting format is the deterministic head's");
        if (wave < NRT) {
            const int lrow = wave * RT + (lane >> 2), c = lane & 3;
            if (lrow < nrow) {
                const float o = head_sum16(z2s + ROWS * kPartPitch, ROWS, lrow, c, hps[(2 + 4) * H2 + c]);
                const float a = action_of1(A, o, c, r0 + lrow, s_noise + lrow * 4);
                A.actions[(size_t)(r0 + lrow) * 4 + c] = a;
                if (ENV) s_act[lrow * 4 + c] = a;
            }
        }
    } else if (wave < kRowWaves) {
        const int lrow = wave + kRowWaves * gq;
        if (lrow < nrow) {
            const int r = r0 + lrow;
            float o[GAUSS ? 8 : 4];
            head16<GAUSS ? 8 : 4, GAUSS ? 8 : 4, RELU>(z2s + lrow * LDA2, hps, gc, slope, m.no_ln, o);
            if (gc < 4) {
                const float a = action_of<GAUSS>(A, o, gc, r, s_noise + lrow * 4);
                A.actions[(size_t)r * 4 + gc] = a;
                if (ENV) s_act[lrow * 4 + gc] = a;
            }
        }
    }
    if (ENV) {
        using namespace hxenv;
        float* s_row = wb1;                    // [ROWS][33] replay rows   (the odd chunk buffer is free since the last barrier)
        float* s_obs = wb1 + ROWS * kRowPitch;  // [ROWS][13] next observations
        // Ring slots: one atomic per workgroup on ONE address, 256 workgroups at about the same moment — its return takes ~3 us.  Which rows
        // are stored depends on the state only (episode step counter against max_step), so wave 0 asks for its slots as soon as its own head
        // row is done, BEFORE the barrier that collects the other rows' actions: the wait of the other waves and the env step hide it.
        bool trunc = false, store = false;
        int rank = 0, nstore = 0;
        unsigned long long base = 0ull;
        if (wave == 0) {
            if (lane < 2 * ROWS && env_e < nrow) {
                uint32_t ep = envT.episode_step();
                ep = ep < 65535u ? ep + 1u : ep;
                trunc = A.o.max_step > 0 && (int)ep >= A.o.max_step;  // train_all.py:346-347
                store = A.o.ring != nullptr && !trunc;
            }
            const unsigned long long bal = __ballot(store && !env_opp);
            rank = __popcll(bal & ((1ull << (lane & ~1)) - 1ull));  // both lanes of a pair get the env's rank
            nstore = __popcll(bal);
            if (lane == 0 && nstore > 0) base = atomicAdd((unsigned long long*)A.o.total, (unsigned long long)nstore);
        }
        __syncthreads();  // actions of all rows in s_act
        STAMP();
        if (wave == 0 && lane < 2 * ROWS) {
            const int e = env_e;
            const bool is_opp = env_opp, own = !env_opp;
            const bool active = e < nrow;
            const int64_t i = (int64_t)r0 + e;
            Stepper<true>& T = envT;
            float4 act = {0.f, 0.f, 0.f, 0.f};
            bool bad_act = false;
            if (active) {
                act = *reinterpret_cast<const float4*>(s_act + e * 4);
                bad_act = sanitize_action(act);
            }
            Wrapped W{};
            V3 eu{}, eu2{};
            bool ended = false;
            unsigned st_kill = 0, st_fs = 0, st_tl = 0, st_fire = 0, st_good = 0, st_lock = 0;
            if (active) {
                T.step(act, is_opp, eu, eu2, W);
                STAMP();
                unsigned ended_own = 0;
                if (own) {
                    ended_own = (A.o.auto_reset && (W.done || trunc)) ? 1u : 0u;
                    st_fire = (T.S.flags & HX_F_FIRED) ? 1u : 0u;
                    st_good = W.success == 1 ? 1u : 0u;
                    st_lock = (T.S.flags & HX_F_LOCKED) ? 1u : 0u;
                    st_kill = (ended_own && (T.S.flags & HX_F_EPISODE_SUCCESS)) ? 1u : 0u;
                    st_fs = (ended_own && (T.S.flags & HX_F_FIRE_SUCCESS)) ? 1u : 0u;
                    st_tl = (ended_own && !W.done) ? 1u : 0u;
                    A.reward[i] = W.reward;
                    A.done[i] = W.done ? 1 : 0;
                    A.success[i] = (int8_t)W.success;
                }
                const unsigned theirs = swap1u(ended_own);
                ended = (own ? ended_own : theirs) != 0u;
            }
            if (store) {  // row = s[13] a[4] s'[13] r done   (Transition, buffer.py:8): each lane of the pair writes its share
                float* row = s_row + rank * kRowPitch;
                if (is_opp) {
#pragma unroll
                    for (int j = 0; j < HX_OBS_DIM; ++j) row[j] = envPrev[j];
                    row[26] = eu.x; row[27] = eu.y; row[28] = eu.z;
                } else {
                    row[13] = act.x; row[14] = act.y; row[15] = act.z; row[16] = act.w;
                    row[17] = W.o0; row[18] = W.o1; row[19] = W.o2;
                    row[20] = eu.x; row[21] = eu.y; row[22] = eu.z;
                    row[23] = W.o6; row[24] = W.o7; row[25] = W.o8;
                    row[29] = W.o12;
                    row[30] = W.reward;
                    row[31] = W.done ? 1.0f : 0.0f;
                }
            }
            // the workgroup's first ring slot, uniform across the wave
            const unsigned long long b0 = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                                          (unsigned)__builtin_amdgcn_readfirstlane((int)(base & 0xFFFFFFFFull));
            const unsigned slot0 = nstore > 0 ? ring_slot(b0, (unsigned long long)A.o.cap, A.inv_cap) : 0u;
            if (lane == 0) {
                s_base = slot0;
                s_nstore = nstore;
            }
            if (store && own && A.o.ring_success) A.o.ring_success[wrap_slot(slot0 + (unsigned)rank, (unsigned)A.o.cap)] = (int8_t)W.success;
            if (active) {
                if (ended) {
                    uint32_t epi = 0u;
                    if (own) {
                        epi = A.o.episode_ctr[i] + 1u;
                        A.o.episode_ctr[i] = epi;
                    }
                    T.reset(is_opp, A.o.randomize != 0, A.o.seed, A.o.env_id0 + (uint32_t)i, epi, eu, eu2, W);
                }
                T.store(A.state, A.stride, r0, (uint32_t)e, is_opp);
                float* out = s_obs + e * HX_OBS_DIM;
                if (own) {
                    out[0] = W.o0; out[1] = W.o1; out[2] = W.o2;
                    out[3] = eu.x; out[4] = eu.y; out[5] = eu.z;
                    out[6] = W.o6; out[7] = W.o7; out[8] = W.o8;
                    out[12] = W.o12;
                } else {
                    out[9] = eu.x; out[10] = eu.y; out[11] = eu.z;
                }
            }
            STAMP();
            if (A.o.stats) {
                const bool mine_ = active && own;
                const unsigned vals[HX_STAT_COUNT] = {(mine_ && ended) ? 1u : 0u, st_kill, st_fs, st_tl, st_fire, st_good, st_lock, mine_ ? 1u : 0u,
                                                      (mine_ && bad_act) ? 1u : 0u};
                unsigned mine = 0;
#pragma unroll
                for (int k = 0; k < HX_STAT_COUNT; ++k) {
                    const unsigned c = (unsigned)__popcll(__ballot(vals[k] != 0u));
                    if (lane == k) mine = c;
                }
                if (lane < HX_STAT_COUNT && mine) atomicAdd((unsigned long long*)&A.o.stats[(bid % HX_STAT_WAYS) * HX_STAT_PITCH + lane], (unsigned long long)mine);
            }
        }
        STAMP();
        __syncthreads();  // rows, next observations, s_base / s_nstore
        STAMP();
        for (int k = tid; k < nrow * HX_OBS_DIM; k += kWide) A.obs[(size_t)r0 * HX_OBS_DIM + k] = s_obs[k];
        const int nst = s_nstore;
        if (nst > 0) {  // 16 B per lane, rows contiguous in the ring (modulo wrap)
            const unsigned slot0 = s_base, cap = (unsigned)A.o.cap;
            float4* ring4 = reinterpret_cast<float4*>(A.o.ring);
            for (int k = tid; k < nst * (HX_ROW_WORDS / 4); k += kWide) {
                const int rr = k >> 3, c = (k & 7) * 4;
                const float* src = s_row + rr * kRowPitch + c;
                ring4[(size_t)wrap_slot(slot0 + (unsigned)rr, cap) * (HX_ROW_WORDS / 4) + (k & 7)] = make_float4(src[0], src[1], src[2], src[3]);
            }
        }
    }
    STAMP();
    STAMP_FLUSH(56, (bid == 0 || bid == 200) && tid == 0);
    SPAN_LOG(HX_SPAN_ACT);
}

}  // namespace hxact
