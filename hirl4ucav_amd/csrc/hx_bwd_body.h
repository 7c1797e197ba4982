// hx_bwd_body.h — the backward workgroup of the update's 256 <-> 512 layer (bwd_l2) as a device function over an explicit LDS block and explicit
// workgroup coordinates (gfx950): bwd_l2_kernel (hx_fwdbwd.hip) is a thin wrapper; the front launch (hx_front.hip) runs the TD jobs (launch C of
// learn(): y, loss, dq, LN2 backward, dh1 — HIRL.py:270-286) as further workgroups behind launches A and B.
#pragma once
#include "hx_fwd_body.h"

namespace hxu {

// compact kernel argument (see FwdJobC): 128 bytes per job, two s_load_dwordx16, job = blockIdx.y.  The heads a mode does not
// use are simply not expanded (t1 / t2 for the TD jobs, `soft` for the critic's policy job, `crit` for the actor's policy job).
struct BwdJobC {
    const float* net; float* ws;
    const float* h1_net; float* h1_ws;   // TD: t1; CRITIC_PI: soft; ACTOR_PI: crit
    const float* h2_net; float* h2_ws;   // TD: t2
    const float* src; const float* bonus; const float* bonus_scale;
    uint32_t cfg;   // m:10 | h1.m:10 | h2.m:10
    uint32_t cfg2;  // mode:3 | loss_slot:3
    int32_t rows;
    float gamma, lambda, slope, inv_batch;
    float* losses; int* soft_count;
    uint32_t img_t;  // BF16 instantiations: index of this net's transposed W2 image
};
static_assert(sizeof(BwdJobC) == 128, "two s_load_dwordx16");
struct BwdArgsC {
    BwdJobC job[2];
    const uint16_t* images;  // BF16 instantiations: base of the bf16 W2 images
    int rowmap;              // 1: row tiles -> XCDs as in fwd_l2 (see there)
};
inline BwdJobC pack_bwd(const BwdJob& J, const BwdArgs& A) {
    BwdJobC c{};
    c.net = J.net; c.ws = J.ws.x;
    const Head& h1 = (J.mode == BM_CRITIC_TD || J.mode == BM_SAC_QMIN || J.mode == BM_SAC_POLICY) ? J.t1 : (J.mode == BM_CRITIC_PI ? J.soft : J.crit);
    c.h1_net = h1.net; c.h1_ws = h1.ws.x;
    c.h2_net = J.t2.net; c.h2_ws = J.t2.ws.x;
    c.src = J.src.main; c.bonus = J.bonus; c.bonus_scale = J.bonus_scale;
    c.cfg = mlp_bits(J.m) | (mlp_bits(h1.m) << 10) | (mlp_bits(J.t2.m) << 20);
    c.cfg2 = (uint32_t)J.mode | ((uint32_t)J.loss_slot << 3);
    c.rows = J.rows; c.gamma = J.gamma; c.lambda = J.lambda; c.slope = A.slope; c.inv_batch = A.inv_batch;
    c.losses = A.losses; c.soft_count = A.soft_count; c.img_t = (uint32_t)J.img_t;
    return c;
}
__device__ __forceinline__ BwdJob expand_bwd(const BwdJobC& c) {
    BwdJob J;
    J.net = c.net; J.m = mlp_of(c.cfg & 1023u); J.ws = carve_slot(c.ws, c.rows); J.rows = c.rows;
    J.mode = (int)(c.cfg2 & 7u); J.loss_slot = (int)((c.cfg2 >> 3) & 7u);
    const Head h1{c.h1_net, mlp_of((c.cfg >> 10) & 1023u), carve_slot(c.h1_ws, c.rows)};
    J.t1 = h1; J.soft = h1; J.crit = h1;
    J.t2 = Head{c.h2_net, mlp_of((c.cfg >> 20) & 1023u), carve_slot(c.h2_ws, c.rows)};
    J.src = RowSrc{c.src, nullptr, nullptr, 0, 32};
    J.gamma = c.gamma; J.lambda = c.lambda; J.bonus = c.bonus; J.bonus_scale = c.bonus_scale;
    return J;
}

// one LN1-backward row sum from the per-workgroup partials bwd_l2 left in lnp ([kColWgB][2], stride 2): fixed-order tree
__device__ __forceinline__ float lnp_sum(const float* lp) {
    float v[kColWgB];
#pragma unroll
    for (int c = 0; c < kColWgB; ++c) v[c] = lp[2 * c];
#pragma unroll
    for (int w = 1; w < kColWgB; w *= 2)
#pragma unroll
        for (int c = 0; c < kColWgB; c += 2 * w) v[c] += v[c + w];
    return v[0];
}

// The workgroup's LDS as ONE object (the kernel declares it; hx_front.hip overlays it with the acting workgroups' block in a union).
template <int GRP, bool BF16>
struct BwdLds {
    static constexpr int IMG = (GRP == 3 || GRP == 5) ? 8 : 4;
    static constexpr bool PAIRED = GRP <= 2 || GRP >= 4;
    __attribute__((aligned(16))) float dz2s[BF16 ? kCTB * RT * 2 : RT * LDA2];  // (BF16: only the epilogue's row-sum scratch)
    __attribute__((aligned(16))) __bf16 dz2b[BF16 ? RT * LDB2 : 8];
    __attribute__((aligned(16))) float kred[(kKSB - 1) * kCTB * 256];  // split-K partial tiles
    __attribute__((aligned(16))) float hps[(GRP == 0 ? 3 : (GRP == 4 ? 2 : 1)) * HeadImage<IMG>::kStride];
    __attribute__((aligned(16))) float c1s[GRP == 2 ? H1 * 6 : (GRP == 5 ? H1 * 8 : 4)];  // critic layer 1: g1 be1 W1[:,13..16]  (GRP 5: W1[:,13..16] of both critics)
    float red[16][4];
    float st1s[RT * 2];  // LN1 stats of the tile's rows (epilogue)
    float tq[PAIRED ? (RT / 2) * 4 : 1];
};

// GRP 0: BM_CRITIC_TD jobs, 1: BM_CRITIC_PI, 2: BM_ACTOR_PI / BM_ACTOR_BC, 3: BM_GIVEN — the head gradient was written to ws.dout
// by an earlier kernel, heads up to 8 wide (SAC) (one instantiation per launch keeps the register
// footprint of each below 128 at 16 waves per workgroup); 4: BM_SAC_QMIN — SAC's min(Q1, Q2)(s, a~) selection in the prologue (role 1: the
// other critic's head) instead of a launch of its own; 5: BM_SAC_POLICY — the policy's head gradient in the prologue (role 1: dL/da from
// both critics' layer-1 backward), 8-wide head.
// Latency structure (what matters at B = 128, one workgroup per CU): EVERY global load of the workgroup — the W2 fragment
// of the MFMA phase, the z2 rows, labels, the other nets' rows, all head parameters, the epilogue's z1 — is issued at
// entry; there is ONE wait; head parameters are shared through LDS; the rest runs out of registers and LDS.
// BF16 (the bf16 update path): dh1 = dz2 W2 on v_mfma_f32_16x16x32_bf16 — every row's dz2 is rounded to bf16 once into the LDS tile (the copy
// published for wgrad stays fp32), the wave's B fragments come straight from the net's TRANSPOSED bf16 image (two 16-byte loads per lane
// instead of sixteen strided dword loads); heads, losses, LayerNorm backward and the epilogue stay fp32.
template <int GRP, bool RELU, bool BF16, int FRONT>
__device__ __forceinline__ void bwd_l2_body(const BwdArgsC& AC, const int bx, const int by, BwdLds<GRP, BF16>& SL, const FrontSync& X) {
    static_assert(FRONT == 0 || (FRONT == 3 && GRP == 0), "in-launch consumer: the TD jobs (launch C)");
    float* const dz2s = SL.dz2s;
    __bf16* const dz2b = SL.dz2b;
    float* const kred = SL.kred;
    constexpr int IMG = (GRP == 3 || GRP == 5) ? 8 : 4;  // head width of this instantiation's LDS images
    // head width known at compile time: the critic jobs (GRP 0, 1) have ONE output, the actor jobs (GRP 2) four; GRP 3 (SAC: policy 8 wide,
    // Q heads 1) keeps the run-time width.  A run-time trip count over dout[] costs a select chain per step (no indexed registers).
    constexpr int NOUT = (GRP <= 1 || GRP == 4) ? 1 : (GRP == 2 ? 4 : (GRP == 5 ? 8 : 0));
    constexpr int OUTW = NOUT ? NOUT : IMG;
    typedef HeadImage<IMG> Img;
    constexpr int kHpStride = Img::kStride;
    float* const hps = SL.hps;
    float* const c1s = SL.c1s;
    auto& red = SL.red;
    float* const st1s = SL.st1s;
    // TD job (GRP 0): EIGHT rows per workgroup, a wave PAIR per row — wave w (role 0) owns the row's own head, loss gradient and LN2
    // backward, wave w + 8 (role 1) the two target heads; min(Q1', Q2') crosses through LDS.  Twice the workgroups (256 at B = 128, two
    // jobs): half the row bytes per CU (rows the previous launches produced on all eight XCDs arrive at ~19 B/clk/CU, and 96 KB of them
    // were in front of this prologue), and the three heads of a row no longer run one after the other on one wave.  The MFMA tile keeps
    // its 16 rows (8 of them zero): that phase is the short one.
    // The critic-PI job (GRP 1) pairs the same way (role 1: the soft head), the actor jobs (GRP 2) too (role 1: dL/da from the critic's
    // layer-1 backward, four dot products over 256 hidden units).
    constexpr bool PAIRED = GRP <= 2 || GRP >= 4;
    constexpr int RTB = PAIRED ? RT / 2 : RT;
    float* const tq = SL.tq;

    const int b = bx;
    const BwdJobC& jc = AC.job[by];
    const BwdJob J = expand_bwd(jc);
    struct { float slope, inv_batch; float* losses; int* soft_count; } A{jc.slope, jc.inv_batch, jc.losses, jc.soft_count};
    // rowmap (see fwd_l2): the workgroups of the 16-row tile t of the forward launches run on XCD t % 8.  PAIRED kernels split it into two
    // 8-row tiles 2t, 2t + 1: b = x + 8 k -> row tile 2 x + (k & 1), column workgroup k >> 1 (B = 128: 16 tiles of 8 rows; any other row
    // count keeps the plain order)
    int rt = b / kColWgB, nt = b % kColWgB;
    if (AC.rowmap) {
        const int ntile_ = (jc.rows + RTB - 1) / RTB;
        if (PAIRED && ntile_ == 16) { rt = 2 * (b & 7) + ((b >> 3) & 1); nt = b >> 4; }
        else if (!PAIRED) { rt = b % ntile_; nt = b / ntile_; }
    }
    const int r0 = rt * RTB;
    const int nrow = min(RTB, J.rows - r0);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float slope = A.slope;
    const bool lead = nt == 0;  // the column-tile-0 workgroup of a row tile also publishes dz2 / st2 / dout / losses
    const int role = PAIRED ? wave / RTB : 0, prow = PAIRED ? wave % RTB : wave;
    const bool live = role == 0 && prow < nrow;   // wave owns row prow of the tile (own head, gradient, LN2 backward)
    const bool tlive = role == 1 && prow < nrow;  // PAIRED: wave owns the target heads of row prow
    const size_t R = (size_t)(r0 + (prow < nrow ? prow : 0));
    const int ct = wave % kCTB, kq = wave / kCTB;
    const int n0 = nt * kNTB + ct * 16;
    STAMP_DECL;
    STAMP();

    // ---------------- issue phase ----------------
    BFrag<BF16 ? 16 : H2 / kKSB> bfrag;   // (BF16: unused)
    constexpr int NSLB = (H2 / kKSB) / 32;  // BF16: 32-wide slabs of n per wave (K = 512 over the kKSB K-parts)
    uint4 bqb[BF16 ? NSLB : 1];
    // Row loads are UNCONDITIONAL (R is clamped to a valid row; a wave without a row never uses them): behind `if (live)` the compiler
    // zero-fills the registers, loads under a branch and — where the two versions merge — WAITS for the loads in the middle of the issue phase.
    RowReg<H2> z, za, zb;
    float lab0 = 0.f, lab1 = 0.f, tgt[4] = {0.f, 0.f, 0.f, 0.f};
    RowReg<H1> cdh, cz, cdh2, cz2;  // (cdh2 / cz2: GRP 5, the second critic)
    float cst0 = 0.f, cst1 = 0.f, cs1 = 0.f, cs2 = 0.f;
    float paux[GRP == 5 ? 13 : 1] = {};  // GRP 5: a[4], sigma eps[4], clamp mask[4], entropy of the row (the Gaussian head's aux row)
    // PAIRED: role 0 asks for the row of its own net (z), role 1 for the two target nets' (za, zb); both behind one scalar branch each, the
    // skipped registers left unset (never used by that role)
    const bool role1 = PAIRED && __builtin_amdgcn_readfirstlane(wave) >= RTB;
    if (FRONT != 3 && !role1) z.load(J.ws.z2 + R * H2);  // (FRONT 3: the rows this launch produces are asked for behind the wait below)
    Img pv0, pv1, pv2;
    pv0.fetch(J.net, J.m, tid);
    float bonus = 0.f, bonus_scale = 0.f, dgiv[GRP == 3 ? 8 : 1] = {};
    if (GRP == 0) {
        if (FRONT != 3 && role1) {
            za.load(J.t1.ws.z2 + R * H2);
            zb.load(J.t2.ws.z2 + R * H2);
        }
        const float* row = src_row(J.src, (int)R);
        lab0 = row[30];
        lab1 = row[31];
        pv1.fetch(J.t1.net, J.t1.m, tid);
        pv2.fetch(J.t2.net, J.t2.m, tid);
        if (J.bonus) {  // SAC: + alpha * entropy(s')  SAC/agent.py:205-206 (multiplied where it is used: no wait here)
            bonus = J.bonus[R];
            bonus_scale = J.bonus_scale[threadIdx.x & 0];
        }
    }
    if (GRP == 3) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) dgiv[jj] = J.ws.dout[R * OW + jj];  // (the row pitch is 8: all in bounds; masked below)
    }
    if (GRP == 1) {
        if (role1 && J.soft.net) za.load(J.soft.ws.z2 + R * H2);  // same net as J.net: shares the LDS image
    }
    if (GRP == 4) {  // the other critic on (s, a~): its own LDS image
        if (role1) za.load(J.t1.ws.z2 + R * H2);
        pv1.fetch(J.t1.net, J.t1.m, tid);
    }
    float c1v[2] = {0.f, 0.f};
    if (GRP == 5) {
        if (role1) {  // the pair's second wave owns dL/da: both critics' dh1 and z1 rows
            cdh.load(J.t1.ws.dh1 + R * H1);
            cz.load(J.t1.ws.z1 + R * H1);
            cdh2.load(J.t2.ws.dh1 + R * H1);
            cz2.load(J.t2.ws.z1 + R * H1);
        } else {
#pragma unroll
            for (int jj = 0; jj < 13; ++jj) paux[jj] = J.bonus[R * 16 + jj];
            bonus_scale = J.bonus_scale[threadIdx.x & 0];  // alpha
        }
        c1v[0] = J.t1.net[J.t1.m.W1() + (tid >> 2) * J.t1.m.in + 13 + (tid & 3)];  // W1[k][13..16] of each critic, one float per thread
        c1v[1] = J.t2.net[J.t2.m.W1() + (tid >> 2) * J.t2.m.in + 13 + (tid & 3)];
    }
    if (GRP == 2) {
        if (J.mode == BM_ACTOR_PI) {
            const Head& C = J.crit;
            if (role1) {  // the pair's second wave owns dL/da
                cdh.load(C.ws.dh1 + R * H1);
                cz.load(C.ws.z1 + R * H1);
                cst0 = C.ws.st1[R * 2];
                cst1 = C.ws.st1[R * 2 + 1];
                const float* lp = C.ws.lnp + R * (2 * kColWgB);
                cs1 = lnp_sum(lp) * (1.0f / H1);
                cs2 = lnp_sum(lp + 1) * (1.0f / H1);
            }
            // g1 | be1 (512 floats) by threads 0..511; W1[k][13..16] (1024 floats) one per thread
            if (tid < 2 * H1) c1v[0] = C.net[C.m.g1() + tid];
            c1v[1] = C.net[C.m.W1() + (tid >> 2) * C.m.in + 13 + (tid & 3)];
        } else {
            const float* row = src_row(J.src, (int)R);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) tgt[jj] = row[13 + jj];
        }
    }
    // epilogue operands of the waves that finish the tile (kq == 0): z1, g1, be1 of their 4 rows x 1 column; LN1 stats via LDS
    constexpr bool kEpiPrefetch = true;  // (with 64-column workgroups the TD instantiation had no registers to spare for this)
    float ez1[4] = {0.f, 0.f, 0.f, 0.f}, eg1 = 0.f, ebe1 = 0.f;
    if (kEpiPrefetch && kq == 0) {
        const int r = lane & 15, g = lane >> 4;
        eg1 = J.net[J.m.g1() + n0 + r];
        ebe1 = J.net[J.m.be1() + n0 + r];
        if (FRONT != 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = 4 * g + q;
                ez1[q] = J.ws.z1[(size_t)(r0 + (row < nrow ? row : 0)) * H1 + n0 + r];  // unconditional, clamped (rows >= nrow are never stored)
            }
        }
    }
    float st1v = tid & 1 ? 1.0f : 0.0f;
    if (FRONT != 3 && tid < nrow * 2) st1v = J.ws.st1[(size_t)r0 * 2 + tid];
    // the W2 fragment of the MFMA phase (16 loads per lane, needed last) goes out behind the prologue's operands, not in front of them
    if constexpr (BF16) {  // block (tile of 16 columns of dh1, slab of 32 n) of the transposed image: 512 elements, lane l's 16 bytes at + 8 l
        const uint16_t* blk = AC.images + (size_t)jc.img_t * kImgElems + (size_t)((nt * kCTB + ct) * 16 + kq * NSLB) * 512 + lane * 8;
#pragma unroll
        for (int i = 0; i < NSLB; ++i) bqb[i] = *reinterpret_cast<const uint4*>(blk + i * 512);
    } else {
        bfrag.load(J.net + J.m.W2() + (size_t)(kq * (H2 / kKSB)) * H1 + n0 + (lane & 15), H1);
    }
    if constexpr (FRONT == 3) {
        // Launch C inside the front launch: everything requested so far is older than this launch (weights, labels, the pre-drawn tiles).  The z2 rows of
        // this job's net and its z1 / LN1 statistics come from launch A's workgroups of THIS launch, the target critics' z2 rows from launch B's — they
        // count themselves in per 16-row tile (fwd_l2_body, X.with_c); row tile 0's target-actor counter stands for the cleared loss accumulators.
        const int ft = r0 / RT;
        if (tid == 0) {
            const unsigned* fo = X.flags + 16 * (1 + by) + ft;
            const unsigned* ft2 = X.flags + 48 + ft;
            int spins = 0;
            while ((int)(__hip_atomic_load(fo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - X.c_target) < 0 ||
                   (int)(__hip_atomic_load(ft2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - X.t_target) < 0 ||
                   (int)(__hip_atomic_load(X.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - X.arrive) < 0) {
                if (++spins > 4000000) { atomicOr(X.status, 2u); break; }  // (launch C riding: the waiters can outnumber the CUs — include/hirl4ucav.h hx_hirl_front; opt-in, one process per GPU)
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (!role1) z.load_agent(J.ws.z2 + R * H2);
        else {
            za.load_agent(J.t1.ws.z2 + R * H2);
            zb.load_agent(J.t2.ws.z2 + R * H2);
        }
        if (kEpiPrefetch && kq == 0) {
            const int r = lane & 15, g = lane >> 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = 4 * g + q;
                ez1[q] = ld_agent(J.ws.z1 + (size_t)(r0 + (row < nrow ? row : 0)) * H1 + n0 + r);
            }
        }
        if (tid < nrow * 2) st1v = ld_agent(J.ws.st1 + (size_t)r0 * 2 + tid);
    }
    // ---------------- one wait: publish the shared operands in LDS ----------------
    if (tid < RT * 2) st1s[tid] = st1v;
    pv0.store(hps, J.net, J.m, tid);
    if (GRP == 0) {
        pv1.store(hps + kHpStride, J.t1.net, J.t1.m, tid);
        pv2.store(hps + 2 * kHpStride, J.t2.net, J.t2.m, tid);
    }
    if (GRP == 2 && J.mode == BM_ACTOR_PI) {
        if (tid < 2 * H1) c1s[tid] = c1v[0];
        c1s[2 * H1 + tid] = c1v[1];
    }
    if (GRP == 4) pv1.store(hps + kHpStride, J.t1.net, J.t1.m, tid);
    if (GRP == 5) {
        c1s[tid] = c1v[0];
        c1s[4 * H1 + tid] = c1v[1];
    }
    __syncthreads();
    STAMP();

    // ---------------- prologue: head, loss gradient, LN2 backward (registers + LDS only) ----------------
    float part[4] = {0.f, 0.f, 0.f, 0.f};  // loss partials of this row
    int cnt = 0;
    float* drow = dz2s + (BF16 ? 0 : wave * LDA2);  // fp32 only (BF16 rows go to dz2b through store_row_bf16)
    // BF16: lane's elements n = (i * 64 + lane) * 4 + c of the row -> four bf16 (8 bytes) at [wave][n]
    auto store_row_bf16 = [&](const RowReg<H2>& v) {
        typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const v4bf q = {(__bf16)v.v[4 * i], (__bf16)v.v[4 * i + 1], (__bf16)v.v[4 * i + 2], (__bf16)v.v[4 * i + 3]};
            *reinterpret_cast<uint2*>(dz2b + wave * LDB2 + (i * 64 + lane) * 4) = __builtin_bit_cast(uint2, q);
        }
    };
    RowReg<H2> xh, y;
    float mean = 0.f, rstd = 0.f, o[OUTW] = {};
    if constexpr (PAIRED) {  // the pair's two halves side by side, then one barrier
        if (tlive) {
            if constexpr (GRP == 0) {
                RowReg<H2> xa, ya;
                float m1, s1, q1[1], q2[1];
                head_regs<1, IMG, RELU>(za, hps + kHpStride, 1, slope, xa, ya, m1, s1, q1, J.t1.m.no_ln);
                head_regs<1, IMG, RELU>(zb, hps + 2 * kHpStride, 1, slope, xa, ya, m1, s1, q2, J.t2.m.no_ln);
                if (lane == 0) tq[prow] = fminf(q1[0], q2[0]);
            } else if constexpr (GRP == 1) {
                if (J.soft.net) {
                    RowReg<H2> xa, ya;
                    float m1, s1, qs[1];
                    head_regs<1, IMG, RELU>(za, hps, 1, slope, xa, ya, m1, s1, qs, J.m.no_ln);
                    if (lane == 0) tq[prow] = qs[0];
                }
            } else if constexpr (GRP == 4) {
                RowReg<H2> xa, ya;
                float m1, s1, qo[1];
                head_regs<1, IMG, RELU>(za, hps + kHpStride, 1, slope, xa, ya, m1, s1, qo, J.t1.m.no_ln);
                if (lane == 0) tq[prow] = qo[0];
            } else if constexpr (GRP == 5) {
                // dL/da_j = sum over both critics and their 256 hidden units of dz1[k] W1[k][13 + j]; plain stacks: dz1 = dh1 relu'(z1)
                // (policy_dout_kernel's order: critic 1's four units of the lane, then critic 2's, then the wave sums)
                float da[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int hsel = 0; hsel < 2; ++hsel) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int k = lane * 4 + c;
                        const float dz1 = act_bwd<true>(hsel ? cdh2.v[c] : cdh.v[c], hsel ? cz2.v[c] : cz.v[c], 0.0f);
                        const float4 w4 = *reinterpret_cast<const float4*>(c1s + hsel * 4 * H1 + 4 * k);
                        da[0] += dz1 * w4.x; da[1] += dz1 * w4.y; da[2] += dz1 * w4.z; da[3] += dz1 * w4.w;
                    }
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) da[jj] = wave_sum(da[jj]);
                if (lane < 4) tq[prow * 4 + lane] = lane == 0 ? da[0] : lane == 1 ? da[1] : lane == 2 ? da[2] : da[3];
            } else if (J.mode == BM_ACTOR_PI) {
                // dL/da_j = sum_k dz1_c[k] W1c[k][13 + j], dz1_c = LN1 backward of the critic's dh1 (row sums from lnp)
                float da[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c) {  // RowReg<256> maps v[c] -> hidden unit k = lane*4 + c
                    const int k = lane * 4 + c;
                    const float g1 = c1s[k], be1 = c1s[H1 + k];
                    const float xh1 = (cz.v[c] - cst0) * cst1;
                    const float dxh = act_bwd<RELU>(cdh.v[c], g1 * xh1 + be1, slope) * g1;
                    const float dz1 = cst1 * (dxh - cs1 - xh1 * cs2);
                    const float4 w4 = *reinterpret_cast<const float4*>(c1s + 2 * H1 + 4 * k);
                    da[0] += dz1 * w4.x; da[1] += dz1 * w4.y; da[2] += dz1 * w4.z; da[3] += dz1 * w4.w;
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) da[jj] = wave_sum(da[jj]);
                if (lane < 4) tq[prow * 4 + lane] = lane == 0 ? da[0] : lane == 1 ? da[1] : lane == 2 ? da[2] : da[3];
            }
        } else if (live) {
            head_regs<OUTW, IMG, RELU>(z, hps, NOUT ? NOUT : J.m.out, slope, xh, y, mean, rstd, o, J.m.no_ln);
        }
        __syncthreads();
    }
    if (!live) {  // padded rows (and the target-head waves' rows 8..15 of the MFMA tile) contribute zeros
        RowReg<H2> zero;
#pragma unroll
        for (int i = 0; i < 8; ++i) zero.v[i] = 0.0f;
        if constexpr (BF16) store_row_bf16(zero);
        else zero.store_lds(drow);
    } else {
        if constexpr (!PAIRED) head_regs<OUTW, IMG, RELU>(z, hps, NOUT ? NOUT : J.m.out, slope, xh, y, mean, rstd, o, J.m.no_ln);
        float dout[OUTW] = {};
        if constexpr (GRP == 3) {  // head gradient supplied by a previous kernel (SAC: min-selected critics, sampled policy)
#pragma unroll
            for (int jj = 0; jj < OUTW; ++jj) dout[jj] = dgiv[jj < (GRP == 3 ? 8 : 1) ? jj : 0];
        } else if constexpr (GRP == 0) {
            const float qmin = tq[prow];  // min(Q1', Q2') of this row, from the pair's other wave
            // HIRL.py:270-274; with `bonus` SAC's r + (1 - d) gamma (min Q' + alpha H')  SAC/agent.py:202-210
            const float target = J.bonus ? lab0 + (1.0f - lab1) * (J.gamma * (qmin + bonus * bonus_scale))
                                         : lab0 + (J.gamma * qmin) * (1.0f - lab1);
            const float diff = o[0] - target;
            dout[0] = 2.0f * diff * A.inv_batch;  // d mse / dq
            part[0] += diff * diff * A.inv_batch;
        } else if constexpr (GRP == 4) {
            // torch.min(Q1, Q2)'s subgradient: the smaller head takes -1 / B, a tie is shared (q_select_kernel)   SAC/agent.py:380-383
            const float other = tq[prow];
            const float w = o[0] < other ? 1.0f : (o[0] == other ? 0.5f : 0.0f);
            dout[0] = -w * A.inv_batch;
            if (J.loss_slot == 0) part[2] += -fminf(o[0], other) * A.inv_batch;  // the -min(Q) / B part of the policy loss (logged), once per row
        } else if constexpr (GRP == 5) {
            const float ab = bonus_scale * A.inv_batch;  // alpha / B
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) sac_policy_dout(tq[prow * 4 + jj], paux[jj], paux[4 + jj], paux[8 + jj], ab, dout[jj], dout[4 + jj]);
            part[2] += -bonus_scale * paux[12] * A.inv_batch;  // -alpha H / B (logged)
        } else if constexpr (GRP == 1) {
            dout[0] = -A.inv_batch;            // rl_loss = -mean(Q1(s, pi(s)))  HIRL.py:297
            part[3] += -o[0] * A.inv_batch;
            if (J.soft.net) cnt += (tq[prow] > o[0]) ? 1 : 0;  // (soft_Q > rl_Q)  HIRL.py:303 — the soft head came from the pair's other wave
        } else if (J.mode == BM_ACTOR_PI) {
            // dL/da (the pair's other wave) through the policy's tanh
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float a = fast_tanh(o[jj]);
                dout[jj] = tq[prow * 4 + jj] * (1.0f - a * a);
            }
        } else {  // BM_ACTOR_BC: bc_loss = lambda * mse(actor(s_bc), a_bc)  HIRL.py:310-311
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float a = fast_tanh(o[jj]);
                const float diff = a - tgt[jj];
                dout[jj] = (2.0f * J.lambda * 0.25f * A.inv_batch) * diff * (1.0f - a * a);
                part[2] += J.lambda * 0.25f * A.inv_batch * diff * diff;
                if (jj == 3) part[1] += J.lambda * A.inv_batch * diff * diff;  // bc_fire_loss (logging), HIRL.py:317-319
            }
        }
        // dh2 = dout W3, through act' and LN2 backward
        RowReg<H2> g, dx;
        g.load(hps);
#pragma unroll
        for (int i = 0; i < 8; ++i) dx.v[i] = 0.0f;
        if (NOUT) {
#pragma unroll
            for (int jj = 0; jj < OUTW; ++jj) {
                RowReg<H2> w;
                w.load(hps + (2 + jj) * H2);
#pragma unroll
                for (int i = 0; i < 8; ++i) dx.v[i] += dout[jj] * w.v[i];
            }
        } else {
            for (int jj = 0; jj < J.m.out; ++jj) {
                RowReg<H2> w;
                w.load(hps + (2 + jj) * H2);
#pragma unroll
                for (int i = 0; i < 8; ++i) dx.v[i] += dout[jj] * w.v[i];
            }
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            dx.v[i] = act_bwd<RELU>(dx.v[i], y.v[i], slope) * g.v[i];
            s1 += dx.v[i];
            s2 += dx.v[i] * xh.v[i];
        }
        s1 = wave_sum(s1) * (1.0f / H2);
        s2 = wave_sum(s2) * (1.0f / H2);
        if (J.m.no_ln) s1 = s2 = 0.0f;  // identity "norm": dz2 = dy2
#pragma unroll
        for (int i = 0; i < 8; ++i) dx.v[i] = rstd * (dx.v[i] - s1 - xh.v[i] * s2);
        if constexpr (BF16) store_row_bf16(dx);
        else dx.store_lds(drow);
        if (lead) {
            dx.store(J.ws.dz2 + R * H2);
            if (lane == 0) {
                J.ws.st2[R * 2] = mean;
                J.ws.st2[R * 2 + 1] = rstd;
            }
            if constexpr (GRP == 5) {  // eight head gradients; no outv (nothing reads the policy's pre-activations downstream)
                const float d0 = dout[0], d1 = dout[1], d2 = dout[2], d3 = dout[3], d4 = dout[4], d5 = dout[5], d6 = dout[6], d7 = dout[7];
                if (lane < 8) J.ws.dout[R * OW + lane] = lane == 0 ? d0 : lane == 1 ? d1 : lane == 2 ? d2 : lane == 3 ? d3 : lane == 4 ? d4 : lane == 5 ? d5 : lane == 6 ? d6 : d7;
            } else if (GRP != 3 && lane < 4) {
                J.ws.dout[R * OW + lane] = dout[lane < OUTW ? lane : 0];
                if (J.mode != BM_ACTOR_PI) J.ws.outv[R * OW + lane] = (J.m.out == 4) ? fast_tanh(o[lane < OUTW ? lane : 0]) : o[lane < OUTW ? lane : 0];
            }
        }
    }
    if (lead && lane == 0) {
        red[wave][0] = part[0]; red[wave][1] = part[1]; red[wave][2] = part[2]; red[wave][3] = part[3];
        if (cnt) atomicAdd(A.soft_count, cnt);
    }
    __syncthreads();
    STAMP();
    if (lead && wave == 0) {  // lane -> (row w = lane & 15, partial c = lane >> 4): one DPP sum over each 16-lane row
        const int c = lane >> 4;
        const float p = sum16(red[lane & 15][c]);
        if ((lane & 15) == 0) {
            if (J.mode == BM_CRITIC_TD && c == 0) atomicAdd(&A.losses[J.loss_slot], p);
            if (J.mode == BM_CRITIC_PI && c == 3) atomicAdd(&A.losses[3], p);
            if (J.mode == BM_ACTOR_BC && c == 2) atomicAdd(&A.losses[2], p);
            if (J.mode == BM_ACTOR_BC && c == 1) atomicAdd(&A.losses[4], p);
            if ((GRP == 4 || GRP == 5) && c == 2 && p != 0.0f) atomicAdd(&A.losses[2], p);  // SAC: the policy loss's logged parts
        }
    }
    if (GRP == 5 && bx == 0 && by == 0 && wave == 1) {
        // The mean entropy feeds the log-alpha step, so it must not depend on arrival order: the per-row entropies were written by the
        // Gaussian head's launch, one wave adds them in a fixed order (policy_dout_kernel's).
        float s = 0.0f;
        for (int rr = lane; rr < J.rows; rr += 64) s += J.bonus[(size_t)rr * 16 + 12];
        s = wave_sum(s);
        if (lane == 0) A.losses[4] = s * A.inv_batch;
    }
    // ---------------- dh1 tile on fp32 MFMA: wave = (column tile ct, K quarter kq), K = 512 ----------------
    {
        const int r = lane & 15, g = lane >> 4;
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (BF16) {
#pragma unroll
            for (int i = 0; i < NSLB; ++i) acc = mfma16_bf16(*reinterpret_cast<const uint4*>(dz2b + r * LDB2 + 32 * (kq * NSLB + i) + 8 * g), bqb[i], acc);
        } else {
            acc = tile_a_lds_b_frag<BF16 ? 16 : H2 / kKSB>(dz2s + kq * (H2 / kKSB), LDA2, bfrag, acc);
        }
        if (kq) *reinterpret_cast<float4*>(kred + (((kq - 1) * kCTB + ct) * 64 + lane) * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        __syncthreads();  // partial tiles visible; dz2s is dead from here on
        STAMP();
        float* ps = dz2s;  // reused as [kCTB column tiles][16 rows][2]
        if (!kEpiPrefetch && kq == 0) {
            eg1 = J.net[J.m.g1() + n0 + r];
            ebe1 = J.net[J.m.be1() + n0 + r];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (4 * g + q < nrow) ez1[q] = J.ws.z1[(size_t)(r0 + 4 * g + q) * H1 + n0 + r];
        }
        if (kq == 0) {
            // epilogue: store dh1 and this tile's share of the LN1-backward row sums (consumed by wgrad / the actor's
            // backward, which then need no cross-column reduction of their own)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float part[kKSB];  // fixed-order tree over the K-parts
                part[0] = acc[q];
#pragma unroll
                for (int k = 1; k < kKSB; ++k) part[k] = kred[(((k - 1) * kCTB + ct) * 64 + lane) * 4 + q];
#pragma unroll
                for (int w = 1; w < kKSB; w *= 2)
#pragma unroll
                    for (int k = 0; k < kKSB; k += 2 * w) part[k] += part[k + w];
                const float v = part[0];
                const int row = 4 * g + q;
                float p1 = 0.0f, p2 = 0.0f;
                if (row < nrow) {
                    J.ws.dh1[(size_t)(r0 + row) * H1 + n0 + r] = v;
                    const float xh = (ez1[q] - st1s[row * 2]) * st1s[row * 2 + 1];
                    const float dxh = act_bwd<RELU>(v, eg1 * xh + ebe1, slope) * eg1;
                    p1 = J.m.no_ln ? 0.0f : dxh;
                    p2 = J.m.no_ln ? 0.0f : dxh * xh;
                }
                const float a1 = sum16(p1), a2 = sum16(p2);
                if (r == 0) {
                    ps[(ct * RT + row) * 2] = a1;
                    ps[(ct * RT + row) * 2 + 1] = a2;
                }
            }
        }
        __syncthreads();
        if (tid < nrow * 2) {
            float v = ps[tid];
#pragma unroll
            for (int c = 1; c < kCTB; ++c) v += ps[c * RT * 2 + tid];
            J.ws.lnp[(size_t)(r0 + (tid >> 1)) * (2 * kColWgB) + nt * 2 + (tid & 1)] = v;
        }
        STAMP();
        STAMP_FLUSH(16, bx == 3 && tid == 0);
        SPAN_LOG(HX_SPAN_BWD);
    }
}

}  // namespace hxu
