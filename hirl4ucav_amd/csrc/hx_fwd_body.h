// hx_fwd_body.h — the forward workgroup of the update's 256 -> 512 layer (fwd_l2_kernel of hx_fwdbwd.hip) as a device function over an explicit
// LDS block and explicit workgroup coordinates, so that hx_front.hip can run launches A and B of learn() as workgroups of the act + env launch.
#pragma once
#include "hx_update.h"

namespace hxu {

// What the kernel actually receives: 64 bytes per job (ONE s_load_dwordx16), the job picked by by.  A kernel argument
// block of 1.7 KB read field by field behind branches cost a chain of 6-8 dependent scalar-load round trips before the first
// vector load went out (~1.5-2 us of a ~10 us launch); the compact form is one round trip, and everything else is scalar ALU.
struct FwdJobC {
    const float* net; const float* src; const float* noise; const float* prev_net;
    float* ws; float* prev_ws;
    uint32_t cfg;  // m:10 | prev.m:10 | act_mode:2 | save:1 | col0:6 | img:3
    int32_t rows;
    float noise_clamp;
    float slope;   // (per launch; carried in every job so that the job's own 64 bytes are all a workgroup waits for)
};
static_assert(sizeof(FwdJobC) == 64, "one s_load_dwordx16");
struct FwdArgsC {
    FwdJobC job[6];
    float slope;
    int zero_nf;
    float* zero_f;
    int* zero_i;
    const uint16_t* images;  // BF16 instantiations: base of the bf16 W2 images (one more scalar load beside the job's own, not behind it)
    int rowmap;              // 1: bx -> (row tile = b % tiles, column workgroup = b / tiles): a row tile's workgroups share an XCD
};
inline FwdJobC pack_fwd(const FwdJob& J) {
    FwdJobC c{};
    c.net = J.net; c.src = J.src.main; c.noise = J.noise; c.prev_net = J.prev.net;
    c.ws = J.ws.x; c.prev_ws = J.prev.ws.x;
    c.cfg = mlp_bits(J.m) | (mlp_bits(J.prev.m) << 10) | ((uint32_t)J.act_mode << 20) | ((uint32_t)(J.save ? 1 : 0) << 22) | ((uint32_t)J.col0 << 23) |
            ((uint32_t)(J.img & 7) << 29);
    c.rows = J.rows; c.noise_clamp = J.noise_clamp;
    return c;
}
__device__ __forceinline__ FwdJob expand_fwd(const FwdJobC& c) {
    FwdJob J;
    J.net = c.net; J.m = mlp_of(c.cfg & 1023u);
    J.src = RowSrc{c.src, nullptr, nullptr, 0, 32};
    J.col0 = (int)((c.cfg >> 23) & 63u); J.img = (int)(c.cfg >> 29); J.act_mode = (int)((c.cfg >> 20) & 3u);
    J.prev.net = c.prev_net; J.prev.m = mlp_of((c.cfg >> 10) & 1023u); J.prev.ws = carve_slot(c.prev_ws, c.rows);
    J.noise = c.noise; J.noise_clamp = c.noise_clamp;
    J.ws = carve_slot(c.ws, c.rows);
    J.rows = c.rows; J.save = (int)((c.cfg >> 22) & 1u);
    return J;
}

__device__ __forceinline__ int tiles_of(int rows) { return (rows + RT - 1) / RT; }

// NT = 64 / 32: columns per workgroup in latency mode (B = 128): CT = NT/16 column tiles x KS = 16/CT K-parts over the 16 waves,
//                partial sums meet in LDS.  64 when the launch has three or more nets (192+ workgroups), 32 for one or two nets
//                (then 128-256 workgroups still run in one round and each carries half the MFMA work).
// NT = 256     : one 16-column tile per wave, full K (throughput mode, thousands of rows: the prologue is recomputed 2x per
//                row tile instead of 8x or 16x)
struct NoSample {};
// SAMPLE (launch A of hx_hirl_*_sampled, batch <= 256): the minibatch is drawn here (draw_fused) and every workgroup gathers its 16 rows
// straight from the replay / expert rings; the workgroups of job 0 also leave the row tiles, the indices and the smoothing noise for the
// later launches.
// BF16 (the bf16 update path, BASELINE.json configs[4]): the 256 -> 512 product on v_mfma_f32_16x16x32_bf16 — h1 rounded to bf16 once into an
//                LDS tile, the wave's B fragments straight from the net's bf16 image into registers (one contiguous kilobyte per load, no
//                LDS staging of W2), fp32 accumulation; layer 1, LayerNorm, the previous net's head and everything saved for the backward
//                pass stay fp32.
// The workgroup's LDS as ONE object (the kernel declares it; hx_front.hip overlays it with the acting workgroups' block in a union).
template <int NT, bool SAMPLE, bool BF16>
struct FwdLds {
    static constexpr bool WIDE = NT == 256;
    static constexpr int CT = WIDE ? 1 : NT / 16, KS = WIDE ? 1 : 16 / CT;  // column tiles / K-parts per workgroup (latency mode)
    static constexpr int KRED = WIDE ? 4 : (KS - 1) * CT * 256;
    // W2 tile of the workgroup's NT columns, [NT][LDA1] (latency modes): requested with COALESCED loads (a column's 16 or 32 threads cover 256
    // or 512 contiguous bytes) and turned into MFMA operand order through LDS.  Straight into registers in operand order, adjacent lanes
    // are adjacent columns, 1 KB apart in the row-major matrix: 64 separate 16-byte requests per load, 4,096 per workgroup.
    static constexpr int kW2S = (WIDE || BF16) ? 4 : NT * LDA1;
    __attribute__((aligned(16))) float lds[RT * LDA1 + RT * XP + RT * 2 + KRED + H1 * 17 + 8 + kW2S];
    __attribute__((aligned(16))) __bf16 h1b[BF16 ? RT * LDB1 : 8];  // BF16: the A operand of the MFMA phase
    uint32_t s_hkey[SAMPLE ? 2 : 1][SAMPLE ? kFusedSlots : 1];
    int s_hown[SAMPLE ? 2 : 1][SAMPLE ? kFusedSlots : 1];
    int s_fin[SAMPLE ? 2 : 1][SAMPLE ? kFusedBatchMax : 1];
};

// FRONT (hx_front.hip: launches A and B of learn() as workgroups of the act + env launch, B waiting for A in-launch):
//   0  a launch of its own (fwd_l2_kernel)
//   1  producer: job 0 (the target actor) writes its z2 tile with agent-scope relaxed stores and counts itself in on its row tile's counter X.flags[rt]
//      (tools/ubench/handoff_probe.hip: the only hand-off form that does not cost a cache flush on this machine); the LAST of the tile's column
//      workgroups then evaluates the net's head for the tile's 16 rows — tanh, smoothing noise, clamp: what every consumer workgroup of a launch of its own
//      repeats for itself, 2.7 us of CU time each — writes the action rows and counts once more
//   2  consumer: a job that feeds on the previous net's head waits (bounded) for the counter of its row tile and reads the action rows with agent-scope loads
//   (3: hx_bwd_body.h — launch C's workgroups wait for the critic jobs of launch A and the target critics of launch B, which count themselves in too: with_c)
struct FrontSync {
    unsigned* flags;   // [row tiles] monotonic counters: + 1 per column workgroup of the producer job, + 1 once the tile's action rows are written
    unsigned arrive;   // value a row tile's counter shows when the LAST column workgroup of the producer job has counted itself in
    unsigned target;   // value it reaches once the tile's action rows are there: what the consumers wait for
    unsigned* status;  // sticky: bit 0 = a consumer gave up waiting
    const float* noise;  // [4] the call's target-smoothing draw (or null) and its clamp: applied by the producer's last column workgroup
    float noise_clamp;
    // launch C (the critics' backward, hx_bwd_body.h) as further workgroups of the same launch: the critic jobs of launch A (jobs 1, 2) and the target
    // critics of launch B (jobs 0, 1) then publish what C reads — z2 rows, z1, LN1 statistics — with agent-scope stores and count themselves in on
    // flags[16 + 16 * (job - 1) + rt] / flags[48 + rt]; C's workgroups wait for c_target (one job's column workgroups) / t_target (both target jobs')
    unsigned with_c, c_target, t_target;
};
template <int NT, bool RELU, bool SAMPLE, bool BF16, int FRONT, typename SAT>
__device__ __forceinline__ void fwd_l2_body(const FwdArgsC& A, const SAT& SA, const int bx, const int by, FwdLds<NT, SAMPLE, BF16>& SL, const FrontSync& X) {
    typedef FwdLds<NT, SAMPLE, BF16> Lds;
    constexpr bool WIDE = Lds::WIDE;
    constexpr int NTW = NT;
    constexpr int CT = Lds::CT, KS = Lds::KS, KRED = Lds::KRED;
    float* const lds = SL.lds;
    __bf16* const h1b = SL.h1b;
    auto& s_hkey = SL.s_hkey;
    auto& s_hown = SL.s_hown;
    auto& s_fin = SL.s_fin;
    constexpr int NSL = 8 / KS;  // BF16: 32-wide k-slabs per wave (K = 256 in 8 slabs over the KS K-parts)
    float* h1s = lds;
    float* xs = lds + RT * LDA1;
    float* sts = xs + RT * XP;
    float* kred = sts + RT * 2;   // [KS - 1 K-parts][CT column tiles][64 lanes][4]
    float* w1s = kred + KRED;     // W1 [256][in], staged with coalesced loads (a per-thread row walk is 17 scattered requests)
    float* w2s = w1s + H1 * 17 + 8;

    // job = by; row tile / column tile from bx
    const int b = bx;
    const FwdJobC& jc = A.job[by];
    const FwdJob J = expand_fwd(jc);
    // Workgroups are dealt round-robin over the 8 XCDs (b % 8), each with its own L2.  rowmap: the row tile is b % tiles, so at B = 128 (8 row
    // tiles) every workgroup of row tile rt runs on XCD rt — where the launches before this one left that tile's rows, and where the launches
    // after it will look for what this one writes: rows another XCD has just written come back at ~19 B/clk/CU, the XCD's own at ~35.  The
    // weights then come from every XCD's share of the Infinity Cache instead of one L2 slice per column workgroup (clean lines: cheap).
    const int ntile_ = tiles_of(J.rows);
    const int rt = A.rowmap ? b % ntile_ : b / (H2 / NTW), nt = A.rowmap ? b / ntile_ : b % (H2 / NTW);
    const int r0 = rt * RT;
    const int nrow = min(RT, J.rows - r0);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float slope = jc.slope;
    const int in = J.m.in;
    // operands that do not depend on the prologue are requested first: their latency hides behind the gather
    // this thread's share of the W2 tile: column tid / TPC, 16-byte piece tid % TPC of each K section of TPC * 4 floats
    constexpr int TPC = WIDE ? 16 : kWide / NT;       // threads per column: 16 (NT = 64) or 32 (NT = 32)
    constexpr int NW2 = WIDE ? 1 : H1 / (TPC * 4);    // loads per thread: 4 or 2
    v4f w2v[NW2];  // (native vectors: an array of HIP float4 stays an alloca)
    uint4 bq[BF16 ? NSL : 1];  // BF16: this wave's B fragments (column tile ct, slabs kq NSL ..) from the image
    // layer 1 runs on MFMA: wave w owns hidden units 16 w .. 16 w + 15 of all 16 rows; lane (lr, lg) ends up with rows 4 lg .. 4 lg + 3 of unit u
    const int lr = lane & 15, lg = lane >> 4, u = wave * 16 + lr;
    STAMP_DECL;
    STAMP();
    // Every global operand of the prologue is requested before the first one is consumed: W1 (one or two float4 per thread),
    // the layer-1 vectors and this thread's element of the 16 x XP input tile travel together — one round trip, not three.
    const float4* W1v = reinterpret_cast<const float4*>(J.net + J.m.W1());
    const int n4 = H1 * in / 4;  // 832 or 1088 float4
    const float4 wv0 = tid < n4 ? W1v[tid] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 wv1 = tid + kWide < n4 ? W1v[tid + kWide] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float bias1 = J.net[J.m.b1() + u], g1v = J.net[J.m.g1() + u], be1v = J.net[J.m.be1() + u];
    // b2 of this wave's column of the z2 tile: asked for HERE — behind the MFMA phase's barrier the load stood alone in front of the tile's stores
    const float bias2 = J.net[J.m.b2() + nt * NTW + (WIDE ? wave * 16 : (wave % CT) * 16) + (lane & 15)];
    // 1. input tile xs[16][XP]: thread -> (row, column); columns 13..16 carry the action of a 17-wide net, the rest is zero
    const int xr = tid / XP, xc = tid % XP;
    const bool head_mode = in == 17 && J.act_mode != 0 && J.act_mode != 3;
    float xv = 0.0f;
    float4 tile_piece = make_float4(0.f, 0.f, 0.f, 0.f);  // SAMPLE: this thread's 16 bytes of the row tile its workgroup publishes
    if constexpr (SAMPLE) {
        // the draw needs *total and LDS only; W1, the vectors and the W2 fragment are already on their way
        draw_fused(SA, J.rows, s_hkey, s_hown, s_fin);
        if (tid < RT * XP && xr < nrow) {
            const int r = r0 + xr;
            const float* row = (r < SA.n_main ? SA.ring : SA.expert_ring) + (size_t)s_fin[0][r] * 32;
            if (xc < 13) xv = row[J.col0 + xc];
            else if (in == 17 && xc < 17 && J.act_mode == 0) xv = row[xc];
        }
        if (by == 0 && nt < 2 && tid < nrow * 8) {  // column workgroup 0 publishes rows[r0 ..], column workgroup 1 bc_rows[r0 ..]
            const int r = r0 + (tid >> 3);
            if (nt == 0) tile_piece = reinterpret_cast<const float4*>((r < SA.n_main ? SA.ring : SA.expert_ring) + (size_t)s_fin[0][r] * 32)[tid & 7];
            else if (SA.bc_rows) tile_piece = reinterpret_cast<const float4*>(SA.bc_table + (size_t)s_fin[1][r] * 32)[tid & 7];
        }
    } else if (tid < RT * XP && xr < nrow) {
        if (xc < 13) xv = src_row(J.src, r0 + xr)[J.col0 + xc];
        else if (in == 17 && xc < 17) {
            if (J.act_mode == 0) xv = src_row(J.src, r0 + xr)[xc];                             // replayed action, row cols 13..16
            else if (J.act_mode == 3) xv = J.noise[(size_t)(r0 + xr) * 4 + (xc - 13)];        // action rows of an earlier kernel (SAC)
        }
    }
    // the W2 fragment of the MFMA phase: 64 separate 16-byte requests per load (adjacent lanes are adjacent COLUMNS, 1 KB apart in the
    // row-major matrix) — behind the prologue's own operands, not in front of them
    if constexpr (BF16) {
        // column tile (of 16) and first slab of this wave; block (tile, slab) of the image is 512 elements, lane l's 16 bytes at + 8 l
        const int ctile = WIDE ? nt * 16 + wave : nt * CT + wave % CT;
        const int sl0 = WIDE ? 0 : (wave / CT) * NSL;
        const uint16_t* blk = A.images + (size_t)J.img * kImgElems + (size_t)(ctile * 8 + sl0) * 512 + lane * 8;
#pragma unroll
        for (int i = 0; i < NSL; ++i) bq[i] = *reinterpret_cast<const uint4*>(blk + i * 512);
    } else if (!WIDE) {
        const float* wcol = J.net + J.m.W2() + (size_t)(nt * NT + tid / TPC) * H1 + (tid % TPC) * 4;
#pragma unroll
        for (int i = 0; i < NW2; ++i) w2v[i] = *reinterpret_cast<const v4f*>(wcol + i * TPC * 4);
    }
    STAMP();
    if constexpr (FRONT == 2) {
        if (head_mode) {  // the previous net's z2 comes from workgroups of THIS launch (everything else this workgroup reads is older: already requested)
            if (tid == 0) {
                int spins = 0;
                while ((int)(__hip_atomic_load(&X.flags[rt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - X.target) < 0) {
                    if (++spins > 4000000) { atomicOr(X.status, 1u); break; }  // (~1 s, never seen: include/hirl4ucav.h hx_hirl_front, WHY THE WAITS END)
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (tid < nrow * 4) xs[(tid >> 2) * XP + 13 + (tid & 3)] = ld_agent(J.prev.ws.outv + (size_t)(r0 + (tid >> 2)) * OW + (tid & 3));
        }
    }
    if (FRONT != 2 && head_mode && wave < nrow) {
        // head of the previous net: wave w owns row w (its loads go out right behind the ones above, nothing waited on yet)
        const int r = wave;
        RowReg<H2> xh, y;
        float mean, rstd, o[4];
        if (J.prev.m.out == 4) head_row4<RELU>(J.prev.ws.z2 + (size_t)(r0 + r) * H2, J.prev.net, J.prev.m, slope, xh, y, mean, rstd, o);
        else head_row<4, RELU>(J.prev.ws.z2 + (size_t)(r0 + r) * H2, J.prev.net, J.prev.m, slope, xh, y, mean, rstd, o);
        if (lane < 4) {
            float a = fast_tanh(lane == 0 ? o[0] : lane == 1 ? o[1] : lane == 2 ? o[2] : o[3]);  // Actor.forward's tanh, HIRL.py:140
            if (J.noise) {              // target smoothing, HIRL.py:264-267
                const float e = fminf(fmaxf(J.noise[lane], -J.noise_clamp), J.noise_clamp);
                a = fminf(fmaxf(a + e, -1.0f), 1.0f);
            }
            xs[r * XP + 13 + lane] = a;
            if (nt == 0) J.prev.ws.outv[(size_t)(r0 + r) * OW + lane] = a;
        }
        if (nt == 0 && lane == 0) {
            J.prev.ws.st2[(size_t)(r0 + r) * 2] = mean;
            J.prev.ws.st2[(size_t)(r0 + r) * 2 + 1] = rstd;
        }
    }
    STAMP();
    if (tid < n4) reinterpret_cast<float4*>(w1s)[tid] = wv0;
    if (tid + kWide < n4) reinterpret_cast<float4*>(w1s)[tid + kWide] = wv1;
    if (tid < RT * XP && !(head_mode && xc >= 13 && xc < 17 && xr < nrow)) xs[tid] = xv;  // those four belong to the head wave
    __syncthreads();
    STAMP();

    // 2. z1[row 4 lg + q][u] = b1[u] + sum_k x[row][k] W1[u][k] on fp32 MFMA, K = 20 (13 or 17 used) in five steps: lane (lr, lg) feeds
    //    x[lr][4 m + lg] and W1[u][4 m + lg] from LDS (10 reads and 5 MFMAs per lane instead of 85 reads and 68 FMAs)
    float z1[4];
    {
        v4f acc = {bias1, bias1, bias1, bias1};
        const float* wrow = w1s + u * in + lg;   // columns >= in of xs are zero; W1 is masked (the LDS words behind a row are not zeros)
        const float* xrow = xs + lr * XP + lg;
#pragma unroll
        for (int mm = 0; mm < 5; ++mm) {
            const float wv = wrow[4 * mm];
            acc = mfma16(xrow[4 * mm], 4 * mm + lg < in ? wv : 0.0f, acc);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            z1[q] = acc[q];
            h1s[(4 * lg + q) * LDA1 + u] = z1[q];
        }
    }
    __syncthreads();
    STAMP();
    // 3. LN1 statistics: wave w owns row w
    {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = h1s[wave * LDA1 + i * 64 + lane];
        float mean, rstd;
        row_stats<4>(v, H1, mean, rstd);
        if (J.m.no_ln) { mean = 0.0f; rstd = 1.0f; }
        if (lane == 0) {
            sts[wave * 2] = mean;
            sts[wave * 2 + 1] = rstd;
        }
    }
    __syncthreads();
    STAMP();
    // 4. h1 = act(LN1(z1))
    {
        const float g = g1v, be = be1v;
        const bool save = J.save && nt == 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * lg + r;
            const float h = act_f<RELU>(g * ((z1[r] - sts[row * 2]) * sts[row * 2 + 1]) + be, slope);
            if (BF16) h1b[row * LDB1 + u] = (__bf16)h;  // v_cvt_pk_bf16_f32: round to nearest even
            else h1s[row * LDA1 + u] = h;
            if (save && row < nrow) {
                if (FRONT != 0 && X.with_c) st_agent(&J.ws.z1[(size_t)(r0 + row) * H1 + u], z1[r]);  // (launch C of the same launch reads z1 and st1)
                else J.ws.z1[(size_t)(r0 + row) * H1 + u] = z1[r];
                J.ws.h1[(size_t)(r0 + row) * H1 + u] = h;
            }
        }
        if (save) {
            if (tid < nrow * XP) J.ws.x[(size_t)r0 * XP + tid] = xs[tid];
            if (tid < nrow * 2) {
                if (FRONT != 0 && X.with_c) st_agent(&J.ws.st1[(size_t)r0 * 2 + tid], sts[tid]);
                else J.ws.st1[(size_t)r0 * 2 + tid] = sts[tid];
            }
        }
        if (!WIDE && !BF16) {
#pragma unroll
            for (int i = 0; i < NW2; ++i) *reinterpret_cast<v4f*>(w2s + (tid / TPC) * LDA1 + (tid % TPC) * 4 + i * TPC * 4) = w2v[i];
        }
    }
    __syncthreads();
    STAMP();
    // 5. z2 tile on fp32 MFMA
    if (WIDE) {
        const int n0 = nt * NTW + wave * 16;
        const int r = lane & 15, g = lane >> 4;
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (BF16) {
#pragma unroll
            for (int i = 0; i < NSL; ++i) acc = mfma16_bf16(*reinterpret_cast<const uint4*>(h1b + r * LDB1 + 32 * i + 8 * g), bq[i], acc);
        } else {
            acc = tile_a_lds_bt_global<H1>(h1s, LDA1, J.net + J.m.W2() + (size_t)(n0 + r) * H1, acc);
        }
        const float bias = bias2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = 4 * g + q;
            if (row < nrow) J.ws.z2[(size_t)(r0 + row) * H2 + n0 + r] = acc[q] + bias;
        }
    } else {
        // wave = (column tile ct, K part kq); partial sums meet in LDS
        const int ct = wave % CT, kq = wave / CT;
        const int n0 = nt * NT + ct * 16;
        const int r = lane & 15, g = lane >> 4;
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (BF16) {  // lane (r, g): A[row r][32 sl + 8 g ..+7] from the bf16 tile, B from registers
#pragma unroll
            for (int i = 0; i < NSL; ++i) acc = mfma16_bf16(*reinterpret_cast<const uint4*>(h1b + r * LDB1 + 32 * (kq * NSL + i) + 8 * g), bq[i], acc);
        } else {   // A (h1) and B (W2 tile) fragments both from LDS: lane (r, g) reads 16 bytes at [row / column r][kq K/KS + 16 i + 4 g]
            const float* ap = h1s + r * LDA1 + kq * (H1 / KS) + 4 * g;
            const float* bp = w2s + (ct * 16 + r) * LDA1 + kq * (H1 / KS) + 4 * g;
#pragma unroll
            for (int i = 0; i < H1 / KS / 16; ++i) {
                const float4 a4 = *reinterpret_cast<const float4*>(ap + 16 * i);
                const float4 b4 = *reinterpret_cast<const float4*>(bp + 16 * i);
                acc = mfma16(a4.x, b4.x, acc);
                acc = mfma16(a4.y, b4.y, acc);
                acc = mfma16(a4.z, b4.z, acc);
                acc = mfma16(a4.w, b4.w, acc);
            }
        }
        if (kq) *reinterpret_cast<float4*>(kred + (((kq - 1) * CT + ct) * 64 + lane) * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        __syncthreads();
        if (kq == 0) {
            const float bias = bias2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float part[KS];  // fixed-order tree over the K-parts
                part[0] = acc[q];
#pragma unroll
                for (int k = 1; k < KS; ++k) part[k] = kred[(((k - 1) * CT + ct) * 64 + lane) * 4 + q];
#pragma unroll
                for (int w = 1; w < KS; w *= 2)
#pragma unroll
                    for (int k = 0; k < KS; k += 2 * w) part[k] += part[k + w];
                const int row = 4 * g + q;
                if (row < nrow) {
                    if ((FRONT == 1 && by == 0) || (FRONT != 0 && X.with_c)) st_agent(&J.ws.z2[(size_t)(r0 + row) * H2 + n0 + r], part[0] + bias);
                    else J.ws.z2[(size_t)(r0 + row) * H2 + n0 + r] = part[0] + bias;
                }
            }
        }
        STAMP();
        STAMP_FLUSH(SAMPLE ? 8 : 0, bx == 5 && tid == 0);
    }
    SPAN_LOG(FRONT == 1 ? HX_SPAN_FRONT_A : (FRONT == 2 ? HX_SPAN_FRONT_B : HX_SPAN_FWD));
    if constexpr (SAMPLE) {
        if (by == 0) {  // what hx_sample_batch leaves behind: row tiles, indices, noise — read by the launches after this one
            if (FRONT == 1) {
                float* dst = nt == 0 ? SA.rows : SA.bc_rows;
                if (nt < 2 && dst && tid < nrow * 8) {
                    dst += ((size_t)r0 * 8 + tid) * 4;
                    st_agent(dst, tile_piece.x); st_agent(dst + 1, tile_piece.y); st_agent(dst + 2, tile_piece.z); st_agent(dst + 3, tile_piece.w);
                }
            } else {
                if (nt == 0 && tid < nrow * 8) reinterpret_cast<float4*>(SA.rows)[(size_t)r0 * 8 + tid] = tile_piece;
                if (nt == 1 && SA.bc_rows && tid < nrow * 8) reinterpret_cast<float4*>(SA.bc_rows)[(size_t)r0 * 8 + tid] = tile_piece;
            }
            if (b == 2) {
                if (tid < J.rows) {
                    SA.idx[tid] = s_fin[0][tid];
                    if (SA.idx_bc) SA.idx_bc[tid] = s_fin[1][tid];
                }
                if (tid < 4 && SA.noise) SA.noise[tid] = smoothing_noise(SA, tid);  // the (4,) target-smoothing draw, HIRL.py:265
            }
        }
    }
    if constexpr (FRONT == 1) {
        if (bx == 0 && by == 0) {  // the accumulators launch C adds to are cleared BEFORE this workgroup counts itself in (C waits for row tile 0's counter too)
            if ((int)threadIdx.x < A.zero_nf) st_agent(&A.zero_f[threadIdx.x], 0.0f);
            if (threadIdx.x == 0 && A.zero_i) *A.zero_i = 0;
        }
    }
    if constexpr (FRONT == 1 || FRONT == 2) {
        if (X.with_c && ((FRONT == 1 && (by == 1 || by == 2)) || (FRONT == 2 && by < 2))) {  // read by launch C of this launch: count in once the rows have left the CU
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(&X.flags[(FRONT == 1 ? 16 * by : 48) + rt], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if constexpr (FRONT == 1) {
        if (by == 0) {  // everything this workgroup published has left the CU: count it in
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            int* last = reinterpret_cast<int*>(lds);  // (every wave is past its last LDS access)
            if (tid == 0) *last = __hip_atomic_fetch_add(&X.flags[rt], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == X.arrive;
            __syncthreads();
            if (*last) {  // the tile's last column workgroup: all 512 columns of its 16 rows are there — the head once, for every consumer
                if (wave < nrow) {
                    const int r = wave;
                    RowReg<H2> xh, y;
                    float mean, rstd, o[4];
                    if (J.m.out == 4) head_row4<RELU, true>(J.ws.z2 + (size_t)(r0 + r) * H2, J.net, J.m, slope, xh, y, mean, rstd, o);
                    else head_row<4, RELU, true>(J.ws.z2 + (size_t)(r0 + r) * H2, J.net, J.m, slope, xh, y, mean, rstd, o);
                    if (lane < 4) {
                        float a = fast_tanh(lane == 0 ? o[0] : lane == 1 ? o[1] : lane == 2 ? o[2] : o[3]);  // Actor.forward's tanh, HIRL.py:140
                        if (X.noise) {          // target smoothing, HIRL.py:264-267
                            const float e = fminf(fmaxf(X.noise[lane], -X.noise_clamp), X.noise_clamp);
                            a = fminf(fmaxf(a + e, -1.0f), 1.0f);
                        }
                        st_agent(&J.ws.outv[(size_t)(r0 + r) * OW + lane], a);
                    }
                    if (lane == 0) {
                        J.ws.st2[(size_t)(r0 + r) * 2] = mean;
                        J.ws.st2[(size_t)(r0 + r) * 2 + 1] = rstd;
                    }
                }
                __builtin_amdgcn_s_waitcnt(0);
                __syncthreads();
                if (tid == 0) __hip_atomic_fetch_add(&X.flags[rt], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // accumulators of LATER launches are cleared here, at the end: their kernel-argument words are off every workgroup's critical path
    if (FRONT != 1 && bx == 0 && by == 0) {
        if ((int)threadIdx.x < A.zero_nf) A.zero_f[threadIdx.x] = 0.0f;
        if (threadIdx.x == 0 && A.zero_i) *A.zero_i = 0;
    }
}

}  // namespace hxu
