// hx_fwdbwd.hip — forward and backward kernels of the update's 256 <-> 512 layer and everything fused around it (gfx950).
//   fwd_l2   z2 = act(LN(x W1^T + b1)) W2^T + b2 for up to 6 independent nets per launch  (Actor.forward / Critic.forward, HIRL.py:55-97,126-140)
//   bwd_l2   head + loss gradient + LN2 backward, dh1 = dz2 W2                               (the backward of HIRL.py:276-286,291-324)
// Structure and measurements: hx_update.h, DESIGN.md section 4.
#include "hx_update.h"

using namespace hxnn;
using namespace hxu;

namespace {

// What the kernel actually receives: 64 bytes per job (ONE s_load_dwordx16), the job picked by blockIdx.y.  A kernel argument
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
    int rowmap;              // 1: blockIdx.x -> (row tile = b % tiles, column workgroup = b / tiles): a row tile's workgroups share an XCD
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
template <int NT, bool RELU, bool SAMPLE, bool BF16 = false>
__global__ __launch_bounds__(kWide) void fwd_l2_kernel(FwdArgsC A, typename std::conditional<SAMPLE, SampleDev, NoSample>::type SA) {
    constexpr bool WIDE = NT == 256;
    __shared__ uint32_t s_hkey[SAMPLE ? 2 : 1][SAMPLE ? kFusedSlots : 1];
    __shared__ int s_hown[SAMPLE ? 2 : 1][SAMPLE ? kFusedSlots : 1];
    __shared__ int s_fin[SAMPLE ? 2 : 1][SAMPLE ? kFusedBatchMax : 1];
    constexpr int NTW = NT;
    constexpr int CT = WIDE ? 1 : NT / 16, KS = WIDE ? 1 : 16 / CT;  // column tiles / K-parts per workgroup (latency mode)
    constexpr int KRED = WIDE ? 4 : (KS - 1) * CT * 256;
    // W2 tile of the workgroup's NT columns, [NT][LDA1] (latency modes): requested with COALESCED loads (a column's 16 or 32 threads cover 256
    // or 512 contiguous bytes) and turned into MFMA operand order through LDS.  Straight into registers in operand order, adjacent lanes
    // are adjacent columns, 1 KB apart in the row-major matrix: 64 separate 16-byte requests per load, 4,096 per workgroup.
    constexpr int kW2S = (WIDE || BF16) ? 4 : NT * LDA1;
    __shared__ __attribute__((aligned(16))) float lds[RT * LDA1 + RT * XP + RT * 2 + KRED + H1 * 17 + 8 + kW2S];
    __shared__ __attribute__((aligned(16))) __bf16 h1b[BF16 ? RT * LDB1 : 8];  // BF16: the A operand of the MFMA phase
    constexpr int NSL = 8 / KS;  // BF16: 32-wide k-slabs per wave (K = 256 in 8 slabs over the KS K-parts)
    float* h1s = lds;
    float* xs = lds + RT * LDA1;
    float* sts = xs + RT * XP;
    float* kred = sts + RT * 2;   // [KS - 1 K-parts][CT column tiles][64 lanes][4]
    float* w1s = kred + KRED;     // W1 [256][in], staged with coalesced loads (a per-thread row walk is 17 scattered requests)
    float* w2s = w1s + H1 * 17 + 8;

    // job = blockIdx.y; row tile / column tile from blockIdx.x
    const int b = blockIdx.x;
    const FwdJobC& jc = A.job[blockIdx.y];
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
        if (blockIdx.y == 0 && nt < 2 && tid < nrow * 8) {  // column workgroup 0 publishes rows[r0 ..], column workgroup 1 bc_rows[r0 ..]
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
    if (head_mode && wave < nrow) {
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
                J.ws.z1[(size_t)(r0 + row) * H1 + u] = z1[r];
                J.ws.h1[(size_t)(r0 + row) * H1 + u] = h;
            }
        }
        if (save) {
            if (tid < nrow * XP) J.ws.x[(size_t)r0 * XP + tid] = xs[tid];
            if (tid < nrow * 2) J.ws.st1[(size_t)r0 * 2 + tid] = sts[tid];
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
        const float bias = J.net[J.m.b2() + n0 + r];
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
            const float bias = J.net[J.m.b2() + n0 + r];
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
                if (row < nrow) J.ws.z2[(size_t)(r0 + row) * H2 + n0 + r] = part[0] + bias;
            }
        }
        STAMP();
        STAMP_FLUSH(SAMPLE ? 8 : 0, blockIdx.x == 5 && tid == 0);
    }
    SPAN_LOG(HX_SPAN_FWD);
    if constexpr (SAMPLE) {
        if (blockIdx.y == 0) {  // what hx_sample_batch leaves behind: row tiles, indices, noise — read by the launches after this one
            if (nt == 0 && tid < nrow * 8) reinterpret_cast<float4*>(SA.rows)[(size_t)r0 * 8 + tid] = tile_piece;
            if (nt == 1 && SA.bc_rows && tid < nrow * 8) reinterpret_cast<float4*>(SA.bc_rows)[(size_t)r0 * 8 + tid] = tile_piece;
            if (b == 2) {
                if (tid < J.rows) {
                    SA.idx[tid] = s_fin[0][tid];
                    if (SA.idx_bc) SA.idx_bc[tid] = s_fin[1][tid];
                }
                if (tid < 4 && SA.noise) {  // the (4,) target-smoothing draw, HIRL.py:265 (sample_kernel's arithmetic)
                    uint32_t uu[4];
                    philox4x32_10(0xFFFFFFF0u, SA.call, 2u, 0u, (uint32_t)SA.seed, (uint32_t)(SA.seed >> 32), uu);
                    const float ua = u01(uu[tid & 2]), ub = u01(uu[(tid & 2) + 1]);
                    const float rad = sqrtf(-2.0f * __logf(ua)), ang = 6.28318530717958647692f * ub;
                    SA.noise[tid] = SA.sigma * ((tid & 1) ? rad * __sinf(ang) : rad * __cosf(ang));
                }
            }
        }
    }
    // accumulators of LATER launches are cleared here, at the end: their kernel-argument words are off every workgroup's critical path
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        if ((int)threadIdx.x < A.zero_nf) A.zero_f[threadIdx.x] = 0.0f;
        if (threadIdx.x == 0 && A.zero_i) *A.zero_i = 0;
    }
}

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
template <int GRP, bool RELU, bool BF16 = false>
__global__ __launch_bounds__(kWide) void bwd_l2_kernel(BwdArgsC AC) {
    __shared__ __attribute__((aligned(16))) float dz2s[BF16 ? kCTB * RT * 2 : RT * LDA2];  // (BF16: only the epilogue's row-sum scratch)
    __shared__ __attribute__((aligned(16))) __bf16 dz2b[BF16 ? RT * LDB2 : 8];
    __shared__ __attribute__((aligned(16))) float kred[(kKSB - 1) * kCTB * 256];  // split-K partial tiles
    constexpr int IMG = (GRP == 3 || GRP == 5) ? 8 : 4;  // head width of this instantiation's LDS images
    // head width known at compile time: the critic jobs (GRP 0, 1) have ONE output, the actor jobs (GRP 2) four; GRP 3 (SAC: policy 8 wide,
    // Q heads 1) keeps the run-time width.  A run-time trip count over dout[] costs a select chain per step (no indexed registers).
    constexpr int NOUT = (GRP <= 1 || GRP == 4) ? 1 : (GRP == 2 ? 4 : (GRP == 5 ? 8 : 0));
    constexpr int OUTW = NOUT ? NOUT : IMG;
    typedef HeadImage<IMG> Img;
    constexpr int kHpStride = Img::kStride;
    __shared__ __attribute__((aligned(16))) float hps[(GRP == 0 ? 3 : (GRP == 4 ? 2 : 1)) * kHpStride];
    __shared__ __attribute__((aligned(16))) float c1s[GRP == 2 ? H1 * 6 : (GRP == 5 ? H1 * 8 : 4)];  // critic layer 1: g1 be1 W1[:,13..16]  (GRP 5: W1[:,13..16] of both critics)
    __shared__ float red[16][4];
    __shared__ float st1s[RT * 2];  // LN1 stats of the tile's rows (epilogue)
    // TD job (GRP 0): EIGHT rows per workgroup, a wave PAIR per row — wave w (role 0) owns the row's own head, loss gradient and LN2
    // backward, wave w + 8 (role 1) the two target heads; min(Q1', Q2') crosses through LDS.  Twice the workgroups (256 at B = 128, two
    // jobs): half the row bytes per CU (rows the previous launches produced on all eight XCDs arrive at ~19 B/clk/CU, and 96 KB of them
    // were in front of this prologue), and the three heads of a row no longer run one after the other on one wave.  The MFMA tile keeps
    // its 16 rows (8 of them zero): that phase is the short one.
    // The critic-PI job (GRP 1) pairs the same way (role 1: the soft head), the actor jobs (GRP 2) too (role 1: dL/da from the critic's
    // layer-1 backward, four dot products over 256 hidden units).
    constexpr bool PAIRED = GRP <= 2 || GRP >= 4;
    constexpr int RTB = PAIRED ? RT / 2 : RT;
    __shared__ float tq[PAIRED ? RTB * 4 : 1];

    const int b = blockIdx.x;
    const BwdJobC& jc = AC.job[blockIdx.y];
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
    if (!role1) z.load(J.ws.z2 + R * H2);
    Img pv0, pv1, pv2;
    pv0.fetch(J.net, J.m, tid);
    float bonus = 0.f, bonus_scale = 0.f, dgiv[GRP == 3 ? 8 : 1] = {};
    if (GRP == 0) {
        if (role1) {
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
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = 4 * g + q;
            ez1[q] = J.ws.z1[(size_t)(r0 + (row < nrow ? row : 0)) * H1 + n0 + r];  // unconditional, clamped (rows >= nrow are never stored)
        }
    }
    const float st1v = tid < nrow * 2 ? J.ws.st1[(size_t)r0 * 2 + tid] : (tid & 1 ? 1.0f : 0.0f);
    // the W2 fragment of the MFMA phase (16 loads per lane, needed last) goes out behind the prologue's operands, not in front of them
    if constexpr (BF16) {  // block (tile of 16 columns of dh1, slab of 32 n) of the transposed image: 512 elements, lane l's 16 bytes at + 8 l
        const uint16_t* blk = AC.images + (size_t)jc.img_t * kImgElems + (size_t)((nt * kCTB + ct) * 16 + kq * NSLB) * 512 + lane * 8;
#pragma unroll
        for (int i = 0; i < NSLB; ++i) bqb[i] = *reinterpret_cast<const uint4*>(blk + i * 512);
    } else {
        bfrag.load(J.net + J.m.W2() + (size_t)(kq * (H2 / kKSB)) * H1 + n0 + (lane & 15), H1);
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
    if (GRP == 5 && blockIdx.x == 0 && blockIdx.y == 0 && wave == 1) {
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
        STAMP_FLUSH(16, blockIdx.x == 3 && tid == 0);
        SPAN_LOG(HX_SPAN_BWD);
    }
}

int fwd_row_tiles(const FwdArgs& a) {
    int n = 0;
    for (int j = 0; j < a.njobs; ++j) n += (a.job[j].rows + RT - 1) / RT;
    return n;
}
int bwd_blocks(const BwdArgs& a, int rows_per_wg) {  // per job (every job of a launch has the same row count)
    return ((a.job[0].rows + rows_per_wg - 1) / rows_per_wg) * kColWgB;
}
template <int GRP>
void launch_bwd_t(const BwdArgs& G, hipStream_t st) {
    BwdArgsC C{};
    for (int j = 0; j < G.njobs; ++j) C.job[j] = pack_bwd(G.job[j], G);
    C.images = G.images;
    static const int rowmap = getenv("HX_XCD_ROWMAP") ? atoi(getenv("HX_XCD_ROWMAP")) : 3;  // bit 1: bwd_l2
    C.rowmap = (rowmap >> 1) & 1;
    const dim3 grid(bwd_blocks(G, (GRP <= 2 || GRP >= 4) ? RT / 2 : RT), G.njobs);
    if constexpr (GRP <= 2) {  // the bf16 update path covers the HIRL / TD3 / BC jobs (GRP 3 = SAC's given head gradients: fp32)
        static const int dbg_off = getenv("HX_DBG_BF16_OFF") ? atoi(getenv("HX_DBG_BF16_OFF")) : 0;
        if (G.images && !(dbg_off & 2)) {
            if (G.slope == 0.0f) hipLaunchKernelGGL((bwd_l2_kernel<GRP, true, true>), grid, dim3(kWide), 0, st, C);
            else hipLaunchKernelGGL((bwd_l2_kernel<GRP, false, true>), grid, dim3(kWide), 0, st, C);
            return;
        }
    }
    if (G.slope == 0.0f) hipLaunchKernelGGL((bwd_l2_kernel<GRP, true>), grid, dim3(kWide), 0, st, C);
    else hipLaunchKernelGGL((bwd_l2_kernel<GRP, false>), grid, dim3(kWide), 0, st, C);
}

}  // namespace

namespace hxu {

// column tiling by size: enough row tiles to fill the chip -> wide workgroups (less prologue recomputation)
void launch_fwd(const FwdArgs& F, hipStream_t st) {
    FwdArgsC C{};
    for (int j = 0; j < F.njobs; ++j) {  // every job of a launch has the same row count (the minibatch)
        C.job[j] = pack_fwd(F.job[j]);
        C.job[j].slope = F.slope;
    }
    C.slope = F.slope; C.zero_nf = F.zero_nf; C.zero_f = F.zero_f; C.zero_i = F.zero_i; C.images = F.images;
    static const int rowmap = getenv("HX_XCD_ROWMAP") ? atoi(getenv("HX_XCD_ROWMAP")) : 3;  // bit 0: fwd_l2 (0: the plain order everywhere)
    C.rowmap = rowmap & 1;
    const int tiles = fwd_row_tiles(F), per_job = tiles / F.njobs;
    const bool relu = F.slope == 0.0f;  // compile-time ReLU instantiations (hx_nn.h act_f)
    static const int dbg_off = getenv("HX_DBG_BF16_OFF") ? atoi(getenv("HX_DBG_BF16_OFF")) : 0;
    const bool bf16 = F.images != nullptr && !(dbg_off & 1);
#define HX_FWD_T(NT_, RELU_, BF16_) hipLaunchKernelGGL((fwd_l2_kernel<NT_, RELU_, false, BF16_>), grid, dim3(kWide), 0, st, C, NoSample{})
#define HX_FWD(NT_) do { const dim3 grid(per_job * (H2 / NT_), F.njobs); \
        if (bf16) { if (relu) HX_FWD_T(NT_, true, true); else HX_FWD_T(NT_, false, true); } \
        else { if (relu) HX_FWD_T(NT_, true, false); else HX_FWD_T(NT_, false, false); } } while (0)
    if (F.sample) {  // (the callers checked: three or four jobs of at most 256 rows -> the 64-column tiling)
        const dim3 grid(per_job * (H2 / kNT), F.njobs);
        if (bf16) {
            if (relu) hipLaunchKernelGGL((fwd_l2_kernel<kNT, true, true, true>), grid, dim3(kWide), 0, st, C, *F.sample);
            else hipLaunchKernelGGL((fwd_l2_kernel<kNT, false, true, true>), grid, dim3(kWide), 0, st, C, *F.sample);
        } else {
            if (relu) hipLaunchKernelGGL((fwd_l2_kernel<kNT, true, true>), grid, dim3(kWide), 0, st, C, *F.sample);
            else hipLaunchKernelGGL((fwd_l2_kernel<kNT, false, true>), grid, dim3(kWide), 0, st, C, *F.sample);
        }
        return;
    }
    if (tiles >= 128) HX_FWD(256);
    else if (tiles * (H2 / 32) <= 256) HX_FWD(32);  // one or two nets at B = 128: 32-column workgroups still fit the chip in one round
    else HX_FWD(kNT);
#undef HX_FWD_T
#undef HX_FWD
}

void launch_bwd(int grp, const BwdArgs& G, hipStream_t st) {
    switch (grp) {
        case 0: launch_bwd_t<0>(G, st); break;
        case 1: launch_bwd_t<1>(G, st); break;
        case 2: launch_bwd_t<2>(G, st); break;
        case 4: launch_bwd_t<4>(G, st); break;
        case 5: launch_bwd_t<5>(G, st); break;
        default: launch_bwd_t<3>(G, st); break;
    }
}

}  // namespace hxu

HX_DEFINE_DEBUG_COLLECTORS(fwdbwd, 0, 32)
