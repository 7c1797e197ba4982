// hx_update.hip — actor/critic kernels of the HIRL (TD3+BC) / TD3 update and batched policy inference (gfx950).
//
// What it replaces (reference file:line):
//   Actor.forward / Critic.forward / onlyQ1     hirl/agents/HIRL.py:55-97,126-140          (U1, U2)
//   Agent.chooseAction*                         hirl/agents/HIRL.py:192-212                (U5)   -> hx_actor_act
//   Agent.learn                                 hirl/agents/HIRL.py:221-334                (U7-U11) -> hx_hirl_*
//   TD3.Agent.learn                             hirl/agents/TD3.py:201-260                 (U12)  (slope 0.01, bc off)
//   soft_update                                 hirl/agents/HIRL.py:11-13                  (U3)   -> hx_polyak
//   optim.Adam(lr) defaults                     hirl/agents/HIRL.py:50,123                        -> hx_adam
//   UniformMemory.sample + minibatch assembly   hirl/utils/buffer.py:38-48, HIRL.py:223-251 (U6, U7): rows are gathered
//                                               by index straight from the device-resident replay rings
//
// Structure (DESIGN.md "update kernels"): B = 128 is 0.49 GFLOP per learn(), so the ~1,770 eager ops of the reference collapse into
// 4 launches (critic-only call) or 8 (call with the delayed actor step), minibatch draw and optimizer steps included.  Every launch is
//   fwd_l2   z2 = act(LN(x W1^T + b1)) W2^T + b2 for up to 4 independent nets; layer 1 (K = 13 / 17) on MFMA from LDS-staged operands, the
//            previous net's LN2/final/tanh "head" recomputed per workgroup when the input action is another net's output, the
//            256 -> 512 GEMM tiled 16 rows x 32 / 64 columns per workgroup on fp32 MFMA (v_mfma_f32_16x16x4_f32: fp32 products and
//            sums, parity at 1e-5); the first forward launch can also draw and gather the minibatch (SAMPLE);
//   bwd_l2   head + loss gradient + LN2 backward in the prologue (8 rows per workgroup, a wave pair per row), dh1 = dz2 W2 on MFMA;
//   wgrad    dW2 = dz2^T h1 on MFMA (16 x 128 per workgroup, a tile per wave pair over row halves) + the vector / LN / layer-1 gradients
//            (16 columns x 64 row groups per workgroup): 112 workgroups per job, written into a flat gradient buffer with the parameter layout (one all-reduce message per phase when
//            sharded); on one GPU the same threads apply Adam, the Polyak step of the target and refresh the W2 images (ADAM);
//   adam / polyak  elementwise over the flat buffers, 16 B per lane (sharded path, SAC).
// The kernels issue an instruction nearly every cycle of their life (16 waves per CU): their run time follows the instruction count — and the
// bytes their prologue asks for (a CU gets ~19 B/clk of lines another XCD has just written), and the workgroup count (a launch is as long
// as its slowest workgroup plus the drain of its stores).
#include <cstdlib>
#include <type_traits>

#include "hx_common.h"
#include "hx_nn.h"
#include "hx_env_dev.h"

using namespace hxnn;

// In-kernel phase stamps for diagnosis only (make STAMPS=1): s_memrealtime ticks (10 ns) between phases of ONE
// workgroup land in hx_dbg; the shipped build compiles them out.
#ifdef HX_STAMPS
__device__ float hx_dbg[80];
#define STAMP_DECL unsigned long long TS_[16]; int tsn_ = 0
#define STAMP() TS_[tsn_++] = __builtin_amdgcn_s_memrealtime()
#define STAMP_FLUSH(base, cond) do { if (cond) { for (int i_ = 1; i_ < tsn_; ++i_) hx_dbg[(base) + i_] = (float)(TS_[i_] - TS_[i_ - 1]); hx_dbg[(base)] = (float)tsn_; } } while (0)
// life span of EVERY workgroup of every launch (first stamp .. now), appended to a log: where a learn() spends its time BETWEEN workgroups
constexpr int kSpanCap = 8192;
__device__ unsigned long long hx_span[kSpanCap][2];
__device__ unsigned hx_span_tag[kSpanCap];
__device__ unsigned hx_span_n;
#define SPAN_LOG() do { if (threadIdx.x == 0) { const unsigned long long e_ = __builtin_amdgcn_s_memrealtime(); const unsigned i_ = atomicAdd(&hx_span_n, 1u); \
    if (i_ < (unsigned)kSpanCap) { hx_span[i_][0] = TS_[0]; hx_span[i_][1] = e_; hx_span_tag[i_] = (unsigned)__LINE__; } } } while (0)
#else
#define STAMP_DECL
#define STAMP()
#define STAMP_FLUSH(base, cond)
#define SPAN_LOG()
#endif

namespace {

constexpr int kThreads = 256;
constexpr int kWide = 1024;  // fwd_l2 / bwd_l2 workgroups: 16 waves = one per row of the tile in the prologue, and
                             // CT column tiles x 16/CT K-parts (split-K, LDS reduce) in the MFMA phase.  B = 128 runs ONE
                             // workgroup per CU, so 4 waves per SIMD are what hides the prologue's load/reduce latency.
constexpr int kNT = 64;  // fwd_l2: z2 columns per workgroup (4 column tiles x 16, 4 K-quarters) when a launch carries three or more nets
// bwd_l2: dh1 columns per workgroup.  Its launches have one or two jobs, so 32 columns (2 column tiles x 8 K-parts per
// workgroup, 64-128 workgroups) still fit the chip in one round and halve the MFMA work on each workgroup's critical path.
constexpr int kNTB = 32, kCTB = kNTB / 16, kKSB = 16 / kCTB, kColWgB = H1 / kNTB;
constexpr int OW = 8;    // row pitch of the per-slot head output / head gradient arrays

// minibatch row r comes from main[idx[r]] if r < nb else from exp[idx[r]]; idx == nullptr: row r of `main` itself
struct RowSrc {
    const float* main;
    const float* exp;
    const int* idx;
    int nb;
    int pitch;  // floats per source row (32 for replay rows, 13 for a plain observation matrix)
};
__device__ __forceinline__ const float* src_row(const RowSrc& s, int r) {
    if (!s.idx) return s.main + (size_t)r * s.pitch;
    const int i = s.idx[r];
    return (r < s.nb ? s.main : s.exp) + (size_t)i * s.pitch;
}

// per-evaluation scratch (one "slot" = one net evaluated on one batch), R rows
struct Slot {
    float* x;     // [R][XP]   input rows (state ++ action)
    float* z1;    // [R][H1]
    float* st1;   // [R][2]    mean, rstd of LN1
    float* h1;    // [R][H1]
    float* z2;    // [R][H2]
    float* st2;   // [R][2]
    float* outv;  // [R][8]    head output (tanh(o) for actors, q for critics); 8 = widest head (SAC policy: mean ++ log_std)
    float* dz2;   // [R][H2]
    float* dh1;   // [R][H1]
    float* dout;  // [R][8]    gradient wrt the head pre-activation o
    float* lnp;   // [R][kColWgB][2] LN1-backward row sums (sum dxhat, sum dxhat*xhat) over each workgroup's columns of dh1
};

// a slot is ONE allocation carved in a fixed order: kernels receive its base pointer only and rebuild the field pointers with a
// dozen scalar adds — 2 dwords of kernel argument per slot instead of 22
__host__ __device__ inline Slot carve_slot(float* base, int rows) {
    Slot s;
    float* p = base;
    s.x = p; p += (size_t)rows * XP;
    s.z1 = p; p += (size_t)rows * H1;
    s.st1 = p; p += (size_t)rows * 2;
    s.h1 = p; p += (size_t)rows * H1;
    s.z2 = p; p += (size_t)rows * H2;
    s.st2 = p; p += (size_t)rows * 2;
    s.outv = p; p += (size_t)rows * OW;
    s.dz2 = p; p += (size_t)rows * H2;
    s.dh1 = p; p += (size_t)rows * H1;
    s.dout = p; p += (size_t)rows * OW;
    s.lnp = p; p += (size_t)rows * (2 * kColWgB);
    return s;
}
// Mlp <-> 10 bits
__host__ __device__ inline uint32_t mlp_bits(const Mlp& m) { return (uint32_t)m.in | ((uint32_t)m.out << 5) | ((uint32_t)(m.no_ln ? 1 : 0) << 9); }
__host__ __device__ inline Mlp mlp_of(uint32_t b) { return Mlp{(int)(b & 31u), (int)((b >> 5) & 15u), (int)((b >> 9) & 1u)}; }

struct Head {  // a previous net whose output is (part of) this net's input
    const float* net;
    Mlp m;
    Slot ws;
};

// ---------------------------------------------------------------------------------------------------------------
// row helpers: a wave owns one row; lane holds n = (i*64 + lane)*4 + c  (i < PER4, c < 4): 16-B coalesced loads
// ---------------------------------------------------------------------------------------------------------------
template <int N>
struct RowReg {
    static constexpr int PER4 = N / 256;
    float v[PER4 * 4];
    __device__ __forceinline__ void load(const float* __restrict__ p) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int i = 0; i < PER4; ++i) {
            const float4 t = reinterpret_cast<const float4*>(p)[i * 64 + lane];
            v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ p) const {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int i = 0; i < PER4; ++i) reinterpret_cast<float4*>(p)[i * 64 + lane] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
    }
    // LDS variant with a pitch that keeps 16-B alignment (pitch % 4 == 0)
    __device__ __forceinline__ void store_lds(float* p) const { store(p); }
};

// head of an MLP block for ONE row held by a wave: LN2 stats of z2, h2 = act(LN2(z2)), o[j] = h2 . W3[j] + b3[j].
// Leaves xhat and y (pre-activation) in registers for the backward prologue.
template <int OUTMAX, bool RELU>
__device__ __forceinline__ void head_row(const float* __restrict__ z2row, const float* __restrict__ net, const Mlp m, float slope,
                                         RowReg<H2>& xhat, RowReg<H2>& y, float& mean, float& rstd, float (&o)[OUTMAX]) {
    RowReg<H2> z, g, be;
    z.load(z2row);
    row_stats<8>(z.v, H2, mean, rstd);
    if (m.no_ln) { mean = 0.0f; rstd = 1.0f; }
    g.load(net + m.g2());
    be.load(net + m.be2());
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        xhat.v[i] = (z.v[i] - mean) * rstd;
        y.v[i] = g.v[i] * xhat.v[i] + be.v[i];
    }
#pragma unroll
    for (int j = 0; j < OUTMAX; ++j) {
        float acc = 0.0f;
        if (j < m.out) {
            RowReg<H2> w;
            w.load(net + m.W3() + j * H2);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += act_f<RELU>(y.v[i], slope) * w.v[i];
            acc = wave_sum(acc) + net[m.b3() + j];
        }
        o[j] = acc;
    }
}

// The same head with EVERY operand requested before the first one is used (four outputs: the deterministic actor).  head_row asks for W3's
// rows one at a time under `if (j < m.out)`: dependent round trips to L2 on the critical path of a forward workgroup whose input action
// is another net's output (-0.2 us per step; staging the 12 KB through LDS as bwd_l2 does costs a barrier more than it saves: +0.35 us).
// Same arithmetic in the same order: same bits.
template <bool RELU>
__device__ __forceinline__ void head_row4(const float* __restrict__ z2row, const float* __restrict__ net, const Mlp m, float slope,
                                          RowReg<H2>& xhat, RowReg<H2>& y, float& mean, float& rstd, float (&o)[4]) {
    RowReg<H2> z, g, be, w0, w1, w2, w3;
    z.load(z2row);
    g.load(net + m.g2());
    be.load(net + m.be2());
    w0.load(net + m.W3());
    w1.load(net + m.W3() + H2);
    w2.load(net + m.W3() + 2 * H2);
    w3.load(net + m.W3() + 3 * H2);
    const int lane = threadIdx.x & 63;
    const float b3 = net[m.b3() + (lane & 3)];
    row_stats<8>(z.v, H2, mean, rstd);
    if (m.no_ln) { mean = 0.0f; rstd = 1.0f; }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        xhat.v[i] = (z.v[i] - mean) * rstd;
        y.v[i] = g.v[i] * xhat.v[i] + be.v[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) a0 += act_f<RELU>(y.v[i], slope) * w0.v[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) a1 += act_f<RELU>(y.v[i], slope) * w1.v[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) a2 += act_f<RELU>(y.v[i], slope) * w2.v[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) a3 += act_f<RELU>(y.v[i], slope) * w3.v[i];
    o[0] = wave_sum(a0) + __shfl(b3, 0);
    o[1] = wave_sum(a1) + __shfl(b3, 1);
    o[2] = wave_sum(a2) + __shfl(b3, 2);
    o[3] = wave_sum(a3) + __shfl(b3, 3);
}

// ---- counter-based RNG shared by the acting / sampling kernels ------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u01(uint32_t u) { return ((float)(u >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// ---- the minibatch draw inside launch A (HxSample, hx_hirl_learn_sampled) --------------------------------------------------------
// What sample_kernel computes for batch <= 512 — the two index streams side by side, "without replacement" by the hash set in LDS
// (key = group | index, owner = lowest row that drew it), the redraw rounds, Philox4x32-10(seed; row, call, stream, round) — repeated by
// EVERY workgroup of the launch (a few hundred instructions and three barriers, under the workgroup's own W1 / W2 requests); the set is
// smaller (batch <= 256 -> 1,024 slots), which changes no result: a key's owner does not depend on where the table keeps it.
struct SampleDev {
    const unsigned long long* total;
    const float* ring; const float* expert_ring; const float* bc_table;
    float* rows; float* bc_rows; float* noise; int* idx; int* idx_bc;
    long long cap;
    uint32_t expert_len, bc_len;
    int32_t n_main;
    uint32_t call;
    uint64_t seed;
    float sigma;
};
constexpr int kFusedSlots = 1024, kFusedBatchMax = 256;
__device__ __forceinline__ uint32_t fused_hash(uint32_t k) { return (k * 2654435761u) >> 22; }  // top 10 bits

__device__ __forceinline__ void draw_fused(const SampleDev& S, int B, uint32_t (*hkey)[kFusedSlots], int (*hown)[kFusedSlots], int (*fin)[kFusedBatchMax]) {
    const int tid = threadIdx.x;
    const int t = tid & 511, stream = tid >> 9;  // 0: replay / expert rows, 1: BC rows
    const unsigned long long tot = *S.total;
    const uint32_t len_main = (uint32_t)(tot < (unsigned long long)S.cap ? tot : (unsigned long long)S.cap);
    const uint32_t k0 = (uint32_t)S.seed, k1 = (uint32_t)(S.seed >> 32);
    const bool live = (stream == 0 || S.idx_bc != nullptr) && t < B;
    const bool main_grp = t < S.n_main;
    const uint32_t len = stream == 1 ? S.bc_len : (main_grp ? len_main : S.expert_len);
    const uint32_t grp = (stream == 1 || main_grp) ? 0u : 0x80000000u;  // groups: [0, n_main) and [n_main, batch)
    uint32_t* keys = hkey[stream];
    int* owns = hown[stream];
    for (int e = t; e < kFusedSlots; e += 512) {
        keys[e] = 0xFFFFFFFFu;
        owns[e] = 0x7FFFFFFF;
    }
    __syncthreads();
    uint32_t key = 0;
    bool dup = live;
    for (int round = 0; round < 128; ++round) {
        if (dup) {
            uint32_t u[4];
            philox4x32_10((uint32_t)t, S.call, (uint32_t)stream, (uint32_t)round, k0, k1, u);
            key = grp | (len ? __umulhi(u[0], len) : 0u);
            uint32_t h = fused_hash(key);
            for (int probe = 0; probe < kFusedSlots; ++probe) {
                const uint32_t k = atomicCAS(&keys[h], 0xFFFFFFFFu, key);
                if (k == 0xFFFFFFFFu || k == key) break;
                h = (h + 1) & (kFusedSlots - 1);
            }
            atomicMin(&owns[h], t);
        }
        __syncthreads();
        if (live) {
            uint32_t h = fused_hash(key);
            for (int probe = 0; probe < kFusedSlots && keys[h] != key; ++probe) h = (h + 1) & (kFusedSlots - 1);
            dup = owns[h] != t;
        }
        if (!__syncthreads_or(dup)) break;
    }
    if (live) fin[stream][t] = (int)(key & 0x7FFFFFFFu);
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// fwd_l2
// ---------------------------------------------------------------------------------------------------------------
struct FwdJob {
    const float* net;  // MLP block to evaluate
    Mlp m;
    RowSrc src;
    int col0;      // state columns [col0, col0+13) of the source row (0 = s, 17 = s')
    int act_mode;  // in == 17 only: 0 = action from source row cols 13..16, 1 = tanh(head(prev)) (+ clamped noise, clamp +-1),
                   // 3 = action rows [rows][4] given through `noise`
    Head prev;
    const float* noise;  // [4] one draw shared by the whole batch (HIRL.py:265) or nullptr
    float noise_clamp;
    Slot ws;
    int rows;
    int save;  // write x, z1, st1, h1 (needed by the backward pass)
};
struct FwdArgs {
    FwdJob job[6];
    int njobs;
    float slope;
    // accumulators cleared by this launch (consumed by LATER launches on the same stream): replaces memset nodes
    float* zero_f;
    int zero_nf;
    int* zero_i;
    const SampleDev* sample;  // launch A of the *_sampled entry points: draw and gather inside this launch
};

// What the kernel actually receives: 64 bytes per job (ONE s_load_dwordx16), the job picked by blockIdx.y.  A kernel argument
// block of 1.7 KB read field by field behind branches cost a chain of 6-8 dependent scalar-load round trips before the first
// vector load went out (~1.5-2 us of a ~10 us launch); the compact form is one round trip, and everything else is scalar ALU.
struct FwdJobC {
    const float* net; const float* src; const float* noise; const float* prev_net;
    float* ws; float* prev_ws;
    uint32_t cfg;  // m:10 | prev.m:10 | act_mode:2 | save:1 | col0:6
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
};
inline FwdJobC pack_fwd(const FwdJob& J) {
    FwdJobC c{};
    c.net = J.net; c.src = J.src.main; c.noise = J.noise; c.prev_net = J.prev.net;
    c.ws = J.ws.x; c.prev_ws = J.prev.ws.x;
    c.cfg = mlp_bits(J.m) | (mlp_bits(J.prev.m) << 10) | ((uint32_t)J.act_mode << 20) | ((uint32_t)(J.save ? 1 : 0) << 22) | ((uint32_t)J.col0 << 23);
    c.rows = J.rows; c.noise_clamp = J.noise_clamp;
    return c;
}
__device__ __forceinline__ FwdJob expand_fwd(const FwdJobC& c) {
    FwdJob J;
    J.net = c.net; J.m = mlp_of(c.cfg & 1023u);
    J.src = RowSrc{c.src, nullptr, nullptr, 0, 32};
    J.col0 = (int)(c.cfg >> 23); J.act_mode = (int)((c.cfg >> 20) & 3u);
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
template <int NT, bool RELU, bool SAMPLE>
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
    constexpr int kW2S = WIDE ? 4 : NT * LDA1;
    __shared__ __attribute__((aligned(16))) float lds[RT * LDA1 + RT * XP + RT * 2 + KRED + H1 * 17 + 8 + kW2S];
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
    const int rt = b / (H2 / NTW), nt = b % (H2 / NTW);
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
    if (!WIDE) {
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
            h1s[row * LDA1 + u] = h;
            if (save && row < nrow) {
                J.ws.z1[(size_t)(r0 + row) * H1 + u] = z1[r];
                J.ws.h1[(size_t)(r0 + row) * H1 + u] = h;
            }
        }
        if (save) {
            if (tid < nrow * XP) J.ws.x[(size_t)r0 * XP + tid] = xs[tid];
            if (tid < nrow * 2) J.ws.st1[(size_t)r0 * 2 + tid] = sts[tid];
        }
        if (!WIDE) {
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
        acc = tile_a_lds_bt_global<H1>(h1s, LDA1, J.net + J.m.W2() + (size_t)(n0 + r) * H1, acc);
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
        {   // A (h1) and B (W2 tile) fragments both from LDS: lane (r, g) reads 16 bytes at [row / column r][kq K/KS + 16 i + 4 g]
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
    SPAN_LOG();
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

// LDS image of one net's head parameters: g2[512] be2[512] W3[out][512] (padded to OUTMAX rows) b3[out]
template <int OUTMAX>
struct HeadImage {
    static constexpr int kStride = (2 + OUTMAX) * H2 + 8;
    static constexpr int kPer = ((2 + OUTMAX) * (H2 / 4) + kWide - 1) / kWide;  // float4 per thread to stage it
    v4f v[kPer];  // (a native vector type: HIP's float4 is a struct whose copies become memcpy calls, and two of them in an array stay an
                  //  alloca — in scratch, or promoted into 32 KB of LDS — instead of registers)
    float b3v;
    // g2, be2, W3 rows are contiguous in the parameter block from g2()
    __device__ __forceinline__ void fetch(const float* __restrict__ net, const Mlp& m, int tid) {
        // Loads from clamped (valid) addresses, unconditional INSIDE a wave; store() keeps only the live ones.  (A load under a per-lane
        // condition merges with its zero default through register copies that WAIT for the data — in the middle of the caller's issue
        // phase.)  Whole waves past the image's end skip theirs behind a scalar branch: the address pipeline takes 16 lanes per clock
        // whatever they ask for, and a one-output image is 6 waves' worth of the 16.
        // (a skipping wave leaves its members unset — it never stores them; a default value would be merged with the loaded one through a
        //  register copy that waits for the data, the very thing this function avoids)
        const int w0 = __builtin_amdgcn_readfirstlane(tid);
        if (w0 < m.out) b3v = net[m.b3() + (tid < m.out ? tid : 0)];
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int e = tid + i * kWide;
            if (w0 + i * kWide < (2 + m.out) * (H2 / 4)) v[i] = reinterpret_cast<const v4f*>(net + m.g2())[e < (2 + m.out) * (H2 / 4) ? e : 0];
        }
    }
    __device__ __forceinline__ void store(float* hp, const float* __restrict__ net, const Mlp& m, int tid) const {
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int e = tid + i * kWide;
            if (e < (2 + m.out) * (H2 / 4)) reinterpret_cast<v4f*>(hp)[e] = v[i];
        }
        (void)net;
        if (tid < m.out) hp[(2 + OUTMAX) * H2 + tid] = b3v;
    }
};
// head from registers + the LDS image: LN2 stats of z, y = g2 xhat + be2, o[j] = act(y) . W3[j] + b3[j]
// OUTMAX = how many outputs are computed, IMG = head width the LDS image was laid out for (HeadImage<IMG>)
template <int OUTMAX, int IMG, bool RELU>
__device__ __forceinline__ void head_regs(const RowReg<H2>& z, const float* hp, int out, float slope, RowReg<H2>& xhat, RowReg<H2>& y,
                                          float& mean, float& rstd, float (&o)[OUTMAX], int no_ln = 0) {
    RowReg<H2> g, be;
    row_stats<8>(z.v, H2, mean, rstd);
    if (no_ln) { mean = 0.0f; rstd = 1.0f; }
    g.load(hp);
    be.load(hp + H2);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        xhat.v[i] = (z.v[i] - mean) * rstd;
        y.v[i] = g.v[i] * xhat.v[i] + be.v[i];
    }
#pragma unroll
    for (int j = 0; j < OUTMAX; ++j) {
        float acc = 0.0f;
        if (j < out) {
            RowReg<H2> w;
            w.load(hp + (2 + j) * H2);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += act_f<RELU>(y.v[i], slope) * w.v[i];
            acc = wave_sum(acc) + hp[(2 + IMG) * H2 + j];
        }
        o[j] = acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// act_fused: the whole policy for 16 observation rows in ONE workgroup — layer 1 + LN1 (VALU), z2 = h1 W2^T for all 512
// columns (two 16-column MFMA tiles per wave, K = 256; W2 streams through two LDS buffers in 16-wide k-chunks, register-
// prefetched two chunks ahead), then LN2 + final layer + tanh + exploration noise + clamp with one wave per row straight from the LDS copy
// of z2.  No z2 round trip through HBM, no second launch.
// chooseAction / chooseActionSmallNoise / chooseActionNoNoise, HIRL.py:192-212.
// ---------------------------------------------------------------------------------------------------------------
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
};

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
// The bf16 image of W2 is stored in the ORDER THE ACTING KERNEL READS IT: for each (column tile of 16, k-slab of 32) one contiguous
// 1 KB block holding lane 0..63's 16 bytes — lane (r = column in the tile, g = k group): elements k = 32 slab + 8 g .. + 7 of column r.
// A wave's B-fragment load is then ONE contiguous kilobyte (8 full cache lines); from the row-major image every 4-lane quad of the same
// load touched four different columns = four 16-byte requests, and the 256 KB image took ~5 us to reach the registers.
// the fp32 image: one 1 KB block per (column tile of 16, k-chunk of 16), lane (r, g): k = 16 chunk + 4 g .. + 3 of column r
__host__ __device__ inline uint32_t w2f_image_index(uint32_t col, uint32_t k) {
    return ((((col >> 4) * 16u + (k >> 4)) * 4u + ((k >> 2) & 3u)) * 16u + (col & 15u)) * 4u + (k & 3u);
}
__host__ __device__ inline uint32_t w2_image_index(uint32_t col, uint32_t k) {
    return ((((col >> 4) * 8u + (k >> 5)) * 4u + ((k >> 3) & 3u)) * 16u + (col & 15u)) * 8u + (k & 7u);
}
constexpr int LDB1 = H1 + 16;  // bf16 h1 tile pitch (elements) = 136 dwords = 8 mod 64: the ds_read_b128 A-operand read is conflict-free

constexpr int ACT_KC = 16;            // k-chunk of W2 staged through LDS (64 B per column), double-buffered
constexpr int ACT_LDW = ACT_KC + 8;   // pitch = 8 mod 16 dwords: conflict-free ds_read_b128 (see LDA1)
constexpr int ACT_NCH = H1 / ACT_KC;  // 16 chunks

__device__ __forceinline__ float pick8(const float (&o)[8], int i) {
    return i == 0 ? o[0] : i == 1 ? o[1] : i == 2 ? o[2] : i == 3 ? o[3] : i == 4 ? o[4] : i == 5 ? o[5] : i == 6 ? o[6] : o[7];
}
// one standard-normal draw per (row, component j) of the acting kernels: Philox4x32-10(seed; row, call, tag) + Box-Muller
__device__ __forceinline__ float philox_normal(uint32_t row, uint32_t call, uint32_t tag, uint64_t seed, int j) {
    uint32_t u[4];
    philox4x32_10(row, call, tag, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), u);
    const float ua = u01(u[j & 2]), ub = u01(u[(j & 2) + 1]);
    const float rad = sqrtf(-2.0f * logf(ua)), ang = 6.28318530717958647692f * ub;
    return (j & 1) ? rad * sinf(ang) : rad * cosf(ang);
}

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
template <int NRT, bool GAUSS, bool ENV, bool BF16, bool RELU, bool F32I = false>
__global__ __launch_bounds__(kWide) void act_fused_kernel(ActFusedArgs A) {
    static_assert(!(BF16 && F32I), "one image format at a time");
    constexpr int ROWS = NRT * RT;
    __shared__ float s_act[ENV ? ROWS * 4 : 4];
    __shared__ float s_noise[ROWS * 4];  // exploration noise of the workgroup's rows, drawn by the last wave(s) under the prologue's loads
    __shared__ unsigned s_base;  // ring slot of the workgroup's first row
    __shared__ int s_nstore;
    static_assert(2 * H2 * ACT_LDW >= ROWS * LDA2, "the z2 tile reuses the W2 chunk buffers");
    // fp32: two W2 chunk buffers (reused for z2 and, in the env tail, the replay rows / next observations)
    // bf16: the z2 tile, then the replay rows / next observations, and the bf16 h1 tile
    constexpr bool IMG = BF16 || F32I;  // W2 comes from an image straight into registers: no chunk buffers in LDS
    constexpr int kTileA = IMG ? ROWS * LDA2 : H2 * ACT_LDW;
    constexpr int kTileB = IMG ? (ENV ? ROWS * (hxenv::kRowPitch + HX_OBS_DIM) : 4) : H2 * ACT_LDW;
    __shared__ __attribute__((aligned(16))) float lds[ROWS * LDA1 + ROWS * XP + ROWS * 2 + H1 * 13 + kTileA + kTileB];
    __shared__ __attribute__((aligned(16))) __bf16 h1b[BF16 ? ROWS * LDB1 : 8];
    float* h1s = lds;
    float* xs = h1s + ROWS * LDA1;
    float* sts = xs + ROWS * XP;
    float* w1s = sts + ROWS * 2;
    float* wb0 = w1s + H1 * 13;        // [H2][ACT_LDW]: even k-chunks of W2, every column
    float* wb1 = wb0 + kTileA;         // odd k-chunks
    float* z2s = wb0;                  // [ROWS][LDA2] once the last chunk has been multiplied
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r0 = blockIdx.x * ROWS;
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
    if constexpr (!BF16 && !F32I) {
        ACT_LOAD(ra0, rb0, 0);
        ACT_LOAD(ra1, rb1, 1);
    }
    // head parameters (g2, be2, W3, b3): requested now, parked in 4-8 registers, laid out in LDS once h1 is dead
    typedef HeadImage<GAUSS ? 8 : 4> Img;
    static_assert(Img::kStride <= ROWS * LDA1 + ROWS * XP + ROWS * 2 + H1 * 13, "the head image reuses the prologue's LDS");
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
    if constexpr (!BF16 && !F32I) {  // behind the prologue's own operands
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
    const int cw = BF16 ? ((wave + (int)blockIdx.x) & 15) : wave;
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
#pragma unroll
    for (int t = 0; t < NRT; ++t) {  // LN1 statistics: wave w owns rows w, 16 + w
        const int row = t * RT + wave;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = h1s[row * LDA1 + i * 64 + lane];
        float mean, rstd;
        row_stats<4>(v, H1, mean, rstd);
        if (m.no_ln) { mean = 0.0f; rstd = 1.0f; }
        if (lane == 0) {
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
            const float hv = act_f<RELU>(g1v * ((z1[t][r] - sts[row * 2]) * sts[row * 2 + 1]) + be1v, slope);
            if (BF16) h1b[row * LDB1 + u] = (__bf16)hv;  // v_cvt_pk_bf16_f32: round to nearest even
            else h1s[row * LDA1 + u] = hv;
        }
    if (!BF16 && !F32I) {
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
                    const uint4 aq = *reinterpret_cast<const uint4*>(h1b + (t * RT + r) * LDB1 + 32 * sl + 8 * g);
                    const v8bf a8 = __builtin_bit_cast(v8bf, aq);
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, __builtin_bit_cast(v8bf, bq[0][sl]), acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, __builtin_bit_cast(v8bf, bq[1][sl]), acc[t][1], 0, 0, 0);
                }
            }
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
        for (int c = 0; c < ((BF16 || F32I) ? 0 : ACT_NCH); c += 4) {
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
    __syncthreads();
    STAMP();
    // ENV: the env lanes (wave 0, two lanes per env: hx_env_dev.h "Pair") request their state words and the current observation
    // now — the head phase hides the round trip
    hxenv::Stepper<true> envT;
    float envPrev[HX_OBS_DIM];
    const int env_e = lane >> 1;            // env of this lane inside the workgroup's rows
    const bool env_opp = (lane & 1) != 0;   // this lane owns the opponent aircraft
    if (ENV && wave == 0 && env_e < nrow) {
        envT.load(A.state, A.stride, r0, (uint32_t)env_e, env_opp);
#pragma unroll
        for (int j = 0; j < HX_OBS_DIM; ++j) envPrev[j] = (A.o.ring && env_opp) ? A.obs[((size_t)r0 + env_e) * HX_OBS_DIM + j] : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < NRT; ++t) {  // head: wave w owns rows w, 16 + w
        const int lr = t * RT + wave;
        if (lr >= nrow) continue;
        const int r = r0 + lr;
        RowReg<H2> xh, y, z;
        float mean, rstd;
        z.load(z2s + lr * LDA2);
        if (!GAUSS) {
            float o[4];
            head_regs<4, 4, RELU>(z, hps, m.out, slope, xh, y, mean, rstd, o, m.no_ln);
            if (lane < 4) {
                float a = fast_tanh(lane == 0 ? o[0] : lane == 1 ? o[1] : lane == 2 ? o[2] : o[3]);  // no dynamic register index
                if (A.noise) {
                    a = fminf(fmaxf(a + A.noise[(A.noise_per_row ? (size_t)r * 4 : 0) + lane], -1.0f), 1.0f);
                } else if (A.sigma > 0.0f) {
                    a = fminf(fmaxf(a + A.sigma * s_noise[lr * 4 + lane], -1.0f), 1.0f);
                }
                A.actions[(size_t)r * 4 + lane] = a;
                if (ENV) s_act[lr * 4 + lane] = a;
            }
        } else {
            float o[8];
            head_regs<8, 8, RELU>(z, hps, m.out, slope, xh, y, mean, rstd, o, m.no_ln);
            if (lane < 4) {
                // (selected from VALUES: a select chain over the elements of the array itself is folded back into a run-time index, which
                //  puts the array in scratch or — promoted — in 32 KB of LDS)
                const float o0 = o[0], o1 = o[1], o2 = o[2], o3 = o[3], o4 = o[4], o5 = o[5], o6 = o[6], o7 = o[7];
                const float mu = lane == 0 ? o0 : lane == 1 ? o1 : lane == 2 ? o2 : o3;
                float a = mu;
                if (A.mode != 0) {
                    const float ls = fminf(fmaxf(lane == 0 ? o4 : lane == 1 ? o5 : lane == 2 ? o6 : o7, -20.0f), 2.0f);  // model.py:65-66
                    const float e = A.mode == 1 ? A.noise[(size_t)r * 4 + lane] : s_noise[lr * 4 + lane];
                    a = mu + expf(ls) * e;
                }
                a = tanhf(a);
                A.actions[(size_t)r * 4 + lane] = a;
                if (ENV) s_act[lr * 4 + lane] = a;
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
                if (lane < HX_STAT_COUNT && mine) atomicAdd((unsigned long long*)&A.o.stats[lane], (unsigned long long)mine);
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
    STAMP_FLUSH(56, (blockIdx.x == 0 || blockIdx.x == 200) && tid == 0);
    SPAN_LOG();
}

// 16 rows per workgroup fill the chip up to 4,096 rows; from 8,192 rows on 32 rows per workgroup reuse every W2 chunk twice
// The env tail pays while the launch is ONE round of workgroups (256 CUs x 16 or 32 rows): beyond that every extra round repeats
// the ~8 us tail, and the env kernel on its own (thousands of envs per launch, 10-14 us) is the cheaper way.
constexpr int64_t kFuseEnvMax = 8192;

template <bool GAUSS, bool BF16, bool RELU, bool F32I = false>
static void launch_act_t(const ActFusedArgs& H, hipStream_t st) {
    const bool env = H.state != nullptr;
    // 32 rows per workgroup: from 8,192 rows on (fp32: below that, 16-row workgroups fill the chip and the fp32 MFMA work per workgroup
    // is the longer pole); bf16 from 2,048 rows on: there the launch is bound by every workgroup pulling W2 through L2, not by MFMA
    static const int nrt2_bf16 = getenv("HX_ACT_BF16_NRT2_ROWS") ? atoi(getenv("HX_ACT_BF16_NRT2_ROWS")) : 8192;  // tuning knob
    if (H.rows >= (BF16 ? nrt2_bf16 : 8192)) {
        const dim3 grid((unsigned)((H.rows + 2 * RT - 1) / (2 * RT)));
        if (env) hipLaunchKernelGGL((act_fused_kernel<2, GAUSS, true, BF16, RELU, F32I>), grid, dim3(kWide), 0, st, H);
        else hipLaunchKernelGGL((act_fused_kernel<2, GAUSS, false, BF16, RELU, F32I>), grid, dim3(kWide), 0, st, H);
    } else {
        const dim3 grid((unsigned)((H.rows + RT - 1) / RT));
        if (env) hipLaunchKernelGGL((act_fused_kernel<1, GAUSS, true, BF16, RELU, F32I>), grid, dim3(kWide), 0, st, H);
        else hipLaunchKernelGGL((act_fused_kernel<1, GAUSS, false, BF16, RELU, F32I>), grid, dim3(kWide), 0, st, H);
    }
}
template <bool GAUSS>
static void launch_act(const ActFusedArgs& H, hipStream_t st) {
    // the activation is a compile-time ReLU when the slope is 0 (HIRL, SAC; the Gaussian policy is a Linear-ReLU stack by definition)
    if (GAUSS || H.slope == 0.0f) {
        if (H.w2b) launch_act_t<GAUSS, true, true>(H, st);
        else if (H.w2f) launch_act_t<GAUSS, false, true, true>(H, st);
        else launch_act_t<GAUSS, false, true>(H, st);
    } else {
        if (H.w2b) launch_act_t<false, true, false>(H, st);
        else if (H.w2f) launch_act_t<false, false, false, true>(H, st);
        else launch_act_t<false, false, false>(H, st);
    }
}

// bf16 image of a [n] fp32 array (round to nearest even): the policy's W2 for the BF16 acting kernels
__global__ __launch_bounds__(kThreads) void pack_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int n) {
    const int i = (blockIdx.x * kThreads + threadIdx.x) * 2;  // row-major element (column i / 256, k = i % 256); pairs stay adjacent in the image
    if (i + 1 < n) {
        const float2 v = *reinterpret_cast<const float2*>(src + i);
        typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
        const v2bf r = {(__bf16)v.x, (__bf16)v.y};
        *reinterpret_cast<unsigned*>(dst + w2_image_index((uint32_t)i / H1, (uint32_t)i % H1)) = __builtin_bit_cast(unsigned, r);
    }
}

__global__ __launch_bounds__(kThreads) void pack_f32i_kernel(const float* __restrict__ src, float* __restrict__ dst, int n) {
    const int i = (blockIdx.x * kThreads + threadIdx.x) * 4;  // four consecutive k of one column: adjacent in the image too
    if (i < n) *reinterpret_cast<float4*>(dst + w2f_image_index((uint32_t)i / H1, (uint32_t)i % H1)) = *reinterpret_cast<const float4*>(src + i);
}

// ---------------------------------------------------------------------------------------------------------------
// bwd_l2: head + loss gradient + LN2 backward (prologue), dh1 = dz2 W2 (MFMA)
// ---------------------------------------------------------------------------------------------------------------
enum { BM_CRITIC_TD = 0, BM_CRITIC_PI = 1, BM_ACTOR_PI = 2, BM_ACTOR_BC = 3, BM_GIVEN = 4 };

struct BwdJob {
    const float* net;
    Mlp m;
    Slot ws;
    int rows;
    int mode;
    // BM_CRITIC_TD: y = r + gamma min(Q1', Q2') (1 - d)     HIRL.py:270-274
    Head t1, t2;  // target critic heads evaluated on (s', a')
    RowSrc src;   // minibatch rows (reward col 30, done col 31; BC target action cols 13..16)
    float gamma;
    // BM_CRITIC_PI: optional soft-weight count  HIRL.py:299-306
    Head soft;  // critic Q1 evaluated on (s, bc_actor(s)); net == nullptr: off
    // BM_ACTOR_PI: dL/da from the critic's layer-1 backward
    Head crit;    // the critic Q1 slot evaluated on (s, pi) (its dh1, z1, st1 are read)
    float lambda; // BM_ACTOR_BC: loss_lambda (HIRL.py:182)
    // SAC: per-row target bonus (entropy of the next action, scaled by *bonus_scale = alpha) and which losses[] slot a TD job feeds
    const float* bonus;
    const float* bonus_scale;
    int loss_slot;
};
struct BwdArgs {
    BwdJob job[2];
    int njobs;
    float slope;
    float inv_batch;  // 1 / B
    float* losses;    // [8]: critic, actor, bc, rl, bc_fire, bc_weight, -, -
    int* soft_count;
};

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
    uint32_t pad_;
};
static_assert(sizeof(BwdJobC) == 128, "two s_load_dwordx16");
struct BwdArgsC {
    BwdJobC job[2];
};
inline BwdJobC pack_bwd(const BwdJob& J, const BwdArgs& A) {
    BwdJobC c{};
    c.net = J.net; c.ws = J.ws.x;
    const Head& h1 = J.mode == BM_CRITIC_TD ? J.t1 : (J.mode == BM_CRITIC_PI ? J.soft : J.crit);
    c.h1_net = h1.net; c.h1_ws = h1.ws.x;
    c.h2_net = J.t2.net; c.h2_ws = J.t2.ws.x;
    c.src = J.src.main; c.bonus = J.bonus; c.bonus_scale = J.bonus_scale;
    c.cfg = mlp_bits(J.m) | (mlp_bits(h1.m) << 10) | (mlp_bits(J.t2.m) << 20);
    c.cfg2 = (uint32_t)J.mode | ((uint32_t)J.loss_slot << 3);
    c.rows = J.rows; c.gamma = J.gamma; c.lambda = J.lambda; c.slope = A.slope; c.inv_batch = A.inv_batch;
    c.losses = A.losses; c.soft_count = A.soft_count;
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
// footprint of each below 128 at 16 waves per workgroup).
// Latency structure (what matters at B = 128, one workgroup per CU): EVERY global load of the workgroup — the W2 fragment
// of the MFMA phase, the z2 rows, labels, the other nets' rows, all head parameters, the epilogue's z1 — is issued at
// entry; there is ONE wait; head parameters are shared through LDS; the rest runs out of registers and LDS.
template <int GRP, bool RELU>
__global__ __launch_bounds__(kWide) void bwd_l2_kernel(BwdArgsC AC) {
    __shared__ __attribute__((aligned(16))) float dz2s[RT * LDA2];
    __shared__ __attribute__((aligned(16))) float kred[(kKSB - 1) * kCTB * 256];  // split-K partial tiles
    constexpr int IMG = GRP == 3 ? 8 : 4;  // head width of this instantiation's LDS images
    // head width known at compile time: the critic jobs (GRP 0, 1) have ONE output, the actor jobs (GRP 2) four; GRP 3 (SAC: policy 8 wide,
    // Q heads 1) keeps the run-time width.  A run-time trip count over dout[] costs a select chain per step (no indexed registers).
    constexpr int NOUT = GRP <= 1 ? 1 : (GRP == 2 ? 4 : 0);
    constexpr int OUTW = NOUT ? NOUT : IMG;
    typedef HeadImage<IMG> Img;
    constexpr int kHpStride = Img::kStride;
    __shared__ __attribute__((aligned(16))) float hps[(GRP == 0 ? 3 : 1) * kHpStride];
    __shared__ __attribute__((aligned(16))) float c1s[GRP == 2 ? H1 * 6 : 4];  // critic layer 1: g1 be1 W1[:,13..16]
    __shared__ float red[16][4];
    __shared__ float st1s[RT * 2];  // LN1 stats of the tile's rows (epilogue)
    // TD job (GRP 0): EIGHT rows per workgroup, a wave PAIR per row — wave w (role 0) owns the row's own head, loss gradient and LN2
    // backward, wave w + 8 (role 1) the two target heads; min(Q1', Q2') crosses through LDS.  Twice the workgroups (256 at B = 128, two
    // jobs): half the row bytes per CU (rows the previous launches produced on all eight XCDs arrive at ~19 B/clk/CU, and 96 KB of them
    // were in front of this prologue), and the three heads of a row no longer run one after the other on one wave.  The MFMA tile keeps
    // its 16 rows (8 of them zero): that phase is the short one.
    // The critic-PI job (GRP 1) pairs the same way (role 1: the soft head), the actor jobs (GRP 2) too (role 1: dL/da from the critic's
    // layer-1 backward, four dot products over 256 hidden units).
    constexpr bool PAIRED = GRP <= 2;
    constexpr int RTB = PAIRED ? RT / 2 : RT;
    __shared__ float tq[PAIRED ? RTB * 4 : 1];

    const int b = blockIdx.x;
    const BwdJobC& jc = AC.job[blockIdx.y];
    const BwdJob J = expand_bwd(jc);
    struct { float slope, inv_batch; float* losses; int* soft_count; } A{jc.slope, jc.inv_batch, jc.losses, jc.soft_count};
    const int rt = b / kColWgB, nt = b % kColWgB;
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
    BFrag<H2 / kKSB> bfrag;
    // Row loads are UNCONDITIONAL (R is clamped to a valid row; a wave without a row never uses them): behind `if (live)` the compiler
    // zero-fills the registers, loads under a branch and — where the two versions merge — WAITS for the loads in the middle of the issue phase.
    RowReg<H2> z, za, zb;
    float lab0 = 0.f, lab1 = 0.f, tgt[4] = {0.f, 0.f, 0.f, 0.f};
    RowReg<H1> cdh, cz;
    float cst0 = 0.f, cst1 = 0.f, cs1 = 0.f, cs2 = 0.f;
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
    float c1v[2] = {0.f, 0.f};
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
    bfrag.load(J.net + J.m.W2() + (size_t)(kq * (H2 / kKSB)) * H1 + n0 + (lane & 15), H1);
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
    __syncthreads();
    STAMP();

    // ---------------- prologue: head, loss gradient, LN2 backward (registers + LDS only) ----------------
    float part[4] = {0.f, 0.f, 0.f, 0.f};  // loss partials of this row
    int cnt = 0;
    float* drow = dz2s + wave * LDA2;
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
        zero.store_lds(drow);
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
        dx.store_lds(drow);
        if (lead) {
            dx.store(J.ws.dz2 + R * H2);
            if (lane == 0) {
                J.ws.st2[R * 2] = mean;
                J.ws.st2[R * 2 + 1] = rstd;
            }
            if (GRP != 3 && lane < 4) {
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
        }
    }
    // ---------------- dh1 tile on fp32 MFMA: wave = (column tile ct, K quarter kq), K = 512 ----------------
    {
        const int r = lane & 15, g = lane >> 4;
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        acc = tile_a_lds_b_frag<H2 / kKSB>(dz2s + kq * (H2 / kKSB), LDA2, bfrag, acc);
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
        SPAN_LOG();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// SAC (hirl/agents/SAC): small per-row kernels around the shared fwd_l2 / bwd_l2<3> / wgrad machinery.  One wave per row.
// ---------------------------------------------------------------------------------------------------------------
// GaussianPolicy.sample (SAC/model.py:69-82) from the policy's z2 rows: mean, log_std = chunk(head), clamp(log_std, -20, 2),
// x = mean + exp(log_std) eps, a = tanh(x), entropy = -sum_j (log N(x_j) - log(1 - a_j^2 + 1e-6)).
struct GaussArgs {
    const float* net;
    Mlp m;
    const float* z2;
    const float* eps;  // [rows][4] standard-normal draws; nullptr with mode 2 -> Philox; mode 0 ignores it
    int rows, mode;    // 0: exploit tanh(mean) (agent.py:191-196), 1: sample with eps, 2: sample with Philox(seed; row, call)
    float* act;        // [rows][4]
    float* ent;        // [rows] or nullptr
    float* aux;        // [rows][16] or nullptr: a[4], sigma*eps[4], clamp pass-through mask[4], entropy
    uint64_t seed;
    uint32_t row0, call;
};
__global__ __launch_bounds__(kThreads) void gauss_head_kernel(GaussArgs A) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= A.rows) return;
    RowReg<H2> xh, y;
    float mean_, rstd_, o[8];
    head_row<8, true>(A.z2 + (size_t)r * H2, A.net, A.m, 0.0f, xh, y, mean_, rstd_, o);
    const int j = lane & 3;
    const float mu = pick8(o, j), ls_raw = pick8(o, 4 + j);
    const float ls = fminf(fmaxf(ls_raw, -20.0f), 2.0f);  // model.py:65-66
    const float sd = expf(ls);
    float e = 0.0f;
    if (A.mode == 1) e = A.eps[(size_t)r * 4 + j];
    if (A.mode == 2) {
        uint32_t u[4];
        philox4x32_10(A.row0 + (uint32_t)r, A.call, 0x53414331u, 0u, (uint32_t)A.seed, (uint32_t)(A.seed >> 32), u);
        const float ua = u01(u[j & 2]), ub = u01(u[(j & 2) + 1]);
        const float rad = sqrtf(-2.0f * logf(ua)), ang = 6.28318530717958647692f * ub;
        e = (j & 1) ? rad * sinf(ang) : rad * cosf(ang);
    }
    const float se = sd * e;
    const float a = tanhf(A.mode == 0 ? mu : mu + se);
    // Normal(mean, std).log_prob(x) with x - mean = sd * eps, minus the tanh correction   model.py:77-78
    const float logp = (-(se * se) / (2.0f * (sd * sd)) - ls - 0.91893853320467274f) - logf(1.0f - a * a + 1e-6f);
    float h = lane < 4 ? -logp : 0.0f;
    h = sum16(h);  // lanes 0..3 sit in the first 16-lane row
    if (lane < 4) {
        A.act[(size_t)r * 4 + j] = a;
        if (A.aux) {
            float* x = A.aux + (size_t)r * 16;
            x[j] = a;
            x[4 + j] = se;
            x[8 + j] = (ls_raw >= -20.0f && ls_raw <= 2.0f) ? 1.0f : 0.0f;
            if (lane == 0) x[12] = h;
        }
    }
    if (lane == 0 && A.ent) A.ent[r] = h;
}

// min(Q1, Q2)(s, a~) for the policy loss (SAC/agent.py:380-383): writes each head's output gradient -w/B (w = 1 for the smaller
// head, 1/2 each on a tie: torch.min's subgradient) and accumulates the -min(Q)/B part of the policy loss.
struct QSelArgs {
    const float* net1;
    const float* net2;
    Mlp m;
    Slot s1, s2;
    int rows;
    float inv_batch;
    float* losses;
};
__global__ __launch_bounds__(kThreads) void q_select_kernel(QSelArgs A) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= A.rows) return;
    RowReg<H2> xh, y;
    float mean_, rstd_, q1[1], q2[1];
    head_row<1, true>(A.s1.z2 + (size_t)r * H2, A.net1, A.m, 0.0f, xh, y, mean_, rstd_, q1);
    if (lane == 0) { A.s1.st2[r * 2] = mean_; A.s1.st2[r * 2 + 1] = rstd_; }
    head_row<1, true>(A.s2.z2 + (size_t)r * H2, A.net2, A.m, 0.0f, xh, y, mean_, rstd_, q2);
    if (lane == 0) {
        A.s2.st2[r * 2] = mean_; A.s2.st2[r * 2 + 1] = rstd_;
        const float w1 = q1[0] < q2[0] ? 1.0f : (q1[0] == q2[0] ? 0.5f : 0.0f);
        A.s1.dout[(size_t)r * OW] = -w1 * A.inv_batch;
        A.s2.dout[(size_t)r * OW] = -(1.0f - w1) * A.inv_batch;
        atomicAdd(&A.losses[2], -fminf(q1[0], q2[0]) * A.inv_batch);
    }
}

// Gradient of the policy loss mean(-min Q - alpha H) (SAC/agent.py:404-406) wrt the policy head's 8 pre-activations: the
// critics' input gradients wrt the action (both heads; the unselected one carries zeros) chained through a = tanh(mean + sigma eps),
// plus the entropy term.  Also accumulates -alpha H / B (policy loss) and stores mean H (for the alpha step).
struct PDoutArgs {
    const float* q1net;
    const float* q2net;
    Mlp mq;
    Slot c1, c2, pol;
    const float* aux;
    const float* alpha_state;  // [4]: log_alpha, m, v, alpha
    int rows;
    float inv_batch;
    float* losses;
};
__global__ __launch_bounds__(kThreads) void policy_dout_kernel(PDoutArgs A) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= A.rows) return;
    float da[4] = {0.f, 0.f, 0.f, 0.f};
    for (int hsel = 0; hsel < 2; ++hsel) {
        const Slot& C = hsel ? A.c2 : A.c1;
        const float* net = hsel ? A.q2net : A.q1net;
        RowReg<H1> dh, z;
        dh.load(C.dh1 + (size_t)r * H1);
        z.load(C.z1 + (size_t)r * H1);
#pragma unroll
        for (int c = 0; c < 4; ++c) {  // hidden unit k = lane*4 + c; plain stack: dz1 = dh1 * relu'(z1)
            const int k = lane * 4 + c;
            const float dz1 = act_bwd<true>(dh.v[c], z.v[c], 0.0f);
            const float* w = net + A.mq.W1() + k * A.mq.in + 13;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) da[jj] += dz1 * w[jj];
        }
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) da[jj] = wave_sum(da[jj]);
    const float alpha = A.alpha_state[3];
    if (lane < 4) {
        const float* x = A.aux + (size_t)r * 16;
        const float a = x[lane], se = x[4 + lane], mask = x[8 + lane];
        const float t = 1.0f - a * a;
        const float dHdx = (-2.0f * a * t) / (t + 1e-6f);
        const float dLdx = (lane == 0 ? da[0] : lane == 1 ? da[1] : lane == 2 ? da[2] : da[3]) * t - alpha * A.inv_batch * dHdx;
        A.pol.dout[(size_t)r * OW + lane] = dLdx;
        A.pol.dout[(size_t)r * OW + 4 + lane] = (dLdx * se - alpha * A.inv_batch) * mask;
        if (lane == 0) atomicAdd(&A.losses[2], -alpha * x[12] * A.inv_batch);  // logged only
    }
    // The mean entropy feeds the log-alpha step, so it must not depend on arrival order: the per-row entropies were written by
    // the head kernel before this launch, one wave adds them in a fixed order.
    if (blockIdx.x == 0 && wave == 0) {
        float s = 0.0f;
        for (int rr = lane; rr < A.rows; rr += 64) s += A.aux[(size_t)rr * 16 + 12];
        s = wave_sum(s);
        if (lane == 0) A.losses[4] = s * A.inv_batch;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// wgrad: all parameter gradients of one MLP block from up to two slots (scaled), into the flat gradient buffer
// ---------------------------------------------------------------------------------------------------------------
struct WgJob {
    const float* net;  // parameters (W3, g2, be2, g1, be1 are read)
    float* grad;       // gradient block, same layout
    Mlp m;
    Slot ws[2];
    int rows[2];
    int nslots;
    int wmode[2];  // per slot: 0 = scale 1, 1 = scale (1 - w), 2 = scale w
    // ADAM instantiation: the optimizer step of this block in the same launch (single GPU: no exchange between gradient and step)
    float* p; float* mom; float* var;  // parameters (== net) and Adam moments, same layout
    float* target;                 // nullptr, or the target network's block: soft_update with the new parameters (HIRL.py:11-13)
    uint16_t* w2b;                 // nullptr, or the bf16 image of W2 to refresh
    float* w2f;                    // nullptr, or the fp32 image of W2 to refresh
};
struct WgAdam {
    float b1, b2, eps, step_size, bc2_sqrt, tau;
    int finish_actor, use_bc;  // thread 0 of the launch finishes actor_loss / bc_weight (HIRL.py:321,334)
    float* losses;
    float* wstate;
};
struct WgArgs {
    WgJob job[2];
    int njobs;
    float slope;
    WgAdam ad;
    // effective BC weight  w: 0 = given, 1 = estimate from soft_count (HIRL.py:304-306), 2 = reuse *wstate
    int w_kind;
    float w_given, warm, inv_batch;
    const int* soft_count;
    const float* wstate;
};

// torch.optim.Adam (defaults) on one element, and soft_update.  Contraction is OFF in these two: HIP's __fmul_rn / __fsub_rn are plain
// operators, which the compiler may or may not fuse depending on the surrounding kernel — and the fused wgrad + Adam launch must
// round exactly like adam_kernel (the one-call and the staged update paths are compared bit for bit).
#pragma clang fp contract(off)
__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float g, float b1, float b2, float eps, float step_size, float bc2_sqrt) {
    m = __builtin_fmaf(g, 1.0f - b1, m * b1);
    v = __builtin_fmaf(g * g, 1.0f - b2, v * b2);
    const float denom = __builtin_amdgcn_sqrtf(v) * __builtin_amdgcn_rcpf(bc2_sqrt) + eps;  // hardware sqrt / reciprocal (1 ulp each)
    p = p - step_size * (m * __builtin_amdgcn_rcpf(denom));
}
__device__ __forceinline__ float polyak_update(float target, float p, float tau) { return target * (1.0f - tau) + p * tau; }  // HIRL.py:13
#pragma clang fp contract(fast)
__device__ __forceinline__ float effective_w(int kind, float given, float warm, float inv_batch, const int* count, const float* wstate) {
    float w = given;
    if (kind == 1) w = (float)(*count) * inv_batch + warm;
    if (kind == 2) w = *wstate;
    return w > 1.0f ? 1.0f : w;  // HIRL.py:308
}

// compact kernel argument (see FwdJobC): 128 bytes per job; everything a workgroup needs before its first vector load is in its own job
struct WgJobC {
    const float* net; float* grad;
    float* ws0; float* ws1;
    float* mom; float* var; float* target; uint16_t* w2b; float* w2f;
    uint32_t cfg;  // m:10 | nslots:2 | wmode0:2 | wmode1:2 | w_kind:2 | adam.finish_actor:1 | adam.use_bc:1
    int32_t rows0, rows1;
    float slope, w_given, warm, inv_batch;
    float b1, b2, eps, step_size, bc2_sqrt, tau;
    uint32_t pad_;
    const int* soft_count; float* wstate; float* losses;
};
static_assert(sizeof(WgJobC) == 152, "WgJobC layout");
struct WgArgsC {
    WgJobC job[2];
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
        if (J.target) J.target[idx] = polyak_update(t, p, a.tau);
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
    A.slope = c.slope; A.w_kind = (int)((c.cfg >> 16) & 3u); A.w_given = c.w_given; A.warm = c.warm; A.inv_batch = c.inv_batch;
    A.soft_count = c.soft_count; A.wstate = c.wstate;
    A.ad.b1 = c.b1; A.ad.b2 = c.b2; A.ad.eps = c.eps; A.ad.step_size = c.step_size; A.ad.bc2_sqrt = c.bc2_sqrt; A.ad.tau = c.tau;
    A.ad.finish_actor = (int)((c.cfg >> 18) & 1u); A.ad.use_bc = (int)((c.cfg >> 19) & 1u);
    A.ad.losses = c.losses; A.ad.wstate = c.wstate;
}
inline WgJobC pack_wg(const WgJob& J, const WgArgs& A) {
    WgJobC c{};
    c.net = J.net; c.grad = J.grad; c.ws0 = J.ws[0].x; c.ws1 = J.nslots > 1 ? J.ws[1].x : J.ws[0].x;
    c.mom = J.mom; c.var = J.var; c.target = J.target; c.w2b = J.w2b; c.w2f = J.w2f;
    c.cfg = mlp_bits(J.m) | ((uint32_t)J.nslots << 10) | ((uint32_t)J.wmode[0] << 12) | ((uint32_t)J.wmode[1] << 14) | ((uint32_t)A.w_kind << 16) |
            ((uint32_t)(A.ad.finish_actor ? 1 : 0) << 18) | ((uint32_t)(A.ad.use_bc ? 1 : 0) << 19);
    c.rows0 = J.rows[0]; c.rows1 = J.nslots > 1 ? J.rows[1] : J.rows[0];
    c.slope = A.slope; c.w_given = A.w_given; c.warm = A.warm; c.inv_batch = A.inv_batch;
    c.b1 = A.ad.b1; c.b2 = A.ad.b2; c.eps = A.ad.eps; c.step_size = A.ad.step_size; c.bc2_sqrt = A.ad.bc2_sqrt; c.tau = A.ad.tau;
    c.soft_count = A.soft_count; c.wstate = const_cast<float*>(A.wstate); c.losses = A.ad.losses;
    return c;
}

template <bool ADAM, bool RELU>
__global__ __launch_bounds__(kWide) void wgrad_kernel(WgArgsC AC) {
    __shared__ __attribute__((aligned(16))) float lds[kWgRowChunk * XP + kWgRowChunk * 12 + kWgRG * kWgCols * kRedP];
    float* xs = lds;                          // [chunk][XP]   inputs (layer-1 job)
    float* rinfo = lds + kWgRowChunk * XP;    // [chunk][<=12] per-row scalars
    float* red = rinfo + kWgRowChunk * 12;    // [16][64][kRedP]  cross-row-group reduction

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
    if (ADAM && A.ad.finish_actor && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {  // what adam_kernel's thread 0 does on an actor step
        if (A.ad.use_bc) {
            A.ad.losses[1] = A.ad.losses[2] * w + A.ad.losses[3] * (1.0f - w);  // HIRL.py:321
            A.ad.losses[5] = w;
            *A.ad.wstate = w;
        } else {
            A.ad.losses[1] = A.ad.losses[3];  // TD3.py:236
        }
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
                    const __bf16 bv = (__bf16)ae[q].p;
                    J.w2b[w2_image_index((uint32_t)(n0 + 4 * g + q0 + q), (uint32_t)(k0 + r))] = __builtin_bit_cast(uint16_t, bv);
                }
                if (J.w2f) J.w2f[w2f_image_index((uint32_t)(n0 + 4 * g + q0 + q), (uint32_t)(k0 + r))] = ae[q].p;
            }
        }
        STAMP();
        STAMP_FLUSH(32, blockIdx.x == 0 && blockIdx.y == 0 && tid == 0);
        SPAN_LOG();
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
        SPAN_LOG();
        return;
    }
    // layer 1: hidden unit k; dz1 = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat)) with the row means taken from the
    // per-workgroup partial sums bwd_l2 left in lnp -> no cross-column work here.
    {
        const int kb = (b - kWgTilesPerBlock - kWgVecWgs) * kWgCols;
        const int k = kb + cl;
        const int in = J.m.in;
        const float g1 = J.net[J.m.g1() + k], be1 = J.net[J.m.be1() + k];
        STAMP();
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
                STAMP();
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
                STAMP();
                const int er = tid < nr ? tid : 0;
                v2f st1v;  // (unset in a skipping wave, which never stores it)
                v4f l0, l1, l2, l3;
                if (w0 < nr) {
                    st1v = *reinterpret_cast<const v2f*>(S.st1 + (size_t)(c0 + er) * 2);
                    const v4f* lp4 = reinterpret_cast<const v4f*>(S.lnp + (size_t)(c0 + er) * (2 * kColWgB));  // [8 column workgroups][2]
                    l0 = lp4[0]; l1 = lp4[1]; l2 = lp4[2]; l3 = lp4[3];
                }
                STAMP();
                if (lae_pending) {
                    lae.fetch(J, lidx);
                    lae_pending = false;
                }
                STAMP();
                __syncthreads();
                STAMP();
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
                STAMP();
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
        STAMP();
        float* my = red + (rg * kWgCols + cl) * kRedP;
        my[0] = db1; my[1] = dg; my[2] = dbe;
#pragma unroll
        for (int i = 0; i < 17; ++i) my[3 + i] = dw1[i];
        STAMP();
        __syncthreads();
        STAMP();
        if (llive) {
            float v = sum_groups(red + ocol * kRedP + oitem, ohalf);
            if ((oitem == 1 || oitem == 2) && J.m.no_ln) v = 0.0f;
#ifdef HX_STAMPS
            asm volatile("" ::"v"(v));
            STAMP();
#endif
            if (ohalf == 0) {
                J.grad[lidx] = v;
                if (ADAM) lae.apply(J, A.ad, lidx, v);
            }
        }
        STAMP();
        STAMP_FLUSH(48, b == kWgTilesPerBlock + kWgVecWgs && j == 0 && tid == 0);
        SPAN_LOG();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam defaults restated) and Polyak, elementwise, 16 B per lane
// ---------------------------------------------------------------------------------------------------------------
struct AdamArgs {
    float* p; const float* g; float* m; float* v;
    int n;
    float b1, b2, eps, step_size, bc2_sqrt, gscale;
    // bookkeeping done by thread 0 of block 0 on actor steps: actor_loss and the stored BC weight
    int finish_actor;
    int w_kind; float w_given, warm, inv_batch;
    const int* soft_count; float* wstate; float* losses; int use_bc;
    // SAC policy step: also the log-alpha step (SAC/agent.py:322-325, 408-414)
    float* alpha_state;  // [4]: log_alpha, m, v, alpha; nullptr = not a SAC policy step
    float target_entropy, alpha_step_size;
    // soft_update of this network's target in the same pass (HIRL.py:11-13,327-330): nothing reads the targets between this
    // Adam step and the end of learn(), so target <- (1 - tau) target + tau p_new here equals the reference's separate pass
    float* target;       // nullptr = no Polyak this call
    float tau;
    // bf16 image of this block's W2 kept current by the step that changes it (elements [w2_lo, w2_lo + 512*256) of p): the BF16
    // acting kernels read it; nullptr = none
    uint16_t* w2b;
    float* w2f;  // fp32 image of W2 to refresh (same range)
    int w2_lo;
    // merged actor message of a sharded run: g holds the summed dL_rl, g2 the summed dL_bc, *countf the summed
    // soft count; the step uses g = w g2 + (1 - w) g with w from the GLOBAL count.  nullptr: g is the finished gradient.
    const float* g2;
    const float* countf;
};

// the BC weight of this call (HIRL.py:299-308); countf: the soft count as a float word of the all-reduced message
__device__ __forceinline__ float adam_w(const AdamArgs& A) {
    if (A.countf && A.w_kind == 1) {
        const float w = *A.countf * A.inv_batch + A.warm;
        return w > 1.0f ? 1.0f : w;
    }
    return effective_w(A.w_kind, A.w_given, A.warm, A.inv_batch, A.soft_count, A.wstate);
}

__global__ __launch_bounds__(kThreads) void adam_kernel(AdamArgs A) {
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
            *reinterpret_cast<uint2*>(A.w2b + w2_image_index(e / H1, e % H1)) = __builtin_bit_cast(uint2, r);
        }
        if (A.w2f && i >= A.w2_lo && i < A.w2_lo + H2 * H1) {
            const uint32_t e = (uint32_t)(i - A.w2_lo);
            *reinterpret_cast<float4*>(A.w2f + w2f_image_index(e / H1, e % H1)) = p4;
        }
        if (A.target) {
            float4 t4 = *reinterpret_cast<const float4*>(A.target + i);
            t4.x = polyak_update(t4.x, p4.x, A.tau);
            t4.y = polyak_update(t4.y, p4.y, A.tau);
            t4.z = polyak_update(t4.z, p4.z, A.tau);
            t4.w = polyak_update(t4.w, p4.w, A.tau);
            *reinterpret_cast<float4*>(A.target + i) = t4;
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
__global__ __launch_bounds__(kThreads) void polyak_kernel(float* target, const float* source, int n, float tau, float* target2 = nullptr,
                                                          const float* source2 = nullptr, int n2 = 0) {
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


// ---------------------------------------------------------------------------------------------------------------
// minibatch sampling on the device: UniformMemory.sample (buffer.py:45 random.sample, without replacement),
// np.random.choice(N_exp, B, replace=False) (HIRL.py:249) and the (4,) target-smoothing noise (HIRL.py:265).
// One workgroup; Philox4x32-10 keyed by `seed`, counter (row, call, stream, round).  Duplicates inside a group are
// redrawn until none is left (normally one round: B << len), at most 128 rounds; a duplicate surviving that — only
// possible when a group asks for nearly the whole table — is accepted (the with-replacement draw of SURVEY.md quirk 13).
// ---------------------------------------------------------------------------------------------------------------
struct SampleArgs {
    const unsigned long long* total;  // transitions ever stored in the main ring
    long long cap, expert_len, bc_len;
    int batch, n_main;
    uint64_t seed;
    uint32_t call;
    float sigma;
    int* idx;
    int* idx_bc;
    float* noise;
    // gather: the sampled rows are copied ONCE into compact [batch][32] tiles that every update kernel then reads with
    // plain row addressing (no index indirection, no page-scattered loads on their critical paths)
    const float* ring;
    const float* expert_ring;
    const float* bc_table;
    float* rows;
    float* bc_rows;
    int do_sample;  // 0: idx / idx_bc are inputs (parity tests, the N = 1 facade), only gather
};

// 1024 threads.  With B <= 512 the two index streams (replay / expert rows, BC rows) are drawn side by side by the two halves
// of the workgroup.  "Without replacement" = inside a group, of several rows that drew the same index the lowest row keeps it
// and the others redraw.  The check is a hash set in LDS (key = group | index, owner = lowest row that drew it: atomicCAS +
// atomicMin, linear probing, load <= 0.3), O(1) per row and round instead of an all-pairs scan.  The gather reads the indices
// from LDS.
constexpr int kSampleSlots = 4096;
__device__ __forceinline__ uint32_t sample_hash(uint32_t k) { return (k * 2654435761u) >> 20; }  // top 12 bits

__global__ __launch_bounds__(1024) void sample_kernel(SampleArgs A) {
    __shared__ uint32_t hkey[2][kSampleSlots];
    __shared__ int hown[2][kSampleSlots];
    __shared__ int fin[2][1024];
    const int tid = threadIdx.x;
    const int B = A.batch;
    const int np = B <= 512 ? 2 : 1;            // streams drawn in parallel
    const int width = 1024 / np;                // threads per stream
    const int t = tid % width, tab = tid / width;
    if (A.do_sample) {
        const unsigned long long tot = *A.total;
        const uint32_t len_main = (uint32_t)(tot < (unsigned long long)A.cap ? tot : (unsigned long long)A.cap);
        const uint32_t k0 = (uint32_t)A.seed, k1 = (uint32_t)(A.seed >> 32);
        for (int pass = 0; pass < 2 / np; ++pass) {
            const int stream = np == 2 ? tab : pass;  // 0: replay / expert rows, 1: BC rows
            int* out = stream == 1 ? A.idx_bc : A.idx;
            const bool live = out != nullptr && t < B;
            const bool main_grp = t < A.n_main;
            const uint32_t len = stream == 1 ? (uint32_t)A.bc_len : (main_grp ? len_main : (uint32_t)A.expert_len);
            const uint32_t grp = (stream == 1 || main_grp) ? 0u : 0x80000000u;  // groups: [0, n_main) and [n_main, batch)
            uint32_t* keys = hkey[tab];
            int* owns = hown[tab];
            for (int e = t; e < kSampleSlots; e += width) {
                keys[e] = 0xFFFFFFFFu;
                owns[e] = 0x7FFFFFFF;
            }
            __syncthreads();
            uint32_t key = 0;
            bool dup = live;
            for (int round = 0; round < 128; ++round) {
                if (dup) {
                    uint32_t u[4];
                    philox4x32_10((uint32_t)t, A.call, (uint32_t)stream, (uint32_t)round, k0, k1, u);
                    key = grp | (len ? __umulhi(u[0], len) : 0u);
                    uint32_t h = sample_hash(key);
                    for (int probe = 0; probe < kSampleSlots; ++probe) {  // bounded: the set never fills (<= B + redraws keys)
                        const uint32_t k = atomicCAS(&keys[h], 0xFFFFFFFFu, key);
                        if (k == 0xFFFFFFFFu || k == key) break;
                        h = (h + 1) & (kSampleSlots - 1);
                    }
                    atomicMin(&owns[h], t);
                }
                __syncthreads();
                if (live) {  // every row looks its index up again: a redraw of a lower row may have taken it over
                    uint32_t h = sample_hash(key);
                    for (int probe = 0; probe < kSampleSlots && keys[h] != key; ++probe) h = (h + 1) & (kSampleSlots - 1);
                    dup = owns[h] != t;
                }
                if (!__syncthreads_or(dup)) break;  // nobody redraws: done (the common case after the first round)
            }
            if (live) {
                const int v = (int)(key & 0x7FFFFFFFu);
                out[t] = v;
                fin[stream][t] = v;
            }
            __syncthreads();
        }
        if (tid < 4 && A.noise) {
            uint32_t u[4];
            philox4x32_10(0xFFFFFFF0u, A.call, 2u, 0u, k0, k1, u);
            const float ua = u01(u[tid & 2]), ub = u01(u[(tid & 2) + 1]);
            const float rad = sqrtf(-2.0f * __logf(ua)), ang = 6.28318530717958647692f * ub;
            A.noise[tid] = A.sigma * ((tid & 1) ? rad * __sinf(ang) : rad * __cosf(ang));
        }
    } else {
        for (int e = tid; e < B; e += 1024) {
            fin[0][e] = A.idx[e];
            if (A.idx_bc) fin[1][e] = A.idx_bc[e];
        }
        __syncthreads();
    }
    // gather: 8 lanes per row, one 16-B piece each; both tiles' loads are in flight together
    const bool bc = A.bc_rows && A.bc_table && A.idx_bc;
    for (int e = tid; e < B * 8; e += 1024) {
        const int r = e >> 3, c = e & 7;
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f), q = m;
        if (A.rows) m = reinterpret_cast<const float4*>((r < A.n_main ? A.ring : A.expert_ring) + (size_t)fin[0][r] * 32)[c];
        if (bc) q = reinterpret_cast<const float4*>(A.bc_table + (size_t)fin[1][r] * 32)[c];
        if (A.rows) reinterpret_cast<float4*>(A.rows)[e] = m;
        if (bc) reinterpret_cast<float4*>(A.bc_rows)[e] = q;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
constexpr size_t kSlotFloats = XP + H1 + 2 + H1 + H2 + 2 + OW + H2 + H1 + OW + 2 * kColWgB;  // per row

enum { S_TA = 0, S_C1, S_C2, S_TC1, S_TC2, S_API, S_ABC, S_BCS, S_CPI, S_CSOFT, S_COUNT };

int fwd_row_tiles(const FwdArgs& a) {
    int n = 0;
    for (int j = 0; j < a.njobs; ++j) n += (a.job[j].rows + RT - 1) / RT;
    return n;
}
// column tiling by size: enough row tiles to fill the chip -> wide workgroups (less prologue recomputation)
void launch_fwd(const FwdArgs& F, hipStream_t st) {
    FwdArgsC C{};
    for (int j = 0; j < F.njobs; ++j) {  // every job of a launch has the same row count (the minibatch)
        C.job[j] = pack_fwd(F.job[j]);
        C.job[j].slope = F.slope;
    }
    C.slope = F.slope; C.zero_nf = F.zero_nf; C.zero_f = F.zero_f; C.zero_i = F.zero_i;
    const int tiles = fwd_row_tiles(F), per_job = tiles / F.njobs;
    const bool relu = F.slope == 0.0f;  // compile-time ReLU instantiations (hx_nn.h act_f)
#define HX_FWD(NT_) do { const dim3 grid(per_job * (H2 / NT_), F.njobs); \
        if (relu) hipLaunchKernelGGL((fwd_l2_kernel<NT_, true, false>), grid, dim3(kWide), 0, st, C, NoSample{}); \
        else hipLaunchKernelGGL((fwd_l2_kernel<NT_, false, false>), grid, dim3(kWide), 0, st, C, NoSample{}); } while (0)
    if (F.sample) {  // (the callers checked: three or four jobs of at most 256 rows -> the 64-column tiling)
        const dim3 grid(per_job * (H2 / kNT), F.njobs);
        if (relu) hipLaunchKernelGGL((fwd_l2_kernel<kNT, true, true>), grid, dim3(kWide), 0, st, C, *F.sample);
        else hipLaunchKernelGGL((fwd_l2_kernel<kNT, false, true>), grid, dim3(kWide), 0, st, C, *F.sample);
        return;
    }
    if (tiles >= 128) HX_FWD(256);
    else if (tiles * (H2 / 32) <= 256) HX_FWD(32);  // one or two nets at B = 128: 32-column workgroups still fit the chip in one round
    else HX_FWD(kNT);
#undef HX_FWD
}
int bwd_blocks(const BwdArgs& a, int rows_per_wg) {  // per job (every job of a launch has the same row count)
    return ((a.job[0].rows + rows_per_wg - 1) / rows_per_wg) * kColWgB;
}
template <bool ADAM>
void launch_wg(const WgArgs& W, hipStream_t st) {
    WgArgsC C{};
    for (int j = 0; j < W.njobs; ++j) C.job[j] = pack_wg(W.job[j], W);
    if (W.slope == 0.0f) hipLaunchKernelGGL((wgrad_kernel<ADAM, true>), dim3(kWgPerJob, W.njobs), dim3(kWide), 0, st, C);
    else hipLaunchKernelGGL((wgrad_kernel<ADAM, false>), dim3(kWgPerJob, W.njobs), dim3(kWide), 0, st, C);
}
template <int GRP>
void launch_bwd(const BwdArgs& G, hipStream_t st) {
    BwdArgsC C{};
    for (int j = 0; j < G.njobs; ++j) C.job[j] = pack_bwd(G.job[j], G);
    if (G.slope == 0.0f) hipLaunchKernelGGL((bwd_l2_kernel<GRP, true>), dim3(bwd_blocks(G, GRP <= 2 ? RT / 2 : RT), G.njobs), dim3(kWide), 0, st, C);
    else hipLaunchKernelGGL((bwd_l2_kernel<GRP, false>), dim3(bwd_blocks(G, GRP <= 2 ? RT / 2 : RT), G.njobs), dim3(kWide), 0, st, C);
}

const Mlp kActor{13, 4, 0};
const Mlp kQ{17, 1, 0};
const Mlp kPolicy{13, 8, 1};  // SAC GaussianPolicy: Linear-ReLU stack, head = mean ++ log_std (SAC/model.py:58-60)
const Mlp kQs{17, 1, 1};      // SAC Q head (SAC/model.py:21-23)

}  // namespace

extern "C" {

/* diagnostic builds only (make STAMPS=1): copy the 64 phase-stamp floats to host memory; returns -1 otherwise */
int hx_debug_stamps(float* host_out) {
#ifdef HX_STAMPS
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(hx_dbg), 80 * sizeof(float)) == hipSuccess ? 0 : -2;
#else
    (void)host_out;
    return -1;
#endif
}
/* diagnostic builds only: the workgroup life-span log (start, end in 10 ns ticks; tag = source line of the exit) -> host, then cleared */
int hx_debug_spans(unsigned long long* host_spans /* [8192][2] */, unsigned* host_tags /* [8192] */, unsigned* host_n) {
#ifdef HX_STAMPS
    unsigned zero = 0u;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (hipMemcpyFromSymbol(host_n, HIP_SYMBOL(hx_span_n), sizeof(unsigned)) != hipSuccess) return -2;
    if (hipMemcpyFromSymbol(host_spans, HIP_SYMBOL(hx_span), sizeof(unsigned long long) * 2 * kSpanCap) != hipSuccess) return -2;
    if (hipMemcpyFromSymbol(host_tags, HIP_SYMBOL(hx_span_tag), sizeof(unsigned) * kSpanCap) != hipSuccess) return -2;
    return hipMemcpyToSymbol(HIP_SYMBOL(hx_span_n), &zero, sizeof(unsigned)) == hipSuccess ? 0 : -2;
#else
    (void)host_spans; (void)host_tags; (void)host_n;
    return -1;
#endif
}
int hx_actor_param_count(void) { return kActor.size(); }
int hx_critic_param_count(void) { return 2 * kQ.padded(); }
int64_t hx_hirl_workspace_floats(int32_t batch) { return (int64_t)S_COUNT * kSlotFloats * batch + 64; }
int64_t hx_act_workspace_floats(int64_t rows) { (void)rows; return 0; }  // the acting kernels keep z2 in LDS: no workspace any more

/* chooseAction / chooseActionSmallNoise / chooseActionNoNoise for `rows` observations (HIRL.py:192-212):
 * actions = clamp(actor(obs) + noise, -1, 1).  noise_mode 0: none, 1: noise[4] shared by all rows, 2: noise[rows][4],
 * 3: N(0, sigma^2) per row and component from Philox(seed; row0 + row, call).  ws: unused since the whole policy runs in one kernel (may be NULL). */
static int actor_act_impl(const float* actor, const uint16_t* w2b, const float* w2f, const float* obs, int64_t rows, float* actions, int32_t noise_mode, const float* noise,
                 float sigma, uint64_t seed, uint32_t row0, uint32_t call, float slope, void* stream) {
    HX_REQUIRE(actor && obs && actions && rows > 0, "hx_actor_act: bad arguments");
    HX_REQUIRE(noise_mode >= 0 && noise_mode <= 3 && (noise || (noise_mode != 1 && noise_mode != 2)), "hx_actor_act: bad noise mode");
    ActFusedArgs H{actor, kActor, const_cast<float*>(obs), (int)rows, slope, actions, (noise_mode == 1 || noise_mode == 2) ? noise : nullptr,
                   noise_mode == 2, noise_mode == 3 ? sigma : 0.0f, 0, seed, row0, call, nullptr, 0, nullptr, nullptr, nullptr, HxStepOpts{}, 0.0, w2b, w2f};
    launch_act<false>(H, (hipStream_t)stream);
    HX_CHECK_LAUNCH("hx_actor_act");
    return 0;
}

int hx_actor_act(const float* actor, const float* obs, int64_t rows, float* actions, int32_t noise_mode, const float* noise,
                 float sigma, uint64_t seed, uint32_t row0, uint32_t call, float slope, float* ws, void* stream) {
    (void)ws;
    return actor_act_impl(actor, nullptr, nullptr, obs, rows, actions, noise_mode, noise, sigma, seed, row0, call, slope, stream);
}
/* The same with the 256 -> 512 layer on bf16 MFMA (BASELINE.json configs[4]): w2_bf16 = hx_pack_w2_bf16 image of full2.weight. */
int hx_actor_act_bf16(const float* actor, const uint16_t* w2_bf16, const float* obs, int64_t rows, float* actions, int32_t noise_mode,
                      const float* noise, float sigma, uint64_t seed, uint32_t row0, uint32_t call, float slope, void* stream) {
    HX_REQUIRE(w2_bf16 && (reinterpret_cast<uintptr_t>(w2_bf16) & 15u) == 0, "hx_actor_act_bf16: w2_bf16 must be a 16-byte aligned bf16 image of W2");
    return actor_act_impl(actor, w2_bf16, nullptr, obs, rows, actions, noise_mode, noise, sigma, seed, row0, call, slope, stream);
}
/* bf16 image (round to nearest even) of an MLP block's W2 [512][256]; in_dim = 13 (actor / policy) or 17 (Q head) locates it. */
int hx_pack_w2_bf16(const float* net, int32_t in_dim, uint16_t* w2_bf16, void* stream) {
    HX_REQUIRE(net && w2_bf16 && (in_dim == 13 || in_dim == 17), "hx_pack_w2_bf16: bad arguments");
    const Mlp m{in_dim, 1, 0};
    const int n = H2 * H1;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3((n / 2 + kThreads - 1) / kThreads), dim3(kThreads), 0, (hipStream_t)stream, net + m.W2(), w2_bf16, n);
    HX_CHECK_LAUNCH("hx_pack_w2_bf16");
    return 0;
}

static int check_step_args(const float* state, int64_t n, int64_t stride, const float* obs_io, const float* actions, const float* reward,
                           const uint8_t* done, const int8_t* success, const HxStepOpts& o, const char* who) {
    HX_REQUIRE(state && obs_io && actions && reward && done && success && n > 0 && stride >= n, "%s: bad buffers", who);
    HX_REQUIRE(n < (int64_t)1 << 31, "%s: at most 2^31 - 1 envs per launch", who);
    HX_REQUIRE(!o.auto_reset || o.episode_ctr, "%s: auto_reset needs episode_ctr", who);
    HX_REQUIRE(stride < ((int64_t)1 << 25), "%s: stride must be below 2^25 envs", who);
    HX_REQUIRE(!o.ring || (o.cap >= 512 && o.cap < ((int64_t)1 << 31) && o.total && (reinterpret_cast<uintptr_t>(o.ring) & 15u) == 0),
               "%s: ring needs 512 <= cap < 2^31, total and 16-byte alignment", who);
    return 0;
}

/* chooseAction + HarfangEnv.step for n envs in ONE launch (train_all.py:343-345): actions = clamp(actor(obs_io) + noise, -1, 1) as
 * hx_actor_act, then hx_env_step with those actions in the tail of the same kernel — obs_io in: current observation, out: next. */
static int actor_act_step_impl(const float* actor, const uint16_t* w2b, const float* w2f, float* state, int64_t n, int64_t stride, float* obs_io, float* actions, int32_t noise_mode,
                      const float* noise, float sigma, uint64_t seed, uint32_t row0, uint32_t call, float slope, float* reward,
                      uint8_t* done, int8_t* success, const HxStepOpts* opts, void* stream) {
    HX_REQUIRE(actor, "hx_actor_act_step: null actor");
    HX_REQUIRE(noise_mode >= 0 && noise_mode <= 3 && (noise || (noise_mode != 1 && noise_mode != 2)), "hx_actor_act_step: bad noise mode");
    const HxStepOpts o = opts ? *opts : HxStepOpts{};
    if (int rc = check_step_args(state, n, stride, obs_io, actions, reward, done, success, o, "hx_actor_act_step")) return rc;
    if (n > kFuseEnvMax) {  // more than one round of workgroups: the env step is cheaper as a launch of its own
        if (int rc = actor_act_impl(actor, w2b, w2f, obs_io, n, actions, noise_mode, noise, sigma, seed, row0, call, slope, stream)) return rc;
        return hx_env_step(state, n, stride, actions, obs_io, reward, done, success, opts, stream);
    }
    ActFusedArgs H{actor, kActor, obs_io, (int)n, slope, actions, (noise_mode == 1 || noise_mode == 2) ? noise : nullptr,
                   noise_mode == 2, noise_mode == 3 ? sigma : 0.0f, 0, seed, row0, call, state, stride, reward, done, success, o,
                   o.cap > 0 ? 1.0 / (double)o.cap : 0.0, w2b, w2f};
    launch_act<false>(H, (hipStream_t)stream);
    HX_CHECK_LAUNCH("hx_actor_act_step");
    return 0;
}

int hx_actor_act_step(const float* actor, float* state, int64_t n, int64_t stride, float* obs_io, float* actions, int32_t noise_mode,
                      const float* noise, float sigma, uint64_t seed, uint32_t row0, uint32_t call, float slope, float* reward,
                      uint8_t* done, int8_t* success, const HxStepOpts* opts, void* stream) {
    return actor_act_step_impl(actor, nullptr, nullptr, state, n, stride, obs_io, actions, noise_mode, noise, sigma, seed, row0, call, slope, reward, done,
                               success, opts, stream);
}
int hx_actor_act_step_bf16(const float* actor, const uint16_t* w2_bf16, float* state, int64_t n, int64_t stride, float* obs_io, float* actions,
                           int32_t noise_mode, const float* noise, float sigma, uint64_t seed, uint32_t row0, uint32_t call, float slope,
                           float* reward, uint8_t* done, int8_t* success, const HxStepOpts* opts, void* stream) {
    HX_REQUIRE(w2_bf16 && (reinterpret_cast<uintptr_t>(w2_bf16) & 15u) == 0, "hx_actor_act_step_bf16: w2_bf16 must be a 16-byte aligned bf16 image of W2");
    return actor_act_step_impl(actor, w2_bf16, nullptr, state, n, stride, obs_io, actions, noise_mode, noise, sigma, seed, row0, call, slope, reward, done,
                               success, opts, stream);
}

/* The fp32 policy from the re-ordered fp32 image of W2 (hx_pack_w2_f32i): bit-identical to hx_actor_act / hx_actor_act_step. */
int hx_pack_w2_f32i(const float* net, int32_t in_dim, float* w2_f32i, void* stream) {
    HX_REQUIRE(net && w2_f32i && (in_dim == 13 || in_dim == 17) && (reinterpret_cast<uintptr_t>(w2_f32i) & 15u) == 0, "hx_pack_w2_f32i: bad arguments");
    const Mlp m{in_dim, 1, 0};
    const int n = H2 * H1;
    hipLaunchKernelGGL(pack_f32i_kernel, dim3((n / 4 + kThreads - 1) / kThreads), dim3(kThreads), 0, (hipStream_t)stream, net + m.W2(), w2_f32i, n);
    HX_CHECK_LAUNCH("hx_pack_w2_f32i");
    return 0;
}
int hx_actor_act_f32i(const float* actor, const float* w2_f32i, const float* obs, int64_t rows, float* actions, int32_t noise_mode,
                      const float* noise, float sigma, uint64_t seed, uint32_t row0, uint32_t call, float slope, void* stream) {
    HX_REQUIRE(w2_f32i && (reinterpret_cast<uintptr_t>(w2_f32i) & 15u) == 0, "hx_actor_act_f32i: w2_f32i must be a 16-byte aligned fp32 image of W2");
    return actor_act_impl(actor, nullptr, w2_f32i, obs, rows, actions, noise_mode, noise, sigma, seed, row0, call, slope, stream);
}
int hx_actor_act_step_f32i(const float* actor, const float* w2_f32i, float* state, int64_t n, int64_t stride, float* obs_io, float* actions,
                           int32_t noise_mode, const float* noise, float sigma, uint64_t seed, uint32_t row0, uint32_t call, float slope,
                           float* reward, uint8_t* done, int8_t* success, const HxStepOpts* opts, void* stream) {
    HX_REQUIRE(w2_f32i && (reinterpret_cast<uintptr_t>(w2_f32i) & 15u) == 0, "hx_actor_act_step_f32i: w2_f32i must be a 16-byte aligned fp32 image of W2");
    return actor_act_step_impl(actor, nullptr, w2_f32i, state, n, stride, obs_io, actions, noise_mode, noise, sigma, seed, row0, call, slope, reward, done,
                               success, opts, stream);
}

static void make_slots(const HxNets* N, int B, Slot* s) {
    for (int i = 0; i < S_COUNT; ++i) s[i] = carve_slot(N->ws + (size_t)i * kSlotFloats * B, B);
}

/* Stage 1 (every call): TD target, critic forward, critic gradients -> grad_critic, losses[0].  HIRL.py:259-286.
 * actor_fwd: 1 = also run the delayed actor step's forward passes that do not depend on the critic update (actor(s),
 * actor(s_bc)), 2 = plus bc_actor(s) for the soft estimate — they ride in launch A instead of a launch of their own.
 */
// torch.optim.Adam's per-step scalars (defaults: betas (0.9, 0.999), eps 1e-8) for the fused wgrad + Adam launch
static WgAdam make_adam(const HxNets* N, const HxHyper* Hy, float lr, int step, bool finish_actor) {
    const double b1 = 0.9, b2 = 0.999;
    const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
    WgAdam a{};
    a.b1 = (float)b1; a.b2 = (float)b2; a.eps = 1e-8f;
    a.step_size = (float)(lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.tau = Hy->tau;
    a.finish_actor = finish_actor ? 1 : 0;
    a.use_bc = Hy->use_bc;
    a.losses = N->losses;
    a.wstate = N->wstate;
    return a;
}

// adam_step > 0: the critic's optimizer step (and, with polyak, the soft_update of its target) rides in the wgrad launch
// An HxSample for a launch-A draw: validated, then either the device-side description (batch <= 256: *fused = true, launch A draws and
// gathers) or the sampling launch right here (larger batches).  rows / bc_rows / noise: the tiles the later launches read.
static int prepare_draw(const HxSample* S, int B, float* rows, float* bc_rows, float* noise, void* stream, SampleDev* SD, bool* fused) {
    HX_REQUIRE(S->total && S->cap > 0 && S->ring && S->idx && rows && S->n_main >= 0 && S->n_main <= B,
               "hx_*_sampled: the draw needs total, cap, ring, idx and the output tile rows");
    HX_REQUIRE(S->n_main == B || S->expert_ring, "hx_*_sampled: expert rows requested without an expert ring");
    HX_REQUIRE(!S->bc_table == !S->idx_bc && (!S->bc_table || bc_rows), "hx_*_sampled: bc_table, idx_bc and bc_rows go together");
    *fused = B <= kFusedBatchMax;
    if (*fused) {
        *SD = SampleDev{(const unsigned long long*)S->total, S->ring, S->expert_ring ? S->expert_ring : S->ring, S->bc_table, rows,
                        S->bc_table ? bc_rows : nullptr, noise, S->idx, S->idx_bc, (long long)S->cap, (uint32_t)S->expert_len, (uint32_t)S->bc_len,
                        S->n_main, S->call, S->seed, S->sigma};
        return 0;
    }
    return hx_sample_batch(S->total, S->cap, S->ring, S->expert_ring, S->expert_len, S->bc_table, S->bc_len, B, S->n_main, 1, S->seed, S->call,
                           S->sigma, S->idx, S->idx_bc, noise, rows, S->bc_table ? bc_rows : nullptr, stream);
}

static int critic_grads_impl(const HxNets* N, const HxBatch* Bt, const HxHyper* Hy, int32_t actor_fwd, void* stream, int adam_step, bool polyak,
                             const HxSample* S = nullptr) {
    HX_REQUIRE(N && Bt && Hy && Bt->batch > 0 && Bt->batch % 16 == 0, "hx_hirl_critic_grads: batch must be a positive multiple of 16");
    hipStream_t st = (hipStream_t)stream;
    const int B = Bt->batch;
    SampleDev SD{};
    bool fused = false;
    if (S) {  // the minibatch is drawn by this call: inside launch A (batch <= 256) or by the sampling launch first
        HX_REQUIRE(Bt->noise, "hx_hirl_*_sampled: the draw needs the output word noise[4]");
        if (int rc = prepare_draw(S, B, const_cast<float*>(Bt->rows), const_cast<float*>(Bt->bc_rows), const_cast<float*>(Bt->noise), stream, &SD, &fused)) return rc;
    }
    Slot s[S_COUNT];
    make_slots(N, B, s);
    const RowSrc src{Bt->rows, nullptr, nullptr, 0, 32};
    const float* tc1 = N->target_critic;
    const float* tc2 = N->target_critic + kQ.padded();
    {   // launch A: targetActor(s'), critic Q1/Q2 (s, a)
        FwdArgs F{};
        F.njobs = 3; F.slope = Hy->slope;
        F.zero_f = N->losses; F.zero_nf = 1;  // critic_loss accumulator
        F.job[0] = FwdJob{N->target_actor, kActor, src, 17, 0, Head{}, nullptr, 0.f, s[S_TA], B, 0};
        F.job[1] = FwdJob{N->critic, kQ, src, 0, 0, Head{}, nullptr, 0.f, s[S_C1], B, 1};
        F.job[2] = FwdJob{N->critic + kQ.padded(), kQ, src, 0, 0, Head{}, nullptr, 0.f, s[S_C2], B, 1};
        if (actor_fwd) {  // the delayed actor step's forwards ride along, split so that NEITHER launch exceeds 256 workgroups
            F.zero_nf = 5; F.zero_i = N->soft_count;  // + actor / bc / rl / bc_fire accumulators and the soft count
            F.job[3] = FwdJob{N->actor, kActor, src, 0, 0, Head{}, nullptr, 0.f, s[S_API], B, 1};
            F.njobs = 4;
        }
        F.sample = fused ? &SD : nullptr;
        launch_fwd(F, st);
    }
    {   // launch B: targetCritic Q1/Q2 (s', clamp(targetActor(s') + clamp(noise)))  [+ actor(s_bc), bc_actor(s)]
        FwdArgs F{};
        F.njobs = 2; F.slope = Hy->slope;
        const Head prev{N->target_actor, kActor, s[S_TA]};
        F.job[0] = FwdJob{tc1, kQ, src, 17, 1, prev, Bt->noise, Hy->noise_clamp, s[S_TC1], B, 0};
        F.job[1] = FwdJob{tc2, kQ, src, 17, 1, prev, Bt->noise, Hy->noise_clamp, s[S_TC2], B, 0};
        if (actor_fwd && Hy->use_bc) {
            const RowSrc bcsrc{Bt->bc_rows ? Bt->bc_rows : Bt->rows, nullptr, nullptr, 0, 32};
            int n = 2;
            F.job[n++] = FwdJob{N->actor, kActor, bcsrc, 0, 0, Head{}, nullptr, 0.f, s[S_ABC], B, 1};
            if (actor_fwd == 2) F.job[n++] = FwdJob{N->bc_actor, kActor, src, 0, 0, Head{}, nullptr, 0.f, s[S_BCS], B, 0};
            F.njobs = n;
        }
        launch_fwd(F, st);
    }
    {   // launch C: y, loss, dq, LN2 backward, dh1 for both heads
        BwdArgs G{};
        G.njobs = 2; G.slope = Hy->slope; G.inv_batch = 1.0f / B; G.losses = N->losses; G.soft_count = N->soft_count;
        for (int h = 0; h < 2; ++h) {
            BwdJob& J = G.job[h];
            J = BwdJob{};
            J.net = N->critic + h * kQ.padded(); J.m = kQ; J.ws = s[S_C1 + h]; J.rows = B; J.mode = BM_CRITIC_TD;
            J.t1 = Head{tc1, kQ, s[S_TC1]}; J.t2 = Head{tc2, kQ, s[S_TC2]}; J.src = src; J.gamma = Hy->gamma;
        }
        launch_bwd<0>(G, st);
    }
    {   // launch D: all critic parameter gradients
        WgArgs W{};
        W.njobs = 2; W.slope = Hy->slope; W.w_kind = 0; W.w_given = 0.f; W.inv_batch = 1.0f / B;
        W.soft_count = N->soft_count; W.wstate = N->wstate;
        for (int h = 0; h < 2; ++h) {
            WgJob& J = W.job[h];
            J = WgJob{};
            J.net = N->critic + h * kQ.padded(); J.grad = N->grad_critic + h * kQ.padded(); J.m = kQ;
            J.ws[0] = s[S_C1 + h]; J.rows[0] = B; J.nslots = 1; J.wmode[0] = 0;
            if (adam_step > 0) {
                J.p = N->critic + h * kQ.padded();
                J.mom = N->m_critic + h * kQ.padded(); J.var = N->v_critic + h * kQ.padded();
                J.target = polyak ? N->target_critic + h * kQ.padded() : nullptr;
            }
        }
        if (adam_step > 0) {
            W.ad = make_adam(N, Hy, Hy->lr_critic, adam_step, false);
            launch_wg<true>(W, st);
        } else {
            launch_wg<false>(W, st);
        }
    }
    HX_CHECK_LAUNCH("hx_hirl_critic_grads");
    return 0;
}
int hx_hirl_critic_grads(const HxNets* N, const HxBatch* Bt, const HxHyper* Hy, int32_t actor_fwd, void* stream) {
    return critic_grads_impl(N, Bt, Hy, actor_fwd, stream, 0, false);
}
int hx_hirl_critic_grads_sampled(const HxNets* N, const HxBatch* Bt, const HxHyper* Hy, const HxSample* S, int32_t actor_fwd, void* stream) {
    HX_REQUIRE(S, "hx_hirl_critic_grads_sampled: null sample description");
    return critic_grads_impl(N, Bt, Hy, actor_fwd, stream, 0, false, S);
}


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
    if (which != 0 && N->actor_w2_bf16) {
        A.w2b = N->actor_w2_bf16;
        A.w2_lo = kActor.W2();
    }
    if (msg) {  // merged actor message: [dL_rl | dL_bc | count ...]
        HX_REQUIRE(which == 1 && (reinterpret_cast<uintptr_t>(msg) & 15u) == 0, "hx_adam_mixed: actor step only, 16-byte aligned message");
        A.g = msg;
        if (Hy->use_bc) {
            A.g2 = msg + kActor.padded();
            A.countf = msg + 2 * kActor.padded();
        }
    }
    HX_REQUIRE((((uintptr_t)A.p | (uintptr_t)A.g | (uintptr_t)A.m | (uintptr_t)A.v) & 15u) == 0, "hx_adam: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(adam_kernel, dim3((A.n / 4 + kThreads) / kThreads), dim3(kThreads), 0, (hipStream_t)stream, A);
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
int64_t hx_actor_message_floats(void) { return 2 * (int64_t)kActor.padded() + 64; }

/* Stage 2a (delayed actor step, HIRL.py:291-319): actor / bc_actor forward, Q1 with the UPDATED critic, the soft
 * count, backward down to dz2/dh1 of the actor.  Leaves soft_count and losses[2..4] ready; no parameter gradient yet. */
int hx_hirl_actor_backward(const HxNets* N, const HxBatch* Bt, const HxHyper* Hy, int32_t estimate_soft, int32_t fwd_done, void* stream) {
    HX_REQUIRE(N && Bt && Hy && Bt->batch > 0 && Bt->batch % 16 == 0, "hx_hirl_actor_backward: batch must be a positive multiple of 16");
    hipStream_t st = (hipStream_t)stream;
    const int B = Bt->batch;
    Slot s[S_COUNT];
    make_slots(N, B, s);
    const RowSrc src{Bt->rows, nullptr, nullptr, 0, 32};
    const RowSrc bcsrc{Bt->bc_rows ? Bt->bc_rows : Bt->rows, nullptr, nullptr, 0, 32};
    const bool bc = Hy->use_bc != 0, soft = bc && estimate_soft;
    if (!fwd_done) {   // launch F: actor(s), actor(s_bc), bc_actor(s)  (hx_hirl_critic_grads(actor_fwd) can carry them instead)
        FwdArgs F{};
        F.slope = Hy->slope;
        F.zero_f = N->losses + 1; F.zero_nf = 4; F.zero_i = N->soft_count;  // actor / bc / rl / bc_fire accumulators + soft count
        int n = 0;
        F.job[n++] = FwdJob{N->actor, kActor, src, 0, 0, Head{}, nullptr, 0.f, s[S_API], B, 1};
        if (bc) F.job[n++] = FwdJob{N->actor, kActor, bcsrc, 0, 0, Head{}, nullptr, 0.f, s[S_ABC], B, 1};
        if (soft) F.job[n++] = FwdJob{N->bc_actor, kActor, src, 0, 0, Head{}, nullptr, 0.f, s[S_BCS], B, 0};
        F.njobs = n;
        launch_fwd(F, st);
    }
    {   // launch G: Q1(s, pi(s)) and Q1(s, bc_actor(s)) with the updated critic
        FwdArgs F{};
        F.slope = Hy->slope;
        int n = 0;
        F.job[n++] = FwdJob{N->critic, kQ, src, 0, 1, Head{N->actor, kActor, s[S_API]}, nullptr, 0.f, s[S_CPI], B, 1};
        if (soft) F.job[n++] = FwdJob{N->critic, kQ, src, 0, 1, Head{N->bc_actor, kActor, s[S_BCS]}, nullptr, 0.f, s[S_CSOFT], B, 0};
        F.njobs = n;
        launch_fwd(F, st);
    }
    {   // launch H: rl_loss, soft count, critic backward down to dh1 (gradient wrt the action comes next)
        BwdArgs G{};
        G.njobs = 1; G.slope = Hy->slope; G.inv_batch = 1.0f / B; G.losses = N->losses; G.soft_count = N->soft_count;
        BwdJob& J = G.job[0];
        J = BwdJob{};
        J.net = N->critic; J.m = kQ; J.ws = s[S_CPI]; J.rows = B; J.mode = BM_CRITIC_PI;
        if (soft) J.soft = Head{N->critic, kQ, s[S_CSOFT]};
        launch_bwd<1>(G, st);
    }
    {   // launch I: actor backward for the RL batch (through tanh and the critic's input gradient) and the BC batch
        BwdArgs G{};
        G.slope = Hy->slope; G.inv_batch = 1.0f / B; G.losses = N->losses; G.soft_count = N->soft_count;
        int n = 0;
        {
            BwdJob& J = G.job[n++];
            J = BwdJob{};
            J.net = N->actor; J.m = kActor; J.ws = s[S_API]; J.rows = B; J.mode = BM_ACTOR_PI;
            J.crit = Head{N->critic, kQ, s[S_CPI]};
        }
        if (bc) {
            BwdJob& J = G.job[n++];
            J = BwdJob{};
            J.net = N->actor; J.m = kActor; J.ws = s[S_ABC]; J.rows = B; J.mode = BM_ACTOR_BC;
            J.src = bcsrc; J.lambda = Hy->loss_lambda;
        }
        G.njobs = n;
        launch_bwd<2>(G, st);
    }
    HX_CHECK_LAUNCH("hx_hirl_actor_backward");
    return 0;
}

/* Stage 2b: actor parameter gradients grad_actor = w * dL_bc + (1 - w) * dL_rl (HIRL.py:321-324).
 * w_kind 0: w_given (linear / fixed schedule, train_all.py:328-333); 1: soft estimate soft_count / batch + warm
 * (HIRL.py:304-306; soft_count may have been all-reduced and `batch` is then the global batch); 2: stored weight. */
static int actor_wgrad_impl(const HxNets* N, const HxHyper* Hy, int32_t batch, int32_t count_batch, int32_t w_kind, float w_given,
                            float warm, void* stream, int adam_step, bool polyak) {
    HX_REQUIRE(N && Hy && batch > 0 && count_batch > 0, "hx_hirl_actor_wgrad: bad arguments");
    Slot s[S_COUNT];
    make_slots(N, batch, s);
    const bool bc = Hy->use_bc != 0;
    WgArgs W{};
    W.njobs = 1; W.slope = Hy->slope; W.w_kind = bc ? w_kind : 0; W.w_given = bc ? w_given : 0.0f; W.warm = warm;
    W.inv_batch = 1.0f / count_batch; W.soft_count = N->soft_count; W.wstate = N->wstate;
    WgJob& J = W.job[0];
    J = WgJob{};
    J.net = N->actor; J.grad = N->grad_actor; J.m = kActor;
    J.ws[0] = s[S_API]; J.rows[0] = batch; J.wmode[0] = 1;
    J.nslots = 1;
    if (bc) { J.ws[1] = s[S_ABC]; J.rows[1] = batch; J.wmode[1] = 2; J.nslots = 2; }
    if (adam_step > 0) {  // actor.optimizer.step() (+ soft_update of targetActor, + the bf16 image) in the same launch
        J.p = N->actor; J.mom = N->m_actor; J.var = N->v_actor;
        J.target = polyak ? N->target_actor : nullptr;
        J.w2b = N->actor_w2_bf16;
        J.w2f = N->actor_w2_f32i;
        W.ad = make_adam(N, Hy, Hy->lr_actor, adam_step, true);
        launch_wg<true>(W, (hipStream_t)stream);
    } else {
        launch_wg<false>(W, (hipStream_t)stream);
    }
    HX_CHECK_LAUNCH("hx_hirl_actor_wgrad");
    return 0;
}
int hx_hirl_actor_wgrad(const HxNets* N, const HxHyper* Hy, int32_t batch, int32_t count_batch, int32_t w_kind, float w_given,
                        float warm, void* stream) {
    return actor_wgrad_impl(N, Hy, batch, count_batch, w_kind, w_given, warm, stream, 0, false);
}

__global__ void count_to_float_kernel(const int* count, float* out) {
    if (threadIdx.x == 0) out[0] = count ? (float)*count : 0.0f;
}
/* Stage 2b of a sharded run with ONE exchange for the actor phase: the UNWEIGHTED gradients of the two actor losses and the local soft
 * count go into one message, msg = [dL_rl | dL_bc | count, 0...] (hx_actor_message_floats()); after its all-reduce hx_adam_mixed
 * forms w from the global count and combines.  TD3 (use_bc = 0): dL_rl only. */
int hx_hirl_actor_wgrad_split(const HxNets* N, const HxHyper* Hy, int32_t batch, float* msg, void* stream) {
    HX_REQUIRE(N && Hy && batch > 0 && msg && (reinterpret_cast<uintptr_t>(msg) & 15u) == 0, "hx_hirl_actor_wgrad_split: bad arguments");
    Slot s[S_COUNT];
    make_slots(N, batch, s);
    const bool bc = Hy->use_bc != 0;
    WgArgs W{};
    W.njobs = bc ? 2 : 1; W.slope = Hy->slope; W.w_kind = 0; W.w_given = 0.0f; W.inv_batch = 1.0f / batch;
    W.soft_count = N->soft_count; W.wstate = N->wstate;
    for (int j = 0; j < W.njobs; ++j) {
        WgJob& J = W.job[j];
        J = WgJob{};
        J.net = N->actor; J.grad = msg + j * kActor.padded(); J.m = kActor;
        J.ws[0] = s[j == 0 ? S_API : S_ABC]; J.rows[0] = batch; J.wmode[0] = 0; J.nslots = 1;
    }
    launch_wg<false>(W, (hipStream_t)stream);
    hipLaunchKernelGGL(count_to_float_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, bc ? N->soft_count : nullptr, msg + 2 * kActor.padded());
    HX_CHECK_LAUNCH("hx_hirl_actor_wgrad_split");
    return 0;
}



/* Minibatch assembly for one learn() call (replaces UniformMemory.sample buffer.py:38-48, the buffer/expert mixing and the
 * BC draw of HIRL.py:223-251, and the noise draw HIRL.py:265).  do_sample = 1: draw idx[batch] (rows < n_main index the main
 * ring, whose live length min(*total, cap) is read on the device; the rest the expert ring), idx_bc[batch], noise[4] =
 * sigma N(0,1) with Philox4x32-10(seed; row, call), without replacement inside each group.  do_sample = 0: idx / idx_bc are
 * inputs.  Either way the selected rows are then copied into the compact tiles rows[batch][32] / bc_rows[batch][32] that
 * the update stages read. */
int hx_sample_batch(const uint64_t* total, int64_t cap, const float* ring, const float* expert_ring, int64_t expert_len,
                    const float* bc_table, int64_t bc_len, int32_t batch, int32_t n_main, int32_t do_sample, uint64_t seed,
                    uint32_t call, float sigma, int32_t* idx, int32_t* idx_bc, float* noise, float* rows, float* bc_rows,
                    void* stream) {
    HX_REQUIRE(idx && ring && rows && batch > 0 && batch <= 1024 && n_main >= 0 && n_main <= batch, "hx_sample_batch: bad arguments");
    HX_REQUIRE(!do_sample || (total && cap > 0), "hx_sample_batch: sampling needs total and cap");
    HX_REQUIRE(n_main == batch || expert_ring, "hx_sample_batch: expert rows requested without an expert ring");
    SampleArgs A{(const unsigned long long*)total, cap, expert_len, bc_len, batch, n_main, seed, call, sigma, idx, idx_bc, noise,
                 ring, expert_ring ? expert_ring : ring, bc_table, rows, bc_rows, do_sample};
    hipLaunchKernelGGL(sample_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, A);
    HX_CHECK_LAUNCH("hx_sample_batch");
    return 0;
}

/* soft_update of both targets (HIRL.py:327-330) */
int hx_polyak(const HxNets* N, const HxHyper* Hy, void* stream) {
    HX_REQUIRE(N && Hy, "hx_polyak: bad arguments");
    const int nc = 2 * kQ.padded(), na = kActor.size();
    hipLaunchKernelGGL(polyak_kernel, dim3((nc / 4 + kThreads) / kThreads + (na / 4 + kThreads) / kThreads), dim3(kThreads), 0, (hipStream_t)stream,
                       N->target_critic, N->critic, nc, Hy->tau, N->target_actor, N->actor, na);
    HX_CHECK_LAUNCH("hx_polyak");
    return 0;
}

/* One whole Agent.learn on a single GPU (no gradient exchange): the stages above back to back in ONE host call.
 * actor_phase: this is an actorTrainable call (HIRL.py:291); do_polyak: its update_count hits target_update_freq
 * (HIRL.py:327).  critic_step / actor_step: 1-based Adam step numbers of this call. */
int hx_hirl_learn(const HxNets* N, const HxBatch* Bt, const HxHyper* Hy, int32_t critic_step, int32_t actor_phase,
                  int32_t actor_step, int32_t do_polyak, int32_t w_kind, float w_given, float warm, void* stream) {
    // 4 launches on a critic-only call, 8 on an actor call: the two optimizer steps (and the Polyak passes of the calls that move the
    // targets) ride in the wgrad launches, the actor's critic-independent forwards in launches A and B
    return hx_hirl_learn_sampled(N, Bt, Hy, nullptr, critic_step, actor_phase, actor_step, do_polyak, w_kind, w_given, warm, stream);
}
/* The same with the minibatch drawn and gathered inside the first launch (sample == NULL: the tiles of Bt are inputs, as above). */
int hx_hirl_learn_sampled(const HxNets* N, const HxBatch* Bt, const HxHyper* Hy, const HxSample* S, int32_t critic_step, int32_t actor_phase,
                          int32_t actor_step, int32_t do_polyak, int32_t w_kind, float w_given, float warm, void* stream) {
    HX_REQUIRE(critic_step >= 1 && (!actor_phase || actor_step >= 1), "hx_hirl_learn: Adam steps are 1-based");
    int rc = critic_grads_impl(N, Bt, Hy, actor_phase ? (w_kind == 1 ? 2 : 1) : 0, stream, critic_step, do_polyak != 0, S);
    if (rc || !actor_phase) return rc;
    if ((rc = hx_hirl_actor_backward(N, Bt, Hy, w_kind == 1, 1, stream))) return rc;
    return actor_wgrad_impl(N, Hy, Bt->batch, Bt->batch, w_kind, w_given, warm, stream, actor_step, do_polyak != 0);
}

/* BC.Agent.train_actor (hirl/agents/BC.py:160-185): one behaviour-cloning step of the actor on the BC minibatch
 * Bt->bc_rows — loss = mse(actor(s_bc), a_bc) (no loss_lambda here), backward, actor.optimizer.step().
 * losses[2] receives the loss.  Hy->slope = 0.01 reproduces BC.py's LeakyReLU actor. */
int hx_bc_train_actor(const HxNets* N, const HxBatch* Bt, const HxHyper* Hy, int32_t step, void* stream) {
    HX_REQUIRE(N && Bt && Hy && Bt->bc_rows && Bt->batch > 0 && Bt->batch % 16 == 0 && step >= 1, "hx_bc_train_actor: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int B = Bt->batch;
    Slot s[S_COUNT];
    make_slots(N, B, s);
    const RowSrc bcsrc{Bt->bc_rows, nullptr, nullptr, 0, 32};
    {
        FwdArgs F{};
        F.njobs = 1; F.slope = Hy->slope;
        F.zero_f = N->losses + 1; F.zero_nf = 4;
        F.job[0] = FwdJob{N->actor, kActor, bcsrc, 0, 0, Head{}, nullptr, 0.f, s[S_ABC], B, 1};
        launch_fwd(F, st);
    }
    {
        BwdArgs G{};
        G.njobs = 1; G.slope = Hy->slope; G.inv_batch = 1.0f / B; G.losses = N->losses; G.soft_count = N->soft_count;
        BwdJob& J = G.job[0];
        J = BwdJob{};
        J.net = N->actor; J.m = kActor; J.ws = s[S_ABC]; J.rows = B; J.mode = BM_ACTOR_BC; J.src = bcsrc; J.lambda = 1.0f;
        launch_bwd<2>(G, st);
    }
    {
        WgArgs W{};
        W.njobs = 1; W.slope = Hy->slope; W.w_kind = 0; W.w_given = 0.f; W.inv_batch = 1.0f / B;
        W.soft_count = N->soft_count; W.wstate = N->wstate;
        WgJob& J = W.job[0];
        J = WgJob{};
        J.net = N->actor; J.grad = N->grad_actor; J.m = kActor; J.ws[0] = s[S_ABC]; J.rows[0] = B; J.nslots = 1; J.wmode[0] = 0;
        launch_wg<false>(W, st);
    }
    HX_CHECK_LAUNCH("hx_bc_train_actor");
    return hx_adam(N, Hy, 2, step, 1.0f, 0, 0.0f, 0.0f, B, stream);
}

/* ------------------------------------------------------------------------------------------------------------------
 * SAC (hirl/agents/SAC/agent.py, the non-imitative branch train_sac.py uses).
 * Slots: 0 policy(s'), 1 policy(s), 2/3 Q1/Q2(s, a), 4/5 target Q1/Q2(s', a'), 6/7 Q1/Q2(s, a~)
 * ------------------------------------------------------------------------------------------------------------------ */
enum { SS_PN = 0, SS_PC, SS_Q1, SS_Q2, SS_T1, SS_T2, SS_Q1P, SS_Q2P };

int hx_sac_policy_param_count(void) { return kPolicy.padded(); }
int64_t hx_sac_workspace_floats(int32_t batch) { return hx_hirl_workspace_floats(batch) + 32 * (int64_t)batch; }

struct SacAux {
    float* act_n; float* ent_n; float* act_c; float* aux_c;
};
static SacAux sac_aux(const HxSacNets* N, int B) {
    float* p = N->ws + (size_t)S_COUNT * kSlotFloats * B;
    return SacAux{p, p + 4 * B, p + 5 * B, p + 9 * B};  // [B][4], [B], [B][4], [B][16]
}
static void sac_slots(const HxSacNets* N, int B, Slot* s) {
    for (int i = 0; i < S_COUNT; ++i) s[i] = carve_slot(N->ws + (size_t)i * kSlotFloats * B, B);
}

/* SacAgent.explore / exploit (SAC/agent.py:183-196) for `rows` observations.  mode 0: exploit = tanh(mean); 1: sample with the
 * standard-normal draws eps[rows][4]; 2: sample with Philox(seed; row0 + row, call).  ws: unused since the whole policy runs in one kernel (may be NULL). */
static int sac_act_impl(const float* policy, const float* w2f, const float* obs, int64_t rows, float* actions, int32_t mode, const float* eps,
                        uint64_t seed, uint32_t row0, uint32_t call, void* stream) {
    HX_REQUIRE(policy && obs && actions && rows > 0 && mode >= 0 && mode <= 2 && (mode != 1 || eps), "hx_sac_act: bad arguments");
    ActFusedArgs H{policy, kPolicy, const_cast<float*>(obs), (int)rows, 0.0f, actions, mode == 1 ? eps : nullptr, 1, 0.0f, mode, seed, row0, call,
                   nullptr, 0, nullptr, nullptr, nullptr, HxStepOpts{}, 0.0, nullptr, w2f};
    launch_act<true>(H, (hipStream_t)stream);
    HX_CHECK_LAUNCH("hx_sac_act");
    return 0;
}
int hx_sac_act(const float* policy, const float* obs, int64_t rows, float* actions, int32_t mode, const float* eps, uint64_t seed,
               uint32_t row0, uint32_t call, float* ws, void* stream) {
    (void)ws;
    return sac_act_impl(policy, nullptr, obs, rows, actions, mode, eps, seed, row0, call, stream);
}
int hx_sac_act_f32i(const float* policy, const float* w2_f32i, const float* obs, int64_t rows, float* actions, int32_t mode, const float* eps,
                    uint64_t seed, uint32_t row0, uint32_t call, void* stream) {
    HX_REQUIRE(w2_f32i && (reinterpret_cast<uintptr_t>(w2_f32i) & 15u) == 0, "hx_sac_act_f32i: w2_f32i must be a 16-byte aligned fp32 image of W2");
    return sac_act_impl(policy, w2_f32i, obs, rows, actions, mode, eps, seed, row0, call, stream);
}

/* SacAgent.explore / exploit + HarfangEnv.step in one launch (train_sac.py:238-241): hx_sac_act, then hx_env_step in the kernel's tail. */
static int sac_act_step_impl(const float* policy, const float* w2f, float* state, int64_t n, int64_t stride, float* obs_io, float* actions, int32_t mode,
                             const float* eps, uint64_t seed, uint32_t row0, uint32_t call, float* reward, uint8_t* done, int8_t* success,
                             const HxStepOpts* opts, void* stream) {
    HX_REQUIRE(policy && mode >= 0 && mode <= 2 && (mode != 1 || eps), "hx_sac_act_step: bad arguments");
    const HxStepOpts o = opts ? *opts : HxStepOpts{};
    if (int rc = check_step_args(state, n, stride, obs_io, actions, reward, done, success, o, "hx_sac_act_step")) return rc;
    if (n > kFuseEnvMax) {
        if (int rc = sac_act_impl(policy, w2f, obs_io, n, actions, mode, eps, seed, row0, call, stream)) return rc;
        return hx_env_step(state, n, stride, actions, obs_io, reward, done, success, opts, stream);
    }
    ActFusedArgs H{policy, kPolicy, obs_io, (int)n, 0.0f, actions, mode == 1 ? eps : nullptr, 1, 0.0f, mode, seed, row0, call,
                   state, stride, reward, done, success, o, o.cap > 0 ? 1.0 / (double)o.cap : 0.0, nullptr, w2f};
    launch_act<true>(H, (hipStream_t)stream);
    HX_CHECK_LAUNCH("hx_sac_act_step");
    return 0;
}
int hx_sac_act_step(const float* policy, float* state, int64_t n, int64_t stride, float* obs_io, float* actions, int32_t mode,
                    const float* eps, uint64_t seed, uint32_t row0, uint32_t call, float* reward, uint8_t* done, int8_t* success,
                    const HxStepOpts* opts, void* stream) {
    return sac_act_step_impl(policy, nullptr, state, n, stride, obs_io, actions, mode, eps, seed, row0, call, reward, done, success, opts, stream);
}
int hx_sac_act_step_f32i(const float* policy, const float* w2_f32i, float* state, int64_t n, int64_t stride, float* obs_io, float* actions,
                         int32_t mode, const float* eps, uint64_t seed, uint32_t row0, uint32_t call, float* reward, uint8_t* done,
                         int8_t* success, const HxStepOpts* opts, void* stream) {
    HX_REQUIRE(w2_f32i && (reinterpret_cast<uintptr_t>(w2_f32i) & 15u) == 0, "hx_sac_act_step_f32i: w2_f32i must be a 16-byte aligned fp32 image of W2");
    return sac_act_step_impl(policy, w2_f32i, state, n, stride, obs_io, actions, mode, eps, seed, row0, call, reward, done, success, opts, stream);
}

/* Critic half of SacAgent.learn (SAC/agent.py:278-313): [Polyak of the target critics first when polyak_first], a', H' =
 * policy.sample(s') with eps_next, y = r + (1 - d) gamma (min Q_target(s', a') + alpha H'), q1_loss / q2_loss -> losses[0..1],
 * grad_critic.  Also evaluates policy(s) for the policy half.  Follow with hx_sac_adam(which = 0). */
static int sac_critic_grads_impl(const HxSacNets* N, const HxSacBatch* Bt, const HxHyper* Hy, const HxSample* S, int32_t polyak_first, void* stream,
                                 int adam_step = 0) {
    HX_REQUIRE(N && Bt && Hy && Bt->rows && Bt->batch > 0 && Bt->batch % 16 == 0, "hx_sac_critic_grads: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int B = Bt->batch;
    SampleDev SD{};
    bool fused = false;
    if (S) {  // memory.sample(batch_size) inside the first forward launch (or, batch > 256, by the sampling launch right here)
        HX_REQUIRE(!S->bc_table && !S->idx_bc, "hx_sac_critic_grads_sampled: SAC has no BC minibatch");
        if (int rc = prepare_draw(S, B, const_cast<float*>(Bt->rows), nullptr, nullptr, stream, &SD, &fused)) return rc;
    }
    Slot s[S_COUNT];
    sac_slots(N, B, s);
    const SacAux X = sac_aux(N, B);
    const RowSrc src{Bt->rows, nullptr, nullptr, 0, 32};
    const int nq = 2 * kQs.padded();
    if (polyak_first)  // soft_update(critic_target, critic) BEFORE the update, agent.py:278-279
        hipLaunchKernelGGL(polyak_kernel, dim3((nq / 4 + kThreads) / kThreads), dim3(kThreads), 0, st, N->target_critic, N->critic, nq, Hy->tau);
    const float* q1 = N->critic; const float* q2 = N->critic + kQs.padded();
    const float* t1 = N->target_critic; const float* t2 = N->target_critic + kQs.padded();
    {   // policy(s'), policy(s), Q1/Q2(s, a)
        FwdArgs F{};
        F.njobs = 4; F.slope = 0.0f;
        F.zero_f = N->losses; F.zero_nf = 5;
        F.job[0] = FwdJob{N->policy, kPolicy, src, 17, 0, Head{}, nullptr, 0.f, s[SS_PN], B, 0};
        F.job[1] = FwdJob{N->policy, kPolicy, src, 0, 0, Head{}, nullptr, 0.f, s[SS_PC], B, 1};
        F.job[2] = FwdJob{q1, kQs, src, 0, 0, Head{}, nullptr, 0.f, s[SS_Q1], B, 1};
        F.job[3] = FwdJob{q2, kQs, src, 0, 0, Head{}, nullptr, 0.f, s[SS_Q2], B, 1};
        F.sample = fused ? &SD : nullptr;
        launch_fwd(F, st);
    }
    {   // a', H' = policy.sample(s')
        GaussArgs G{N->policy, kPolicy, s[SS_PN].z2, Bt->eps_next, B, Bt->eps_next ? 1 : 2, X.act_n, X.ent_n, nullptr, Bt->seed, 0x40000000u, Bt->call};
        hipLaunchKernelGGL(gauss_head_kernel, dim3((unsigned)((B + 3) / 4)), dim3(kThreads), 0, st, G);
    }
    {   // target Q1/Q2 (s', a')
        FwdArgs F{};
        F.njobs = 2; F.slope = 0.0f;
        F.job[0] = FwdJob{t1, kQs, src, 17, 3, Head{}, X.act_n, 0.f, s[SS_T1], B, 0};
        F.job[1] = FwdJob{t2, kQs, src, 17, 3, Head{}, X.act_n, 0.f, s[SS_T2], B, 0};
        launch_fwd(F, st);
    }
    {   // y, losses, dq, dh1
        BwdArgs G{};
        G.njobs = 2; G.slope = 0.0f; G.inv_batch = 1.0f / B; G.losses = N->losses; G.soft_count = nullptr;
        for (int h = 0; h < 2; ++h) {
            BwdJob& J = G.job[h];
            J = BwdJob{};
            J.net = h ? q2 : q1; J.m = kQs; J.ws = s[SS_Q1 + h]; J.rows = B; J.mode = BM_CRITIC_TD;
            J.t1 = Head{t1, kQs, s[SS_T1]}; J.t2 = Head{t2, kQs, s[SS_T2]}; J.src = src; J.gamma = Hy->gamma;
            J.bonus = X.ent_n; J.bonus_scale = N->alpha_state + 3; J.loss_slot = h;
        }
        launch_bwd<0>(G, st);
    }
    {
        WgArgs W{};
        W.njobs = 2; W.slope = 0.0f; W.w_kind = 0; W.inv_batch = 1.0f / B; W.soft_count = nullptr; W.wstate = nullptr;
        for (int h = 0; h < 2; ++h) {
            WgJob& J = W.job[h];
            J = WgJob{};
            J.net = h ? q2 : q1; J.grad = N->grad_critic + h * kQs.padded(); J.m = kQs;
            J.ws[0] = s[SS_Q1 + h]; J.rows[0] = B; J.nslots = 1; J.wmode[0] = 0;
            if (adam_step > 0) {  // q1_optim.step() / q2_optim.step() ride in the wgrad launch (one GPU): the thread that produced a gradient steps it
                J.p = N->critic + h * kQs.padded();
                J.mom = N->m_critic + h * kQs.padded(); J.var = N->v_critic + h * kQs.padded();
            }
        }
        if (adam_step > 0) {
            const double b1 = 0.9, b2 = 0.999;
            const double bc1 = 1.0 - pow(b1, adam_step), bc2 = 1.0 - pow(b2, adam_step);
            W.ad = WgAdam{};
            W.ad.b1 = (float)b1; W.ad.b2 = (float)b2; W.ad.eps = 1e-8f;
            W.ad.step_size = (float)(Hy->lr_critic / bc1);
            W.ad.bc2_sqrt = (float)sqrt(bc2);
            W.ad.losses = N->losses;
            launch_wg<true>(W, st);
        } else {
            launch_wg<false>(W, st);
        }
    }
    HX_CHECK_LAUNCH("hx_sac_critic_grads");
    return 0;
}
int hx_sac_critic_grads(const HxSacNets* N, const HxSacBatch* Bt, const HxHyper* Hy, int32_t polyak_first, void* stream) {
    return sac_critic_grads_impl(N, Bt, Hy, nullptr, polyak_first, stream);
}
/* The same with memory.sample (SAC/agent.py:286-296) drawn and gathered inside its first launch (HxSample without a BC table; Bt->rows is
 * the output tile): bit-identical to hx_sample_batch followed by hx_sac_critic_grads. */
int hx_sac_critic_grads_sampled(const HxSacNets* N, const HxSacBatch* Bt, const HxHyper* Hy, const HxSample* S, int32_t polyak_first, void* stream) {
    HX_REQUIRE(S, "hx_sac_critic_grads_sampled: null sample description");
    return sac_critic_grads_impl(N, Bt, Hy, S, polyak_first, stream);
}

/* Policy half (SAC/agent.py:315-319, 376-406): a~, H = policy.sample(s) with eps_cur, Q1/Q2(s, a~) with the UPDATED critics,
 * policy_loss = mean(-min Q - alpha H) -> losses[2], mean entropy -> losses[4], grad_policy.  Follow with hx_sac_adam(which = 1). */
/* One GPU: hx_sac_critic_grads[_sampled] + hx_sac_adam(which = 0, grad_scale 1) as ONE call with the optimizer step inside the weight-gradient
 * launch (sample may be null: the minibatch was assembled by the caller).  Same Adam on the same gradients: bit-identical to the two calls. */
int hx_sac_critic_step(const HxSacNets* N, const HxSacBatch* Bt, const HxHyper* Hy, const HxSample* S, int32_t polyak_first, int32_t step, void* stream) {
    HX_REQUIRE(step >= 1, "hx_sac_critic_step: step is 1-based");
    return sac_critic_grads_impl(N, Bt, Hy, S, polyak_first, stream, step);
}
int hx_sac_policy_grads(const HxSacNets* N, const HxSacBatch* Bt, const HxHyper* Hy, void* stream) {
    HX_REQUIRE(N && Bt && Hy && Bt->rows && Bt->batch > 0 && Bt->batch % 16 == 0, "hx_sac_policy_grads: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int B = Bt->batch;
    Slot s[S_COUNT];
    sac_slots(N, B, s);
    const SacAux X = sac_aux(N, B);
    const RowSrc src{Bt->rows, nullptr, nullptr, 0, 32};
    const float* q1 = N->critic; const float* q2 = N->critic + kQs.padded();
    {
        GaussArgs G{N->policy, kPolicy, s[SS_PC].z2, Bt->eps_cur, B, Bt->eps_cur ? 1 : 2, X.act_c, nullptr, X.aux_c, Bt->seed, 0x80000000u, Bt->call};
        hipLaunchKernelGGL(gauss_head_kernel, dim3((unsigned)((B + 3) / 4)), dim3(kThreads), 0, st, G);
    }
    {
        FwdArgs F{};
        F.njobs = 2; F.slope = 0.0f;
        F.job[0] = FwdJob{q1, kQs, src, 0, 3, Head{}, X.act_c, 0.f, s[SS_Q1P], B, 1};
        F.job[1] = FwdJob{q2, kQs, src, 0, 3, Head{}, X.act_c, 0.f, s[SS_Q2P], B, 1};
        launch_fwd(F, st);
    }
    {
        QSelArgs Q{q1, q2, kQs, s[SS_Q1P], s[SS_Q2P], B, 1.0f / B, N->losses};
        hipLaunchKernelGGL(q_select_kernel, dim3((unsigned)((B + 3) / 4)), dim3(kThreads), 0, st, Q);
    }
    {   // both critics backward down to dh1 with the given head gradients
        BwdArgs G{};
        G.njobs = 2; G.slope = 0.0f; G.inv_batch = 1.0f / B; G.losses = N->losses;
        for (int h = 0; h < 2; ++h) {
            BwdJob& J = G.job[h];
            J = BwdJob{};
            J.net = h ? q2 : q1; J.m = kQs; J.ws = s[SS_Q1P + h]; J.rows = B; J.mode = BM_GIVEN;
        }
        launch_bwd<3>(G, st);
    }
    {
        PDoutArgs P{q1, q2, kQs, s[SS_Q1P], s[SS_Q2P], s[SS_PC], X.aux_c, N->alpha_state, B, 1.0f / B, N->losses};
        hipLaunchKernelGGL(policy_dout_kernel, dim3((unsigned)((B + 3) / 4)), dim3(kThreads), 0, st, P);
    }
    {
        BwdArgs G{};
        G.njobs = 1; G.slope = 0.0f; G.inv_batch = 1.0f / B; G.losses = N->losses;
        BwdJob& J = G.job[0];
        J = BwdJob{};
        J.net = N->policy; J.m = kPolicy; J.ws = s[SS_PC]; J.rows = B; J.mode = BM_GIVEN;
        launch_bwd<3>(G, st);
    }
    {
        WgArgs W{};
        W.njobs = 1; W.slope = 0.0f; W.w_kind = 0; W.inv_batch = 1.0f / B;
        WgJob& J = W.job[0];
        J = WgJob{};
        J.net = N->policy; J.grad = N->grad_policy; J.m = kPolicy; J.ws[0] = s[SS_PC]; J.rows[0] = B; J.nslots = 1; J.wmode[0] = 0;
        launch_wg<false>(W, st);
    }
    HX_CHECK_LAUNCH("hx_sac_policy_grads");
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
    if (which == 1) {
        A.alpha_state = N->alpha_state;
        A.target_entropy = target_entropy;
        A.alpha_step_size = (float)(Hy->lr_actor / bc1);
    }
    hipLaunchKernelGGL(adam_kernel, dim3((A.n / 4 + kThreads) / kThreads), dim3(kThreads), 0, (hipStream_t)stream, A);
    HX_CHECK_LAUNCH("hx_sac_adam");
    return 0;
}

}  // extern "C"
