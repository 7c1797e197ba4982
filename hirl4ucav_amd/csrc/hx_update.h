// hx_update.h — what the translation units of the actor/critic side share (gfx950): workspace slots, row helpers, the heads, the
// counter-based RNG and the in-launch minibatch draw, the job descriptions the host sequencing (hx_hirl.hip, hx_sac.hip) hands to the
// launchers of the kernel files, and the optimizer arithmetic.
//
//   hx_fwdbwd.hip   fwd_l2_kernel, bwd_l2_kernel            launch_fwd, launch_bwd
//   hx_wgrad.hip    wgrad_kernel, adam_kernel, polyak_kernel launch_wg, launch_adam, launch_polyak; hx_adam*, hx_polyak, hx_sac_adam
//   hx_act.hip      act_fused_kernel + the W2 image packers  hx_actor_act*, hx_sac_act*, hx_pack_w2_*
//   hx_sampler.hip  sample_kernel                            hx_sample_batch
//   hx_sac.hip      gauss_head / q_select / policy_dout      hx_sac_critic_*, hx_sac_policy_grads
//   hx_hirl.hip     host sequencing of Agent.learn           hx_hirl_*, hx_bc_train_actor
//
// What the kernels replace (reference file:line):
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
#pragma once
#include <cstdlib>
#include <type_traits>

#include "hx_common.h"
#include "hx_nn.h"

// In-kernel phase stamps for diagnosis only (make stamps): s_memrealtime ticks (10 ns) between phases of ONE workgroup land in the
// translation unit's own hx_dbg (each kernel file owns a range of the 80 stamp words, hx_debug_stamps gathers them); the shipped build
// compiles them out.
#ifdef HX_STAMPS
static __device__ float hx_dbg[80];
#define STAMP_DECL unsigned long long TS_[16]; int tsn_ = 0
#define STAMP() TS_[tsn_++] = __builtin_amdgcn_s_memrealtime()
#define STAMP_FLUSH(base, cond) do { if (cond) { for (int i_ = 1; i_ < tsn_; ++i_) hx_dbg[(base) + i_] = (float)(TS_[i_] - TS_[i_ - 1]); hx_dbg[(base)] = (float)tsn_; } } while (0)
// life span of EVERY workgroup of every launch (first stamp .. now), appended to the file's log: where a learn() spends its time BETWEEN
// workgroups.  tag = HX_SPAN_* (which kernel) | blockIdx.x << 8 | blockIdx.y << 24
constexpr int kSpanCap = 8192;
static __device__ unsigned long long hx_span[kSpanCap][2];
static __device__ unsigned hx_span_tag[kSpanCap];
static __device__ unsigned hx_span_n;
#define SPAN_LOG(tag) do { if (threadIdx.x == 0) { const unsigned long long e_ = __builtin_amdgcn_s_memrealtime(); const unsigned i_ = atomicAdd(&hx_span_n, 1u); \
    if (i_ < (unsigned)kSpanCap) { hx_span[i_][0] = TS_[0]; hx_span[i_][1] = e_; hx_span_tag[i_] = (unsigned)(tag) | (blockIdx.x << 8) | (blockIdx.y << 24); } } } while (0)
// the file's share of hx_debug_stamps / hx_debug_spans (hx_core.hip): stamp words [lo, hi) and the span log, which is cleared
#define HX_DEFINE_DEBUG_COLLECTORS(name, lo, hi) \
    namespace hx { \
    int dbg_stamps_##name(float* host80) { \
        float tmp[80]; \
        if (hipMemcpyFromSymbol(tmp, HIP_SYMBOL(hx_dbg), sizeof(tmp)) != hipSuccess) return -2; \
        for (int i = (lo); i < (hi); ++i) host80[i] = tmp[i]; \
        return 0; \
    } \
    int dbg_spans_##name(unsigned long long* spans, unsigned* tags, unsigned* n, unsigned cap) { \
        unsigned k = 0, zero = 0u; \
        if (hipMemcpyFromSymbol(&k, HIP_SYMBOL(hx_span_n), sizeof(unsigned)) != hipSuccess) return -2; \
        if (k > (unsigned)kSpanCap) k = kSpanCap; \
        if (k > cap - *n) k = cap - *n; \
        if (k && hipMemcpyFromSymbol(spans + 2 * (size_t)*n, HIP_SYMBOL(hx_span), sizeof(unsigned long long) * 2 * k) != hipSuccess) return -2; \
        if (k && hipMemcpyFromSymbol(tags + *n, HIP_SYMBOL(hx_span_tag), sizeof(unsigned) * k) != hipSuccess) return -2; \
        *n += k; \
        return hipMemcpyToSymbol(HIP_SYMBOL(hx_span_n), &zero, sizeof(unsigned)) == hipSuccess ? 0 : -2; \
    } \
    }
#else
#define STAMP_DECL
#define STAMP()
#define STAMP_FLUSH(base, cond)
#define SPAN_LOG(tag)
#define HX_DEFINE_DEBUG_COLLECTORS(name, lo, hi)
#endif
enum { HX_SPAN_FWD = 1, HX_SPAN_ACT = 2, HX_SPAN_BWD = 3, HX_SPAN_WGRAD = 4, HX_SPAN_FRONT_A = 5, HX_SPAN_FRONT_B = 6 };

namespace hxu {
using namespace hxnn;

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
// agent-scope relaxed accesses: data handed from one workgroup to another INSIDE a launch (hx_front.hip) goes past the non-coherent levels with
// these and needs no fence (tools/ubench/handoff_probe.hip: a release / acquire fence costs a cache write-back / invalidate on this machine)
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

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
    __device__ __forceinline__ void load_agent(const float* p) {  // the same elements through agent-scope loads
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int i = 0; i < PER4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) v[4 * i + c] = ld_agent(p + (i * 64 + lane) * 4 + c);
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
template <int OUTMAX, bool RELU, bool AGENT = false>
__device__ __forceinline__ void head_row(const float* __restrict__ z2row, const float* __restrict__ net, const Mlp m, float slope,
                                         RowReg<H2>& xhat, RowReg<H2>& y, float& mean, float& rstd, float (&o)[OUTMAX]) {
    RowReg<H2> z, g, be;
    if (AGENT) z.load_agent(z2row);
    else z.load(z2row);
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
template <bool RELU, bool AGENT = false>
__device__ __forceinline__ void head_row4(const float* __restrict__ z2row, const float* __restrict__ net, const Mlp m, float slope,
                                          RowReg<H2>& xhat, RowReg<H2>& y, float& mean, float& rstd, float (&o)[4]) {
    RowReg<H2> z, g, be, w0, w1, w2, w3;
    if (AGENT) z.load_agent(z2row);
    else z.load(z2row);
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
    // guard > 0 (hx_front.hip: the draw runs BESIDE the env step of the same launch): `total` points at a snapshot taken before that env step and the
    // `guard` ring slots the step may overwrite are not drawn — the population is every transition that is in the ring before AND after the step
    // (slot_of_draw below); 0: UniformMemory.sample over the whole buffer (buffer.py:45)
    uint32_t guard;
};
// draw k of `live` -> ring slot.  Without a guard the slot IS the draw.  With one, the window [h, h + guard) mod cap behind the ring head h is left out:
// its wrapped part [0, e0) shifts every draw up, and a draw at or beyond h jumps the rest of the window.
struct DrawMap {
    uint32_t live, e0, h, jump;
};
__device__ __forceinline__ DrawMap draw_map(unsigned long long tot, unsigned long long cap, uint32_t guard) {
    DrawMap m;
    if (guard == 0u) {  // (the default: no 64-bit division on the path of every launch-A workgroup)
        m.live = (uint32_t)(tot < cap ? tot : cap);
        m.e0 = 0u; m.h = 0xFFFFFFFFu; m.jump = 0u;
        return m;
    }
    if (tot < cap) {
        const unsigned long long over = tot + guard > cap ? tot + guard - cap : 0ull;
        m.e0 = (uint32_t)(over < tot ? over : tot);
        m.live = (uint32_t)tot - m.e0;
        m.h = 0xFFFFFFFFu; m.jump = 0u;
    } else {
        const uint32_t g = guard < cap ? guard : (uint32_t)cap;
        m.h = (uint32_t)(tot % cap);
        const unsigned long long over = (unsigned long long)m.h + g > cap ? (unsigned long long)m.h + g - cap : 0ull;
        m.e0 = (uint32_t)over;
        m.live = (uint32_t)cap - g;
        m.jump = g - m.e0;
        if (g == 0u) m.h = 0xFFFFFFFFu;
    }
    return m;
}
__device__ __forceinline__ uint32_t slot_of_draw(const DrawMap& m, uint32_t k) {
    uint32_t s = k + m.e0;
    if (s >= m.h) s += m.jump;
    return s;
}
// the (4,) target-smoothing draw of a learn() call, HIRL.py:265 (sample_kernel's arithmetic): component k
template <typename S>
__device__ __forceinline__ float smoothing_noise(const S& SA, int k) {
    uint32_t uu[4];
    philox4x32_10(0xFFFFFFF0u, SA.call, 2u, 0u, (uint32_t)SA.seed, (uint32_t)(SA.seed >> 32), uu);
    const float ua = u01(uu[k & 2]), ub = u01(uu[(k & 2) + 1]);
    const float rad = sqrtf(-2.0f * __logf(ua)), ang = 6.28318530717958647692f * ub;
    return SA.sigma * ((k & 1) ? rad * __sinf(ang) : rad * __cosf(ang));
}
constexpr int kFusedSlots = 1024, kFusedBatchMax = 256;
__device__ __forceinline__ uint32_t fused_hash(uint32_t k) { return (k * 2654435761u) >> 22; }  // top 10 bits

__device__ __forceinline__ void draw_fused(const SampleDev& S, int B, uint32_t (*hkey)[kFusedSlots], int (*hown)[kFusedSlots], int (*fin)[kFusedBatchMax]) {
    const int tid = threadIdx.x;
    const int t = tid & 511, stream = tid >> 9;  // 0: replay / expert rows, 1: BC rows
    const unsigned long long tot = *S.total;
    const DrawMap dm = draw_map(tot, (unsigned long long)S.cap, S.guard);
    const uint32_t len_main = dm.live;
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
    if (live) fin[stream][t] = (stream == 0 && main_grp) ? (int)slot_of_draw(dm, key & 0x7FFFFFFFu) : (int)(key & 0x7FFFFFFFu);
    __syncthreads();
}

// One workgroup = one whole minibatch assembly (what hx_sample_batch does as a launch of its own): draw, indices, smoothing noise, row tiles.
// hx_hirl_learn_back runs it as an extra workgroup of the critics' wgrad launch (224 workgroups: CUs to spare) for the NEXT front launch (hx_front.hip), whose workgroups then start from
// finished tiles instead of each repeating the draw (3-4 us of every launch-A workgroup, tools/ubench/front_spans.py) and chasing an index into
// the ring.  scratch: 18 KB of LDS.
__device__ __forceinline__ void predraw_wg(const SampleDev& S, int B, float* scratch) {
    auto hkey = reinterpret_cast<uint32_t (*)[kFusedSlots]>(scratch);
    auto hown = reinterpret_cast<int (*)[kFusedSlots]>(scratch + 2 * kFusedSlots);
    auto fin = reinterpret_cast<int (*)[kFusedBatchMax]>(scratch + 4 * kFusedSlots);
    draw_fused(S, B, hkey, hown, fin);
    const int tid = threadIdx.x;
    if (tid < B) {
        S.idx[tid] = fin[0][tid];
        if (S.idx_bc) S.idx_bc[tid] = fin[1][tid];
    }
    if (tid < 4 && S.noise) S.noise[tid] = smoothing_noise(S, tid);
    const bool bc = S.bc_rows && S.bc_table && S.idx_bc;
    for (int e = tid; e < B * 8; e += kWide) {  // 8 lanes per row, one 16-byte piece each; both tiles' loads in flight together
        const int r = e >> 3, c = e & 7;
        const float4 m = reinterpret_cast<const float4*>((r < S.n_main ? S.ring : S.expert_ring) + (size_t)fin[0][r] * 32)[c];
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bc) q = reinterpret_cast<const float4*>(S.bc_table + (size_t)fin[1][r] * 32)[c];
        reinterpret_cast<float4*>(S.rows)[e] = m;
        if (bc) reinterpret_cast<float4*>(S.bc_rows)[e] = q;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// job descriptions: what the host sequencing hands to launch_fwd / launch_bwd / launch_wg
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
    int img;   // bf16 update path: which image of FwdArgs::images holds this net's W2 (IM_*; forward images are 0..7)
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
    const uint16_t* images;   // nullptr: the 256 -> 512 product on fp32 MFMA; else the bf16 W2 images (HxNets.w2_bf16_all): bf16 MFMA
};

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
// the TRANSPOSED bf16 image, for dh1 = dz2 W2 (B[k = n][col = k1] = W2[n][k1]): one 1 KB block per (tile of 16 columns k1, slab of 32 n),
// lane (r = k1 in the tile, g): n = 32 slab + 8 g .. + 7
__host__ __device__ inline uint32_t w2t_image_index(uint32_t k1, uint32_t n) {
    return ((((k1 >> 4) * 16u + (n >> 5)) * 4u + ((n >> 3) & 3u)) * 16u + (k1 & 15u)) * 8u + (n & 7u);
}
constexpr int LDB1 = H1 + 16;  // bf16 h1 tile pitch (elements) = 136 dwords = 8 mod 64: the ds_read_b128 A-operand read is conflict-free
constexpr int LDB2 = H2 + 16;  // bf16 dz2 tile pitch = 264 dwords = 8 mod 64
// bf16 update path (BASELINE.json configs[4] "bf16 actor/critic"): HxNets.w2_bf16_all holds one image per use of a W2 —
// forward images (w2_image_index) first, so that a FwdJobC's 3-bit field can name them, then the transposed ones (w2t_image_index)
enum { IM_ACTOR = 0, IM_C1, IM_C2, IM_TA, IM_TC1, IM_TC2, IM_BC, IM_ACTOR_T, IM_C1_T, IM_C2_T, IM_COUNT };
constexpr size_t kImgElems = (size_t)H2 * H1;
// x = hi + mid + lo, exactly: three bf16 numbers (8 significand bits each, round to nearest even; the two remainders are exact in fp32)
__device__ __forceinline__ void split3_bf16(float x, uint16_t& hi, uint16_t& mid, uint16_t& lo) {
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    const __bf16 l = (__bf16)r2;
    hi = __builtin_bit_cast(uint16_t, h); mid = __builtin_bit_cast(uint16_t, m); lo = __builtin_bit_cast(uint16_t, l);
}
__device__ __forceinline__ v8bf as_v8bf(const uint4& q) { return __builtin_bit_cast(v8bf, q); }
__device__ __forceinline__ v4f mfma16_bf16(const uint4& a, const uint4& b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_v8bf(a), as_v8bf(b), c, 0, 0, 0);
}
// eight floats -> eight bf16 (round to nearest even, v_cvt_pk_bf16_f32), packed in MFMA fragment order
__device__ __forceinline__ uint4 pack8_bf16(const float (&v)[8]) {
    typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
    uint4 q;
    q.x = __builtin_bit_cast(unsigned, v2bf{(__bf16)v[0], (__bf16)v[1]});
    q.y = __builtin_bit_cast(unsigned, v2bf{(__bf16)v[2], (__bf16)v[3]});
    q.z = __builtin_bit_cast(unsigned, v2bf{(__bf16)v[4], (__bf16)v[5]});
    q.w = __builtin_bit_cast(unsigned, v2bf{(__bf16)v[6], (__bf16)v[7]});
    return q;
}

enum { BM_CRITIC_TD = 0, BM_CRITIC_PI = 1, BM_ACTOR_PI = 2, BM_ACTOR_BC = 3, BM_GIVEN = 4,
       BM_SAC_QMIN = 5,     // SAC policy loss, a critic's side: dq = -w / B, w = 1 for the smaller of Q1 / Q2 (s, a~), 1/2 each on a tie  (SAC/agent.py:380-383)
       BM_SAC_POLICY = 6 }; // ... the policy's side: the critics' dL/da chained through a = tanh(mean + sigma eps), plus the entropy term (:404-406)

// The gradient of SAC's policy loss wrt the policy head's pre-activations (mean_j, log_std_j) of one action component, shared by
// policy_dout_kernel and bwd_l2<5> so that both round alike (contraction off, the multiply-adds spelled out):
//   da = dL/da_j from the critics, a = tanh(x), se = sigma eps, mask = log_std clamp pass-through, ab = alpha / B
#pragma clang fp contract(off)
__device__ __forceinline__ void sac_policy_dout(float da, float a, float se, float mask, float ab, float& d_mean, float& d_logstd) {
    const float t = 1.0f - a * a;
    const float dHdx = (-2.0f * a * t) / (t + 1e-6f);
    const float dLdx = da * t - ab * dHdx;
    d_mean = dLdx;
    d_logstd = (dLdx * se - ab) * mask;
}
#pragma clang fp contract(fast)

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
    int img_t;  // bf16 update path: which image of BwdArgs::images holds this net's TRANSPOSED W2 (IM_*_T)
    // BM_SAC_QMIN: t1 = the OTHER critic evaluated on (s, a~) (its z2 rows); loss_slot 0: this job also logs -min(Q1, Q2) / B into losses[2]
    // BM_SAC_POLICY: t1 / t2 = the two critics' slots on (s, a~) (dh1, z1 read), bonus = the Gaussian head's aux rows [rows][16]
    //                (a[4], sigma eps[4], clamp mask[4], entropy), bonus_scale = &alpha
};
struct BwdArgs {
    BwdJob job[2];
    int njobs;
    float slope;
    float inv_batch;  // 1 / B
    float* losses;    // [8]: critic, actor, bc, rl, bc_fire, bc_weight, -, -
    int* soft_count;
    const uint16_t* images;  // nullptr: dh1 = dz2 W2 on fp32 MFMA; else the bf16 images: dz2 rounded to bf16, W2^T from its image
};

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
    uint16_t* w2tb;                // nullptr, or the transposed bf16 image of W2 to refresh (bf16 update path)
    uint16_t* tgt_w2b;             // nullptr, or the bf16 image of the TARGET's W2: follows the Polyak step
    int w2b_x9;                    // w2b is the FIRST of three images, hi | mid | lo (HxNets.actor_w2_x9): the exact split of every weight
};
struct WgAdam {
    float b1, b2, eps, step_size, bc2_sqrt, tau;
    int finish_actor, use_bc;  // thread 0 of the launch finishes actor_loss / bc_weight (HIRL.py:321,334)
    float* losses;
    float* wstate;
    // SAC policy step: thread 0 of the launch also steps log_alpha (SAC/agent.py:322-325, 408-414) with the mean entropy in losses[4]
    float* alpha_state;  // [4]: log_alpha, m, v, alpha; nullptr = not a SAC policy step
    float target_entropy, alpha_step_size;
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
    int bf16;  // dW2 = dz2^T h1 with both operands rounded to bf16 (fp32 accumulate): the bf16 update path
    float* count_out;  // nullptr, or where thread 0 of the launch leaves (float)*soft_count: the merged actor message's count word
    const SampleDev* predraw;  // hx_hirl_learn_back: one more workgroup draws and gathers the NEXT front launch's minibatch (predraw_wg), or null
    int predraw_batch;
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
    int w2b_x9;  // w2b is the first of three images hi | mid | lo (HxNets.actor_w2_x9)
    int w2_lo;
    // bf16 update path: up to two W2 ranges of p (the critic's two heads) with their forward / transposed images and the images of the
    // target's W2 (written when `target` is stepped here); nseg = 0: none
    int nseg;
    int seg_lo[2];
    uint16_t* seg_w2b[2];
    uint16_t* seg_w2tb[2];
    uint16_t* seg_tgt_w2b[2];
    // merged actor message of a sharded run: g holds the summed dL_rl, g2 the summed dL_bc, *countf the summed
    // soft count; the step uses g = w g2 + (1 - w) g with w from the GLOBAL count.  nullptr: g is the finished gradient.
    const float* g2;
    const float* countf;
    const uint32_t* guard;  // nullptr, or a word that must be 0 for the step to happen (HxNets.xchg_status: a failed exchange)
};

constexpr size_t kSlotFloats = XP + H1 + 2 + H1 + H2 + 2 + OW + H2 + H1 + OW + 2 * kColWgB;  // per row

enum { S_TA = 0, S_C1, S_C2, S_TC1, S_TC2, S_API, S_ABC, S_BCS, S_CPI, S_CSOFT, S_COUNT };

const Mlp kActor{13, 4, 0};
const Mlp kQ{17, 1, 0};
const Mlp kPolicy{13, 8, 1};  // SAC GaussianPolicy: Linear-ReLU stack, head = mean ++ log_std (SAC/model.py:58-60)
const Mlp kQs{17, 1, 1};      // SAC Q head (SAC/model.py:21-23)

__device__ __forceinline__ float pick8(const float (&o)[8], int i) {
    return i == 0 ? o[0] : i == 1 ? o[1] : i == 2 ? o[2] : i == 3 ? o[3] : i == 4 ? o[4] : i == 5 ? o[5] : i == 6 ? o[6] : o[7];
}

// An HxSample for a launch-A draw: validated, then either the device-side description (batch <= 256: *fused = true, launch A draws and
// gathers) or the sampling launch right here (larger batches).  rows / bc_rows / noise: the tiles the later launches read.
inline int prepare_draw(const HxSample* S, int B, float* rows, float* bc_rows, float* noise, void* stream, SampleDev* SD, bool* fused, bool launch_now = true) {
    HX_REQUIRE(S->total && S->cap > 0 && S->ring && S->idx && rows && S->n_main >= 0 && S->n_main <= B,
               "hx_*_sampled: the draw needs total, cap, ring, idx and the output tile rows");
    HX_REQUIRE(S->n_main == B || S->expert_ring, "hx_*_sampled: expert rows requested without an expert ring");
    HX_REQUIRE(!S->bc_table == !S->idx_bc && (!S->bc_table || bc_rows), "hx_*_sampled: bc_table, idx_bc and bc_rows go together");
    *fused = B <= kFusedBatchMax;
    HX_REQUIRE(!S->guard || *fused, "hx_*_sampled: HxSample.guard is honoured by the fused draw only (batch <= 256)");
    HX_REQUIRE(!S->guard || 2 * (int64_t)S->guard <= S->cap, "hx_*_sampled: a guard of more than half the ring leaves too little to draw from (cap >= 2 n)");
    if (*fused) {
        *SD = SampleDev{(const unsigned long long*)S->total, S->ring, S->expert_ring ? S->expert_ring : S->ring, S->bc_table, rows,
                        S->bc_table ? bc_rows : nullptr, noise, S->idx, S->idx_bc, (long long)S->cap, (uint32_t)S->expert_len, (uint32_t)S->bc_len,
                        S->n_main, S->call, S->seed, S->sigma, S->guard};
        return 0;
    }
    if (!launch_now) return 0;
    return hx_sample_batch(S->total, S->cap, S->ring, S->expert_ring, S->expert_len, S->bc_table, S->bc_len, B, S->n_main, 1, S->seed, S->call,
                           S->sigma, S->idx, S->idx_bc, noise, rows, S->bc_table ? bc_rows : nullptr, stream);
}


// launchers (defined beside their kernels)
void launch_fwd(const FwdArgs& F, hipStream_t st);                          // hx_fwdbwd.hip
void set_fwd_nt(int nt, int skip, int count);                                                  // hx_fwdbwd.hip (hx_debug_set_fwd_nt)
void launch_bwd(int grp, const BwdArgs& G, hipStream_t st);                 // hx_fwdbwd.hip: grp = bwd_l2_kernel's GRP (0..3)
// hx_front.hip: the act + env + insert workgroups of hx_actor_act_step_f32i (32 rows each) and the workgroups of launches A and B as ONE launch
int launch_front(const float* actor, const float* w2f, const uint16_t* w2x, const uint16_t* w2b, float* state, int64_t n, int64_t stride, float* obs_io, float* actions, int32_t noise_mode,
                 const float* noise, float sigma, uint64_t seed, uint32_t row0, uint32_t call, float slope, float* reward, uint8_t* done, int8_t* success,
                 const HxStepOpts& o, const FwdArgs& FA, const FwdArgs& FB, const BwdArgs* GC, const HxFront& front, hipStream_t st);
void launch_wg(const WgArgs& W, bool adam, hipStream_t st);                 // hx_wgrad.hip: adam = the optimizer step rides in the launch
void launch_adam(const AdamArgs& A, hipStream_t st);                        // hx_wgrad.hip
void launch_polyak(float* target, const float* source, int n, float tau, float* target2, const float* source2, int n2, hipStream_t st,
                   const uint32_t* guard = nullptr);  // guard: a word that must be 0 for the step to happen (a failed exchange)
// bf16 image of the W2 [512][256] at `w2`: forward order (w2_image_index) or transposed (w2t_image_index)             hx_act.hip
void launch_pack_bf16(const float* w2, uint16_t* image, bool transposed, hipStream_t st);

}  // namespace hxu
