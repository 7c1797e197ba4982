// hx_act.hip — batched policy inference (gfx950): the whole policy for 16 / 32 observation rows in ONE workgroup, optionally with the
// env step of the same rows in the kernel's tail, and the re-ordered images of W2 it reads.
//   Agent.chooseAction / chooseActionSmallNoise / chooseActionNoNoise   hirl/agents/HIRL.py:192-212   (U5)  -> hx_actor_act*
//   chooseAction + HarfangEnv.step                                      hirl/train_all.py:343-345           -> hx_actor_act_step*
//   SacAgent.explore / exploit                                          hirl/agents/SAC/agent.py:183-196    -> hx_sac_act*
#include <hip/hip_ext.h>

#include "hx_act_body.h"

using namespace hxnn;
using namespace hxu;
using namespace hxact;

namespace {

// the per-tile acting workgroup lives in hx_act_body.h (hx_front.hip runs it beside learn()'s first launches)
template <int NRT, bool GAUSS, bool ENV, bool BF16, bool RELU, bool F32I = false, bool X3 = false>
__global__ __launch_bounds__(kWide) void act_fused_kernel(ActFusedArgs A) {
    __shared__ ActLds<NRT, ENV, BF16, F32I, X3> SL;
    act_fused_body<NRT, GAUSS, ENV, BF16, RELU, F32I, X3>(A, (int)blockIdx.x, SL);
}

// 16 rows per workgroup fill the chip up to 4,096 rows; from 8,192 rows on 32 rows per workgroup reuse every W2 chunk twice.
// The env tail on wave 0 pays while the launch is ONE round of workgroups (256 CUs x 16 or 32 rows: kFuseEnvMax, hx_act.h); beyond that the
// persistent kernel of hx_actp.hip takes the launch (its env tail runs on all 16 waves, once per workgroup).

// ENV launches carry HxStepOpts: with ev_start / ev_stop set the launch is stamped with the kernel's own begin / end (bench.py's live
// roofline of the act + env launch, as hx_env_step does for the env kernel)
template <typename K>
static void launch_act_k(K kernel, dim3 grid, const ActFusedArgs& H, hipStream_t st) {
    if (H.state && H.o.ev_start && H.o.ev_stop)
        hipExtLaunchKernelGGL(kernel, grid, dim3(kWide), 0, st, (hipEvent_t)H.o.ev_start, (hipEvent_t)H.o.ev_stop, 0, H);
    else
        hipLaunchKernelGGL(kernel, grid, dim3(kWide), 0, st, H);
}
template <bool GAUSS, bool BF16, bool RELU, bool F32I = false, bool X3 = false>
static void launch_act_t(const ActFusedArgs& H, hipStream_t st) {
    const bool env = H.state != nullptr;
    // 32 rows per workgroup: fp32 MFMA from 8,192 rows on (below that, 16-row workgroups fill the chip and the fp32 MFMA work per workgroup
    // is the longer pole); bf16 and the exact-split format [r5]: as soon as 16-row workgroups would need a second round (4,097 rows: there the launch is
    // bound by every workgroup pulling its images through L2, not by MFMA)
    static const int nrt2_bf16 = getenv("HX_ACT_BF16_NRT2_ROWS") ? atoi(getenv("HX_ACT_BF16_NRT2_ROWS")) : 4097;  // tuning knob; [r5] 4,097 (was 8,192): 5,120 rows 22.6 -> 14.3 us (profiles/r05_act_x9_tiling_time.txt)
    static const int nrt2_f32 = getenv("HX_ACT_F32_NRT2_ROWS") ? atoi(getenv("HX_ACT_F32_NRT2_ROWS")) : 8192;     // tuning knob (tools/ubench/merge_probe.sh)
    // tuning knob; default [r5]: 32-row workgroups as soon as 16-row ones would need a second round (4,097 .. 8,192 rows: 5,120 rows 30.5 -> 19.2 us, 8,192 rows
    // 31.7 -> 21.5 us act + env + insert, tools/ubench/act_x9_tiling_time.py, profiles/r05_act_x9_tiling_time.txt; fp32 MFMA from the fp32 image 35.0 / 27.9);
    // up to 4,096 rows 256 16-row workgroups fill the chip; beyond 8,192 the persistent kernel runs
    static const int nrt2_x9 = getenv("HX_ACT_X9_NRT2_ROWS") ? atoi(getenv("HX_ACT_X9_NRT2_ROWS")) : 4097;
    {
        if (H.rows >= (X3 ? nrt2_x9 : BF16 ? nrt2_bf16 : nrt2_f32)) {
            const dim3 grid((unsigned)((H.rows + 2 * RT - 1) / (2 * RT)));
            if (env) launch_act_k(act_fused_kernel<2, GAUSS, true, BF16, RELU, F32I, X3>, grid, H, st);
            else launch_act_k(act_fused_kernel<2, GAUSS, false, BF16, RELU, F32I, X3>, grid, H, st);
            return;
        }
    }
    {
        const dim3 grid((unsigned)((H.rows + RT - 1) / RT));
        if (env) launch_act_k(act_fused_kernel<1, GAUSS, true, BF16, RELU, F32I, X3>, grid, H, st);
        else launch_act_k(act_fused_kernel<1, GAUSS, false, BF16, RELU, F32I, X3>, grid, H, st);
    }
}
// HX_ACT_PERSIST=0 keeps act_fused_kernel at every size (A/B timing, bit-identity tests of the two kernels)
static bool persist_enabled() {
    const char* e = getenv("HX_ACT_PERSIST");
    return !(e && e[0] == '0');
}
template <bool GAUSS>
static void launch_act(const ActFusedArgs& H, hipStream_t st) {
    // beyond one round of workgroups: persistent workgroups that fetch W2 once (hx_actp.hip)
    if (H.rows > kFuseEnvMax && persist_enabled() && launch_act_persist(H, GAUSS, st)) return;
    // the activation is a compile-time ReLU when the slope is 0 (HIRL, SAC; the Gaussian policy is a Linear-ReLU stack by definition)
    if (!GAUSS && H.w2b && H.x9) {  // fp32 through the exact bf16 split (deterministic head only)
        if (H.slope == 0.0f) launch_act_t<false, false, true, false, true>(H, st);
        else launch_act_t<false, false, false, false, true>(H, st);
        return;
    }
    if (GAUSS || H.slope == 0.0f) {
        if (!GAUSS && H.w2b) launch_act_t<false, true, true>(H, st);  // (no entry point hands a Gaussian policy a plain bf16 image: its bf16-core format is the exact split above)
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

// the TRANSPOSED bf16 image (w2t_image_index): what bwd_l2's dh1 = dz2 W2 reads as its B operand
__global__ __launch_bounds__(kThreads) void pack_bf16_t_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int n) {
    const int i = blockIdx.x * kThreads + threadIdx.x;  // row-major element (row n_ = i / 256, k = i % 256): coalesced reads
    if (i < n) {
        const __bf16 v = (__bf16)src[i];
        dst[w2t_image_index((uint32_t)i % H1, (uint32_t)i / H1)] = __builtin_bit_cast(uint16_t, v);
    }
}

// the hi | mid | lo images of W2 (split3_bf16), each in the bf16 image order
__global__ __launch_bounds__(kThreads) void pack_x9_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int n) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) {
        uint16_t hi, mid, lo;
        split3_bf16(src[i], hi, mid, lo);
        const uint32_t ix = w2_image_index((uint32_t)i / H1, (uint32_t)i % H1);
        dst[ix] = hi; dst[kImgElems + ix] = mid; dst[2 * kImgElems + ix] = lo;
    }
}

__global__ __launch_bounds__(kThreads) void pack_f32i_kernel(const float* __restrict__ src, float* __restrict__ dst, int n) {
    const int i = (blockIdx.x * kThreads + threadIdx.x) * 4;  // four consecutive k of one column: adjacent in the image too
    if (i < n) *reinterpret_cast<float4*>(dst + w2f_image_index((uint32_t)i / H1, (uint32_t)i % H1)) = *reinterpret_cast<const float4*>(src + i);
}

}  // namespace

namespace hxu {
void launch_pack_bf16(const float* w2, uint16_t* image, bool transposed, hipStream_t st) {
    const int n = H2 * H1;
    if (transposed) hipLaunchKernelGGL(pack_bf16_t_kernel, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0, st, w2, image, n);
    else hipLaunchKernelGGL(pack_bf16_kernel, dim3((n / 2 + kThreads - 1) / kThreads), dim3(kThreads), 0, st, w2, image, n);
}
}  // namespace hxu

extern "C" {

int64_t hx_act_workspace_floats(int64_t rows) { (void)rows; return 0; }  // the acting kernels keep z2 in LDS: no workspace any more

/* chooseAction / chooseActionSmallNoise / chooseActionNoNoise for `rows` observations (HIRL.py:192-212):
 * actions = clamp(actor(obs) + noise, -1, 1).  noise_mode 0: none, 1: noise[4] shared by all rows, 2: noise[rows][4],
 * 3: N(0, sigma^2) per row and component from Philox(seed; row0 + row, call).  ws: unused since the whole policy runs in one kernel (may be NULL). */
static int actor_act_impl(const float* actor, const uint16_t* w2b, const float* w2f, const float* obs, int64_t rows, float* actions, int32_t noise_mode, const float* noise,
                 float sigma, uint64_t seed, uint32_t row0, uint32_t call, float slope, void* stream) {
    HX_REQUIRE(actor && obs && actions && rows > 0, "hx_actor_act: bad arguments");
    const Mlp mA{13, 4, (noise_mode & 16) ? 1 : 0};  // + 16: layerNorm = False
    const int x9 = (w2b && (noise_mode & 32)) ? 1 : 0;  // + 32 (internal, the *_x9 entry points): w2b is the first of the hi | mid | lo images
    noise_mode &= 15;
    HX_REQUIRE(noise_mode >= 0 && noise_mode <= 3 && (noise || (noise_mode != 1 && noise_mode != 2)), "hx_actor_act: bad noise mode");
    ActFusedArgs H{actor, mA, const_cast<float*>(obs), (int)rows, slope, actions, (noise_mode == 1 || noise_mode == 2) ? noise : nullptr,
                   noise_mode == 2, noise_mode == 3 ? sigma : 0.0f, 0, seed, row0, call, nullptr, 0, nullptr, nullptr, nullptr, HxStepOpts{}, 0.0, w2b, w2f, x9};
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

/* chooseAction + HarfangEnv.step for n envs in ONE launch (train_all.py:343-345): actions = clamp(actor(obs_io) + noise, -1, 1) as
 * hx_actor_act, then hx_env_step with those actions in the tail of the same kernel — obs_io in: current observation, out: next. */
static int actor_act_step_impl(const float* actor, const uint16_t* w2b, const float* w2f, float* state, int64_t n, int64_t stride, float* obs_io, float* actions, int32_t noise_mode,
                      const float* noise, float sigma, uint64_t seed, uint32_t row0, uint32_t call, float slope, float* reward,
                      uint8_t* done, int8_t* success, const HxStepOpts* opts, void* stream) {
    HX_REQUIRE(actor, "hx_actor_act_step: null actor");
    const int32_t mode_in = noise_mode;
    const Mlp mA{13, 4, (noise_mode & 16) ? 1 : 0};  // + 16: layerNorm = False
    const int x9 = (w2b && (noise_mode & 32)) ? 1 : 0;  // + 32 (internal): see actor_act_impl
    noise_mode &= 15;
    HX_REQUIRE(noise_mode >= 0 && noise_mode <= 3 && (noise || (noise_mode != 1 && noise_mode != 2)), "hx_actor_act_step: bad noise mode");
    const HxStepOpts o = opts ? *opts : HxStepOpts{};
    if (int rc = check_step_args(state, n, stride, obs_io, actions, reward, done, success, o, "hx_actor_act_step")) return rc;
    ActFusedArgs H{actor, mA, obs_io, (int)n, slope, actions, (noise_mode == 1 || noise_mode == 2) ? noise : nullptr,
                   noise_mode == 2, noise_mode == 3 ? sigma : 0.0f, 0, seed, row0, call, state, stride, reward, done, success, o,
                   o.cap > 0 ? 1.0 / (double)o.cap : 0.0, w2b, w2f, x9};
    if (n > kFuseEnvMax) {  // more than one round of 32-row workgroups: the persistent kernel (env tail on all waves), else two launches
        if (persist_enabled() && launch_act_persist(H, false, (hipStream_t)stream)) {
            HX_CHECK_LAUNCH("hx_actor_act_step");
            return 0;
        }
        if (int rc = actor_act_impl(actor, w2b, w2f, obs_io, n, actions, mode_in, noise, sigma, seed, row0, call, slope, stream)) return rc;
        return hx_env_step(state, n, stride, actions, obs_io, reward, done, success, opts, stream);
    }
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

/* The fp32 policy with the 256 -> 512 product through the exact three-way bf16 split of both operands on the bf16 matrix cores (see the header; hx_act.h HX_X9_TERMS). */
int hx_pack_w2_x9(const float* net, int32_t in_dim, uint16_t* w2_x9, void* stream) {
    HX_REQUIRE(net && w2_x9 && (in_dim == 13 || in_dim == 17) && (reinterpret_cast<uintptr_t>(w2_x9) & 15u) == 0, "hx_pack_w2_x9: bad arguments");
    const Mlp m{in_dim, 1, 0};
    const int n = H2 * H1;
    hipLaunchKernelGGL(pack_x9_kernel, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0, (hipStream_t)stream, net + m.W2(), w2_x9, n);
    HX_CHECK_LAUNCH("hx_pack_w2_x9");
    return 0;
}
int hx_actor_act_x9(const float* actor, const uint16_t* w2_x9, const float* obs, int64_t rows, float* actions, int32_t noise_mode,
                    const float* noise, float sigma, uint64_t seed, uint32_t row0, uint32_t call, float slope, void* stream) {
    HX_REQUIRE(w2_x9 && (reinterpret_cast<uintptr_t>(w2_x9) & 15u) == 0 && (noise_mode & 32) == 0, "hx_actor_act_x9: w2_x9 must be the 16-byte aligned hi | mid | lo images of W2");
    return actor_act_impl(actor, w2_x9, nullptr, obs, rows, actions, noise_mode | 32, noise, sigma, seed, row0, call, slope, stream);
}
int hx_actor_act_step_x9(const float* actor, const uint16_t* w2_x9, float* state, int64_t n, int64_t stride, float* obs_io, float* actions,
                         int32_t noise_mode, const float* noise, float sigma, uint64_t seed, uint32_t row0, uint32_t call, float slope,
                         float* reward, uint8_t* done, int8_t* success, const HxStepOpts* opts, void* stream) {
    HX_REQUIRE(w2_x9 && (reinterpret_cast<uintptr_t>(w2_x9) & 15u) == 0 && (noise_mode & 32) == 0, "hx_actor_act_step_x9: w2_x9 must be the 16-byte aligned hi | mid | lo images of W2");
    return actor_act_step_impl(actor, w2_x9, nullptr, state, n, stride, obs_io, actions, noise_mode | 32, noise, sigma, seed, row0, call, slope, reward, done,
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

/* SacAgent.explore / exploit (SAC/agent.py:183-196) for `rows` observations.  mode 0: exploit = tanh(mean); 1: sample with the
 * standard-normal draws eps[rows][4]; 2: sample with Philox(seed; row0 + row, call).  ws: unused since the whole policy runs in one kernel (may be NULL). */
static int sac_act_impl(const float* policy, const float* w2f, const float* obs, int64_t rows, float* actions, int32_t mode, const float* eps,
                        uint64_t seed, uint32_t row0, uint32_t call, void* stream, const uint16_t* w2x = nullptr) {
    HX_REQUIRE(policy && obs && actions && rows > 0 && mode >= 0 && mode <= 2 && (mode != 1 || eps), "hx_sac_act: bad arguments");
    const bool x9 = w2x && rows > kFuseEnvMax && persist_enabled();  // the exact split exists for the Gaussian head in the persistent kernel only
    ActFusedArgs H{policy, kPolicy, const_cast<float*>(obs), (int)rows, 0.0f, actions, mode == 1 ? eps : nullptr, 1, 0.0f, mode, seed, row0, call,
                   nullptr, 0, nullptr, nullptr, nullptr, HxStepOpts{}, 0.0, x9 ? w2x : nullptr, x9 ? nullptr : w2f, x9 ? 1 : 0};
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
                             const HxStepOpts* opts, void* stream, const uint16_t* w2x = nullptr) {
    HX_REQUIRE(policy && mode >= 0 && mode <= 2 && (mode != 1 || eps), "hx_sac_act_step: bad arguments");
    const HxStepOpts o = opts ? *opts : HxStepOpts{};
    if (int rc = check_step_args(state, n, stride, obs_io, actions, reward, done, success, o, "hx_sac_act_step")) return rc;
    const bool x9 = w2x && n > kFuseEnvMax && persist_enabled();
    ActFusedArgs H{policy, kPolicy, obs_io, (int)n, 0.0f, actions, mode == 1 ? eps : nullptr, 1, 0.0f, mode, seed, row0, call,
                   state, stride, reward, done, success, o, o.cap > 0 ? 1.0 / (double)o.cap : 0.0, x9 ? w2x : nullptr, x9 ? nullptr : w2f, x9 ? 1 : 0};
    if (n > kFuseEnvMax) {
        if (persist_enabled() && launch_act_persist(H, true, (hipStream_t)stream)) {
            HX_CHECK_LAUNCH("hx_sac_act_step");
            return 0;
        }
        if (int rc = sac_act_impl(policy, w2f, obs_io, n, actions, mode, eps, seed, row0, call, stream, w2x)) return rc;
        return hx_env_step(state, n, stride, actions, obs_io, reward, done, success, opts, stream);
    }
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

int hx_sac_act_x9(const float* policy, const uint16_t* w2_x9, const float* w2_f32i, const float* obs, int64_t rows, float* actions, int32_t mode,
                  const float* eps, uint64_t seed, uint32_t row0, uint32_t call, void* stream) {
    HX_REQUIRE(w2_x9 && w2_f32i && (reinterpret_cast<uintptr_t>(w2_x9) & 15u) == 0 && (reinterpret_cast<uintptr_t>(w2_f32i) & 15u) == 0,
               "hx_sac_act_x9: w2_x9 (hi | mid | lo images) and w2_f32i (the fallback up to 8,192 rows) must be 16-byte aligned images of W2");
    return sac_act_impl(policy, w2_f32i, obs, rows, actions, mode, eps, seed, row0, call, stream, w2_x9);
}
int hx_sac_act_step_x9(const float* policy, const uint16_t* w2_x9, const float* w2_f32i, float* state, int64_t n, int64_t stride, float* obs_io,
                       float* actions, int32_t mode, const float* eps, uint64_t seed, uint32_t row0, uint32_t call, float* reward, uint8_t* done,
                       int8_t* success, const HxStepOpts* opts, void* stream) {
    HX_REQUIRE(w2_x9 && w2_f32i && (reinterpret_cast<uintptr_t>(w2_x9) & 15u) == 0 && (reinterpret_cast<uintptr_t>(w2_f32i) & 15u) == 0,
               "hx_sac_act_step_x9: w2_x9 (hi | mid | lo images) and w2_f32i (the fallback up to 8,192 rows) must be 16-byte aligned images of W2");
    return sac_act_step_impl(policy, w2_f32i, state, n, stride, obs_io, actions, mode, eps, seed, row0, call, reward, done, success, opts, stream, w2_x9);
}

}  // extern "C"

HX_DEFINE_DEBUG_COLLECTORS(act, 56, 80)
