// hx_env.hip — batched pursuit-lock-launch env step for gfx950 (MI355X).
//
// Two adjacent lanes per env (ally + shared words | opponent; hx_env_dev.h "Pair") or one ("Solo"), struct-of-arrays fp32 state in HBM (word
// w of env i at state[w*stride + i]: every load and store of a wave is contiguous), observation / replay-row tiles staged through LDS so that the
// row-major outputs leave the CU as full-line 16-B-per-lane stores, wave ballots for the store mask and the
// episode statistics, one ring-head atomic per workgroup.
//
// What it replaces (reference file:line):
//   HarfangEnv.step          hirl/environments/HarfangEnv_GYM.py:83-90   (E9)
//   _apply_action            :139-158 (E4)      scripted opponents :342-353, :412-421 (E10, E11)
//   UPDATE_SCENE tick        external Harfang simulator (E5) — re-derived model, docs/DYNAMICS.md
//   _get_observation         :193-268 (E6)      _get_reward :101-137 (E7)      _get_termination :160-169 (E8)
//   reset / random_reset     :34-81, :171-188, :374-406, :440-474 (E2, E3)
//   UniformMemory.store      hirl/utils/buffer.py:20-36 (U6), fused; episode rules train_all.py:341-361 (D1)
//   get_reward/get_termination for expert labelling :299-336 (E13)
//
// Numerics: this file is compiled with -ffp-contract=off; every operation of the model (docs/DYNAMICS.md, model v2) is spelled out — explicit
// fmaf, + - * / sqrt in a fixed order, the model's own asin / acos / atan2 polynomials — so state words, masks, observations and rewards are
// reproducible bit for bit against the scalar oracle.
#include "hx_common.h"
#include <hip/hip_ext.h>

#include "hx_env_block.h"

namespace {

constexpr int kBlock = 256;  // reset / rearm / label / sim_* kernels

using namespace hxenv;

// HarfangEnv.step for EPB envs per workgroup (hx_env_block.h).  PAIR: two adjacent lanes per env (hx_env_dev.h), EPB * 2 threads; else one
// lane per env.
template <bool PAIR, bool INSERT, int EPB>
__global__ __launch_bounds__(EPB*(PAIR ? 2 : 1)) void env_step_kernel(StepArgs A) {
    __shared__ float lds[kEnvBlockLds<INSERT, EPB>];
    __shared__ unsigned s_slot0;  // ring slot of the workgroup's first row
    __shared__ int s_wcount[EPB * (PAIR ? 2 : 1) / 64];
    env_block_step<PAIR, INSERT, EPB>(A, (int64_t)blockIdx.x * EPB, A.n, lds, s_slot0, s_wcount, blockIdx.x);
}

// Launch shape by size (measured on MI355X, profiles/r02a_env_sweep_layouts.jsonl, profiles/r03b_env_sweep_layouts.jsonl).  While a launch is
// latency-bound (< 32,768 envs): two lanes per env (half the dependent chain, ~66 VGPRs instead of ~98) and envs per workgroup so that it has
// about 64..256 workgroups (4,096 envs: 6.4 us with 64 envs per workgroup, 7.6 with 256).  From 32,768 envs on the chip is full and the launch
// is bound by memory requests and by the ring-head atomic — ONE per workgroup, all on one address, ~12 ns each: 256 envs per workgroup — and
// ONE lane per env issues half the wave-instructions per byte: solo 256 beats pair 256 at 32,768 (8.8 vs 9.4 us), 65,536 (11.0 vs 11.6),
// 1M (100.7 vs 108.8) and 4M (378-396 vs 387-391); only 262,144 measured the other way (32.2 vs 29.7).  Up to 256 envs ONE workgroup steps them
// all, which keeps the replay insert order — and with it a whole training run — reproducible bit for bit.
struct Layout {
    bool pair;
    int epb;
};
inline Layout pick_layout(int64_t n, int32_t forced) {
    if (forced) return Layout{(forced >> 8) != 0, (forced & 0xFF) * 4};
    if (n <= 256) return Layout{true, 256};
    if (n <= 8192) return Layout{true, 64};
    if (n < 32768) return Layout{true, 128};
    return Layout{false, 256};
}

template <bool PAIR, bool INSERT, int EPB>
void launch_step_shape(const StepArgs& A, hipStream_t st) {
    const dim3 grid((unsigned)((A.n + EPB - 1) / EPB)), block(EPB * (PAIR ? 2 : 1));
    if (A.o.ev_start && A.o.ev_stop)
        hipExtLaunchKernelGGL((env_step_kernel<PAIR, INSERT, EPB>), grid, block, 0, st, (hipEvent_t)A.o.ev_start, (hipEvent_t)A.o.ev_stop, 0, A);
    else
        hipLaunchKernelGGL((env_step_kernel<PAIR, INSERT, EPB>), grid, block, 0, st, A);
}
template <bool INSERT>
int launch_step(const StepArgs& A, hipStream_t st) {
    const Layout L = pick_layout(A.n, A.o.layout);
    if (L.pair) {
        switch (L.epb) {
            case 32: launch_step_shape<true, INSERT, 32>(A, st); return 0;
            case 64: launch_step_shape<true, INSERT, 64>(A, st); return 0;
            case 128: launch_step_shape<true, INSERT, 128>(A, st); return 0;
            case 256: launch_step_shape<true, INSERT, 256>(A, st); return 0;
            case 512: launch_step_shape<true, INSERT, 512>(A, st); return 0;
        }
    } else {
        switch (L.epb) {
            case 64: launch_step_shape<false, INSERT, 64>(A, st); return 0;
            case 128: launch_step_shape<false, INSERT, 128>(A, st); return 0;
            case 256: launch_step_shape<false, INSERT, 256>(A, st); return 0;
        }
    }
    return -1;
}

struct ResetArgs {
    float* state;
    int64_t n, stride;
    const uint8_t* mask;
    const int32_t* scenario;
    int32_t scenario_all, randomize;
    uint64_t seed;
    uint32_t env_id0;
    uint32_t* episode_ctr;
    float* obs;
};

__global__ __launch_bounds__(kBlock) void env_reset_kernel(ResetArgs A) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= A.n) return;
    if (A.mask && !A.mask[i]) return;
    Env E;
    const uint32_t scen = (uint32_t)(A.scenario ? A.scenario[i] : A.scenario_all);
    const uint32_t epi = A.episode_ctr ? A.episode_ctr[i] : 0u;
    env_reset(E, scen, A.randomize != 0, A.seed, A.env_id0 + (uint32_t)i, epi);
    store_env(E, A.state, A.stride, (int64_t)blockIdx.x * kBlock, threadIdx.x);
    if (A.obs) {
        float o[HX_OBS_DIM];
        observe(E, o);
        for (int j = 0; j < HX_OBS_DIM; ++j) A.obs[i * HX_OBS_DIM + j] = o[j];
    }
}

__global__ __launch_bounds__(kBlock) void env_rearm_kernel(float* state, int64_t n, int64_t stride, const uint8_t* mask) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n || (mask && !mask[i])) return;
    const uint32_t f = __float_as_uint(state[35 * stride + i]) | HX_F_SIM_SLOT;  // df.rearm_machine
    state[35 * stride + i] = __uint_as_float(f);
}

// ---- simulator-level access (the wire-protocol server of environments/wire.py): what the external simulator does between the
// client's SET_PLANE_* / FIRE_MISSILE calls and its GET_* read-backs (dogfight_client.py), without the wrapper's latches ----
__global__ __launch_bounds__(kBlock) void sim_tick_kernel(float* state, int64_t n, int64_t stride, const float* __restrict__ ally_cmd,
                                                          const float* __restrict__ opp_cmd, const uint8_t* __restrict__ fire) {
    const int64_t i0 = (int64_t)blockIdx.x * kBlock;
    const int64_t i = i0 + threadIdx.x;
    if (i >= n) return;
    Env E;
    load_env(E, state, stride, i0, threadIdx.x);
    sim_core(E, ally_cmd[i * 3], ally_cmd[i * 3 + 1], ally_cmd[i * 3 + 2], opp_cmd[i * 3], opp_cmd[i * 3 + 1], opp_cmd[i * 3 + 2], fire[i] != 0);
    store_env(E, state, stride, i0, threadIdx.x);
}

// out[i][16]: ally position 3, ally Euler (pitch, heading, roll) 3, opponent position 3, opponent Euler 3, target angle in degrees,
// opponent health, target_locked (0/1), missile slot 0 loaded (0/1)
__global__ __launch_bounds__(kBlock) void sim_readback_kernel(const float* state, int64_t n, int64_t stride, float* __restrict__ out) {
    const int64_t i0 = (int64_t)blockIdx.x * kBlock;
    const int64_t i = i0 + threadIdx.x;
    if (i >= n) return;
    Env E;
    load_env(E, state, stride, i0, threadIdx.x);
    const Axes A = quat_axes(E.ally.qw, E.ally.qx, E.ally.qy, E.ally.qz);
    const Axes B = quat_axes(E.opp.qw, E.opp.qx, E.opp.qy, E.opp.qz);
    const V3 ea = euler_rad(A), eo = euler_rad(B);  // radians, as GET_PLANE_STATE reports them
    float* o = out + i * 16;
    o[0] = E.ally.p.x; o[1] = E.ally.p.y; o[2] = E.ally.p.z;
    o[3] = ea.x; o[4] = ea.y; o[5] = ea.z;
    o[6] = E.opp.p.x; o[7] = E.opp.p.y; o[8] = E.opp.p.z;
    o[9] = eo.x; o[10] = eo.y; o[11] = eo.z;
    o[12] = target_angle_deg(geometry(E.ally.p, A.Z, E.opp.p).cosang);
    o[13] = E.s.health;
    o[14] = E.s.lock_timer >= kLockDelay ? 1.0f : 0.0f;
    o[15] = (E.s.flags & HX_F_SIM_SLOT) ? 1.0f : 0.0f;
}

// get_reward / get_termination  HarfangEnv_GYM.py:299-336
__global__ __launch_bounds__(kBlock) void label_kernel(const float* __restrict__ s, const float* __restrict__ a,
                                                       const float* __restrict__ ns, int64_t n, float* reward,
                                                       int8_t* success, uint8_t* done) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float* S = s + i * HX_OBS_DIM;
    const float* N = ns + i * HX_OBS_DIM;
    const float x = N[0] * 10000.0f, y = N[1] * 10000.0f, z = N[2] * 10000.0f;
    const float loc = sqrtf((x * x + y * y) + z * z);
    float r = 0.0f;
    int sc = 0;
    r = r - 0.0001f * loc;
    r = r - N[6] * 10.0f;
    if (a[i * HX_ACT_DIM + 3] > 0.0f) {
        r = r - 8.0f;
        if (S[8] > 0.0f && S[7] < 0.0f) sc = -1;
        else if (S[8] > 0.0f && S[7] > 0.0f) sc = 1;
    }
    // float64 comparisons against 0.1 on fp32 data: `< 0.1` and `<= 0.1` both equal `h < 0.1f`
    const bool low = N[12] < 0.1f;
    if (low) r = r + 600.0f;
    reward[i] = r;
    success[i] = (int8_t)sc;
    done[i] = low ? 1 : 0;
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

// Everything the N = 1 facade reads back after a step, packed into ONE 64-float row (one D2H copy instead of five): words 0..36 the state
// words of env i, 37..49 its observation, 50 reward, 51 done, 52 success.
__global__ void env_pack_row_kernel(const float* __restrict__ state, int64_t stride, int64_t i, const float* __restrict__ obs,
                                    const float* __restrict__ reward, const uint8_t* __restrict__ done, const int8_t* __restrict__ success,
                                    float* __restrict__ out) {
    const int t = threadIdx.x;
    float v = 0.0f;
    if (t < HX_ENV_WORDS) v = state[(int64_t)t * stride + i];
    else if (t < HX_ENV_WORDS + HX_OBS_DIM) v = obs[i * HX_OBS_DIM + (t - HX_ENV_WORDS)];
    else if (t == 50) v = reward[i];
    else if (t == 51) v = (float)done[i];
    else if (t == 52) v = (float)success[i];
    out[t] = v;
}

}  // namespace

extern "C" {

int hx_env_reset(float* state, int64_t n, int64_t stride, const uint8_t* mask, const int32_t* scenario,
                 int32_t scenario_all, int32_t randomize, uint64_t seed, uint32_t env_id0, uint32_t* episode_ctr,
                 float* obs, void* stream) {
    HX_REQUIRE(state && n > 0 && stride >= n, "hx_env_reset: bad state/n/stride");
    HX_REQUIRE(scenario || (scenario_all >= 0 && scenario_all <= 2), "hx_env_reset: scenario must be 0..2");
    ResetArgs A{state, n, stride, mask, scenario, scenario_all, randomize, seed, env_id0, episode_ctr, obs};
    hipLaunchKernelGGL(env_reset_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream, A);
    HX_CHECK_LAUNCH("hx_env_reset");
    return 0;
}

int hx_env_step(float* state, int64_t n, int64_t stride, const float* actions, float* obs_io, float* reward,
                uint8_t* done, int8_t* success, const HxStepOpts* opts, void* stream) {
    HX_REQUIRE(state && actions && obs_io && reward && done && success, "hx_env_step: null buffer");
    HX_REQUIRE(n > 0 && stride >= n, "hx_env_step: bad n/stride");
    HX_REQUIRE((reinterpret_cast<uintptr_t>(actions) & 15u) == 0, "hx_env_step: actions must be 16-byte aligned");
    StepArgs A{state, n, stride, actions, obs_io, reward, done, success, HxStepOpts{}, 0.0};
    if (opts) A.o = *opts;
    HX_REQUIRE(stride < ((int64_t)1 << 25), "hx_env_step: stride must be below 2^25 envs (32-bit byte offsets inside a launch)");
    HX_REQUIRE(!A.o.auto_reset || A.o.episode_ctr, "hx_env_step: auto_reset needs episode_ctr");
    int rc;
    if (A.o.ring) {
        HX_REQUIRE(A.o.cap >= 512 && A.o.cap < ((int64_t)1 << 31) && A.o.total, "hx_env_step: ring needs 512 <= cap < 2^31 and total");
        A.inv_cap = 1.0 / (double)A.o.cap;
        HX_REQUIRE((reinterpret_cast<uintptr_t>(A.o.ring) & 15u) == 0, "hx_env_step: ring must be 16-byte aligned");
        rc = launch_step<true>(A, (hipStream_t)stream);
    } else {
        rc = launch_step<false>(A, (hipStream_t)stream);
    }
    HX_REQUIRE(rc == 0, "hx_env_step: unknown layout %d", A.o.layout);
    HX_CHECK_LAUNCH("hx_env_step");
    return 0;
}

int hx_env_rearm(float* state, int64_t n, int64_t stride, const uint8_t* mask, void* stream) {
    HX_REQUIRE(state && n > 0 && stride >= n, "hx_env_rearm: bad state/n/stride");
    hipLaunchKernelGGL(env_rearm_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream, state, n, stride, mask);
    HX_CHECK_LAUNCH("hx_env_rearm");
    return 0;
}

int hx_sim_tick(float* state, int64_t n, int64_t stride, const float* ally_cmd, const float* opp_cmd, const uint8_t* fire, void* stream) {
    HX_REQUIRE(state && ally_cmd && opp_cmd && fire && n > 0 && stride >= n, "hx_sim_tick: bad arguments");
    hipLaunchKernelGGL(sim_tick_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream, state, n, stride, ally_cmd, opp_cmd, fire);
    HX_CHECK_LAUNCH("hx_sim_tick");
    return 0;
}

int hx_sim_readback(const float* state, int64_t n, int64_t stride, float* out, void* stream) {
    HX_REQUIRE(state && out && n > 0 && stride >= n, "hx_sim_readback: bad arguments");
    hipLaunchKernelGGL(sim_readback_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream, state, n, stride, out);
    HX_CHECK_LAUNCH("hx_sim_readback");
    return 0;
}

int hx_env_pack_row(const float* state, int64_t n, int64_t stride, int64_t i, const float* obs, const float* reward, const uint8_t* done,
                    const int8_t* success, float* out, void* stream) {
    HX_REQUIRE(state && obs && reward && done && success && out && n > 0 && stride >= n && i >= 0 && i < n, "hx_env_pack_row: bad arguments");
    hipLaunchKernelGGL(env_pack_row_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, stride, i, obs, reward, done, success, out);
    HX_CHECK_LAUNCH("hx_env_pack_row");
    return 0;
}

int hx_label_transitions(const float* s, const float* a, const float* ns, int64_t n, float* reward, int8_t* success,
                         uint8_t* done, void* stream) {
    HX_REQUIRE(s && a && ns && reward && success && done && n > 0, "hx_label_transitions: bad arguments");
    hipLaunchKernelGGL(label_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream, s, a, ns, n, reward, success, done);
    HX_CHECK_LAUNCH("hx_label_transitions");
    return 0;
}

}  // extern "C"
