// hx_env.hip — batched pursuit-lock-launch env step for gfx950 (MI355X).
//
// One thread per env, struct-of-arrays fp32 state in HBM (word w of env i at state[w*stride + i]: every load and
// store of a wave is one contiguous 256-B segment), observation / replay-row tiles staged through LDS so that the
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
// Numerics: this file is compiled with -ffp-contract=off; state-evolving arithmetic uses only + - * / sqrt in a
// fixed order, so masks are reproducible bit for bit.  asinf/atan2f/acosf appear only in the observation.
#include "hx_common.h"
#include <hip/hip_ext.h>

#include "hx_env_dev.h"

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

using namespace hxenv;

struct StepArgs {
    float* state;
    int64_t n, stride;
    const float* actions;
    float* obs_io;
    float* reward;
    uint8_t* done;
    int8_t* success;
    HxStepOpts o;
};

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }

// Cooperative, fully coalesced copy of `count` floats between a row-major global tile and LDS.
__device__ __forceinline__ void tile_load(float* __restrict__ lds, const float* __restrict__ g, int count) {
    for (int e = threadIdx.x; e < count; e += kBlock) lds[e] = g[e];
}
__device__ __forceinline__ void tile_store(float* __restrict__ g, const float* __restrict__ lds, int count) {
    for (int e = threadIdx.x; e < count; e += kBlock) g[e] = lds[e];
}

template <bool INSERT>
__global__ __launch_bounds__(kBlock) void env_step_kernel(StepArgs A) {
    // one LDS object: [obs tile 256*13][row tile 256*33 (INSERT)] + small scratch
    constexpr int kObsTile = kBlock * HX_OBS_DIM;
    constexpr int kRowPitch = HX_ROW_WORDS + 1;  // +1: conflict-free per-lane row writes
    __shared__ float lds[kObsTile + (INSERT ? kBlock * kRowPitch : 0)];
    __shared__ unsigned long long s_base;
    __shared__ int s_wcount[kWaves];
    __shared__ unsigned s_stat[kWaves][HX_STAT_COUNT];

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * kBlock;
    const int64_t i = i0 + tid;
    const int nblk = (int)((A.n - i0) < kBlock ? (A.n - i0) : kBlock);
    const bool active = tid < nblk;
    float* s_obs = lds;
    float* s_row = lds + kObsTile;

    if (INSERT) tile_load(s_obs, A.obs_io + i0 * HX_OBS_DIM, nblk * HX_OBS_DIM);

    Env E;
    float4 act = {0.f, 0.f, 0.f, 0.f};
    bool trunc = false, store = false;
    if (active) {
        load_env(E, A.state, A.stride, i0, (uint32_t)tid);
        act = reinterpret_cast<const float4*>(A.actions)[i];
        uint32_t ep = E.counters & 0xFFFFu;
        ep = ep < 65535u ? ep + 1u : ep;
        trunc = A.o.max_step > 0 && (int)ep >= A.o.max_step;  // train_all.py:346-347
        store = INSERT && !trunc;
    }
    // ring slots: ballot -> per-wave rank -> one atomic per workgroup (issued before the arithmetic)
    int rank = 0, nstore = 0;
    if (INSERT) {
        const unsigned long long b = __ballot(store);
        rank = __popcll(b & ((1ull << lane_id()) - 1ull));
        if (lane_id() == 0) s_wcount[wave] = __popcll(b);
    }
    __syncthreads();
    if (INSERT) {
        int before = 0;
        for (int w = 0; w < kWaves; ++w) {
            before += (w < wave) ? s_wcount[w] : 0;
            nstore += s_wcount[w];
        }
        rank += before;
        if (tid == 0 && nstore > 0) s_base = atomicAdd((unsigned long long*)A.o.total, (unsigned long long)nstore);
    }

    float reward = 0.0f;
    int success = 0;
    bool done = false, ended = false;
    Observed O;
    unsigned st_kill = 0, st_fs = 0, st_tl = 0, st_fire = 0, st_good = 0, st_lock = 0;
    if (active) {
        const bool fire = act.w > 0.0f;  // float(action[3] > 0)  HarfangEnv_GYM.py:150
        sim_step(E, act.x, act.y, act.z, fire);
        wrap_step(E, O, reward, success);
        uint32_t ep = E.counters & 0xFFFFu;
        ep = ep < 65535u ? ep + 1u : ep;
        E.counters = (E.counters & 0xFFFF0000u) | ep;
        done = (E.flags & HX_F_DONE) != 0u;
        ended = A.o.auto_reset && (done || trunc);
        st_fire = (E.flags & HX_F_FIRED) ? 1u : 0u;
        st_good = success == 1 ? 1u : 0u;
        st_lock = (E.flags & HX_F_LOCKED) ? 1u : 0u;
        st_kill = (ended && (E.flags & HX_F_EPISODE_SUCCESS)) ? 1u : 0u;
        st_fs = (ended && (E.flags & HX_F_FIRE_SUCCESS)) ? 1u : 0u;
        st_tl = (ended && !done) ? 1u : 0u;
        A.reward[i] = reward;
        A.done[i] = done ? 1 : 0;
        A.success[i] = (int8_t)success;
    }
    if (INSERT) {
        if (store) {
            // row = s[13] a[4] s'[13] r done   (Transition, buffer.py:8; sample() drops step_success :48)
            float* row = s_row + rank * kRowPitch;
            const float* prev = s_obs + tid * HX_OBS_DIM;
#pragma unroll
            for (int j = 0; j < HX_OBS_DIM; ++j) row[j] = prev[j];
            row[13] = act.x;
            row[14] = act.y;
            row[15] = act.z;
            row[16] = act.w;
#pragma unroll
            for (int j = 0; j < HX_OBS_DIM; ++j) row[17 + j] = O.obs[j];
            row[30] = reward;
            row[31] = done ? 1.0f : 0.0f;
        }
    }
    __syncthreads();  // every lane has consumed its previous observation; rows complete; s_base visible
    if (INSERT && store && A.o.ring_success) {
        A.o.ring_success[(s_base + (unsigned long long)rank) % (unsigned long long)A.o.cap] = (int8_t)success;
    }
    if (active) {
        if (ended) {
            const uint32_t scen = (E.flags >> HX_F_SCEN_SHIFT) & 3u;
            const uint32_t epi = A.o.episode_ctr[i] + 1u;
            A.o.episode_ctr[i] = epi;
            env_reset(E, scen, A.o.randomize != 0, A.o.seed, A.o.env_id0 + (uint32_t)i, epi);
            observe(E, O);
        }
        store_env(E, A.state, A.stride, i0, (uint32_t)tid);
        float* out = s_obs + tid * HX_OBS_DIM;
#pragma unroll
        for (int j = 0; j < HX_OBS_DIM; ++j) out[j] = O.obs[j];
    }
    if (A.o.stats) {
        const unsigned vals[HX_STAT_COUNT] = {ended ? 1u : 0u, st_kill, st_fs, st_tl, st_fire, st_good, st_lock, active ? 1u : 0u};
#pragma unroll
        for (int k = 0; k < HX_STAT_COUNT; ++k) {
            const unsigned c = (unsigned)__popcll(__ballot(vals[k] != 0u));
            if (lane_id() == 0) s_stat[wave][k] = c;
        }
    }
    __syncthreads();
    tile_store(A.obs_io + i0 * HX_OBS_DIM, s_obs, nblk * HX_OBS_DIM);
    if (INSERT && nstore > 0) {
        // 16 B per lane, 1 KiB per wave-instruction, rows contiguous in the ring (modulo wrap)
        const unsigned long long base = s_base, cap = (unsigned long long)A.o.cap;
        float4* ring4 = reinterpret_cast<float4*>(A.o.ring);
        for (int e = tid; e < nstore * (HX_ROW_WORDS / 4); e += kBlock) {
            const int r = e >> 3, c = (e & 7) * 4;
            const float* src = s_row + r * kRowPitch + c;
            const float4 v = {src[0], src[1], src[2], src[3]};
            ring4[((base + (unsigned long long)r) % cap) * (HX_ROW_WORDS / 4) + (e & 7)] = v;
        }
    }
    if (A.o.stats && tid < HX_STAT_COUNT) {
        unsigned c = 0;
        for (int w = 0; w < kWaves; ++w) c += s_stat[w][tid];
        if (c) atomicAdd((unsigned long long*)&A.o.stats[tid], (unsigned long long)c);
    }
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
        Observed O;
        observe(E, O);
        for (int j = 0; j < HX_OBS_DIM; ++j) A.obs[i * HX_OBS_DIM + j] = O.obs[j];
    }
}

__global__ __launch_bounds__(kBlock) void env_rearm_kernel(float* state, int64_t n, int64_t stride, const uint8_t* mask) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n || (mask && !mask[i])) return;
    const uint32_t f = __float_as_uint(state[35 * stride + i]) | HX_F_SIM_SLOT;
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
    Observed O;
    observe(E, O);
    float* o = out + i * 16;
    o[0] = E.ally.p.x; o[1] = E.ally.p.y; o[2] = E.ally.p.z;
    float p, h, r;
    euler_of(E.ally, p, h, r);
    o[3] = p; o[4] = h; o[5] = r;
    o[6] = E.opp.p.x; o[7] = E.opp.p.y; o[8] = E.opp.p.z;
    euler_of(E.opp, p, h, r);
    o[9] = p; o[10] = h; o[11] = r;
    o[12] = O.target_angle * 180.0f;
    o[13] = E.health;
    o[14] = E.lock_timer >= kLockDelay ? 1.0f : 0.0f;
    o[15] = (E.flags & HX_F_SIM_SLOT) ? 1.0f : 0.0f;
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
    StepArgs A{state, n, stride, actions, obs_io, reward, done, success, HxStepOpts{}};
    if (opts) A.o = *opts;
    HX_REQUIRE(!A.o.auto_reset || A.o.episode_ctr, "hx_env_step: auto_reset needs episode_ctr");
    if (A.o.ring) {
        HX_REQUIRE(A.o.cap > 0 && A.o.total, "hx_env_step: ring needs cap and total");
        HX_REQUIRE((reinterpret_cast<uintptr_t>(A.o.ring) & 15u) == 0, "hx_env_step: ring must be 16-byte aligned");
        if (A.o.ev_start && A.o.ev_stop)
            hipExtLaunchKernelGGL(env_step_kernel<true>, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream,
                                  (hipEvent_t)A.o.ev_start, (hipEvent_t)A.o.ev_stop, 0, A);
        else
            hipLaunchKernelGGL(env_step_kernel<true>, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream, A);
    } else {
        if (A.o.ev_start && A.o.ev_stop)
            hipExtLaunchKernelGGL(env_step_kernel<false>, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream,
                                  (hipEvent_t)A.o.ev_start, (hipEvent_t)A.o.ev_stop, 0, A);
        else
            hipLaunchKernelGGL(env_step_kernel<false>, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream, A);
    }
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

int hx_label_transitions(const float* s, const float* a, const float* ns, int64_t n, float* reward, int8_t* success,
                         uint8_t* done, void* stream) {
    HX_REQUIRE(s && a && ns && reward && success && done && n > 0, "hx_label_transitions: bad arguments");
    hipLaunchKernelGGL(label_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream, s, a, ns, n, reward, success, done);
    HX_CHECK_LAUNCH("hx_label_transitions");
    return 0;
}

}  // extern "C"
