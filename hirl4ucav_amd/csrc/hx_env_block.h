// hx_env_block.h — HarfangEnv.step for one block of envs, as a device function (gfx950): the body of env_step_kernel, shared with the tail of
// the persistent acting kernel (hx_actp.hip).  What it replaces: hx_env.hip's header (HarfangEnv_GYM.py:83-169,193-268; buffer.py:20-36).
#pragma once
#include "hx_common.h"
#include "hx_env_dev.h"

#pragma clang fp contract(off)
namespace hxenv {

struct StepArgs {
    float* state;
    int64_t n, stride;
    const float* actions;
    float* obs_io;
    float* reward;
    uint8_t* done;
    int8_t* success;
    HxStepOpts o;
    double inv_cap;  // 1 / o.cap
};

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }

// Copy `count` (<= FULL) floats between a row-major global tile and LDS, 16 B per lane with a fixed trip count (every load is
// requested before the first store needs it); the last `count % 4` floats of a ragged final tile go as dwords.
// Both ends are 16-B aligned: tiles start at multiples of 32 envs x 13 floats.
template <int THREADS, int FULL>
__device__ __forceinline__ void tile_copy(float* __restrict__ dst, const float* __restrict__ src, int count, int tid) {
    constexpr int TRIPS = (FULL / 4 + THREADS - 1) / THREADS;
    const int n4 = count >> 2;
#pragma unroll
    for (int it = 0; it < TRIPS; ++it) {
        const int k = tid + it * THREADS;
        if (k < n4) reinterpret_cast<float4*>(dst)[k] = reinterpret_cast<const float4*>(src)[k];
    }
    if (tid < (count & 3)) dst[4 * n4 + tid] = src[4 * n4 + tid];
}
typedef float env_v4f __attribute__((ext_vector_type(4)));
// the same towards global memory with NON-TEMPORAL stores: replay rows and observations are written once and read a step or more later
// (sweep on MI355X, one box, solo 256: 65,536 envs 11.16 -> 11.0 us, 1M envs 105.4 -> 100.7 us; 4M unchanged; -DHX_ENV_NT=0 builds the plain stores,
// =2 adds the state words: no further gain)
template <int THREADS, int FULL>
__device__ __forceinline__ void tile_copy_out(float* __restrict__ dst, const float* __restrict__ src, int count, int tid) {
#if !defined(HX_ENV_NT) || HX_ENV_NT >= 1
    constexpr int TRIPS = (FULL / 4 + THREADS - 1) / THREADS;
    const int n4 = count >> 2;
#pragma unroll
    for (int it = 0; it < TRIPS; ++it) {
        const int k = tid + it * THREADS;
        if (k < n4) __builtin_nontemporal_store(reinterpret_cast<const env_v4f*>(src)[k], reinterpret_cast<env_v4f*>(dst) + k);
    }
    if (tid < (count & 3)) dst[4 * n4 + tid] = src[4 * n4 + tid];
#else
    tile_copy<THREADS, FULL>(dst, src, count, tid);
#endif
}

// HarfangEnv.step for EPB envs per workgroup.  PAIR: two adjacent lanes per env (hx_env_dev.h), EPB * 2 threads; else one lane per
// env.  The row-major observation tile (and with INSERT the replay rows) meet in LDS so that every global access of the
// workgroup is a contiguous run: SoA state words 256 B (128 B x 2 with PAIR) per wave-instruction, tiles 16 B per lane.
// Shared by env_step_kernel (hx_env.hip: one call per workgroup) and the tail of the persistent acting kernel (hx_actp.hip: the workgroup
// steps the envs whose actions it has just produced, EPB at a time).  i0: first env of the block; n_end: one past the last env this
// call may touch; lds: kEnvBlockLds<INSERT, EPB> floats; way: which copy of the statistics counters this workgroup adds to (+ wave).
// Ends with the block's output stores issued and NO barrier behind them: a caller that reuses `lds` / s_slot0 / s_wcount synchronises first.
template <bool INSERT, int EPB>
constexpr int kEnvBlockLds = EPB * HX_OBS_DIM + (INSERT ? EPB * kRowPitch : 0);
template <bool PAIR, bool INSERT, int EPB>
__device__ __forceinline__ void env_block_step(const StepArgs& A, int64_t i0, int64_t n_end, float* lds, unsigned& s_slot0, int* s_wcount, unsigned way) {
    constexpr int THREADS = EPB * (PAIR ? 2 : 1);
    constexpr int WAVES = THREADS / 64;
    constexpr int kObsTile = EPB * HX_OBS_DIM;

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int e = PAIR ? tid >> 1 : tid;       // env of this lane inside the tile
    const bool is_opp = PAIR && (tid & 1);     // the lane that owns the opponent aircraft
    const bool own = !is_opp;                  // the env lane: ally, missile, targeting, wrapper
    const int64_t i = i0 + e;
    const int nblk = (int)((n_end - i0) < EPB ? (n_end - i0) : EPB);
    const bool active = e < nblk;
    float* s_obs = lds;
    float* s_row = lds + kObsTile;

    if (INSERT) tile_copy<THREADS, kObsTile>(s_obs, A.obs_io + i0 * HX_OBS_DIM, nblk * HX_OBS_DIM, tid);

    Stepper<PAIR> T;
    float4 act = {0.f, 0.f, 0.f, 0.f};
    bool trunc = false, store = false, bad_act = false;
    if (active) {
        T.load(A.state, A.stride, i0, (uint32_t)e, is_opp);
        act = reinterpret_cast<const float4*>(A.actions)[i];
        bad_act = sanitize_action(act);
        uint32_t ep = T.episode_step();
        ep = ep < 65535u ? ep + 1u : ep;
        trunc = A.o.max_step > 0 && (int)ep >= A.o.max_step;  // train_all.py:346-347
        store = INSERT && !trunc;
    }
    // ring slots: ballot over the env lanes -> per-wave rank -> one atomic per workgroup (issued before the arithmetic)
    int rank = 0, nstore = 0;
    if (INSERT) {
        const unsigned long long b = __ballot(store && own);
        const unsigned below = PAIR ? (lane_id() & ~1u) : lane_id();  // both lanes of a pair get the env's rank
        rank = __popcll(b & ((1ull << below) - 1ull));
        if (lane_id() == 0) s_wcount[wave] = __popcll(b);
    }
    __syncthreads();
    if (INSERT) {
        int before = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            before += (w < wave) ? s_wcount[w] : 0;
            nstore += s_wcount[w];
        }
        rank += before;
        if (tid == 0 && nstore > 0)
            s_slot0 = ring_slot(atomicAdd((unsigned long long*)A.o.total, (unsigned long long)nstore), (unsigned long long)A.o.cap, A.inv_cap);
    }

    Wrapped W{};
    V3 eu{}, eu2{};
    bool ended = false;
    unsigned st_kill = 0, st_fs = 0, st_tl = 0, st_fire = 0, st_good = 0, st_lock = 0;
    if (active) {
        T.step(act, is_opp, eu, eu2, W);
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
        if (PAIR) {
            const unsigned theirs = swap1u(ended_own);
            ended = (own ? ended_own : theirs) != 0u;
        } else {
            ended = ended_own != 0u;
        }
        if (INSERT && store) {
            // row = s[13] a[4] s'[13] r done   (Transition, buffer.py:8; sample() drops step_success :48)
            float* row = s_row + rank * kRowPitch;
            if (!PAIR || is_opp) {  // the previous observation: copied by the opponent lane (PAIR) while its partner finishes the wrapper
                const float* prev = s_obs + e * HX_OBS_DIM;
#pragma unroll
                for (int j = 0; j < HX_OBS_DIM; ++j) row[j] = prev[j];
            }
            if (own) {
                row[13] = act.x; row[14] = act.y; row[15] = act.z; row[16] = act.w;
                row[17] = W.o0; row[18] = W.o1; row[19] = W.o2;
                row[20] = eu.x; row[21] = eu.y; row[22] = eu.z;
                row[23] = W.o6; row[24] = W.o7; row[25] = W.o8;
                row[29] = W.o12;
                row[30] = W.reward;
                row[31] = W.done ? 1.0f : 0.0f;
            }
            if (!PAIR) { row[26] = eu2.x; row[27] = eu2.y; row[28] = eu2.z; }
            else if (is_opp) { row[26] = eu.x; row[27] = eu.y; row[28] = eu.z; }
        }
    }
    __syncthreads();  // every lane has consumed its previous observation; rows complete; s_slot0 visible
    if (INSERT && store && own && A.o.ring_success) A.o.ring_success[wrap_slot(s_slot0 + (unsigned)rank, (unsigned)A.o.cap)] = (int8_t)W.success;
    if (active) {
        if (ended) {
            uint32_t epi = 0u;
            if (own) {
                epi = A.o.episode_ctr[i] + 1u;
                A.o.episode_ctr[i] = epi;
            }
            T.reset(is_opp, A.o.randomize != 0, A.o.seed, A.o.env_id0 + (uint32_t)i, epi, eu, eu2, W);
        }
        T.store(A.state, A.stride, i0, (uint32_t)e, is_opp);
        float* out = s_obs + e * HX_OBS_DIM;
        if (own) {
            out[0] = W.o0; out[1] = W.o1; out[2] = W.o2;
            out[3] = eu.x; out[4] = eu.y; out[5] = eu.z;
            out[6] = W.o6; out[7] = W.o7; out[8] = W.o8;
            out[12] = W.o12;
        }
        if (!PAIR) { out[9] = eu2.x; out[10] = eu2.y; out[11] = eu2.z; }
        else if (is_opp) { out[9] = eu.x; out[10] = eu.y; out[11] = eu.z; }
    }
    if (A.o.stats) {
        const bool mine = active && own;
        const unsigned vals[HX_STAT_COUNT] = {(mine && ended) ? 1u : 0u, st_kill, st_fs, st_tl, st_fire, st_good, st_lock, mine ? 1u : 0u,
                                              (mine && bad_act) ? 1u : 0u};
        unsigned cnt = 0;
#pragma unroll
        for (int k = 0; k < HX_STAT_COUNT; ++k) {
            const unsigned c = (unsigned)__popcll(__ballot(vals[k] != 0u));
            if (lane_id() == k) cnt = c;
        }
        // every wave adds its own counts, HERE: the atomics' round trip runs under the output stores below (summed per workgroup and issued
        // after them, it was the last thing the launch waited for: 0.6 us of a 65,536-env launch)
        if (lane_id() < HX_STAT_COUNT && cnt)
            atomicAdd((unsigned long long*)&A.o.stats[((way * WAVES + wave) % HX_STAT_WAYS) * HX_STAT_PITCH + lane_id()], (unsigned long long)cnt);
    }
    __syncthreads();
    tile_copy_out<THREADS, kObsTile>(A.obs_io + i0 * HX_OBS_DIM, s_obs, nblk * HX_OBS_DIM, tid);
    if (INSERT && nstore > 0) {
        // 16 B per lane, 1 KiB per wave-instruction, rows contiguous in the ring (modulo wrap)
        const unsigned slot0 = s_slot0, cap = (unsigned)A.o.cap;
        float4* ring4 = reinterpret_cast<float4*>(A.o.ring);
#pragma unroll
        for (int it = 0; it < EPB * (HX_ROW_WORDS / 4) / THREADS; ++it) {  // fixed trip count: every store is issued before the first waits
            const int k = tid + it * THREADS;
            if (k < nstore * (HX_ROW_WORDS / 4)) {
                const int r = k >> 3, c = (k & 7) * 4;
                const float* src = s_row + r * kRowPitch + c;
#if !defined(HX_ENV_NT) || HX_ENV_NT >= 1
                const env_v4f v = {src[0], src[1], src[2], src[3]};
                __builtin_nontemporal_store(v, reinterpret_cast<env_v4f*>(ring4) + (size_t)wrap_slot(slot0 + (unsigned)r, cap) * (HX_ROW_WORDS / 4) + (k & 7));
#else
                const float4 v = {src[0], src[1], src[2], src[3]};
                ring4[(size_t)wrap_slot(slot0 + (unsigned)r, cap) * (HX_ROW_WORDS / 4) + (k & 7)] = v;
#endif
            }
        }
    }
}

}  // namespace hxenv
#pragma clang fp contract(fast)
