/*
 * hirl4ucav.h — C ABI of the MI355X-native hot path of HIRL4UCAV (libhx_mi355.so).
 *
 * The reference (zrc0622/HIRL4UCAV) is pure Python and has no FFI; its "operator interface" for this path is
 * the duck-typed env / agent API.  The Python mirror of that API lives in hirl4ucav_amd/ and
 * calls ONLY the functions below (ctypes).  Each entry point names the reference code it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes.  Every pointer is DEVICE memory owned by the caller unless it says
 *     "host".  The library never allocates caller-visible memory and never synchronises: all work is enqueued
 *     on `stream` (a hipStream_t passed as void*; NULL = the default stream).
 *   - return value: 0 = ok, negative = error (HX_ERR_*); hx_last_error() returns a thread-local message.
 *   - thread-compatible: no global mutable state besides the thread-local error string.
 *
 * Env state layout (struct-of-arrays, fp32 words): word w of env i is state[w * stride + i], stride >= n.
 * HX_ENV_WORDS = 37 words = 148 B per env (SURVEY.md 8d canonical state):
 *     0..2  ally position (x, y = altitude, z)      13..15 opponent position
 *     3..5  ally velocity                           16..18 opponent velocity
 *     6..9  ally attitude quaternion (w, x, y, z)   19..22 opponent quaternion
 *    10..12 ally control levels (pitch, roll, yaw)  23..25 opponent control levels
 *    26..28 missile position   29..31 missile velocity
 *    32 opponent health   33 lock timer [s]   34 missile age [s]
 *    35 flags (u32 bit-cast)   36 counters (u32: lo16 episode step, hi16 opponent-script step)
 */
#ifndef HIRL4UCAV_H
#define HIRL4UCAV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HX_ENV_WORDS 37
#define HX_OBS_DIM 13
#define HX_ACT_DIM 4
#define HX_ROW_WORDS 32 /* replay row: s[13] a[4] s'[13] r done */

/* flags word (state word 35) */
#define HX_F_LOCKED_PREV (1u << 0)     /* Ally_target_locked   HarfangEnv_GYM.py:227 */
#define HX_F_LOCKED (1u << 1)          /* n_Ally_target_locked HarfangEnv_GYM.py:228 */
#define HX_F_SLOT_PREV (1u << 2)       /* missile1_state       HarfangEnv_GYM.py:250 */
#define HX_F_SLOT (1u << 3)            /* n_missile1_state     HarfangEnv_GYM.py:251 */
#define HX_F_FIRED (1u << 4)           /* now_missile_state    HarfangEnv_GYM.py:150-156 */
#define HX_F_FIRE_SUCCESS (1u << 5)    /* fire_success         HarfangEnv_GYM.py:129 */
#define HX_F_EPISODE_SUCCESS (1u << 6) /* episode_success      HarfangEnv_GYM.py:167 */
#define HX_F_DONE (1u << 7)            /* done                 HarfangEnv_GYM.py:163-166 */
#define HX_F_SCEN_SHIFT 8              /* bits 8..9: 0 straight_line, 1 serpentine, 2 circular */
#define HX_F_SERP_POS (1u << 10)
#define HX_F_SERP_LONG (1u << 11)
#define HX_F_M_ACTIVE (1u << 12)
#define HX_F_M_GUIDED (1u << 13)
#define HX_F_SIM_SLOT (1u << 14)

#define HX_ERR_ARG (-1)
#define HX_ERR_HIP (-2)

/* stats[] slots accumulated by hx_env_step (uint64 each) */
enum { HX_STAT_EPISODES = 0, HX_STAT_KILLS, HX_STAT_FIRE_SUCCESS_EPISODES, HX_STAT_TIME_LIMIT, HX_STAT_FIRES,
       HX_STAT_GOOD_FIRES, HX_STAT_LOCKED_STEPS, HX_STAT_ENV_STEPS,
       HX_STAT_NONFINITE_ACTIONS, /* env steps whose action held a NaN / Inf component: the component is taken as 0 (and stored as 0 in the
                                     replay row), so a diverged policy cannot poison the simulator state — the stand-in for the reference's
                                     missing failure detection */
       HX_STAT_COUNT };
/* The counters are kept HX_STAT_WAYS times: workgroup b adds to way b % HX_STAT_WAYS, each way in its own 128-byte line
 * (stats[way * HX_STAT_PITCH + k]); a statistic is the SUM over the ways.  One set of counters put every workgroup's atomics of a launch
 * on one cache line — one memory channel, ~12 ns each: 0.75 us of a 65,536-env launch. */
#define HX_STAT_WAYS 32
#define HX_STAT_PITCH 16

const char* hx_last_error(void);
/* HX_ABI_VERSION of the library that is loaded.  The structs below are part of the ABI: a caller built against another header version must
 * not call in (round 3 widened HxStepOpts.stats from 9 to HX_STAT_WAYS * HX_STAT_PITCH words and appended fields to HxNets / HxHyper without
 * bumping this: a 9-word stats buffer then took atomics up to word 504).  110: round 4 (hx_abi_sizes, hx_rccl_*, hx_allreduce_twostage).
 * 113: round 5.  114: round 6 (hx_rccl_allreduce_bf16). */
#define HX_ABI_VERSION 114
int hx_version(void);
/* sizes[0..7] (host) <- sizeof HxStepOpts, HxNets, HxHyper, HxBatch, HxSample, HxSacNets, HxSacBatch, and the words of a statistics buffer
 * (HX_STAT_WAYS * HX_STAT_PITCH): a binding checks these against its own declarations at load time (hirl4ucav_amd/_lib.py does). */
int hx_abi_sizes(int32_t* sizes8);
/* Kernel-duration events for HxStepOpts.ev_start / ev_stop (bench.py's live roofline measurement). */
void* hx_event_create(void);
int hx_event_destroy(void* ev);
int hx_event_elapsed_us(void* start, void* stop, float* us /* host */);

/* Optional per-call behaviour of hx_env_step.  All pointers device memory (or NULL = feature off). */
typedef struct HxStepOpts {
    int32_t max_step;      /* > 0: episode time limit (train_all.py:159-183 maxStep); the step that reaches it is
                              executed, NOT stored, and ends the episode without done (train_all.py:346-347) */
    int32_t auto_reset;    /* 1: an env whose episode ended (done or time limit) is reset in place and obs_io gets
                              the reset observation (vectorised counterpart of train_all.py:320-323) */
    int32_t randomize;     /* resets use random_reset (HarfangEnv_GYM.py:51-81) instead of reset (:34-49) */
    uint32_t env_id0;      /* global id of env 0 of this shard (Philox counter word 0 = env_id0 + i) */
    uint64_t seed;         /* Philox key */
    uint32_t* episode_ctr; /* [n] per-env episode counter (Philox counter word 1); required if auto_reset */
    float* ring;           /* [cap][HX_ROW_WORDS] replay ring or NULL: fused UniformMemory.store (buffer.py:20-36) */
    int8_t* ring_success;  /* [cap] step_success of each stored row, or NULL */
    int64_t cap;
    uint64_t* total;       /* transitions ever stored; slot = total % cap (buffer.py:36 position) */
    uint64_t* stats;       /* [HX_STAT_WAYS][HX_STAT_PITCH] or NULL: see HX_STAT_WAYS */
    void* ev_start;        /* measurement only (host handles from hx_event_create, or NULL): the launch stamps the kernel's own */
    void* ev_stop;         /* begin / end into them — the duration rocprofv3 reports, without the dispatch gap around it */
    int32_t layout;        /* 0: the library picks the launch shape from n.  HX_LAYOUT(pair, envs_per_block) forces one (tuning,
                              tests): pair = 1 steps each env on two adjacent lanes (ally + shared | opponent), 0 on one lane;
                              envs_per_block in {32, 64, 128, 256, 512} (pair) / {64, 128, 256} (solo).  Same results bit for bit. */
} HxStepOpts;
#define HX_LAYOUT(pair, envs_per_block) ((((pair) ? 1 : 0) << 8) | ((envs_per_block) / 4))

/* HarfangEnv.reset / random_reset (+ Serpentine/Circular variants): HarfangEnv_GYM.py:34-81,171-188,374-406,440-474.
 * mask: NULL = all envs, else only envs with mask[i] != 0.  scenario: per-env ids (NULL = scenario_all for all).
 * Writes the reset observation to obs[i*13..] (obs may be NULL). */
int hx_env_reset(float* state, int64_t n, int64_t stride, const uint8_t* mask, const int32_t* scenario,
                 int32_t scenario_all, int32_t randomize, uint64_t seed, uint32_t env_id0, uint32_t* episode_ctr,
                 float* obs, void* stream);

/* HarfangEnv.step for n envs: _apply_action -> one simulator tick -> _get_observation -> _get_reward ->
 * _get_termination (HarfangEnv_GYM.py:83-90,101-169,193-268; scripted opponents :342-353,412-421).
 * actions [n][4]; obs_io [n][13] in: previous observation (read only when opts->ring), out: next observation;
 * reward [n]; done [n] (0/1); success [n] (-1/0/+1). */
int hx_env_step(float* state, int64_t n, int64_t stride, const float* actions, float* obs_io, float* reward,
                uint8_t* done, int8_t* success, const HxStepOpts* opts /* host, may be NULL */, void* stream);

/* df.rearm_machine before a step (HarfangSerpentineInfiniteEnv.step_test, HarfangEnv_GYM.py:484-486) */
int hx_env_rearm(float* state, int64_t n, int64_t stride, const uint8_t* mask, void* stream);

/* Simulator-level access for the wire-protocol server (hirl4ucav_amd/environments/wire.py), i.e. what the external simulator does
 * for the reference between the client's SET_PLANE_PITCH/ROLL/YAW / FIRE_MISSILE calls and its read-backs (dogfight_client.py:
 * UPDATE_SCENE; GET_PLANE_STATE, GET_HEALTH, GET_MISSILESDEVICE_SLOTS_STATE) — no wrapper latches, no scripted opponent.
 * hx_sim_tick: one tick with the commanded (pitch, roll, yaw) levels of both aircraft [n][3] and the fire flags [n].
 * hx_sim_readback: out[n][16] = ally position 3, ally Euler (pitch, heading, roll) 3, opponent position 3, opponent Euler 3,
 * target angle in degrees, opponent health, target_locked (0/1), missile slot 0 loaded (0/1). */
int hx_sim_tick(float* state, int64_t n, int64_t stride, const float* ally_cmd, const float* opp_cmd, const uint8_t* fire,
                void* stream);
int hx_sim_readback(const float* state, int64_t n, int64_t stride, float* out, void* stream);

/* The N = 1 facade's read-back (hirl4ucav_amd/environments/HarfangEnv_GYM.py: the attributes the reference's wrapper caches after every step,
 * HarfangEnv_GYM.py:193-268) in one 64-float row: out[0..36] the state words of env i, [37..49] obs_io row i, [50] reward, [51] done,
 * [52] success, [53..63] zero — ONE device-to-host copy per step instead of five. */
int hx_env_pack_row(const float* state, int64_t n, int64_t stride, int64_t i, const float* obs, const float* reward, const uint8_t* done,
                    const int8_t* success, float* out /* [64] */, void* stream);

/* get_reward / get_termination for expert labelling (HarfangEnv_GYM.py:299-336, train_all.py:289-306):
 * s, ns [n][13], a [n][4] -> reward [n], success [n], done [n]. */
int hx_label_transitions(const float* s, const float* a, const float* ns, int64_t n, float* reward, int8_t* success,
                         uint8_t* done, void* stream);


/* ------------------------------------------------------------------------------------------------------------
 * Actor / critic side.  Parameters, gradients and Adam moments are FLAT fp32 buffers in the reference's
 * state_dict order (hirl/agents/HIRL.py:19-146): an MLP block is
 *   full{1|3}.weight [256][in], .bias [256], layernorm{1|3}.weight [256], .bias [256], full{2|4}.weight [512][256],
 *   .bias [512], layernorm{2|4}.weight [512], .bias [512], final{|1|2}.weight [out][512], .bias [out]
 * Actor = one block (in 13, out 4) = 138,756 floats; Critic = two blocks (in 17, out 1), each padded from 138,241
 * to 138,244 floats so that the second head stays 16-byte aligned (hx_critic_param_count() = 276,488).
 * ------------------------------------------------------------------------------------------------------------ */
int hx_actor_param_count(void);
int hx_critic_param_count(void);
int64_t hx_hirl_workspace_floats(int32_t batch);
int64_t hx_act_workspace_floats(int64_t rows); /* 0: the acting kernels need no workspace any more; their `ws` argument may be NULL */

/* Agent.chooseAction / chooseActionSmallNoise / chooseActionNoNoise for `rows` observations at once
 * (hirl/agents/HIRL.py:192-212): actions = clamp(actor(obs) + noise, -1, 1).
 * noise_mode 0: none (NoNoise); 1: noise[4] shared by all rows (the reference's one draw per call); 2: noise[rows][4];
 * 3: N(0, sigma^2) per row and component from Philox4x32-10(key = seed; counter = (row0 + row, call)).
 * slope: 0 = ReLU nets (HIRL.py), 0.01 = LeakyReLU nets (TD3.py / BC.py).  ws: unused (may be NULL).
 * noise_mode + 16: the actor was built with layerNorm = False (HIRL.py:135-138): no LayerNorm in its forward (all hx_actor_act* entry points). */
int hx_actor_act(const float* actor, const float* obs, int64_t rows, float* actions, int32_t noise_mode,
                 const float* noise, float sigma, uint64_t seed, uint32_t row0, uint32_t call, float slope, float* ws,
                 void* stream);
/* chooseAction + HarfangEnv.step for n envs in ONE launch (train_all.py:343-345): the actions of hx_actor_act (also written to
 * `actions`), then hx_env_step with them in the tail of the same kernel.  obs_io in: current observations, out: next. */
int hx_actor_act_step(const float* actor, float* state, int64_t n, int64_t stride, float* obs_io, float* actions,
                      int32_t noise_mode, const float* noise, float sigma, uint64_t seed, uint32_t row0, uint32_t call, float slope,
                      float* reward, uint8_t* done, int8_t* success, const HxStepOpts* opts /* host, may be NULL */, void* stream);

/* bf16 policy inference (BASELINE.json configs[4] "bf16 actor/critic + fp32 dynamics"): the 256 -> 512 layer — 96 % of the policy's
 * FLOPs — runs on v_mfma_f32_16x16x32_bf16 with h1 rounded to bf16 once and W2 read from a bf16 image; accumulation, layer 1, both
 * LayerNorms and the head stay fp32, and so do the dynamics, the update and the optimizer.  Tolerance vs the fp32 policy: |da| <= 2e-2 on
 * tanh outputs (tests/test_hirl_gpu.py); vs an fp32 evaluation on the SAME rounded operands: 1e-4.
 * hx_pack_w2_bf16: w2_bf16[512][256] = bf16(W2) of the MLP block at `net` (in_dim 13 or 17), round to nearest even. */
int hx_pack_w2_bf16(const float* net, int32_t in_dim, uint16_t* w2_bf16, void* stream);
int hx_actor_act_bf16(const float* actor, const uint16_t* w2_bf16, const float* obs, int64_t rows, float* actions, int32_t noise_mode,
                      const float* noise, float sigma, uint64_t seed, uint32_t row0, uint32_t call, float slope, void* stream);
int hx_actor_act_step_bf16(const float* actor, const uint16_t* w2_bf16, float* state, int64_t n, int64_t stride, float* obs_io,
                           float* actions, int32_t noise_mode, const float* noise, float sigma, uint64_t seed, uint32_t row0, uint32_t call,
                           float slope, float* reward, uint8_t* done, int8_t* success, const HxStepOpts* opts /* host, may be NULL */,
                           void* stream);

/* fp32 policy inference from a RE-ORDERED fp32 copy of W2 ("image", 512 x 256 floats): the same arithmetic, bit for bit, as
 * hx_actor_act / hx_actor_act_step — but every wave reads its 32 columns of W2 as contiguous kilobytes straight into registers (MFMA
 * operand order) instead of streaming the row-major matrix through LDS with a barrier per 16 k-values.  hx_pack_w2_f32i writes the image
 * of the MLP block at `net`; HxNets.actor_w2_f32i (below) keeps it current through every Adam step of the actor.  The image order is an
 * internal format (w2f_image_index in hx_update.h); both image formats (this and the bf16 one) are for the deterministic policy head. */
int hx_pack_w2_f32i(const float* net, int32_t in_dim, float* w2_f32i, void* stream);
int hx_actor_act_f32i(const float* actor, const float* w2_f32i, const float* obs, int64_t rows, float* actions, int32_t noise_mode,
                      const float* noise, float sigma, uint64_t seed, uint32_t row0, uint32_t call, float slope, void* stream);
int hx_actor_act_step_f32i(const float* actor, const float* w2_f32i, float* state, int64_t n, int64_t stride, float* obs_io,
                           float* actions, int32_t noise_mode, const float* noise, float sigma, uint64_t seed, uint32_t row0, uint32_t call,
                           float slope, float* reward, uint8_t* done, int8_t* success, const HxStepOpts* opts /* host, may be NULL */,
                           void* stream);

/* fp32 policy inference with the 256 -> 512 product on the bf16 matrix cores, EXACTLY: every fp32 operand is the sum of three bf16 numbers
 * (x = hi + mid + lo: 3 x 8 significand bits = fp32's 24, the split loses nothing), every one of the 9 partial products of a pair is exact in
 * fp32, and v_mfma_f32_16x16x32_bf16 accumulates them in fp32 — the small ones in an accumulator of their own that joins hi x hi at the end.
 * The same products as fp32 arithmetic, summed in another order: results within the fp32 kernels' own rounding noise (tests/test_x9_gpu.py
 * measures both against an fp64 evaluation), NOT bit-identical to hx_actor_act.  9 bf16 MFMAs per 32 k cost 144 matrix-core cycles against
 * 256 for fp32 MFMA (the bf16 units are 16 x faster).  w2_x9 = [3][512][256] bf16: hi | mid | lo images of W2 in the bf16 acting kernels'
 * format; hx_pack_w2_x9 writes them, HxNets.actor_w2_x9 keeps them current through every Adam step of the actor.  Deterministic head only. */
int hx_pack_w2_x9(const float* net, int32_t in_dim, uint16_t* w2_x9, void* stream);
int hx_actor_act_x9(const float* actor, const uint16_t* w2_x9, const float* obs, int64_t rows, float* actions, int32_t noise_mode,
                    const float* noise, float sigma, uint64_t seed, uint32_t row0, uint32_t call, float slope, void* stream);
int hx_actor_act_step_x9(const float* actor, const uint16_t* w2_x9, float* state, int64_t n, int64_t stride, float* obs_io,
                         float* actions, int32_t noise_mode, const float* noise, float sigma, uint64_t seed, uint32_t row0, uint32_t call,
                         float slope, float* reward, uint8_t* done, int8_t* success, const HxStepOpts* opts /* host, may be NULL */,
                         void* stream);

/* Minibatch of Agent.learn (HIRL.py:223-251), already assembled by hx_sample_batch into compact row tiles:
 * rows[batch][HX_ROW_WORDS] = s[13] a[4] s'[13] r done (buffer rows first, then expert rows, HIRL.py:229-233);
 * bc_rows[batch][HX_ROW_WORDS]: cols 0..12 state, 13..16 action of the BC minibatch (HIRL.py:248-251; NULL for TD3). */
typedef struct HxBatch {
    const float* rows;
    const float* bc_rows;
    int32_t batch;      /* multiple of 16 */
    const float* noise; /* [4] target-smoothing noise, unclamped: ONE draw for the whole batch (HIRL.py:265) */
} HxBatch;

typedef struct HxNets {
    float* actor; float* critic; float* target_actor; float* target_critic; const float* bc_actor;
    float* grad_actor; float* grad_critic;
    float* m_actor; float* v_actor; float* m_critic; float* v_critic;
    float* losses;       /* [8]: critic_loss, actor_loss, bc_loss, rl_loss, bc_fire_loss, bc_weight (HIRL.py:334) */
    int32_t* soft_count; /* [1] count(soft_Q > rl_Q) (HIRL.py:303) — all-reduce it when the batch is sharded */
    float* wstate;       /* [1] the BC weight in force */
    float* ws;           /* hx_hirl_workspace_floats(batch) */
    uint16_t* actor_w2_bf16; /* NULL, or [512][256] bf16 image of the actor's full2.weight: every Adam step of the actor refreshes it
                                (hx_adam which = 1 / 2), the bf16 acting entry points read it */
    float* actor_w2_f32i;    /* NULL, or the fp32 image of the same matrix (hx_pack_w2_f32i), refreshed likewise; hx_actor_act*_f32i read it */
    uint16_t* w2_bf16_all;   /* NULL: the update computes in fp32 (parity 1e-5 vs the reference).  Else hx_bf16_images_elems() bf16 elements =
                                the bf16 images of every W2 the update reads: the bf16 UPDATE path (below).  Its first 512 x 256 elements are
                                the actor's image in the bf16 acting kernels' format: actor_w2_bf16 must then be NULL or point at them */
    const uint32_t* xchg_status; /* NULL, or the status word of hx_allreduce_oneshot: while it is non-zero (an exchange failed: the summed
                                gradient is garbage) hx_adam / hx_adam_mixed change nothing — no parameter, moment, target or image */
    uint16_t* actor_w2_x9;   /* NULL, or the [3][512][256] hi | mid | lo bf16 images of the actor's full2.weight (hx_pack_w2_x9): refreshed by
                                every Adam step of the actor like the other two; hx_actor_act*_x9 read them */
} HxNets;

/* bf16 update path (BASELINE.json configs[4] "bf16 actor/critic + fp32 dynamics"; SURVEY.md 7 "bf16 config").  With HxNets.w2_bf16_all set,
 * every hx_hirl_* / hx_bc_train_actor call runs the three products of the 256 <-> 512 layer of all five networks (Actor / Critic forward
 * HIRL.py:55-97,126-140 inside learn() HIRL.py:259-325, and their backward passes) on v_mfma_f32_16x16x32_bf16:
 *     z2 = bf16(h1) bf16(W2)^T + b2        dh1 = bf16(dz2) bf16(W2)        dW2 = bf16(dz2)^T bf16(h1)       (fp32 accumulation)
 * Master weights, Adam moments, LayerNorm statistics, layer 1 (K = 13 / 17), the heads, TD targets and loss sums stay fp32.  The images
 * (forward order for z2, transposed for dh1; actor, critic x 2, the three targets, bc_actor) are kept current by every optimizer / Polyak
 * step (hx_hirl_learn*, hx_adam*, hx_polyak, hx_bc_train_actor); after any OTHER write to a network (loading parameters) call
 * hx_pack_update_images.  Tolerance against an fp32 evaluation on the same rounded operands: tests/test_bf16_update_gpu.py. */
int64_t hx_bf16_images_elems(void);
int hx_pack_update_images(const HxNets* nets, void* stream);

typedef struct HxHyper {
    float gamma, tau, lr_actor, lr_critic, slope, noise_clamp, loss_lambda;
    int32_t use_bc; /* 1: HIRL (TD3+BC, HIRL.py), 0: TD3 (TD3.py:201-260) */
    int32_t no_layernorm; /* 1: the agent was built with layerNorm = False: the networks' forward skips both LayerNorms (HIRL.py:70-80,92-97,
                             135-138).  The LayerNorm slots of the flat buffers must hold (1, 0) — the reference constructs the modules either
                             way (HIRL.py:28,33,114,119) and never trains them in this mode — and stay that way (their gradients are zero) */
} HxHyper;

/* Agent.learn, split at the points where a sharded run exchanges data; single-GPU callers run the
 * stages back to back on one stream:
 *   hx_hirl_critic_grads    TD target + critic loss + grad_critic          HIRL.py:259-286   -> [all-reduce grad_critic]
 *   hx_adam(which = 0)      critic.optimizer.step()                         HIRL.py:288
 *   -- every second call (actorTrainable, HIRL.py:291,332) --
 *   hx_hirl_actor_backward  pi, rl_Q with the updated critic, soft count,  HIRL.py:293-319   -> [all-reduce soft_count]
 *                           BC loss, backward down to the actor's dz2/dh1
 *   hx_hirl_actor_wgrad     grad_actor = w dL_bc + (1 - w) dL_rl            HIRL.py:321-324   -> [all-reduce grad_actor]
 *   hx_adam(which = 1)      actor.optimizer.step(); actor_loss, bc_weight   HIRL.py:325,334
 *   hx_polyak               every 3rd actor step (HIRL.py:327-330) */
/* actor_fwd: 0 = critic phase only; 1 = also run the delayed actor step's critic-independent forward passes actor(s),
 * actor(s_bc) in the same first launch; 2 = plus bc_actor(s) for the soft estimate (then pass fwd_done = 1 below). */
int hx_hirl_critic_grads(const HxNets* nets, const HxBatch* batch, const HxHyper* hyper, int32_t actor_fwd, void* stream);
int hx_hirl_actor_backward(const HxNets* nets, const HxBatch* batch, const HxHyper* hyper, int32_t estimate_soft,
                           int32_t fwd_done, void* stream);
/* w_kind 0: w = w_given (linear / fixed, train_all.py:328-333); 1: w = soft_count / count_batch + warm (HIRL.py:304-306);
 * 2: the stored weight.  w is clipped to <= 1 (HIRL.py:308). */
int hx_hirl_actor_wgrad(const HxNets* nets, const HxHyper* hyper, int32_t batch, int32_t count_batch, int32_t w_kind,
                        float w_given, float warm, void* stream);
/* which 0 critic / 1 actor / 2 actor alone (BC pre-training), + 16: also soft_update (HIRL.py:11-13) this network's target with the new
 * parameters in the same launch; step = 1-based Adam step; grad is scaled by grad_scale first (1/world after a SUM). */
int hx_adam(const HxNets* nets, const HxHyper* hyper, int32_t which, int32_t step, float grad_scale, int32_t w_kind,
            float w_given, float warm, int32_t batch, void* stream);
int hx_polyak(const HxNets* nets, const HxHyper* hyper, void* stream); /* both targets in a launch of their own */
/* Sharded runs with ONE exchange for the actor phase (SURVEY.md 8e "all-reduce dL_bc, dL_rl and the count in one message and combine
 * locally"): hx_hirl_actor_wgrad_split writes msg = [dL_rl | dL_bc | soft count as a float, 0...] (hx_actor_message_floats() floats, each
 * gradient hx_actor_param_count() floats padded to a multiple of 4), unweighted; after the all-reduce of msg, hx_adam_mixed forms
 * w = count / batch + warm from the GLOBAL count (w_kind 1; given / stored weight otherwise), g = w dL_bc + (1 - w) dL_rl (HIRL.py:321),
 * and steps the actor (grad_scale = 1 / world; polyak != 0: soft_update of targetActor in the same launch). */
int64_t hx_actor_message_floats(void);
int hx_hirl_actor_wgrad_split(const HxNets* nets, const HxHyper* hyper, int32_t batch, float* msg, void* stream);
int hx_adam_mixed(const HxNets* nets, const HxHyper* hyper, int32_t polyak, int32_t step, float grad_scale, int32_t w_kind, float w_given,
                  float warm, int32_t batch /* global */, const float* msg, void* stream);
/* BC.Agent.train_actor (hirl/agents/BC.py:160-185): one behaviour-cloning step on batch->bc_rows: loss = mse(actor(s_bc),
 * a_bc), backward, actor.optimizer.step(); losses[2] receives the loss.  hyper->slope = 0.01 gives BC.py's LeakyReLU actor.
 * (hx_adam's which = 2 is the actor step without HIRL's actor_loss / bc_weight bookkeeping.) */
int hx_bc_train_actor(const HxNets* nets, const HxBatch* batch, const HxHyper* hyper, int32_t step, void* stream);
/* The stages above back to back in one host call, for a single GPU (no gradient exchange).  actor_phase: this is an
 * actorTrainable call (HIRL.py:291,332); do_polyak: update_count reaches a multiple of 3 in it (HIRL.py:327-330);
 * critic_step / actor_step: 1-based Adam step numbers of this call; w_kind / w_given / warm as in hx_hirl_actor_wgrad. */
int hx_hirl_learn(const HxNets* nets, const HxBatch* batch, const HxHyper* hyper, int32_t critic_step, int32_t actor_phase,
                  int32_t actor_step, int32_t do_polyak, int32_t w_kind, float w_given, float warm, void* stream);

/* Minibatch assembly for one learn() call: replaces UniformMemory.sample (hirl/utils/buffer.py:38-48, random.sample without
 * replacement), the buffer/expert mixing and np.random.choice(replace=False) of HIRL.py:223-251, and the noise draw
 * HIRL.py:265.  do_sample = 1: draws idx[batch] (rows < n_main index `ring`, whose live length min(*total, cap) is read on
 * the device; the rest index `expert_ring`), idx_bc[batch] into bc_table, noise[4] = sigma N(0,1); Philox4x32-10(seed; row,
 * call).  do_sample = 0: idx / idx_bc are inputs (caller-chosen minibatch).  Either way the selected rows are copied into the
 * compact tiles rows[batch][32] and bc_rows[batch][32] (bc_table / idx_bc / bc_rows may be NULL) that HxBatch points at.
 * batch <= 1024. */
int hx_sample_batch(const uint64_t* total, int64_t cap, const float* ring, const float* expert_ring, int64_t expert_len,
                    const float* bc_table, int64_t bc_len, int32_t batch, int32_t n_main, int32_t do_sample, uint64_t seed,
                    uint32_t call, float sigma, int32_t* idx, int32_t* idx_bc, float* noise, float* rows, float* bc_rows,
                    void* stream);
/* The same with the `guard` ring slots behind the head *total left out of the draw (see HxSample.guard / hx_hirl_front). */
int hx_sample_batch_guarded(const uint64_t* total, int64_t cap, const float* ring, const float* expert_ring, int64_t expert_len,
                            const float* bc_table, int64_t bc_len, int32_t batch, int32_t n_main, int32_t do_sample, uint64_t seed,
                            uint32_t call, float sigma, int32_t* idx, int32_t* idx_bc, float* noise, float* rows, float* bc_rows,
                            uint32_t guard, void* stream);

/* The draw of hx_sample_batch(do_sample = 1) as a description instead of a launch: hx_hirl_learn_sampled / hx_hirl_critic_grads_sampled
 * draw the indices and gather the rows inside the FIRST launch of the update (every workgroup repeats the cheap draw, each gathers
 * its own 16 rows straight from the rings) — the same indices, noise, row tiles and results, bit for bit, as hx_sample_batch followed
 * by hx_hirl_learn / hx_hirl_critic_grads, with one launch and one kernel boundary less (UniformMemory.sample buffer.py:38-48 + the
 * minibatch assembly HIRL.py:223-251 + the noise draw HIRL.py:265).  The tiles HxBatch points at (rows, bc_rows, noise) are OUTPUTS
 * then (the later launches read them); idx / idx_bc receive the drawn indices.  batch <= 256 takes the fused path, larger batches run
 * the separate sampling launch inside the same call. */
typedef struct HxSample {
    const uint64_t* total; int64_t cap; const float* ring;   /* main replay ring (hx_env_step's HxStepOpts.total / cap / ring) */
    const float* expert_ring; int64_t expert_len;            /* NULL / 0: no expert rows (n_main == batch) */
    const float* bc_table; int64_t bc_len;                   /* NULL / 0: no BC minibatch (TD3) */
    int32_t n_main;                                          /* rows [0, n_main) from the main ring, the rest from the expert ring */
    uint64_t seed; uint32_t call; float sigma;               /* Philox4x32-10(seed; row, call); noise[4] = sigma N(0, 1) */
    int32_t* idx; int32_t* idx_bc;                           /* [batch] out (idx_bc NULL without a BC table) */
    uint32_t guard;                                          /* 0: UniformMemory.sample over the whole buffer.  > 0 (fused draws only): the `guard`
                                                              * slots behind the ring head *total are not drawn — see hx_hirl_front */
} HxSample;
/* hx_hirl_critic_grads / hx_hirl_learn with the minibatch drawn and gathered in their first launch (arguments as theirs + the draw). */
int hx_hirl_critic_grads_sampled(const HxNets* nets, const HxBatch* batch, const HxHyper* hyper, const HxSample* sample, int32_t actor_fwd,
                                 void* stream);
int hx_hirl_learn_sampled(const HxNets* nets, const HxBatch* batch, const HxHyper* hyper, const HxSample* sample, int32_t critic_step,
                          int32_t actor_phase, int32_t actor_step, int32_t do_polyak, int32_t w_kind, float w_given, float warm, void* stream);

/* The FRONT launch (opt-in; fp32 networks — noise_mode + 32: the policy's W2 from HxNets.actor_w2_x9, the exact three-way bf16 split, as
 * hx_actor_act_step_x9, else from HxNets.actor_w2_f32i — or the bf16 update path, HxNets.w2_bf16_all, with the bf16 acting image
 * HxNets.actor_w2_bf16; n <= 8,192 — with a replay ring any number in the exact-split format and in bf16): hx_actor_act_step_f32i / _x9 / _bf16 for n envs (chooseAction + HarfangEnv.step + replay insert,
 * train_all.py:343-345) AND the first two launches of the learn() call that follows it (targetActor(s'), Q1/Q2(s, a) [+ the actor call's forwards];
 * then targetCritic Q1/Q2 — HIRL.py:259-272) as workgroups of ONE launch: the acting workgroups take 32 rows each and so leave half of the CUs to
 * the update's workgroups, which would otherwise wait for the env step to finish although they depend on nothing it computes EXCEPT the ring it
 * inserts into.  Hence the one change of meaning: the minibatch (`batch`'s tiles, filled before this launch) is drawn from the ring as it stood
 * BEFORE this env step, leaving out the n slots the step may overwrite (HxSample.guard = n) — i.e. uniformly from every transition that is in the
 * buffer both before and after the step (the reference draws after its one-transition append, buffer.py:45; at n envs per step the population
 * differs by the newest and, once the ring is full, the oldest n transitions).  Who fills the tiles: the previous hx_hirl_learn_back (its `next`:
 * one more workgroup of the critics' gradient launch draws and gathers, on a CU that launch leaves idle), or hx_sample_batch_guarded as a launch of its own.
 * Bit-identical to hx_actor_act_step_f32i / _x9 followed by hx_hirl_learn_sampled with HxSample.total read before the step and HxSample.guard = n
 * (and launch B in 64-column workgroups: hx_debug_set_fwd_nt).  The target critics wait IN the launch for the target actor's rows (per-row-tile
 * counters `flags`, agent-scope relaxed accesses, bounded wait ~1 s: bit 0 of *status is set if a launch-B workgroup gives up, bit 1 if a launch-C
 * workgroup does (HxFront.with_c)).
 * WHY THE WAITS END.  Every workgroup of the launch is a whole CU; the acting workgroups and launch A's never wait and leave within ~20 us; the waiting
 * ones are launch B's two target-critic jobs (16 per 16-row tile: 128 at B = 128) and, with launch C riding, its TD jobs (256 more).  While the waiting
 * workgroups are FEWER than the CUs (256) — the default shape: B = 128, launch C on its own — a pending producer always finds a CU under ANY dispatch
 * order that places pending workgroups on free CUs, and a wait ends within the acting workgroups' ~20 us (HirlEngine.front_waiting_workgroups).
 * ONE PROCESS PER GPU: processes that share a GPU share its CUs, and their waiters add up — the argument above is void there.  Three processes
 * free-running the default shapes on one GPU never tripped (100,000 steps each in every acting role; with launch C riding the third process made the
 * waits time out and the status word caught it: profiles/r05_soak_front_shared_gpu.txt); EIGHT did, on a fresh box, and took round 5's test run with
 * them (GPUTEST_r05: 1,024 waiters on 256 CUs).  So the callers decide by design, not by soak [r6]: bench.py and train_all take this entry point only
 * when every rank has a device of its own (world <= visible devices) and run the reference's order otherwise; asked for explicitly on a shared device
 * (bench.py --front, HX_FRONT_SHARED_GPU=1) the status word + fallback below are what stands between a trip and a wrong number.
 * At B = 256 or with launch C riding the waiting workgroups can fill the chip, and there THE ASSUMPTION BEHIND THE WAITS is needed: the workgroups of one
 * launch START in index order (producers have the lower indices), so a waiting workgroup's producers are running or done.  That is what gfx950 /
 * ROCm 7.2 does (free-running soaks of every acting role, launch C riding included: profiles/archive/r04c_front_soak_*.json, profiles/r05_soak_front_roles.jsonl);
 * HIP promises no dispatch order.  Under another order a wait still ends as soon as the producers get a CU (the acting workgroups never wait and leave
 * after ~20 us); only if EVERY resident workgroup were a waiting consumer could a wait run into its bound — then the status word says so, the minibatch
 * of that launch may have been read half-written, and the caller must not go on: read *status at least every few hundred launches (hirl4ucav_amd/
 * train_all.py --status_check_every, default 256: on a trip it reloads its last snapshot and continues in the reference's order in the same process, or
 * exits with code 3; bench.py reads it after the timed region and the second pass — one decision for all ranks — and on a trip repeats the whole run in
 * the reference's order in the same process and labels the line `reference order (front tripped)`) and fall back to hx_actor_act_step_* +
 * hx_hirl_learn_sampled (no in-launch waits; `train_all --loop reference`, `bench.py --no-front`).
 * (Roles taken in START order from a ticket would need no such assumption; measured, that costs 2.8 us of a 51.4 us step: docs/LEVERS.md, Round 5.)
 * hx_hirl_learn_back = the rest of the call (critic backward + gradients + Adam [+ the delayed actor step]) on the same `batch`; next / next_tiles
 * (or NULL): the draw of the NEXT front launch (guard = its n; *next->total is read inside this call's second launch, i.e. after this step's
 * inserts and before the next step's) into tiles of their own. */
typedef struct HxFront {
    uint32_t* flags;       /* [64] device words, zero before the first use (and whenever epoch starts over at 1) */
    uint32_t* status;      /* device word, sticky */
    uint32_t epoch;        /* 1, 2, 3, ... : one per front launch on these flags */
    uint32_t with_c;       /* 0, or 1, 2, 3, ... = this is the k-th front launch on these flags WITH launch C: the critics' backward launch (y, loss, dq, dh1: HIRL.py:270-286) rides in this launch too, waiting in-launch for the forward
                            * workgroups it reads; then pass c_in_front = 1 to hx_hirl_learn_back / hx_hirl_critic_grads_back.  Same bits; it pays only where the
                            * acting workgroups leave the other CUs time to spare (around 8,192 envs in the exact-split format: HirlEngine.front_c_for) */
} HxFront;
int hx_hirl_front(float* state, int64_t n, int64_t stride, float* obs_io, float* actions, int32_t noise_mode,
                  const float* noise, float sigma, uint64_t seed, uint32_t row0, uint32_t call, float* reward, uint8_t* done, int8_t* success,
                  const HxStepOpts* opts, const HxNets* nets, const HxBatch* batch, const HxHyper* hyper, int32_t actor_phase, int32_t w_kind,
                  const HxFront* front, void* stream);
int hx_hirl_learn_back(const HxNets* nets, const HxBatch* batch, const HxHyper* hyper, int32_t critic_step, int32_t actor_phase, int32_t actor_step,
                       int32_t do_polyak, int32_t w_kind, float w_given, float warm, const HxSample* next, const HxBatch* next_tiles, int32_t c_in_front,
                       void* stream);
/* A sharded rank's form: what hx_hirl_critic_grads leaves behind (grad_critic, ready for the exchange; no optimizer step) after a front launch, + the predraw. */
int hx_hirl_critic_grads_back(const HxNets* nets, const HxBatch* batch, const HxHyper* hyper, const HxSample* next, const HxBatch* next_tiles, int32_t c_in_front,
                              void* stream);


/* ------------------------------------------------------------------------------------------------------------
 * SAC (hirl/agents/SAC, the non-imitative branch train_sac.py uses).  Networks are the plain Linear-ReLU stacks of the
 * un-vendored rltorch builder (SAC/model.py:21,58): policy 13 -> 256 -> 512 -> 8 (mean ++ log_std), Q heads 17 -> 256 -> 512
 * -> 1.  They use the SAME flat block layout as above; the LayerNorm slots hold (1, 0) and are never updated.
 * hx_sac_policy_param_count() floats for the policy, hx_critic_param_count() for the twinned Q.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct HxSacNets {
    float* policy; float* critic; float* target_critic;
    float* grad_policy; float* grad_critic;
    float* m_policy; float* v_policy; float* m_critic; float* v_critic;
    float* losses;      /* [8]: q1_loss, q2_loss, policy_loss, entropy_loss, mean entropy, alpha */
    float* alpha_state; /* [4]: log_alpha, its Adam m and v, alpha = exp(log_alpha)  (SAC/agent.py:106-108) */
    float* ws;          /* hx_sac_workspace_floats(batch) */
    float* policy_w2_f32i; /* NULL, or the fp32 image of the policy's W2 (hx_pack_w2_f32i(policy, 13, ...)): hx_sac_adam(which = 1) keeps it
                              current, hx_sac_act*_f32i read it */
    uint16_t* policy_w2_x9; /* NULL, or the hi | mid | lo bf16 images of the policy's W2 (hx_pack_w2_x9(policy, 13, ...), 3 x 512 x 256): the
                               policy's optimizer step keeps them current, hx_sac_act*_x9 read them */
} HxSacNets;

typedef struct HxSacBatch {
    const float* rows;     /* [batch][HX_ROW_WORDS] compact minibatch (hx_sample_batch) */
    int32_t batch;         /* multiple of 16 */
    const float* eps_next; /* [batch][4] standard-normal draws of policy.sample(next_states) (SAC/agent.py:204) */
    const float* eps_cur;  /* [batch][4] draws of policy.sample(states) in calc_policy_loss (SAC/agent.py:380) */
    uint64_t seed;         /* eps_next / eps_cur NULL: the draws come from Philox4x32-10(seed; row, call) inside the kernels */
    uint32_t call;         /* (distinct streams for the two samples), no draw buffers and no launches to fill them */
} HxSacBatch;

int hx_sac_policy_param_count(void);
int64_t hx_sac_workspace_floats(int32_t batch);
/* SacAgent.explore / exploit (SAC/agent.py:183-196): mode 0 exploit = tanh(mean); 1 sample with eps[rows][4]; 2 sample with
 * Philox4x32-10(seed; row0 + row, call).  ws: unused (may be NULL). */
int hx_sac_act(const float* policy, const float* obs, int64_t rows, float* actions, int32_t mode, const float* eps,
               uint64_t seed, uint32_t row0, uint32_t call, float* ws, void* stream);
/* The same from the re-ordered fp32 image of the policy's W2 (see hx_actor_act_f32i): bit-identical results, W2 straight into registers. */
int hx_sac_act_f32i(const float* policy, const float* w2_f32i, const float* obs, int64_t rows, float* actions, int32_t mode, const float* eps,
                    uint64_t seed, uint32_t row0, uint32_t call, void* stream);
int hx_sac_act_step_f32i(const float* policy, const float* w2_f32i, float* state, int64_t n, int64_t stride, float* obs_io, float* actions,
                         int32_t mode, const float* eps, uint64_t seed, uint32_t row0, uint32_t call, float* reward, uint8_t* done,
                         int8_t* success, const HxStepOpts* opts /* host, may be NULL */, void* stream);
/* SacAgent.explore / exploit + HarfangEnv.step in one launch (train_sac.py:238-241). */
/* The same from the exact three-way bf16 split of the 256 -> 512 product's operands (hx_actor_act_x9's format; w2_x9 = hx_pack_w2_x9(policy, 13, ...)):
 * the large-population format of the fp32 policy — beyond 8,192 rows the persistent kernel streams the three images; up to 8,192 rows these
 * entry points fall back to the fp32 image, which must then be given as well (w2_f32i). */
int hx_sac_act_x9(const float* policy, const uint16_t* w2_x9, const float* w2_f32i, const float* obs, int64_t rows, float* actions, int32_t mode,
                  const float* eps, uint64_t seed, uint32_t row0, uint32_t call, void* stream);
int hx_sac_act_step_x9(const float* policy, const uint16_t* w2_x9, const float* w2_f32i, float* state, int64_t n, int64_t stride, float* obs_io,
                       float* actions, int32_t mode, const float* eps, uint64_t seed, uint32_t row0, uint32_t call, float* reward, uint8_t* done,
                       int8_t* success, const HxStepOpts* opts, void* stream);
int hx_sac_act_step(const float* policy, float* state, int64_t n, int64_t stride, float* obs_io, float* actions, int32_t mode,
                    const float* eps, uint64_t seed, uint32_t row0, uint32_t call, float* reward, uint8_t* done, int8_t* success,
                    const HxStepOpts* opts /* host, may be NULL */, void* stream);
/* SacAgent.learn (SAC/agent.py:276-327) in the stages a sharded run separates:
 *   hx_sac_critic_grads   [soft_update of the target critics first on every 3rd call :278-279]; target y = r + (1-d) gamma
 *                         (min Q_target(s', a') + alpha H') :202-210; q1_loss, q2_loss :361-374 -> losses[0..1]; grad_critic
 *   hx_sac_adam(0)        q1_optim.step(), q2_optim.step() :310-313                                  [after all-reduce]
 *   hx_sac_policy_grads   policy_loss = mean(-min Q(s, a~) - alpha H) with the updated critics :376-406 -> losses[2], grad_policy
 *   hx_sac_adam(1)        policy_optim.step() :318-319, then entropy_loss and alpha_optim.step() :322-325,408-414 */
int hx_sac_critic_grads(const HxSacNets* nets, const HxSacBatch* batch, const HxHyper* hyper, int32_t polyak_first, void* stream);
/* hx_sac_critic_grads with the minibatch drawn and gathered in its first launch (HxSample without a BC table; batch->rows is the output
 * tile; batch <= 256 draws in-launch, larger batches run the sampling launch inside the call): same indices, tile and results. */
int hx_sac_critic_grads_sampled(const HxSacNets* nets, const HxSacBatch* batch, const HxHyper* hyper, const HxSample* sample,
                                int32_t polyak_first, void* stream);
/* one GPU: critic gradients AND q1_optim / q2_optim steps (SAC/agent.py:278-313) in one call — the optimizer step rides in the weight-gradient
 * launch; sample may be NULL (minibatch already assembled); step is 1-based.  Bit-identical to hx_sac_critic_grads[_sampled] + hx_sac_adam(0). */
int hx_sac_critic_step(const HxSacNets* nets, const HxSacBatch* batch, const HxHyper* hyper, const HxSample* sample, int32_t polyak_first,
                       int32_t step, void* stream);
int hx_sac_policy_grads(const HxSacNets* nets, const HxSacBatch* batch, const HxHyper* hyper, void* stream);
/* One GPU: the whole SacAgent.learn (SAC/agent.py:276-327) in one call and 9 launches — the soft_update and policy.sample(s) in the launch of
 * policy.sample(s'), every optimizer step (log-alpha included) in its weight-gradient launch, the min(Q1, Q2) selection and the policy's head
 * gradient in the prologues of the backward launches that consume them.  Bit-identical to hx_sac_critic_step +
 * hx_sac_policy_grads + hx_sac_adam(which = 1); sample may be NULL (minibatch already assembled); step is 1-based. */
int hx_sac_learn(const HxSacNets* nets, const HxSacBatch* batch, const HxHyper* hyper, const HxSample* sample, int32_t polyak_first, int32_t step,
                 float target_entropy, void* stream);
/* The SAC front launch (n > 8,192 envs with a replay ring): hx_sac_act_step_x9 (w2_x9 != NULL) / hx_sac_act_step_f32i — explore / exploit + HarfangEnv.step +
 * replay insert (train_sac.py:238-241) — AND the first forward launch of the SacAgent.learn call that follows it (policy(s'), policy(s), Q1/Q2(s, a): SAC/agent.py:
 * 198-210, 276-290) as workgroups of ONE launch that start as the acting ones leave, on the minibatch `batch->rows` holds already.  As for hx_hirl_front the
 * minibatch is drawn from the ring as it stood BEFORE this env step without the n slots the step may overwrite (HxSample.guard = n): by the previous
 * hx_sac_learn_back (its `next`: one more workgroup of the policy's gradient launch draws and gathers into next_rows) or by hx_sample_batch_guarded.
 * Bit-identical to hx_sac_act_step_* followed by hx_sac_learn with HxSample.total read before the step and HxSample.guard = n.
 * hx_sac_learn_back = the rest of hx_sac_learn (8 launches). */
int hx_sac_front(const float* policy, const uint16_t* w2_x9, const float* w2_f32i, float* state, int64_t n, int64_t stride, float* obs_io, float* actions,
                 int32_t mode, const float* eps, uint64_t seed, uint32_t row0, uint32_t call, float* reward, uint8_t* done, int8_t* success,
                 const HxStepOpts* opts, const HxSacNets* nets, const HxSacBatch* batch, void* stream);
int hx_sac_learn_back(const HxSacNets* nets, const HxSacBatch* batch, const HxHyper* hyper, int32_t polyak_first, int32_t step, float target_entropy,
                      const HxSample* next, float* next_rows, void* stream);
int hx_sac_adam(const HxSacNets* nets, const HxHyper* hyper, int32_t which, int32_t step, float grad_scale, float target_entropy,
                void* stream);


/* ------------------------------------------------------------------------------------------------------------
 * Exchange step of a sharded update without a collective library (selectable beside RCCL): one-shot all-reduce over hipIpc peer
 * mappings (hx_xchg.hip).  The reference has no counterpart (single process); SURVEY.md 5 / 8e motivate it: 0.5-1.1 MB messages on a
 * fully connected xGMI fabric.  hx_ipc_*: device memory peers can map; handle64 = 64 bytes (hipIpcMemHandle_t) exchanged by any channel.
 * ------------------------------------------------------------------------------------------------------------ */
int hx_ipc_alloc(int64_t bytes, int32_t finegrained /* 1: flag / status words */, void** dev_ptr);
int hx_ipc_free(void* dev_ptr);
int hx_ipc_export(void* dev_ptr, void* handle64 /* host, out */);
int hx_ipc_import(const void* handle64 /* host */, void** dev_ptr);
int hx_ipc_close(void* dev_ptr);
/* dst[i] = bufs[0][i] + bufs[1][i] + ... (rank order: bit-identical on every rank), n floats (multiple of 4).  bufs / flags: host arrays of
 * `world` (<= 8) device pointers, own memory at index `rank`; every rank calls with the same epoch = 1, 2, 3, ...; messages are
 * double-buffered by the CALLER (epoch parity), see hx_xchg.hip.  *status != 0 afterwards: a peer did not arrive within timeout_ms (1) or
 * reported failure (2) — sticky and global (fail-stop): see hx_xchg.hip.  EXPERIMENTAL until it has run on two physical GPUs. */
int hx_allreduce_oneshot(float* dst, const float* const* bufs, uint32_t* const* flags, uint32_t* status, int32_t world, int32_t rank,
                         int64_t n, uint32_t epoch, int32_t timeout_ms, void* stream);

/* The same sum in TWO stages — reduce-scatter, then all-gather: rank r sums its slice [n r / world, n (r + 1) / world) of every rank's
 * message into reds[r] (its peer-mapped buffer of n floats), then copies the peers' reduced slices: 2 (world - 1) / world x n floats per rank
 * over xGMI instead of world x n.  Two launches; flags2: a second flag word per rank; bf16 != 0: the reduced slices travel as
 * bf16 (the fp32 sum rounded to nearest even, on every rank alike).  Otherwise as hx_allreduce_oneshot, the same fail-stop rules.  EXPERIMENTAL. */
int hx_allreduce_twostage(float* dst, const float* const* bufs, void* const* reds, uint32_t* const* flags, uint32_t* const* flags2, uint32_t* status,
                          int32_t world, int32_t rank, int64_t n, uint32_t epoch, int32_t timeout_ms, int32_t bf16, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * The exchange on RCCL without torch.distributed in the loop (hx_rccl.hip): ncclAllReduce enqueued on the caller's stream.  librccl.so is
 * opened at run time — the copy the process already holds when there is one.  hx_rccl_unique_id on ONE rank, its 128 bytes to every rank by
 * any channel, hx_rccl_init on every rank (collective; one GPU per rank), then hx_rccl_allreduce per message: buf <- sum over ranks, in place
 * (dtype 0 fp32, 1 bf16), the same bits on every rank.  SURVEY.md 8e: the flat critic gradient and the merged actor message.
 * ------------------------------------------------------------------------------------------------------------ */
int hx_rccl_available(void); /* 0: the library and its entry points bind in this process (local, no collective) — ask on every rank before hx_rccl_init */
int hx_rccl_unique_id(uint8_t* id128 /* host, out */);
int hx_rccl_init(const uint8_t* id128 /* host */, int32_t world, int32_t rank, void** comm /* host, out */);
int hx_rccl_allreduce(void* comm, void* buf, int64_t n, int32_t dtype, void* stream);
/* The same exchange with the message on the wire as bf16 (half the bytes; RCCL sums in bf16): buf[i] (fp32) -> scratch[i] (bf16, round to nearest even) ->
 * ncclAllReduce(bf16) -> buf[i] (fp32), three enqueues on `stream`.  Every rank receives the same bits (replicas stay identical); opt-in
 * (`--exchange rccl-bf16`), n a multiple of 4, scratch: n bf16 device words. */
int hx_rccl_allreduce_bf16(void* comm, float* buf, uint16_t* scratch, int64_t n, void* stream);
int hx_rccl_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* HIRL4UCAV_H */
