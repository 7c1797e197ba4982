// hx_hirl.hip — host sequencing of Agent.learn (hirl/agents/HIRL.py:221-334), TD3.Agent.learn (hirl/agents/TD3.py:201-260) and
// BC.Agent.train_actor (hirl/agents/BC.py:160-185): which nets ride in which launch.  4 launches on a critic-only call, 8 on a call with the
// delayed actor step (minibatch draw and optimizer steps included); the stage entry points are what a sharded run calls between exchanges.
#include <cmath>

#include "hx_update.h"

using namespace hxnn;
using namespace hxu;

extern "C" {

int hx_actor_param_count(void) { return kActor.size(); }
int hx_critic_param_count(void) { return 2 * kQ.padded(); }
int64_t hx_hirl_workspace_floats(int32_t batch) { return (int64_t)S_COUNT * kSlotFloats * batch + 64; }
int64_t hx_actor_message_floats(void) { return 2 * (int64_t)kActor.padded() + 64; }
int64_t hx_bf16_images_elems(void) { return (int64_t)IM_COUNT * (int64_t)kImgElems; }

/* Rebuild every bf16 image of the update path from the fp32 networks (after parameters were loaded or written directly). */
int hx_pack_update_images(const HxNets* N, void* stream) {
    HX_REQUIRE(N && N->w2_bf16_all && (reinterpret_cast<uintptr_t>(N->w2_bf16_all) & 15u) == 0, "hx_pack_update_images: w2_bf16_all must be a 16-byte aligned buffer");
    HX_REQUIRE(N->actor && N->critic && N->target_actor && N->target_critic, "hx_pack_update_images: null network");
    hipStream_t st = (hipStream_t)stream;
    uint16_t* im = N->w2_bf16_all;
    launch_pack_bf16(N->actor + kActor.W2(), im + IM_ACTOR * kImgElems, false, st);
    launch_pack_bf16(N->actor + kActor.W2(), im + IM_ACTOR_T * kImgElems, true, st);
    launch_pack_bf16(N->target_actor + kActor.W2(), im + IM_TA * kImgElems, false, st);
    if (N->bc_actor) launch_pack_bf16(N->bc_actor + kActor.W2(), im + IM_BC * kImgElems, false, st);
    for (int h = 0; h < 2; ++h) {
        launch_pack_bf16(N->critic + h * kQ.padded() + kQ.W2(), im + (IM_C1 + h) * kImgElems, false, st);
        launch_pack_bf16(N->critic + h * kQ.padded() + kQ.W2(), im + (IM_C1_T + h) * kImgElems, true, st);
        launch_pack_bf16(N->target_critic + h * kQ.padded() + kQ.W2(), im + (IM_TC1 + h) * kImgElems, false, st);
    }
    HX_CHECK_LAUNCH("hx_pack_update_images");
    return 0;
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

// launch A: targetActor(s'), critic Q1/Q2 (s, a)  [+ actor(s) on an actor call]
static void make_launch_a(const HxNets* N, const HxBatch* Bt, const HxHyper* Hy, int32_t actor_fwd, FwdArgs& F) {
    const Mlp mA{13, 4, Hy->no_layernorm ? 1 : 0}, mQ{17, 1, Hy->no_layernorm ? 1 : 0};  // layerNorm = False: HIRL.py:70-80,92-97,135-138
    const int B = Bt->batch;
    Slot s[S_COUNT];
    make_slots(N, B, s);
    const RowSrc src{Bt->rows, nullptr, nullptr, 0, 32};
    F = FwdArgs{};
    F.njobs = 3; F.slope = Hy->slope;
    F.zero_f = N->losses; F.zero_nf = 1;  // critic_loss accumulator
    F.job[0] = FwdJob{N->target_actor, mA, src, 17, 0, Head{}, nullptr, 0.f, s[S_TA], B, 0, IM_TA};
    F.job[1] = FwdJob{N->critic, mQ, src, 0, 0, Head{}, nullptr, 0.f, s[S_C1], B, 1, IM_C1};
    F.job[2] = FwdJob{N->critic + mQ.padded(), mQ, src, 0, 0, Head{}, nullptr, 0.f, s[S_C2], B, 1, IM_C2};
    if (actor_fwd) {  // the delayed actor step's forwards ride along, split so that NEITHER launch exceeds 256 workgroups
        F.zero_nf = 5; F.zero_i = N->soft_count;  // + actor / bc / rl / bc_fire accumulators and the soft count
        F.job[3] = FwdJob{N->actor, mA, src, 0, 0, Head{}, nullptr, 0.f, s[S_API], B, 1, IM_ACTOR};
        F.njobs = 4;
    }
    F.images = N->w2_bf16_all;
}
// launch B: targetCritic Q1/Q2 (s', clamp(targetActor(s') + clamp(noise)))  [+ actor(s_bc), bc_actor(s)]
static void make_launch_b(const HxNets* N, const HxBatch* Bt, const HxHyper* Hy, int32_t actor_fwd, FwdArgs& F) {
    const Mlp mA{13, 4, Hy->no_layernorm ? 1 : 0}, mQ{17, 1, Hy->no_layernorm ? 1 : 0};
    const int B = Bt->batch;
    Slot s[S_COUNT];
    make_slots(N, B, s);
    const RowSrc src{Bt->rows, nullptr, nullptr, 0, 32};
    const float* tc1 = N->target_critic;
    const float* tc2 = N->target_critic + mQ.padded();
    F = FwdArgs{};
    F.njobs = 2; F.slope = Hy->slope;
    const Head prev{N->target_actor, mA, s[S_TA]};
    F.job[0] = FwdJob{tc1, mQ, src, 17, 1, prev, Bt->noise, Hy->noise_clamp, s[S_TC1], B, 0, IM_TC1};
    F.job[1] = FwdJob{tc2, mQ, src, 17, 1, prev, Bt->noise, Hy->noise_clamp, s[S_TC2], B, 0, IM_TC2};
    if (actor_fwd && Hy->use_bc) {
        const RowSrc bcsrc{Bt->bc_rows ? Bt->bc_rows : Bt->rows, nullptr, nullptr, 0, 32};
        int n = 2;
        F.job[n++] = FwdJob{N->actor, mA, bcsrc, 0, 0, Head{}, nullptr, 0.f, s[S_ABC], B, 1, IM_ACTOR};
        if (actor_fwd == 2) F.job[n++] = FwdJob{N->bc_actor, mA, src, 0, 0, Head{}, nullptr, 0.f, s[S_BCS], B, 0, IM_BC};
        F.njobs = n;
    }
    F.images = N->w2_bf16_all;
}
// launch C: y, loss, dq, LN2 backward, dh1 for both heads
static void make_launch_c(const HxNets* N, const HxBatch* Bt, const HxHyper* Hy, BwdArgs& G) {
    const Mlp mQ{17, 1, Hy->no_layernorm ? 1 : 0};
    const int B = Bt->batch;
    Slot s[S_COUNT];
    make_slots(N, B, s);
    const RowSrc src{Bt->rows, nullptr, nullptr, 0, 32};
    const float* tc1 = N->target_critic;
    const float* tc2 = N->target_critic + mQ.padded();
    G = BwdArgs{};
    G.njobs = 2; G.slope = Hy->slope; G.inv_batch = 1.0f / B; G.losses = N->losses; G.soft_count = N->soft_count;
    for (int h = 0; h < 2; ++h) {
        BwdJob& J = G.job[h];
        J = BwdJob{};
        J.net = N->critic + h * mQ.padded(); J.m = mQ; J.ws = s[S_C1 + h]; J.rows = B; J.mode = BM_CRITIC_TD;
        J.t1 = Head{tc1, mQ, s[S_TC1]}; J.t2 = Head{tc2, mQ, s[S_TC2]}; J.src = src; J.gamma = Hy->gamma;
        J.img_t = IM_C1_T + h;
    }
    G.images = N->w2_bf16_all;
}
// Launch C inside the front launch (HxFront.with_c; its workgroups wait in-launch for launches A and B: hx_bwd_body.h) — the CALLER's choice, told to the
// calls behind it as c_in_front.  Bit-identical either way (tests/test_front_gpu.py); it pays only where the acting workgroups leave the other CUs more
// time than launches A, B AND C need (CU time bounds the front launch): the streaming acting role around 8,192 envs — 67.1 -> 62.2 us per step there,
// 54.1 -> 56.6 at 4,096 envs fp32, 45.6 -> 49.8 bf16, 133.1 -> 136.4 at 131,072 bf16 (tools/ubench/front_c_ab.sh, profiles/archive/r04c_front_c_ab.txt):
// HirlEngine.front_c_for() holds the rule.  HX_FRONT_C = 0 / 1 (A/B knob, read once) overrides the caller in BOTH calls.
static bool front_has_c(bool asked) {
    static const int force = getenv("HX_FRONT_C") ? atoi(getenv("HX_FRONT_C")) : -1;
    return force < 0 ? asked : force != 0;
}
// launches C and D: y, loss, dq, LN2 backward, dh1 for both heads; all critic parameter gradients.
// adam_step > 0: the critic's optimizer step (and, with polyak, the soft_update of its target) rides in the wgrad launch
// skip_c: launch C ran inside the front launch
static int critic_back(const HxNets* N, const HxBatch* Bt, const HxHyper* Hy, void* stream, int adam_step, bool polyak, const SampleDev* predraw = nullptr, bool skip_c = false) {
    hipStream_t st = (hipStream_t)stream;
    const Mlp mQ{17, 1, Hy->no_layernorm ? 1 : 0};
    const int B = Bt->batch;
    Slot s[S_COUNT];
    make_slots(N, B, s);
    if (!skip_c) {
        BwdArgs G;
        make_launch_c(N, Bt, Hy, G);
        launch_bwd(0, G, st);
    }
    {   // launch D: all critic parameter gradients
        WgArgs W{};
        W.njobs = 2; W.slope = Hy->slope; W.w_kind = 0; W.w_given = 0.f; W.inv_batch = 1.0f / B;
        W.soft_count = N->soft_count; W.wstate = N->wstate;
        for (int h = 0; h < 2; ++h) {
            WgJob& J = W.job[h];
            J = WgJob{};
            J.net = N->critic + h * mQ.padded(); J.grad = N->grad_critic + h * mQ.padded(); J.m = mQ;
            J.ws[0] = s[S_C1 + h]; J.rows[0] = B; J.nslots = 1; J.wmode[0] = 0;
            if (adam_step > 0) {
                J.p = N->critic + h * mQ.padded();
                J.mom = N->m_critic + h * mQ.padded(); J.var = N->v_critic + h * mQ.padded();
                J.target = polyak ? N->target_critic + h * mQ.padded() : nullptr;
                if (uint16_t* im = N->w2_bf16_all) {
                    J.w2b = im + (IM_C1 + h) * kImgElems; J.w2tb = im + (IM_C1_T + h) * kImgElems; J.tgt_w2b = im + (IM_TC1 + h) * kImgElems;
                }
            }
        }
        W.bf16 = N->w2_bf16_all != nullptr;
        W.predraw = predraw; W.predraw_batch = B;
        if (adam_step > 0) {
            W.ad = make_adam(N, Hy, Hy->lr_critic, adam_step, false);
            launch_wg(W, true, st);
        } else {
            launch_wg(W, false, st);
        }
    }
    HX_CHECK_LAUNCH("hx_hirl_critic_grads");
    return 0;
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
    FwdArgs F;
    make_launch_a(N, Bt, Hy, actor_fwd, F);
    F.sample = fused ? &SD : nullptr;
    launch_fwd(F, st);
    make_launch_b(N, Bt, Hy, actor_fwd, F);
    launch_fwd(F, st);
    return critic_back(N, Bt, Hy, stream, adam_step, polyak);
}
int hx_hirl_critic_grads(const HxNets* N, const HxBatch* Bt, const HxHyper* Hy, int32_t actor_fwd, void* stream) {
    return critic_grads_impl(N, Bt, Hy, actor_fwd, stream, 0, false);
}
int hx_hirl_critic_grads_sampled(const HxNets* N, const HxBatch* Bt, const HxHyper* Hy, const HxSample* S, int32_t actor_fwd, void* stream) {
    HX_REQUIRE(S, "hx_hirl_critic_grads_sampled: null sample description");
    return critic_grads_impl(N, Bt, Hy, actor_fwd, stream, 0, false, S);
}

/* Stage 2a (delayed actor step, HIRL.py:291-319): actor / bc_actor forward, Q1 with the UPDATED critic, the soft
 * count, backward down to dz2/dh1 of the actor.  Leaves soft_count and losses[2..4] ready; no parameter gradient yet. */
int hx_hirl_actor_backward(const HxNets* N, const HxBatch* Bt, const HxHyper* Hy, int32_t estimate_soft, int32_t fwd_done, void* stream) {
    HX_REQUIRE(N && Bt && Hy && Bt->batch > 0 && Bt->batch % 16 == 0, "hx_hirl_actor_backward: batch must be a positive multiple of 16");
    hipStream_t st = (hipStream_t)stream;
    const Mlp mA{13, 4, Hy->no_layernorm ? 1 : 0}, mQ{17, 1, Hy->no_layernorm ? 1 : 0};  // layerNorm = False: HIRL.py:70-80,92-97,135-138
    (void)mA; (void)mQ;
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
        F.job[n++] = FwdJob{N->actor, mA, src, 0, 0, Head{}, nullptr, 0.f, s[S_API], B, 1, IM_ACTOR};
        if (bc) F.job[n++] = FwdJob{N->actor, mA, bcsrc, 0, 0, Head{}, nullptr, 0.f, s[S_ABC], B, 1, IM_ACTOR};
        if (soft) F.job[n++] = FwdJob{N->bc_actor, mA, src, 0, 0, Head{}, nullptr, 0.f, s[S_BCS], B, 0, IM_BC};
        F.njobs = n;
        F.images = N->w2_bf16_all;
        launch_fwd(F, st);
    }
    {   // launch G: Q1(s, pi(s)) and Q1(s, bc_actor(s)) with the updated critic
        FwdArgs F{};
        F.slope = Hy->slope;
        int n = 0;
        F.job[n++] = FwdJob{N->critic, mQ, src, 0, 1, Head{N->actor, mA, s[S_API]}, nullptr, 0.f, s[S_CPI], B, 1, IM_C1};
        if (soft) F.job[n++] = FwdJob{N->critic, mQ, src, 0, 1, Head{N->bc_actor, mA, s[S_BCS]}, nullptr, 0.f, s[S_CSOFT], B, 0, IM_C1};
        F.njobs = n;
        F.images = N->w2_bf16_all;
        launch_fwd(F, st);
    }
    {   // launch H: rl_loss, soft count, critic backward down to dh1 (gradient wrt the action comes next)
        BwdArgs G{};
        G.njobs = 1; G.slope = Hy->slope; G.inv_batch = 1.0f / B; G.losses = N->losses; G.soft_count = N->soft_count;
        BwdJob& J = G.job[0];
        J = BwdJob{};
        J.net = N->critic; J.m = mQ; J.ws = s[S_CPI]; J.rows = B; J.mode = BM_CRITIC_PI;
        if (soft) J.soft = Head{N->critic, mQ, s[S_CSOFT]};
        J.img_t = IM_C1_T;
        G.images = N->w2_bf16_all;
        launch_bwd(1, G, st);
    }
    {   // launch I: actor backward for the RL batch (through tanh and the critic's input gradient) and the BC batch
        BwdArgs G{};
        G.slope = Hy->slope; G.inv_batch = 1.0f / B; G.losses = N->losses; G.soft_count = N->soft_count;
        int n = 0;
        {
            BwdJob& J = G.job[n++];
            J = BwdJob{};
            J.net = N->actor; J.m = mA; J.ws = s[S_API]; J.rows = B; J.mode = BM_ACTOR_PI;
            J.crit = Head{N->critic, mQ, s[S_CPI]};
            J.img_t = IM_ACTOR_T;
        }
        if (bc) {
            BwdJob& J = G.job[n++];
            J = BwdJob{};
            J.net = N->actor; J.m = mA; J.ws = s[S_ABC]; J.rows = B; J.mode = BM_ACTOR_BC;
            J.src = bcsrc; J.lambda = Hy->loss_lambda;
            J.img_t = IM_ACTOR_T;
        }
        G.njobs = n;
        G.images = N->w2_bf16_all;
        launch_bwd(2, G, st);
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
    const Mlp mA{13, 4, Hy->no_layernorm ? 1 : 0}, mQ{17, 1, Hy->no_layernorm ? 1 : 0};  // layerNorm = False: HIRL.py:70-80,92-97,135-138
    (void)mA; (void)mQ;
    Slot s[S_COUNT];
    make_slots(N, batch, s);
    const bool bc = Hy->use_bc != 0;
    WgArgs W{};
    W.njobs = 1; W.slope = Hy->slope; W.w_kind = bc ? w_kind : 0; W.w_given = bc ? w_given : 0.0f; W.warm = warm;
    W.inv_batch = 1.0f / count_batch; W.soft_count = N->soft_count; W.wstate = N->wstate;
    WgJob& J = W.job[0];
    J = WgJob{};
    J.net = N->actor; J.grad = N->grad_actor; J.m = mA;
    J.ws[0] = s[S_API]; J.rows[0] = batch; J.wmode[0] = 1;
    J.nslots = 1;
    if (bc) { J.ws[1] = s[S_ABC]; J.rows[1] = batch; J.wmode[1] = 2; J.nslots = 2; }
    W.bf16 = N->w2_bf16_all != nullptr;
    if (adam_step > 0) {  // actor.optimizer.step() (+ soft_update of targetActor, + the bf16 image) in the same launch
        J.p = N->actor; J.mom = N->m_actor; J.var = N->v_actor;
        J.target = polyak ? N->target_actor : nullptr;
        J.w2b = N->actor_w2_bf16;
        J.w2f = N->actor_w2_f32i;
        if (N->actor_w2_x9) {
            HX_REQUIRE(!N->actor_w2_bf16 && !N->w2_bf16_all, "hx_hirl_learn: actor_w2_x9 excludes the plain bf16 images (one acting format at a time)");
            J.w2b = N->actor_w2_x9; J.w2b_x9 = 1;
        }
        if (uint16_t* im = N->w2_bf16_all) {
            HX_REQUIRE(!N->actor_w2_bf16 || N->actor_w2_bf16 == im + IM_ACTOR * kImgElems, "hx_hirl_learn: with w2_bf16_all set, actor_w2_bf16 must be NULL or its first image");
            J.w2b = im + IM_ACTOR * kImgElems; J.w2tb = im + IM_ACTOR_T * kImgElems; J.tgt_w2b = im + IM_TA * kImgElems;
        }
        W.ad = make_adam(N, Hy, Hy->lr_actor, adam_step, true);
        launch_wg(W, true, (hipStream_t)stream);
    } else {
        launch_wg(W, false, (hipStream_t)stream);
    }
    HX_CHECK_LAUNCH("hx_hirl_actor_wgrad");
    return 0;
}
int hx_hirl_actor_wgrad(const HxNets* N, const HxHyper* Hy, int32_t batch, int32_t count_batch, int32_t w_kind, float w_given,
                        float warm, void* stream) {
    return actor_wgrad_impl(N, Hy, batch, count_batch, w_kind, w_given, warm, stream, 0, false);
}

/* Stage 2b of a sharded run with ONE exchange for the actor phase: the UNWEIGHTED gradients of the two actor losses and the local soft
 * count go into one message, msg = [dL_rl | dL_bc | count, 0...] (hx_actor_message_floats()); after its all-reduce hx_adam_mixed
 * forms w from the global count and combines.  TD3 (use_bc = 0): dL_rl only. */
int hx_hirl_actor_wgrad_split(const HxNets* N, const HxHyper* Hy, int32_t batch, float* msg, void* stream) {
    HX_REQUIRE(N && Hy && batch > 0 && msg && (reinterpret_cast<uintptr_t>(msg) & 15u) == 0, "hx_hirl_actor_wgrad_split: bad arguments");
    const Mlp mA{13, 4, Hy->no_layernorm ? 1 : 0}, mQ{17, 1, Hy->no_layernorm ? 1 : 0};  // layerNorm = False: HIRL.py:70-80,92-97,135-138
    (void)mA; (void)mQ;
    Slot s[S_COUNT];
    make_slots(N, batch, s);
    const bool bc = Hy->use_bc != 0;
    WgArgs W{};
    W.njobs = bc ? 2 : 1; W.slope = Hy->slope; W.w_kind = 0; W.w_given = 0.0f; W.inv_batch = 1.0f / batch;
    W.soft_count = N->soft_count; W.wstate = N->wstate;
    for (int j = 0; j < W.njobs; ++j) {
        WgJob& J = W.job[j];
        J = WgJob{};
        J.net = N->actor; J.grad = msg + j * mA.padded(); J.m = mA;
        J.ws[0] = s[j == 0 ? S_API : S_ABC]; J.rows[0] = batch; J.wmode[0] = 0; J.nslots = 1;
    }
    W.bf16 = N->w2_bf16_all != nullptr;
    W.soft_count = bc ? N->soft_count : nullptr;
    W.count_out = msg + 2 * mA.padded();  // the message's count word, written by the same launch (was a launch of its own: 3.8 us)
    launch_wg(W, false, (hipStream_t)stream);
    HX_CHECK_LAUNCH("hx_hirl_actor_wgrad_split");
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

/* hx_hirl_learn_sampled in two parts around an env step (include/hirl4ucav.h "front launch"): hx_hirl_front = chooseAction + env step + replay insert of
 * n envs AND launches A and B of the learn() call that follows, as workgroups of ONE launch (hx_front.hip), on the minibatch tiles a predraw left in
 * `batch`; hx_hirl_learn_back = the rest of that call, and (with `next`) the draw + gather of the NEXT front launch's minibatch in its first launch. */
int hx_hirl_front(float* state, int64_t n, int64_t stride, float* obs_io, float* actions, int32_t noise_mode, const float* noise,
                  float sigma, uint64_t seed, uint32_t row0, uint32_t call, float* reward, uint8_t* done, int8_t* success, const HxStepOpts* opts,
                  const HxNets* N, const HxBatch* Bt, const HxHyper* Hy, int32_t actor_phase, int32_t w_kind, const HxFront* front, void* stream) {
    HX_REQUIRE(N && Bt && Hy && front && opts && Bt->batch > 0 && Bt->batch % 16 == 0 && Bt->batch <= kFusedBatchMax && Bt->rows && Bt->noise,
               "hx_hirl_front: the minibatch tiles (a positive multiple of 16 rows, at most 256) and noise[4] must be there");
    HX_REQUIRE(front->flags && front->status && front->epoch >= 1, "hx_hirl_front: flags, status and a 1-based epoch are required");
    const bool bf16 = N->w2_bf16_all != nullptr;  // the bf16 update path: then the acting image is the bf16 one too (bf16 acting beside an fp32 update: separate launches)
    HX_REQUIRE(bf16 ? N->actor_w2_bf16 != nullptr : (!N->actor_w2_bf16 && (N->actor_w2_f32i || N->actor_w2_x9)),
               "hx_hirl_front: fp32 networks with HxNets.actor_w2_f32i / actor_w2_x9, or the bf16 update path (w2_bf16_all) with actor_w2_bf16");
    const int actor_fwd = actor_phase ? (w_kind == 1 ? 2 : 1) : 0;
    FwdArgs FA, FB;
    make_launch_a(N, Bt, Hy, actor_fwd, FA);
    make_launch_b(N, Bt, Hy, actor_fwd, FB);
    const bool x9 = !bf16 && (noise_mode & 32) != 0;  // + 32: the exact-split acting format (HxNets.actor_w2_x9), else fp32 MFMA from HxNets.actor_w2_f32i
    HX_REQUIRE(bf16 || (x9 ? N->actor_w2_x9 != nullptr : N->actor_w2_f32i != nullptr), "hx_hirl_front: the image of the chosen acting format is missing from HxNets");
    noise_mode &= ~32;
    BwdArgs GC;
    const bool with_c = front_has_c(front->with_c != 0);
    HX_REQUIRE(!with_c || front->with_c != 0, "hx_hirl_front: HX_FRONT_C=1 needs a caller that counts its launches with launch C (HxFront.with_c = 1, 2, ...)");
    if (with_c) make_launch_c(N, Bt, Hy, GC);
    return launch_front(N->actor, (x9 || bf16) ? nullptr : N->actor_w2_f32i, x9 ? N->actor_w2_x9 : nullptr, bf16 ? N->actor_w2_bf16 : nullptr, state, n, stride, obs_io, actions, noise_mode, noise, sigma, seed,
                        row0, call, Hy->slope, reward, done, success, *opts, FA, FB, with_c ? &GC : nullptr, *front, (hipStream_t)stream);
}
// the predraw of hx_hirl_learn_back / hx_hirl_critic_grads_back as a device-side description (nothing is launched here)
static int make_predraw(const HxNets* N, const HxBatch* Bt, const HxSample* next, const HxBatch* next_tiles, void* stream, SampleDev* SD) {
    HX_REQUIRE(next_tiles && next_tiles->batch == Bt->batch && next_tiles->rows && next_tiles->noise && next_tiles->rows != Bt->rows && next_tiles->noise != Bt->noise,
               "hx_hirl_learn_back: the next minibatch needs tiles of its own (this call still reads the current ones)");
    (void)N;
    bool fused = false;
    if (int rc = prepare_draw(next, Bt->batch, const_cast<float*>(next_tiles->rows), const_cast<float*>(next_tiles->bc_rows), const_cast<float*>(next_tiles->noise),
                              stream, SD, &fused, /*launch_now=*/false)) return rc;
    HX_REQUIRE(fused, "hx_hirl_learn_back: the predraw covers minibatches of at most 256 rows");
    return 0;
}
/* The sharded rank's form (as hx_hirl_critic_grads after a front launch): launches C and D without the optimizer step; grad_critic is ready for the exchange. */
int hx_hirl_critic_grads_back(const HxNets* N, const HxBatch* Bt, const HxHyper* Hy, const HxSample* next, const HxBatch* next_tiles, int32_t c_in_front,
                              void* stream) {
    HX_REQUIRE(N && Bt && Hy && Bt->batch > 0 && Bt->batch % 16 == 0, "hx_hirl_critic_grads_back: batch must be a positive multiple of 16");
    SampleDev SD{};
    if (next) if (int rc = make_predraw(N, Bt, next, next_tiles, stream, &SD)) return rc;
    return critic_back(N, Bt, Hy, stream, 0, false, next ? &SD : nullptr, front_has_c(c_in_front != 0));
}
int hx_hirl_learn_back(const HxNets* N, const HxBatch* Bt, const HxHyper* Hy, int32_t critic_step, int32_t actor_phase, int32_t actor_step, int32_t do_polyak,
                       int32_t w_kind, float w_given, float warm, const HxSample* next, const HxBatch* next_tiles, int32_t c_in_front, void* stream) {
    HX_REQUIRE(N && Bt && Hy && Bt->batch > 0 && Bt->batch % 16 == 0, "hx_hirl_learn_back: batch must be a positive multiple of 16");
    HX_REQUIRE(critic_step >= 1 && (!actor_phase || actor_step >= 1), "hx_hirl_learn_back: Adam steps are 1-based");
    SampleDev SD{};
    if (next) if (int rc = make_predraw(N, Bt, next, next_tiles, stream, &SD)) return rc;
    int rc = critic_back(N, Bt, Hy, stream, critic_step, do_polyak != 0, next ? &SD : nullptr, front_has_c(c_in_front != 0));
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
    const Mlp mA{13, 4, Hy->no_layernorm ? 1 : 0}, mQ{17, 1, Hy->no_layernorm ? 1 : 0};  // layerNorm = False: HIRL.py:70-80,92-97,135-138
    (void)mA; (void)mQ;
    const int B = Bt->batch;
    Slot s[S_COUNT];
    make_slots(N, B, s);
    const RowSrc bcsrc{Bt->bc_rows, nullptr, nullptr, 0, 32};
    {
        FwdArgs F{};
        F.njobs = 1; F.slope = Hy->slope;
        F.zero_f = N->losses + 1; F.zero_nf = 4;
        F.job[0] = FwdJob{N->actor, mA, bcsrc, 0, 0, Head{}, nullptr, 0.f, s[S_ABC], B, 1, IM_ACTOR};
        F.images = N->w2_bf16_all;
        launch_fwd(F, st);
    }
    {
        BwdArgs G{};
        G.njobs = 1; G.slope = Hy->slope; G.inv_batch = 1.0f / B; G.losses = N->losses; G.soft_count = N->soft_count;
        BwdJob& J = G.job[0];
        J = BwdJob{};
        J.net = N->actor; J.m = mA; J.ws = s[S_ABC]; J.rows = B; J.mode = BM_ACTOR_BC; J.src = bcsrc; J.lambda = 1.0f;
        J.img_t = IM_ACTOR_T;
        G.images = N->w2_bf16_all;
        launch_bwd(2, G, st);
    }
    {
        WgArgs W{};
        W.njobs = 1; W.slope = Hy->slope; W.w_kind = 0; W.w_given = 0.f; W.inv_batch = 1.0f / B;
        W.soft_count = N->soft_count; W.wstate = N->wstate;
        WgJob& J = W.job[0];
        J = WgJob{};
        J.net = N->actor; J.grad = N->grad_actor; J.m = mA; J.ws[0] = s[S_ABC]; J.rows[0] = B; J.nslots = 1; J.wmode[0] = 0;
        W.bf16 = N->w2_bf16_all != nullptr;
        launch_wg(W, false, st);
    }
    HX_CHECK_LAUNCH("hx_bc_train_actor");
    return hx_adam(N, Hy, 2, step, 1.0f, 0, 0.0f, 0.0f, B, stream);
}

}  // extern "C"
