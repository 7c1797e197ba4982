// hx_front.hip — the FRONT launch (gfx950): the env step of a vector loop and the first two launches of the learn() call that follows it, as ONE launch.
//   chooseAction + HarfangEnv.step + memory.store   hirl/train_all.py:343-348   -> the acting workgroups (act_fused_body, hx_act_body.h: 32 rows each)
//   Agent.learn up to the TD target's inputs        hirl/agents/HIRL.py:259-272 -> launch A (targetActor(s'), Q1/Q2(s, a)) and launch B (targetCritic
//                                                                                  Q1/Q2) as further workgroups (fwd_l2_body, hx_fwd_body.h)
// Why: every kernel of the step is 1,024 threads x up to 128 registers — one workgroup per CU — so a launch is as long as its longest workgroup
// chain.  At 4,096 envs the acting launch is 256 workgroups x 16 rows (20.5 us) and launches A + B another 17 us behind it, although they need
// nothing the env step computes (the draw aside).  With 32 rows per acting workgroup the step takes 128 CUs for 22-27 us (tools/ubench/merge_probe.sh,
// x9_32row.sh) and launches A (192-256 workgroups) and B (128-192) run on the other 128 CUs in that shadow.  B needs A's target-actor rows: it waits
// for them IN the launch, per row tile (FrontSync, hx_fwd_body.h) — off the critical path here, because A is through long before the acting
// workgroups are.  Workgroups are dispatched in index order: acting first (the longest), then the update's jobs in FrontCtl::order.
// CU time is what runs out in the shadow (tools/ubench/front_spans.py), so the minibatch is NOT drawn here: the previous hx_hirl_learn_back left its
// tiles behind (predraw_wg, one more workgroup of its wgrad launch), launch B runs in 64-column workgroups whatever its job count, and the target actor's
// head is evaluated once per row tile (FRONT == 1 in hx_fwd_body.h) instead of in every target-critic workgroup.
// Beyond one round of 32-row acting workgroups the acting role is a PERSISTENT kernel on part of the CUs (actp_front_kernel: bf16, weight-stationary;
// actps_front_kernel: fp32 in the exact-split format, one 64-row pass per workgroup).
// Launch C (the critics' backward) can ride too (HxFront.with_c: bwd_l2_body<0, ..., FRONT = 3>, hx_bwd_body.h, waiting in-launch for the jobs of A and B it
// reads): worth it only where the acting workgroups leave the other CUs time for 256 more workgroups — the streaming role at 8,192 envs (hx_hirl.hip).
// The draw's meaning changes (it cannot see this step's inserts and must not read the slots they overwrite): include/hirl4ucav.h hx_hirl_front.
#include <hip/hip_ext.h>

#include "hx_act_body.h"
#include "hx_actp_body.h"
#include "hx_bwd_body.h"

using namespace hxnn;
using namespace hxu;
using namespace hxact;

namespace {

struct FrontCtl {
    int n_act;         // acting workgroups (32 rows each)
    int per;           // workgroups per job of launches A and B (row tiles x column workgroups; both in 64-column workgroups)
    // dispatch order of the update's jobs behind the acting workgroups: job of launch A (0..3) or 16 + job of launch B.  The 128 CUs of the shadow take
    // them round by round (two jobs per round at B = 128); the launches' own order (A's jobs, then B's) is as good as any (tools/ubench/front_order.sh)
    unsigned char order[8];
    FrontSync sync;
    // launch C (the critics' backward: bwd_l2_body<0, ..., 3>) as the LAST workgroups of the launch — n_fwd = per x (jobs of A and B) forward workgroups in
    // front of them, c_per per TD job (0: launch C stays a launch of its own).  They wait in-launch for what they read (hx_bwd_body.h)
    int n_fwd, c_per;
};

// Launch B runs in 64-column workgroups whatever its job count (a launch of its own takes 32-column workgroups for two nets: twice the workgroups,
// each with half the MFMA work — right for an empty chip, wrong in the shadow of the acting workgroups, where CU time is what runs out:
// tools/ubench/front_spans.py).  The K-split of a column tile differs between the two tilings (4 against 8 partial sums), hence the last bits of the
// target critics' z2: hx_debug_set_fwd_nt(64) gives the separate launches the same tiling (tests/test_front_gpu.py).
// X3: the acting workgroups multiply through the exact three-way bf16 split (six partial products, hx_act.h HX_X9_TERMS) (the engine's "f32x9" acting format, H.w2b = the hi | mid | lo images) instead of
// fp32 MFMA from the fp32 image: 21.7 against 26.8 us for the 128 workgroups of 4,096 envs (tools/ubench/x9_32row.sh) — a shorter shadow, but the
// acting workgroups are the launch's longest
// BF16: the bf16 update path (HxNets.w2_bf16_all) with the bf16 acting kernel — both on v_mfma_f32_16x16x32_bf16, as their launches of their own
template <bool RELU, bool X3, bool BF16 = false>
__global__ __launch_bounds__(kWide) void act_front_kernel(ActFusedArgs H, FwdArgsC FA, FwdArgsC FB, FrontCtl C, BwdArgsC GC) {
    static_assert(!(X3 && BF16), "one acting format");
    constexpr int BNT = kNT;
    typedef ActLds<2, true, BF16, !X3 && !BF16, X3> LdsAct;
    typedef FwdLds<kNT, false, BF16> LdsA;
    typedef FwdLds<BNT, false, BF16> LdsB;
    __shared__ union {
        LdsAct act;
        LdsA a;
        LdsB b;
        BwdLds<0, BF16> c;
    } u;
    int b = (int)blockIdx.x;
    if (b < C.n_act) {
        act_fused_body<2, false, true, BF16, RELU, !X3 && !BF16, X3>(H, b, u.act);
        return;
    }
    b -= C.n_act;
    if (b >= C.n_fwd) {
        b -= C.n_fwd;
        const int cj = b / C.c_per;
        bwd_l2_body<0, RELU, BF16, 3>(GC, b - cj * C.c_per, cj, u.c, C.sync);
        return;
    }
    const int k = b / C.per, bx = b - k * C.per, job = C.order[k];
    if (job < 16) fwd_l2_body<kNT, RELU, false, BF16, 1>(FA, NoSample{}, bx, job, u.a, C.sync);
    else fwd_l2_body<BNT, RELU, false, BF16, 2>(FB, NoSample{}, bx, job - 16, u.b, C.sync);
}

// Beyond 8,192 envs (bf16): the acting role is the PERSISTENT weight-stationary kernel of hx_actp.hip on fewer workgroups than CUs — 16,384 envs on 171
// workgroups of three 32-row tiles take 28.4 us against 25.5 on 256 workgroups of two (tools/ubench/actp_time.py with HX_ACT_PERSIST_WGS) — and the
// update's workgroups run on the CUs that leaves free.  tiles: row tiles per acting workgroup.
template <bool RELU>
__global__ __launch_bounds__(kWide) void actp_front_kernel(ActFusedArgs H, FwdArgsC FA, FwdArgsC FB, FrontCtl C, int tiles, BwdArgsC GC) {
    typedef ActpLds<true> LdsAct;
    typedef FwdLds<kNT, false, true> LdsF;
    __shared__ union {
        LdsAct act;
        LdsF f;
        BwdLds<0, true> c;
    } u;
    int b = (int)blockIdx.x;
    if (b < C.n_act) {
        act_persist_bf16_body<true, RELU>(H, tiles, b, C.n_act, u.act);
        return;
    }
    b -= C.n_act;
    if (b >= C.n_fwd) {
        b -= C.n_fwd;
        const int cj = b / C.c_per;
        bwd_l2_body<0, RELU, true, 3>(GC, b - cj * C.c_per, cj, u.c, C.sync);
        return;
    }
    const int k = b / C.per, bx = b - k * C.per, job = C.order[k];
    if (job < 16) fwd_l2_body<kNT, RELU, false, true, 1>(FA, NoSample{}, bx, job, u.f, C.sync);
    else fwd_l2_body<kNT, RELU, false, true, 2>(FB, NoSample{}, bx, job - 16, u.f, C.sync);
}

// fp32 in the exact-split format, 8,192 .. 12,288 envs: the acting role is the STREAMING persistent kernel of hx_actp.hip, one 64-row pass per workgroup — 128 to
// 192 workgroups (37.6 us for a pass) and the rest of the CUs for the update; the per-tile role at 8,192 envs is 256 workgroups with no CU to spare (41.1 us)
template <bool RELU>
__global__ __launch_bounds__(kWide) void actps_front_kernel(ActFusedArgs H, FwdArgsC FA, FwdArgsC FB, FrontCtl C, int tiles, BwdArgsC GC) {
    typedef ActpsLds<1, false, true> LdsAct;
    typedef FwdLds<kNT, false, false> LdsF;
    __shared__ union {
        LdsAct act;
        LdsF f;
        BwdLds<0, false> c;
    } u;
    int b = (int)blockIdx.x;
    if (b < C.n_act) {
        act_persist_stream_body<1, false, true, RELU>(H, tiles, b, C.n_act, u.act);
        return;
    }
    b -= C.n_act;
    if (b >= C.n_fwd) {
        b -= C.n_fwd;
        const int cj = b / C.c_per;
        bwd_l2_body<0, RELU, false, 3>(GC, b - cj * C.c_per, cj, u.c, C.sync);
        return;
    }
    const int k = b / C.per, bx = b - k * C.per, job = C.order[k];
    if (job < 16) fwd_l2_body<kNT, RELU, false, false, 1>(FA, NoSample{}, bx, job, u.f, C.sync);
    else fwd_l2_body<kNT, RELU, false, false, 2>(FB, NoSample{}, bx, job - 16, u.f, C.sync);
}

// SAC (explore + env step + insert, then SacAgent.learn): the policy's acting launch beyond 8,192 envs is the streaming persistent kernel on every CU (one
// 64-row pass each at 16,384 envs); the FIRST forward launch of learn() — policy(s'), policy(s), Q1/Q2(s, a): no dependency inside — rides behind it as further
// workgroups that start as the acting ones leave: one boundary less and no draw in its workgroups (the tiles were pre-drawn).  MODE as act_persist_stream_kernel.
template <int MODE>
__global__ __launch_bounds__(kWide) void actps_sac_front_kernel(ActFusedArgs H, FwdArgsC FA, FrontCtl C, int tiles) {
    typedef ActpsLds<MODE, true, true> LdsAct;
    typedef FwdLds<kNT, false, false> LdsF;
    __shared__ union {
        LdsAct act;
        LdsF f;
    } u;
    int b = (int)blockIdx.x;
    if (b < C.n_act) {
        act_persist_stream_body<MODE, true, true, true>(H, tiles, b, C.n_act, u.act);
        return;
    }
    b -= C.n_act;
    const int k = b / C.per, bx = b - k * C.per;
    fwd_l2_body<kNT, true, false, false, 0>(FA, NoSample{}, bx, (int)C.order[k], u.f, FrontSync{});
}

}  // namespace

namespace hxu {

int launch_front_sac(const ActFusedArgs& H, const FwdArgs& FA, hipStream_t st) {
    HX_REQUIRE(H.state && H.o.ring && H.rows > kFuseEnvMax && (H.w2f || (H.w2b && H.x9)), "hx_sac_front: more than 8,192 envs with a replay ring, the policy's W2 from an image");
    HX_REQUIRE(!FA.sample && FA.njobs >= 3 && FA.njobs <= 8 && FA.slope == 0.0f, "hx_sac_front: the first forward launch of learn() on finished minibatch tiles");
    FwdArgsC CA{};
    for (int j = 0; j < FA.njobs; ++j) { CA.job[j] = pack_fwd(FA.job[j]); CA.job[j].slope = FA.slope; }
    CA.slope = FA.slope; CA.zero_nf = FA.zero_nf; CA.zero_f = FA.zero_f; CA.zero_i = FA.zero_i; CA.images = nullptr; CA.rowmap = 1;
    const int rows = FA.job[0].rows, tiles = (rows + RT - 1) / RT;
    HX_REQUIRE(tiles * FA.njobs < 128, "hx_sac_front: minibatches of at most 256 rows");
    FrontCtl C{};
    C.per = tiles * (H2 / kNT);
    for (int j = 0; j < FA.njobs; ++j) C.order[j] = (unsigned char)j;
    const int npass = (H.rows + 4 * RT - 1) / (4 * RT), per_wg = (npass + 255) / 256;
    C.n_act = (npass + per_wg - 1) / per_wg;
    const dim3 grid((unsigned)(C.n_act + C.per * FA.njobs));
    const bool x9 = H.w2b && H.x9;
#define HX_SACF(MODE_) do { \
        if (H.o.ev_start && H.o.ev_stop) hipExtLaunchKernelGGL((actps_sac_front_kernel<MODE_>), grid, dim3(kWide), 0, st, (hipEvent_t)H.o.ev_start, (hipEvent_t)H.o.ev_stop, 0, H, CA, C, per_wg); \
        else hipLaunchKernelGGL((actps_sac_front_kernel<MODE_>), grid, dim3(kWide), 0, st, H, CA, C, per_wg); } while (0)
    if (x9) HX_SACF(1); else HX_SACF(0);
#undef HX_SACF
    HX_CHECK_LAUNCH("hx_sac_front");
    return 0;
}

int launch_front(const float* actor, const float* w2f, const uint16_t* w2x, const uint16_t* w2b, float* state, int64_t n, int64_t stride, float* obs_io, float* actions, int32_t noise_mode,
                 const float* noise, float sigma, uint64_t seed, uint32_t row0, uint32_t call, float slope, float* reward, uint8_t* done, int8_t* success,
                 const HxStepOpts& o, const FwdArgs& FA, const FwdArgs& FB, const BwdArgs* GC, const HxFront& front, hipStream_t st) {
    HX_REQUIRE(actor && (w2f || w2x || w2b) && ((reinterpret_cast<uintptr_t>(w2f) | reinterpret_cast<uintptr_t>(w2x) | reinterpret_cast<uintptr_t>(w2b)) & 15u) == 0,
               "hx_hirl_front: the actor and a 16-byte aligned image of its W2 (HxNets.actor_w2_x9, actor_w2_f32i or actor_w2_bf16)");
    HX_REQUIRE(!w2b == !FA.images && FA.images == FB.images, "hx_hirl_front: the bf16 acting image goes with the bf16 update path (HxNets.w2_bf16_all), and only with it");
    const Mlp mA{13, 4, (noise_mode & 16) ? 1 : 0};  // + 16: layerNorm = False (as hx_actor_act_step)
    noise_mode &= 15;
    HX_REQUIRE(noise_mode >= 0 && noise_mode <= 3 && (noise || (noise_mode != 1 && noise_mode != 2)), "hx_hirl_front: bad noise mode");
    if (int rc = check_step_args(state, n, stride, obs_io, actions, reward, done, success, o, "hx_hirl_front")) return rc;
    // bf16: beyond how many envs the acting role is the persistent kernel (tuning knob).  Measured, us per step with the per-tile / the persistent acting role:
    // 4,096 envs 44.0 / 46.8; 8,192 envs 56.2 / 50.7 (there the per-tile role is 256 workgroups — no CU left for the update — the persistent one 128 of two tiles)
    static const int64_t persist_rows = getenv("HX_FRONT_PERSIST_ROWS") ? atoll(getenv("HX_FRONT_PERSIST_ROWS")) : 4096;
    // fp32, exact split: from how many envs on the acting role is the streaming persistent kernel, one 64-row pass per workgroup (tuning knob)
    static const int64_t stream_rows = getenv("HX_FRONT_STREAM_ROWS") ? atoll(getenv("HX_FRONT_STREAM_ROWS")) : 8192;
    const bool stream = w2x && o.ring && n >= stream_rows;
    const bool persistent = !stream && (n > kFuseEnvMax || (w2b && o.ring && n > persist_rows));
    HX_REQUIRE(!persistent || (w2b && o.ring), "hx_hirl_front: at most 8,192 envs per launch (one round of 32-row acting workgroups); with a replay ring "
                                               "any number in the exact-split format and in bf16 (persistent acting workgroups)");
    HX_REQUIRE(!FA.sample && FA.njobs >= 3 && FB.njobs >= 2, "hx_hirl_front: launch A reads finished minibatch tiles");
    ActFusedArgs H{actor, mA, obs_io, (int)n, slope, actions, (noise_mode == 1 || noise_mode == 2) ? noise : nullptr,
                   noise_mode == 2, noise_mode == 3 ? sigma : 0.0f, 0, seed, row0, call, state, stride, reward, done, success, o,
                   o.cap > 0 ? 1.0 / (double)o.cap : 0.0, w2b ? w2b : w2x, (w2x || w2b) ? nullptr : w2f, w2x ? 1 : 0};
    FwdArgsC CA{}, CB{};
    for (int j = 0; j < FA.njobs; ++j) { CA.job[j] = pack_fwd(FA.job[j]); CA.job[j].slope = FA.slope; }
    for (int j = 0; j < FB.njobs; ++j) { CB.job[j] = pack_fwd(FB.job[j]); CB.job[j].slope = FB.slope; }
    // Which of the update's workgroups share an XCD (blocks b and b + 8 do, under the round-robin placement observed on gfx950: speed only, never
    // correctness).  1 (default): a ROW TILE's workgroups — the launches behind this one find its rows in their own L2, but every XCD pulls every network
    // of launches A and B through its L2 (36 MB of fetches per launch against 5 MB algorithmic, profiles/pmc_env_traffic.json front_4096).  0
    // (HX_FRONT_ROWMAP=0, tools/ubench/front_rowmap_ab.sh): a 64-COLUMN slice's workgroups — each W2 slice enters ONE L2 — measured, see docs/LEVERS.md
    static const int front_rowmap = getenv("HX_FRONT_ROWMAP") ? atoi(getenv("HX_FRONT_ROWMAP")) : 1;
    CA.slope = FA.slope; CA.zero_nf = FA.zero_nf; CA.zero_f = FA.zero_f; CA.zero_i = FA.zero_i; CA.images = FA.images; CA.rowmap = front_rowmap & 1;
    CB.slope = FB.slope; CB.zero_nf = FB.zero_nf; CB.zero_f = FB.zero_f; CB.zero_i = FB.zero_i; CB.images = FB.images; CB.rowmap = front_rowmap & 1;
    const int rows = FA.job[0].rows, tiles = (rows + RT - 1) / RT;
    HX_REQUIRE(tiles * FB.njobs < 128 && FA.njobs + FB.njobs <= 8, "hx_hirl_front: minibatches of at most 256 rows, at most 8 forward jobs");
    FrontCtl C{};
    C.n_act = (int)((n + 2 * RT - 1) / (2 * RT));
    C.per = tiles * (H2 / kNT);
    {   // launch A's job 0 is the target actor, launch B's jobs 0 and 1 the target critics that feed on it (make_launch_a / make_launch_b)
        int k = 0;
        // HX_FRONT_ORDER (tuning knob, tools/ubench/front_order.sh): 0 = launch A's jobs, then launch B's (default); 1 = the target critics right behind the
        // target actor and Q1, the rest last (+1.1 us per step: measured); 2 = the critic call's five jobs, then the actor call's extras (no difference)
        static const int variant = getenv("HX_FRONT_ORDER") ? atoi(getenv("HX_FRONT_ORDER")) : 0;
        if (variant == 1) {
            C.order[k++] = 0; C.order[k++] = 1;
            for (int j = 0; j < FB.njobs; ++j) C.order[k++] = (unsigned char)(16 + j);
            for (int j = 2; j < FA.njobs; ++j) C.order[k++] = (unsigned char)j;
        } else if (variant == 2) {
            C.order[k++] = 0; C.order[k++] = 1; C.order[k++] = 2; C.order[k++] = 16; C.order[k++] = 17;
            for (int j = 3; j < FA.njobs; ++j) C.order[k++] = (unsigned char)j;
            for (int j = 2; j < FB.njobs; ++j) C.order[k++] = (unsigned char)(16 + j);
        } else {
            for (int j = 0; j < FA.njobs; ++j) C.order[k++] = (unsigned char)j;
            for (int j = 0; j < FB.njobs; ++j) C.order[k++] = (unsigned char)(16 + j);
        }
    }
    static_assert(kNT == 64, "launch B in launch A's tiling");
    {   // a row tile's counter advances by (column workgroups of the target actor's job) + 1 per front launch
        const uint32_t per = (uint32_t)(H2 / kNT) + 1u;
        HX_REQUIRE(FB.job[0].act_mode == 1 && FB.job[0].prev.net == FA.job[0].net, "hx_hirl_front: launch B's first job feeds on launch A's first");
        C.sync = FrontSync{front.flags, (front.epoch - 1u) * per + (per - 1u), front.epoch * per, front.status, FB.job[0].noise, FB.job[0].noise_clamp, 0u, 0u, 0u};
    }
    HX_REQUIRE(tiles <= 16, "hx_hirl_front: at most 16 row tiles (HxFront.flags)");
    C.n_fwd = C.per * (FA.njobs + FB.njobs);
    BwdArgsC CG{};
    if (GC) {  // launch C rides too: its two TD jobs behind the forward workgroups, waiting for launch A's critic jobs (1, 2) and launch B's target critics (0, 1)
        HX_REQUIRE(GC->njobs == 2 && GC->job[0].mode == BM_CRITIC_TD && GC->job[0].rows == rows && FA.job[1].ws.z2 == GC->job[0].ws.z2 && FA.job[2].ws.z2 == GC->job[1].ws.z2 &&
                   FB.job[0].ws.z2 == GC->job[0].t1.ws.z2 && FB.job[1].ws.z2 == GC->job[0].t2.ws.z2, "hx_hirl_front: launch C's TD jobs read launch A's jobs 1, 2 and launch B's jobs 0, 1");
        for (int j = 0; j < 2; ++j) CG.job[j] = pack_bwd(GC->job[j], *GC);
        CG.images = GC->images; CG.rowmap = 1;
        C.c_per = ((rows + RT / 2 - 1) / (RT / 2)) * kColWgB;
        const uint32_t cw = (uint32_t)(H2 / kNT);
        HX_REQUIRE(front.with_c >= 1u, "hx_hirl_front: HxFront.with_c counts the front launches that carry launch C (1, 2, ...)");
        C.sync.with_c = 1u; C.sync.c_target = front.with_c * cw; C.sync.t_target = front.with_c * 2u * cw;  // (their counters advance only in launches with C)
    }
    const unsigned n_c = GC ? 2u * (unsigned)C.c_per : 0u;
    const bool relu = slope == 0.0f;
    if (stream) {  // up to 16,384 envs one 64-row pass per acting workgroup (the CUs they leave serve the update beside them); beyond, passes over every CU and the
                   // update's workgroups behind them (two boundaries less)
        const int npass = (int)((n + 4 * RT - 1) / (4 * RT)), per_wg = (npass + 255) / 256;
        C.n_act = (npass + per_wg - 1) / per_wg;
        const dim3 sgrid((unsigned)(C.n_act + C.n_fwd) + n_c);
        if (o.ev_start && o.ev_stop) {
            if (relu) hipExtLaunchKernelGGL((actps_front_kernel<true>), sgrid, dim3(kWide), 0, st, (hipEvent_t)o.ev_start, (hipEvent_t)o.ev_stop, 0, H, CA, CB, C, per_wg, CG);
            else hipExtLaunchKernelGGL((actps_front_kernel<false>), sgrid, dim3(kWide), 0, st, (hipEvent_t)o.ev_start, (hipEvent_t)o.ev_stop, 0, H, CA, CB, C, per_wg, CG);
        } else {
            if (relu) hipLaunchKernelGGL((actps_front_kernel<true>), sgrid, dim3(kWide), 0, st, H, CA, CB, C, per_wg, CG);
            else hipLaunchKernelGGL((actps_front_kernel<false>), sgrid, dim3(kWide), 0, st, H, CA, CB, C, per_wg, CG);
        }
        HX_CHECK_LAUNCH("hx_hirl_front");
        return 0;
    }
    if (persistent) {
        // up to 32,768 envs two thirds of the CUs act (ceil(tiles / 176) row tiles per workgroup) and the rest serve the update beside them; beyond, the acting
        // workgroups take every CU (the update is a small share there) and the update's workgroups start as the first of them leave: two boundaries less
        static const int wide_wgs = getenv("HX_FRONT_PERSIST_WIDE_WGS") ? atoi(getenv("HX_FRONT_PERSIST_WIDE_WGS")) : 256;  // tuning knob
        static const int part_wgs = getenv("HX_FRONT_PERSIST_WGS") ? atoi(getenv("HX_FRONT_PERSIST_WGS")) : 176;  // tuning knob (tools/ubench/front_persist_wgs.sh)
        const int want = n <= 32768 ? part_wgs : wide_wgs;
        const int ntiles = (int)((n + 2 * RT - 1) / (2 * RT)), tiles_per_wg = (ntiles + want - 1) / want;
        C.n_act = (ntiles + tiles_per_wg - 1) / tiles_per_wg;
        const dim3 pgrid((unsigned)(C.n_act + C.n_fwd) + n_c);
        if (o.ev_start && o.ev_stop) {
            if (relu) hipExtLaunchKernelGGL((actp_front_kernel<true>), pgrid, dim3(kWide), 0, st, (hipEvent_t)o.ev_start, (hipEvent_t)o.ev_stop, 0, H, CA, CB, C, tiles_per_wg, CG);
            else hipExtLaunchKernelGGL((actp_front_kernel<false>), pgrid, dim3(kWide), 0, st, (hipEvent_t)o.ev_start, (hipEvent_t)o.ev_stop, 0, H, CA, CB, C, tiles_per_wg, CG);
        } else {
            if (relu) hipLaunchKernelGGL((actp_front_kernel<true>), pgrid, dim3(kWide), 0, st, H, CA, CB, C, tiles_per_wg, CG);
            else hipLaunchKernelGGL((actp_front_kernel<false>), pgrid, dim3(kWide), 0, st, H, CA, CB, C, tiles_per_wg, CG);
        }
        HX_CHECK_LAUNCH("hx_hirl_front");
        return 0;
    }
    const dim3 grid((unsigned)(C.n_act + C.n_fwd) + n_c);
#define HX_FRONT(RELU_, X3_, BF16_) do { \
        if (o.ev_start && o.ev_stop) hipExtLaunchKernelGGL((act_front_kernel<RELU_, X3_, BF16_>), grid, dim3(kWide), 0, st, (hipEvent_t)o.ev_start, (hipEvent_t)o.ev_stop, 0, H, CA, CB, C, CG); \
        else hipLaunchKernelGGL((act_front_kernel<RELU_, X3_, BF16_>), grid, dim3(kWide), 0, st, H, CA, CB, C, CG); } while (0)
    if (w2b) { if (relu) HX_FRONT(true, false, true); else HX_FRONT(false, false, true); }
    else if (relu) { if (w2x) HX_FRONT(true, true, false); else HX_FRONT(true, false, false); }
    else { if (w2x) HX_FRONT(false, true, false); else HX_FRONT(false, false, false); }
#undef HX_FRONT
    HX_CHECK_LAUNCH("hx_hirl_front");
    return 0;
}

}  // namespace hxu

HX_DEFINE_DEBUG_COLLECTORS(front, 0, 80)
