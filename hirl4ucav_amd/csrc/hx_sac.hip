// hx_sac.hip — SacAgent.learn (hirl/agents/SAC/agent.py:276-414, the non-imitative branch train_sac.py uses) on the shared fwd_l2 / bwd_l2<3> /
// wgrad machinery, plus the small per-row kernels around it (gfx950).
#include <cmath>

#include "hx_act.h"

using namespace hxnn;
using namespace hxu;

namespace {

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
// One launch serves up to TWO evaluations (policy.sample(s') for the TD target and policy.sample(s) for the policy loss: both read z2 rows of
// the first forward launch and the policy as it is before this learn()'s steps) and, behind them, the soft_update of the target critics
// that SacAgent.learn does first on every target_update_interval-th call (agent.py:278-279): nothing reads the targets before the NEXT launch.
struct GaussLaunch {
    GaussArgs g[2];
    int nb0;       // workgroups of g[0]; g[1] follows (rows 0: none)
    int nb_gauss;  // workgroups of g[0] + g[1]; Polyak blocks follow
    float* target; const float* source; int n; float tau;  // soft_update (n = 0: none)
};
__device__ __forceinline__ void gauss_head_rows(const GaussArgs& A, int block);
__global__ __launch_bounds__(kThreads) void gauss_head_kernel(GaussLaunch L) {
    const int b = blockIdx.x;
    if (b >= L.nb_gauss) {  // target <- (1 - tau) target + tau source, 16 B per lane (the arithmetic of polyak_kernel)
        const int i = ((b - L.nb_gauss) * kThreads + threadIdx.x) * 4;
        if (i + 4 <= L.n) {
            float4 t4 = *reinterpret_cast<const float4*>(L.target + i);
            const float4 s4 = *reinterpret_cast<const float4*>(L.source + i);
            t4.x = polyak_update(t4.x, s4.x, L.tau); t4.y = polyak_update(t4.y, s4.y, L.tau);
            t4.z = polyak_update(t4.z, s4.z, L.tau); t4.w = polyak_update(t4.w, s4.w, L.tau);
            *reinterpret_cast<float4*>(L.target + i) = t4;
        } else {
            for (int c = 0; i + c < L.n; ++c) L.target[i + c] = polyak_update(L.target[i + c], L.source[i + c], L.tau);
        }
        return;
    }
    if (b < L.nb0) gauss_head_rows(L.g[0], b);
    else gauss_head_rows(L.g[1], b - L.nb0);
}
__device__ __forceinline__ void gauss_head_rows(const GaussArgs& A, int block) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = block * 4 + wave;
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
        float d_mean, d_logstd;
        sac_policy_dout(lane == 0 ? da[0] : lane == 1 ? da[1] : lane == 2 ? da[2] : da[3], x[lane], x[4 + lane], x[8 + lane], alpha * A.inv_batch, d_mean, d_logstd);
        A.pol.dout[(size_t)r * OW + lane] = d_mean;
        A.pol.dout[(size_t)r * OW + 4 + lane] = d_logstd;
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

}  // namespace

extern "C" {

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
static void launch_gauss(const GaussArgs& g0, const GaussArgs* g1, float* target, const float* source, int n, float tau, hipStream_t st) {
    GaussLaunch L{};
    L.g[0] = g0;
    L.nb0 = (g0.rows + 3) / 4;
    L.nb_gauss = L.nb0;
    if (g1) { L.g[1] = *g1; L.nb_gauss += (g1->rows + 3) / 4; }
    L.target = target; L.source = source; L.n = n; L.tau = tau;
    const int nbp = n > 0 ? (n / 4 + kThreads) / kThreads : 0;
    hipLaunchKernelGGL(gauss_head_kernel, dim3((unsigned)(L.nb_gauss + nbp)), dim3(kThreads), 0, st, L);
}
static void sac_slots(const HxSacNets* N, int B, Slot* s) {
    for (int i = 0; i < S_COUNT; ++i) s[i] = carve_slot(N->ws + (size_t)i * kSlotFloats * B, B);
}

/* Critic half of SacAgent.learn (SAC/agent.py:278-313): [Polyak of the target critics first when polyak_first], a', H' =
 * policy.sample(s') with eps_next, y = r + (1 - d) gamma (min Q_target(s', a') + alpha H'), q1_loss / q2_loss -> losses[0..1],
 * grad_critic.  Also evaluates policy(s) for the policy half.  Follow with hx_sac_adam(which = 0). */
// one_call (hx_sac_learn): the Polyak step rides behind the Gaussian heads' workgroups instead of in a launch of its own, and policy.sample(s)
// of the policy half is evaluated beside policy.sample(s') — the same arithmetic on the same values, two launches less
// launch 1 of SacAgent.learn: policy(s'), policy(s), Q1/Q2(s, a)
static void sac_launch_1(const HxSacNets* N, const HxSacBatch* Bt, FwdArgs& F) {
    const int B = Bt->batch;
    Slot s[S_COUNT];
    sac_slots(N, B, s);
    const RowSrc src{Bt->rows, nullptr, nullptr, 0, 32};
    const float* q1 = N->critic; const float* q2 = N->critic + kQs.padded();
    F = FwdArgs{};
    F.njobs = 4; F.slope = 0.0f;
    F.zero_f = N->losses; F.zero_nf = 5;
    F.job[0] = FwdJob{N->policy, kPolicy, src, 17, 0, Head{}, nullptr, 0.f, s[SS_PN], B, 0};
    F.job[1] = FwdJob{N->policy, kPolicy, src, 0, 0, Head{}, nullptr, 0.f, s[SS_PC], B, 1};
    F.job[2] = FwdJob{q1, kQs, src, 0, 0, Head{}, nullptr, 0.f, s[SS_Q1], B, 1};
    F.job[3] = FwdJob{q2, kQs, src, 0, 0, Head{}, nullptr, 0.f, s[SS_Q2], B, 1};
}
// skip_first: launch 1 has run already (as workgroups of hx_sac_front's launch)
static int sac_critic_grads_impl(const HxSacNets* N, const HxSacBatch* Bt, const HxHyper* Hy, const HxSample* S, int32_t polyak_first, void* stream,
                                 int adam_step = 0, bool one_call = false, bool skip_first = false) {
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
    if (polyak_first && !one_call)  // soft_update(critic_target, critic) BEFORE the update, agent.py:278-279
        launch_polyak(N->target_critic, N->critic, nq, Hy->tau, nullptr, nullptr, 0, st);
    const float* q1 = N->critic; const float* q2 = N->critic + kQs.padded();
    const float* t1 = N->target_critic; const float* t2 = N->target_critic + kQs.padded();
    if (!skip_first) {   // policy(s'), policy(s), Q1/Q2(s, a)
        FwdArgs F;
        sac_launch_1(N, Bt, F);
        F.sample = fused ? &SD : nullptr;
        launch_fwd(F, st);
    }
    {   // a', H' = policy.sample(s')
        const GaussArgs G{N->policy, kPolicy, s[SS_PN].z2, Bt->eps_next, B, Bt->eps_next ? 1 : 2, X.act_n, X.ent_n, nullptr, Bt->seed, 0x40000000u, Bt->call};
        if (one_call) {
            const GaussArgs G2{N->policy, kPolicy, s[SS_PC].z2, Bt->eps_cur, B, Bt->eps_cur ? 1 : 2, X.act_c, nullptr, X.aux_c, Bt->seed, 0x80000000u, Bt->call};
            launch_gauss(G, &G2, N->target_critic, N->critic, polyak_first ? nq : 0, Hy->tau, st);
        } else {
            launch_gauss(G, nullptr, nullptr, nullptr, 0, 0.0f, st);
        }
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
        launch_bwd(0, G, st);
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
            launch_wg(W, true, st);
        } else {
            launch_wg(W, false, st);
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
// adam_step > 0 (hx_sac_learn): policy.sample(s) was evaluated in the critic half's launch, and policy_optim.step() + the log-alpha step ride in
// the policy's weight-gradient launch (the thread that produced a gradient steps it; thread 0 of the launch steps log_alpha)
static int sac_policy_grads_impl(const HxSacNets* N, const HxSacBatch* Bt, const HxHyper* Hy, void* stream, int adam_step, float target_entropy,
                                 const SampleDev* predraw = nullptr) {
    HX_REQUIRE(N && Bt && Hy && Bt->rows && Bt->batch > 0 && Bt->batch % 16 == 0, "hx_sac_policy_grads: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int B = Bt->batch;
    Slot s[S_COUNT];
    sac_slots(N, B, s);
    const SacAux X = sac_aux(N, B);
    const RowSrc src{Bt->rows, nullptr, nullptr, 0, 32};
    const float* q1 = N->critic; const float* q2 = N->critic + kQs.padded();
    if (adam_step == 0) {
        const GaussArgs G{N->policy, kPolicy, s[SS_PC].z2, Bt->eps_cur, B, Bt->eps_cur ? 1 : 2, X.act_c, nullptr, X.aux_c, Bt->seed, 0x80000000u, Bt->call};
        launch_gauss(G, nullptr, nullptr, nullptr, 0, 0.0f, st);
    }
    {
        FwdArgs F{};
        F.njobs = 2; F.slope = 0.0f;
        F.job[0] = FwdJob{q1, kQs, src, 0, 3, Head{}, X.act_c, 0.f, s[SS_Q1P], B, 1};
        F.job[1] = FwdJob{q2, kQs, src, 0, 3, Head{}, X.act_c, 0.f, s[SS_Q2P], B, 1};
        launch_fwd(F, st);
    }
    // one call (adam_step > 0): min(Q1, Q2) is selected in the critics' backward prologue and the policy's head gradient is formed in the
    // policy's backward prologue (bwd_l2<4> / <5>) — q_select_kernel and policy_dout_kernel as launches of their own are the staged form
    static const bool fold_env = !(getenv("HX_SAC_FOLD") && getenv("HX_SAC_FOLD")[0] == '0');  // A/B knob
    const bool fold = adam_step > 0 && fold_env;
    if (!fold) {
        QSelArgs Q{q1, q2, kQs, s[SS_Q1P], s[SS_Q2P], B, 1.0f / B, N->losses};
        hipLaunchKernelGGL(q_select_kernel, dim3((unsigned)((B + 3) / 4)), dim3(kThreads), 0, st, Q);
    }
    {   // both critics backward down to dh1
        BwdArgs G{};
        G.njobs = 2; G.slope = 0.0f; G.inv_batch = 1.0f / B; G.losses = N->losses;
        for (int h = 0; h < 2; ++h) {
            BwdJob& J = G.job[h];
            J = BwdJob{};
            J.net = h ? q2 : q1; J.m = kQs; J.ws = s[SS_Q1P + h]; J.rows = B; J.mode = fold ? BM_SAC_QMIN : BM_GIVEN;
            if (fold) { J.t1 = Head{h ? q1 : q2, kQs, s[SS_Q1P + (1 - h)]}; J.loss_slot = h; }
        }
        launch_bwd(fold ? 4 : 3, G, st);
    }
    if (!fold) {
        PDoutArgs P{q1, q2, kQs, s[SS_Q1P], s[SS_Q2P], s[SS_PC], X.aux_c, N->alpha_state, B, 1.0f / B, N->losses};
        hipLaunchKernelGGL(policy_dout_kernel, dim3((unsigned)((B + 3) / 4)), dim3(kThreads), 0, st, P);
    }
    {
        BwdArgs G{};
        G.njobs = 1; G.slope = 0.0f; G.inv_batch = 1.0f / B; G.losses = N->losses;
        BwdJob& J = G.job[0];
        J = BwdJob{};
        J.net = N->policy; J.m = kPolicy; J.ws = s[SS_PC]; J.rows = B; J.mode = fold ? BM_SAC_POLICY : BM_GIVEN;
        if (fold) {
            J.t1 = Head{q1, kQs, s[SS_Q1P]}; J.t2 = Head{q2, kQs, s[SS_Q2P]};
            J.bonus = X.aux_c; J.bonus_scale = N->alpha_state + 3;
        }
        launch_bwd(fold ? 5 : 3, G, st);
    }
    {
        WgArgs W{};
        W.njobs = 1; W.slope = 0.0f; W.w_kind = 0; W.inv_batch = 1.0f / B;
        WgJob& J = W.job[0];
        J = WgJob{};
        J.net = N->policy; J.grad = N->grad_policy; J.m = kPolicy; J.ws[0] = s[SS_PC]; J.rows[0] = B; J.nslots = 1; J.wmode[0] = 0;
        W.predraw = predraw; W.predraw_batch = B;  // hx_sac_learn_back: one more workgroup assembles the next front launch's minibatch (112 workgroups: CUs to spare)
        if (adam_step > 0) {
            const double b1 = 0.9, b2 = 0.999;
            const double bc1 = 1.0 - pow(b1, adam_step), bc2 = 1.0 - pow(b2, adam_step);
            J.p = N->policy; J.mom = N->m_policy; J.var = N->v_policy;
            J.w2f = N->policy_w2_f32i;  // the acting kernels' images of the policy's W2 follow its optimizer step
            if (N->policy_w2_x9) { J.w2b = N->policy_w2_x9; J.w2b_x9 = 1; }
            W.ad = WgAdam{};
            W.ad.b1 = (float)b1; W.ad.b2 = (float)b2; W.ad.eps = 1e-8f;
            W.ad.step_size = (float)(Hy->lr_actor / bc1);
            W.ad.bc2_sqrt = (float)sqrt(bc2);
            W.ad.losses = N->losses;
            W.ad.alpha_state = N->alpha_state; W.ad.target_entropy = target_entropy; W.ad.alpha_step_size = (float)(Hy->lr_actor / bc1);
            launch_wg(W, true, st);
        } else {
            launch_wg(W, false, st);
        }
    }
    HX_CHECK_LAUNCH("hx_sac_policy_grads");
    return 0;
}
int hx_sac_policy_grads(const HxSacNets* N, const HxSacBatch* Bt, const HxHyper* Hy, void* stream) {
    return sac_policy_grads_impl(N, Bt, Hy, stream, 0, 0.0f);
}
/* One GPU: the whole SacAgent.learn (SAC/agent.py:276-327) in ONE call and 9 launches (the staged sequence takes 14): the Polyak step and
 * policy.sample(s) ride in the launch of policy.sample(s'), both optimizers' steps (and the log-alpha step) in their weight-gradient launches,
 * the min(Q1, Q2) selection and the policy's head gradient in the prologues of the backward launches that consume them.
 * Bit-identical to hx_sac_critic_step + hx_sac_policy_grads + hx_sac_adam(which = 1).  sample may be NULL; step is 1-based. */
int hx_sac_learn(const HxSacNets* N, const HxSacBatch* Bt, const HxHyper* Hy, const HxSample* S, int32_t polyak_first, int32_t step, float target_entropy,
                 void* stream) {
    HX_REQUIRE(step >= 1, "hx_sac_learn: step is 1-based");
    if (int rc = sac_critic_grads_impl(N, Bt, Hy, S, polyak_first, stream, step, true)) return rc;
    return sac_policy_grads_impl(N, Bt, Hy, stream, step, target_entropy);
}

/* hx_sac_learn in two parts around an env step (include/hirl4ucav.h "SAC front launch"): hx_sac_front = hx_sac_act_step_x9 / _f32i for n > 8,192 envs AND the first
 * forward launch of the learn() call behind it as workgroups of ONE launch, on the minibatch tiles a predraw left in `batch`; hx_sac_learn_back = the rest. */
int hx_sac_front(const float* policy, const uint16_t* w2_x9, const float* w2_f32i, float* state, int64_t n, int64_t stride, float* obs_io, float* actions, int32_t mode,
                 const float* eps, uint64_t seed, uint32_t row0, uint32_t call, float* reward, uint8_t* done, int8_t* success, const HxStepOpts* opts,
                 const HxSacNets* N, const HxSacBatch* Bt, void* stream) {
    HX_REQUIRE(policy && (w2_x9 || w2_f32i) && opts && N && Bt && Bt->rows && Bt->batch > 0 && Bt->batch % 16 == 0 && Bt->batch <= kFusedBatchMax && mode >= 0 && mode <= 2 &&
               (mode != 1 || eps), "hx_sac_front: bad arguments");
    if (int rc = hxact::check_step_args(state, n, stride, obs_io, actions, reward, done, success, *opts, "hx_sac_front")) return rc;
    hxact::ActFusedArgs H{policy, kPolicy, obs_io, (int)n, 0.0f, actions, mode == 1 ? eps : nullptr, 1, 0.0f, mode, seed, row0, call,
                          state, stride, reward, done, success, *opts, opts->cap > 0 ? 1.0 / (double)opts->cap : 0.0, w2_x9, w2_x9 ? nullptr : w2_f32i, w2_x9 ? 1 : 0};
    FwdArgs F;
    sac_launch_1(N, Bt, F);
    return launch_front_sac(H, F, (hipStream_t)stream);
}
int hx_sac_learn_back(const HxSacNets* N, const HxSacBatch* Bt, const HxHyper* Hy, int32_t polyak_first, int32_t step, float target_entropy, const HxSample* next,
                      float* next_rows, void* stream) {
    HX_REQUIRE(step >= 1 && N && Bt && Hy, "hx_sac_learn_back: step is 1-based");
    SampleDev SD{};
    if (next) {
        HX_REQUIRE(next_rows && next_rows != Bt->rows && !next->bc_table && !next->idx_bc, "hx_sac_learn_back: the next minibatch needs a tile of its own; SAC has no BC minibatch");
        bool fused = false;
        if (int rc = prepare_draw(next, Bt->batch, next_rows, nullptr, nullptr, stream, &SD, &fused, /*launch_now=*/false)) return rc;
        HX_REQUIRE(fused, "hx_sac_learn_back: the predraw covers minibatches of at most 256 rows");
    }
    if (int rc = sac_critic_grads_impl(N, Bt, Hy, nullptr, polyak_first, stream, step, true, /*skip_first=*/true)) return rc;
    return sac_policy_grads_impl(N, Bt, Hy, stream, step, target_entropy, next ? &SD : nullptr);
}

}  // extern "C"
