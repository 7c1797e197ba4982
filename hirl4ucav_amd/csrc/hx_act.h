// hx_act.h — what the two acting kernels share (gfx950): the launch description and the exploration-noise draw.
//   hx_act.hip    act_fused_kernel: one row tile (16 / 32 rows) per workgroup — up to 8,192 rows (one round of workgroups)
//   hx_actp.hip   act_persist_kernel: persistent workgroups that keep W2 and loop over their row tiles — beyond 8,192 rows
// Agent.chooseAction* hirl/agents/HIRL.py:192-212; SacAgent.explore / exploit hirl/agents/SAC/agent.py:183-196.
#pragma once
#include "hx_update.h"

namespace hxact {
using namespace hxnn;
using namespace hxu;

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
    int x9;               // X3 instantiations: w2b is the first of THREE images hi | mid | lo (hx_pack_w2_x9): the exact bf16 split of W2
};

// one standard-normal draw per (row, component j) of the acting kernels: Philox4x32-10(seed; row, call, tag) + Box-Muller
__device__ __forceinline__ float philox_normal(uint32_t row, uint32_t call, uint32_t tag, uint64_t seed, int j) {
    uint32_t u[4];
    philox4x32_10(row, call, tag, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), u);
    const float ua = u01(u[j & 2]), ub = u01(u[(j & 2) + 1]);
    const float rad = sqrtf(-2.0f * logf(ua)), ang = 6.28318530717958647692f * ub;
    return (j & 1) ? rad * sinf(ang) : rad * cosf(ang);
}

// Rows beyond which the persistent kernel takes over (hx_actp.hip).  Up to here ONE round of 16- / 32-row workgroups covers the rows and
// the env step rides in wave 0 of each; beyond, every further round of workgroups would fetch the whole W2 image again.
constexpr int64_t kFuseEnvMax = 8192;

// hx_actp.hip.  mode: 0 fp32 from the fp32 image (H.w2f), 1 bf16 (H.w2b), 2 the exact 9-term bf16 split (H.w2b = hi | mid | lo, H.x9).
// Returns false when no persistent instantiation covers the request (the caller falls back to act_fused_kernel).
bool launch_act_persist(const ActFusedArgs& H, bool gauss, hipStream_t st);

}  // namespace hxact
