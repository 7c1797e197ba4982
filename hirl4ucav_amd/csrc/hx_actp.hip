// hx_actp.hip — policy inference for MANY rows (gfx950): persistent workgroups that fetch the 256 -> 512 weights ONCE and loop over
// their row tiles, with the env step of the same rows in the launch's tail.
//   Agent.chooseAction / chooseActionSmallNoise / chooseActionNoNoise   hirl/agents/HIRL.py:192-212   (U5)
//   chooseAction + HarfangEnv.step for every env                        hirl/train_all.py:343-345
//   SacAgent.explore / exploit (+ step)                                 hirl/agents/SAC/agent.py:183-196, hirl/train_sac.py:238-241
//
// Why a second kernel.  act_fused_kernel (hx_act.hip) gives every 16 / 32 rows a workgroup of their own, and every workgroup pulls the
// whole W2 image (256 KB bf16, 512 KB fp32) through its L2 again: at 131,072 rows that is 4,096 workgroups x 256 KB = 1 GB per launch and
// 4.5 us of every 6.5 us workgroup (profiles/r03e_stamps.txt).  Here the grid is one workgroup per CU (<= 256) and each owns a CONTIGUOUS
// block of rows:
//   bf16   the wave's B fragments (its 32 columns x all of K: 64 VGPRs) are loaded once and stay in registers for every tile
//          (weight-stationary); the tile loop is software-pipelined over three barriers per 32 rows:
//            P1  z2(t-1) = h1(t-1) W2^T on the bf16 matrix cores        | layer 1 of tile t on the fp32 matrix cores  | noise draw (t-1)
//            P2  LN2 + final layer + tanh + noise of tile t-1 (a wave per row) | LN1 statistics of tile t | the next observation tile -> LDS
//            P3  LN1 + activation of tile t -> bf16 h1
//   fp32 / x9  W2 does not fit the register file (512 / 768 KB): the wave streams its column slices ONCE per pass over 64 rows (four row
//          tiles per B fragment: a quarter of act_fused_kernel<2>'s L2 traffic per row), z2 leaves the accumulators in two halves.
// Per-row arithmetic is the arithmetic of act_fused_kernel (same layer-1 MFMA sequence, same k order, same LayerNorm / head functions): the
// two kernels agree bit for bit (tests/test_hirl_gpu.py::test_act_row_tilings_agree_bit_for_bit, tests/test_actp_gpu.py).
// ENV: once a workgroup's actions are written it steps the SAME rows' envs, 512 at a time on all 16 waves (hx_env_block.h: the body of
// env_step_kernel, two lanes per env, fused replay insert) — act + env + insert is one launch at every size.
#include <hip/hip_ext.h>

#include "hx_actp_body.h"

using namespace hxnn;
using namespace hxu;
using namespace hxact;

namespace {

// the persistent bf16 acting workgroup lives in hx_actp_body.h (hx_front.hip runs it beside learn()'s first launches)
template <bool ENV, bool RELU>
__global__ __launch_bounds__(kWide) void act_persist_bf16_kernel(ActFusedArgs A, int tiles_per_wg) {
    __shared__ ActpLds<ENV> SL;
    act_persist_bf16_body<ENV, RELU>(A, tiles_per_wg, (int)blockIdx.x, (int)gridDim.x, SL);
}

// the streaming persistent acting workgroup (fp32 image / exact split) lives in hx_actp_body.h too
template <int MODE, bool GAUSS, bool ENV, bool RELU>
__global__ __launch_bounds__(kWide) void act_persist_stream_kernel(ActFusedArgs A, int tiles_per_wg) {
    __shared__ ActpsLds<MODE, GAUSS, ENV> SL;
    act_persist_stream_body<MODE, GAUSS, ENV, RELU>(A, tiles_per_wg, (int)blockIdx.x, (int)gridDim.x, SL);
}


template <typename K>
static void launch_k(K kernel, dim3 grid, const ActFusedArgs& H, int tiles_per_wg, hipStream_t st) {
    if (H.state && H.o.ev_start && H.o.ev_stop)
        hipExtLaunchKernelGGL(kernel, grid, dim3(kWide), 0, st, (hipEvent_t)H.o.ev_start, (hipEvent_t)H.o.ev_stop, 0, H, tiles_per_wg);
    else
        hipLaunchKernelGGL(kernel, grid, dim3(kWide), 0, st, H, tiles_per_wg);
}

// one workgroup per CU: the grid that keeps every weight fetch to ONE per CU (a second resident workgroup would not fit beside 128 VGPRs x
// 1,024 threads anyway)
static int persistent_cus() {
    static const int n = [] {
        if (const char* e = getenv("HX_ACT_PERSIST_WGS")) return atoi(e) > 0 ? atoi(e) : 256;  // tuning knob
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) return cus;
        return 256;
    }();
    return n;
}

}  // namespace

namespace hxact {

bool launch_act_persist(const ActFusedArgs& H, bool gauss, hipStream_t st) {
    const bool env = H.state != nullptr;
    if (env && !H.o.ring) return false;  // the env tail is built with the fused replay insert only; without a ring: act (persistent) + hx_env_step
    const bool relu = gauss || H.slope == 0.0f;
    if (!gauss && H.w2b && !H.x9) {
        constexpr int TR = 32;
        const int ntiles = (H.rows + TR - 1) / TR;
        const int per = (ntiles + persistent_cus() - 1) / persistent_cus();
        const dim3 grid((unsigned)((ntiles + per - 1) / per));
        if (env) {
            if (relu) launch_k(act_persist_bf16_kernel<true, true>, grid, H, per, st);
            else launch_k(act_persist_bf16_kernel<true, false>, grid, H, per, st);
        } else {
            if (relu) launch_k(act_persist_bf16_kernel<false, true>, grid, H, per, st);
            else launch_k(act_persist_bf16_kernel<false, false>, grid, H, per, st);
        }
        return true;
    }
    const bool x9 = H.w2b && H.x9;
    if (x9 || H.w2f) {  // fp32 from an image: 64 rows per pass
        constexpr int TR = 64;
        const int ntiles = (H.rows + TR - 1) / TR;
        const int per = (ntiles + persistent_cus() - 1) / persistent_cus();
        const dim3 grid((unsigned)((ntiles + per - 1) / per));
#define HX_STREAM(MODE, G, R) \
        { if (env) launch_k(act_persist_stream_kernel<MODE, G, true, R>, grid, H, per, st); else launch_k(act_persist_stream_kernel<MODE, G, false, R>, grid, H, per, st); }
        if (x9 && gauss) HX_STREAM(1, true, true)
        else if (x9) { if (relu) HX_STREAM(1, false, true) else HX_STREAM(1, false, false) }
        else if (gauss) HX_STREAM(0, true, true)
        else if (relu) HX_STREAM(0, false, true)
        else HX_STREAM(0, false, false)
#undef HX_STREAM
        return true;
    }
    return false;
}

}  // namespace hxact

HX_DEFINE_DEBUG_COLLECTORS(actp, 0, 80)
