// hx_fwdbwd.hip — forward and backward kernels of the update's 256 <-> 512 layer and everything fused around it (gfx950).
//   fwd_l2   z2 = act(LN(x W1^T + b1)) W2^T + b2 for up to 6 independent nets per launch  (Actor.forward / Critic.forward, HIRL.py:55-97,126-140)
//   bwd_l2   head + loss gradient + LN2 backward, dh1 = dz2 W2                               (the backward of HIRL.py:276-286,291-324)
// Structure and measurements: hx_update.h, DESIGN.md section 4.
#include "hx_bwd_body.h"

using namespace hxnn;
using namespace hxu;

namespace {

template <int NT, bool RELU, bool SAMPLE, bool BF16 = false>
__global__ __launch_bounds__(kWide) void fwd_l2_kernel(FwdArgsC A, typename std::conditional<SAMPLE, SampleDev, NoSample>::type SA) {
    __shared__ FwdLds<NT, SAMPLE, BF16> SL;
    fwd_l2_body<NT, RELU, SAMPLE, BF16, 0>(A, SA, (int)blockIdx.x, (int)blockIdx.y, SL, FrontSync{});
}

// (BwdJobC / BwdArgsC, the kernel body: hx_bwd_body.h)
template <int GRP, bool RELU, bool BF16 = false>
__global__ __launch_bounds__(kWide) void bwd_l2_kernel(BwdArgsC AC) {
    __shared__ BwdLds<GRP, BF16> SL;
    bwd_l2_body<GRP, RELU, BF16, 0>(AC, (int)blockIdx.x, (int)blockIdx.y, SL, FrontSync{});
}


int fwd_row_tiles(const FwdArgs& a) {
    int n = 0;
    for (int j = 0; j < a.njobs; ++j) n += (a.job[j].rows + RT - 1) / RT;
    return n;
}
int bwd_blocks(const BwdArgs& a, int rows_per_wg) {  // per job (every job of a launch has the same row count)
    return ((a.job[0].rows + rows_per_wg - 1) / rows_per_wg) * kColWgB;
}
template <int GRP>
void launch_bwd_t(const BwdArgs& G, hipStream_t st) {
    BwdArgsC C{};
    for (int j = 0; j < G.njobs; ++j) C.job[j] = pack_bwd(G.job[j], G);
    C.images = G.images;
    static const int rowmap = getenv("HX_XCD_ROWMAP") ? atoi(getenv("HX_XCD_ROWMAP")) : 3;  // bit 1: bwd_l2
    C.rowmap = (rowmap >> 1) & 1;
    const dim3 grid(bwd_blocks(G, (GRP <= 2 || GRP >= 4) ? RT / 2 : RT) , G.njobs);
    if constexpr (GRP <= 2) {  // the bf16 update path covers the HIRL / TD3 / BC jobs (GRP 3 = SAC's given head gradients: fp32)
        static const int dbg_off = getenv("HX_DBG_BF16_OFF") ? atoi(getenv("HX_DBG_BF16_OFF")) : 0;
        if (G.images && !(dbg_off & 2)) {
            if (G.slope == 0.0f) hipLaunchKernelGGL((bwd_l2_kernel<GRP, true, true>), grid, dim3(kWide), 0, st, C);
            else hipLaunchKernelGGL((bwd_l2_kernel<GRP, false, true>), grid, dim3(kWide), 0, st, C);
            return;
        }
    }
    if (G.slope == 0.0f) hipLaunchKernelGGL((bwd_l2_kernel<GRP, true>), grid, dim3(kWide), 0, st, C);
    else hipLaunchKernelGGL((bwd_l2_kernel<GRP, false>), grid, dim3(kWide), 0, st, C);
}

}  // namespace

namespace hxu {

// column tiling by size: enough row tiles to fill the chip -> wide workgroups (less prologue recomputation)
// hx_debug_set_fwd_nt: 64 = latency-mode launches keep 64-column workgroups whatever their job count (the tiling the front launch gives launch B)
static int g_force_fwd_nt = 0, g_force_skip = 0, g_force_count = 0;
void set_fwd_nt(int nt, int skip, int count) { g_force_fwd_nt = nt; g_force_skip = skip; g_force_count = count; }

void launch_fwd(const FwdArgs& F, hipStream_t st) {
    // HX_FWD_NT64=1 (tuning knob, tools/ubench/nofront_levers.sh): every latency-mode forward launch in 64-column workgroups — the tiling launch B takes inside
    // the front launch, where CU time is what runs out; on an empty chip 128 workgroups of 7.5 us against 256 of 6
    static const bool env64 = getenv("HX_FWD_NT64") && atoi(getenv("HX_FWD_NT64")) != 0;
    bool force64 = env64;
    if (g_force_fwd_nt == kNT) {  // (tests only) the launches [skip, skip + count) after the setter
        if (g_force_skip > 0) --g_force_skip;
        else if (g_force_count > 0) { --g_force_count; force64 = true; }
    }
    FwdArgsC C{};
    for (int j = 0; j < F.njobs; ++j) {  // every job of a launch has the same row count (the minibatch)
        C.job[j] = pack_fwd(F.job[j]);
        C.job[j].slope = F.slope;
    }
    C.slope = F.slope; C.zero_nf = F.zero_nf; C.zero_f = F.zero_f; C.zero_i = F.zero_i; C.images = F.images;
    static const int rowmap = getenv("HX_XCD_ROWMAP") ? atoi(getenv("HX_XCD_ROWMAP")) : 3;  // bit 0: fwd_l2 (0: the plain order everywhere)
    C.rowmap = rowmap & 1;
    const int tiles = fwd_row_tiles(F), per_job = tiles / F.njobs;
    const bool relu = F.slope == 0.0f;  // compile-time ReLU instantiations (hx_nn.h act_f)
    static const int dbg_off = getenv("HX_DBG_BF16_OFF") ? atoi(getenv("HX_DBG_BF16_OFF")) : 0;
    const bool bf16 = F.images != nullptr && !(dbg_off & 1);
#define HX_FWD_T(NT_, RELU_, BF16_) hipLaunchKernelGGL((fwd_l2_kernel<NT_, RELU_, false, BF16_>), grid, dim3(kWide), 0, st, C, NoSample{})
#define HX_FWD(NT_) do { const dim3 grid(per_job * (H2 / NT_), F.njobs); \
        if (bf16) { if (relu) HX_FWD_T(NT_, true, true); else HX_FWD_T(NT_, false, true); } \
        else { if (relu) HX_FWD_T(NT_, true, false); else HX_FWD_T(NT_, false, false); } } while (0)
    if (F.sample) {  // (the callers checked: three or four jobs of at most 256 rows -> the 64-column tiling)
        const dim3 grid(per_job * (H2 / kNT), F.njobs);
        if (bf16) {
            if (relu) hipLaunchKernelGGL((fwd_l2_kernel<kNT, true, true, true>), grid, dim3(kWide), 0, st, C, *F.sample);
            else hipLaunchKernelGGL((fwd_l2_kernel<kNT, false, true, true>), grid, dim3(kWide), 0, st, C, *F.sample);
        } else {
            if (relu) hipLaunchKernelGGL((fwd_l2_kernel<kNT, true, true>), grid, dim3(kWide), 0, st, C, *F.sample);
            else hipLaunchKernelGGL((fwd_l2_kernel<kNT, false, true>), grid, dim3(kWide), 0, st, C, *F.sample);
        }
        return;
    }
    if (tiles >= 128) HX_FWD(256);
    else if (tiles * (H2 / 32) <= 256 && !force64) HX_FWD(32);  // one or two nets at B = 128: 32-column workgroups still fit the chip in one round
    else HX_FWD(kNT);
#undef HX_FWD_T
#undef HX_FWD
}

void launch_bwd(int grp, const BwdArgs& G, hipStream_t st) {
    switch (grp) {
        case 0: launch_bwd_t<0>(G, st); break;
        case 1: launch_bwd_t<1>(G, st); break;
        case 2: launch_bwd_t<2>(G, st); break;
        case 4: launch_bwd_t<4>(G, st); break;
        case 5: launch_bwd_t<5>(G, st); break;
        default: launch_bwd_t<3>(G, st); break;
    }
}

}  // namespace hxu

HX_DEFINE_DEBUG_COLLECTORS(fwdbwd, 0, 32)

/* tests (include/hirl4ucav_debug.h): of the forward launches that follow, numbers [skip, skip + count) keep 64-column workgroups whatever their job
 * count (the tiling hx_hirl_front gives launch B) */
extern "C" int hx_debug_set_fwd_nt(int32_t nt, int32_t skip, int32_t count) {
    HX_REQUIRE((nt == 0 || nt == 64) && skip >= 0 && count >= 0, "hx_debug_set_fwd_nt: nt 0 or 64, skip and count >= 0");
    hxu::set_fwd_nt(nt, skip, count);
    return 0;
}
