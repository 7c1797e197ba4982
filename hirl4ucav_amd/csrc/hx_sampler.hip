// hx_sampler.hip — minibatch sampling as a launch of its own (gfx950); the in-launch form of the same draw is draw_fused (hx_update.h).
#include "hx_update.h"

using namespace hxnn;
using namespace hxu;

namespace {

// ---------------------------------------------------------------------------------------------------------------
// minibatch sampling on the device: UniformMemory.sample (buffer.py:45 random.sample, without replacement),
// np.random.choice(N_exp, B, replace=False) (HIRL.py:249) and the (4,) target-smoothing noise (HIRL.py:265).
// One workgroup; Philox4x32-10 keyed by `seed`, counter (row, call, stream, round).  Duplicates inside a group are
// redrawn until none is left (normally one round: B << len), at most 128 rounds; a duplicate surviving that — only
// possible when a group asks for nearly the whole table — is accepted (the with-replacement draw of SURVEY.md quirk 13).
// ---------------------------------------------------------------------------------------------------------------
struct SampleArgs {
    const unsigned long long* total;  // transitions ever stored in the main ring
    long long cap, expert_len, bc_len;
    int batch, n_main;
    uint64_t seed;
    uint32_t call;
    float sigma;
    int* idx;
    int* idx_bc;
    float* noise;
    // gather: the sampled rows are copied ONCE into compact [batch][32] tiles that every update kernel then reads with
    // plain row addressing (no index indirection, no page-scattered loads on their critical paths)
    const float* ring;
    const float* expert_ring;
    const float* bc_table;
    float* rows;
    float* bc_rows;
    int do_sample;  // 0: idx / idx_bc are inputs (parity tests, the N = 1 facade), only gather
    uint32_t guard; // > 0: the `guard` slots behind the ring head are not drawn (hx_update.h draw_map)
};

// 1024 threads.  With B <= 512 the two index streams (replay / expert rows, BC rows) are drawn side by side by the two halves
// of the workgroup.  "Without replacement" = inside a group, of several rows that drew the same index the lowest row keeps it
// and the others redraw.  The check is a hash set in LDS (key = group | index, owner = lowest row that drew it: atomicCAS +
// atomicMin, linear probing, load <= 0.3), O(1) per row and round instead of an all-pairs scan.  The gather reads the indices
// from LDS.
constexpr int kSampleSlots = 4096;
__device__ __forceinline__ uint32_t sample_hash(uint32_t k) { return (k * 2654435761u) >> 20; }  // top 12 bits

__global__ __launch_bounds__(1024) void sample_kernel(SampleArgs A) {
    __shared__ uint32_t hkey[2][kSampleSlots];
    __shared__ int hown[2][kSampleSlots];
    __shared__ int fin[2][1024];
    const int tid = threadIdx.x;
    const int B = A.batch;
    const int np = B <= 512 ? 2 : 1;            // streams drawn in parallel
    const int width = 1024 / np;                // threads per stream
    const int t = tid % width, tab = tid / width;
    if (A.do_sample) {
        const unsigned long long tot = *A.total;
        const DrawMap dm = draw_map(tot, (unsigned long long)A.cap, A.guard);
        const uint32_t len_main = dm.live;
        const uint32_t k0 = (uint32_t)A.seed, k1 = (uint32_t)(A.seed >> 32);
        for (int pass = 0; pass < 2 / np; ++pass) {
            const int stream = np == 2 ? tab : pass;  // 0: replay / expert rows, 1: BC rows
            int* out = stream == 1 ? A.idx_bc : A.idx;
            const bool live = out != nullptr && t < B;
            const bool main_grp = t < A.n_main;
            const uint32_t len = stream == 1 ? (uint32_t)A.bc_len : (main_grp ? len_main : (uint32_t)A.expert_len);
            const uint32_t grp = (stream == 1 || main_grp) ? 0u : 0x80000000u;  // groups: [0, n_main) and [n_main, batch)
            uint32_t* keys = hkey[tab];
            int* owns = hown[tab];
            for (int e = t; e < kSampleSlots; e += width) {
                keys[e] = 0xFFFFFFFFu;
                owns[e] = 0x7FFFFFFF;
            }
            __syncthreads();
            uint32_t key = 0;
            bool dup = live;
            for (int round = 0; round < 128; ++round) {
                if (dup) {
                    uint32_t u[4];
                    philox4x32_10((uint32_t)t, A.call, (uint32_t)stream, (uint32_t)round, k0, k1, u);
                    key = grp | (len ? __umulhi(u[0], len) : 0u);
                    uint32_t h = sample_hash(key);
                    for (int probe = 0; probe < kSampleSlots; ++probe) {  // bounded: the set never fills (<= B + redraws keys)
                        const uint32_t k = atomicCAS(&keys[h], 0xFFFFFFFFu, key);
                        if (k == 0xFFFFFFFFu || k == key) break;
                        h = (h + 1) & (kSampleSlots - 1);
                    }
                    atomicMin(&owns[h], t);
                }
                __syncthreads();
                if (live) {  // every row looks its index up again: a redraw of a lower row may have taken it over
                    uint32_t h = sample_hash(key);
                    for (int probe = 0; probe < kSampleSlots && keys[h] != key; ++probe) h = (h + 1) & (kSampleSlots - 1);
                    dup = owns[h] != t;
                }
                if (!__syncthreads_or(dup)) break;  // nobody redraws: done (the common case after the first round)
            }
            if (live) {
                const int v = (stream == 0 && main_grp) ? (int)slot_of_draw(dm, key & 0x7FFFFFFFu) : (int)(key & 0x7FFFFFFFu);
                out[t] = v;
                fin[stream][t] = v;
            }
            __syncthreads();
        }
        if (tid < 4 && A.noise) {
            uint32_t u[4];
            philox4x32_10(0xFFFFFFF0u, A.call, 2u, 0u, k0, k1, u);
            const float ua = u01(u[tid & 2]), ub = u01(u[(tid & 2) + 1]);
            const float rad = sqrtf(-2.0f * __logf(ua)), ang = 6.28318530717958647692f * ub;
            A.noise[tid] = A.sigma * ((tid & 1) ? rad * __sinf(ang) : rad * __cosf(ang));
        }
    } else {
        for (int e = tid; e < B; e += 1024) {
            fin[0][e] = A.idx[e];
            if (A.idx_bc) fin[1][e] = A.idx_bc[e];
        }
        __syncthreads();
    }
    // gather: 8 lanes per row, one 16-B piece each; both tiles' loads are in flight together
    const bool bc = A.bc_rows && A.bc_table && A.idx_bc;
    for (int e = tid; e < B * 8; e += 1024) {
        const int r = e >> 3, c = e & 7;
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f), q = m;
        if (A.rows) m = reinterpret_cast<const float4*>((r < A.n_main ? A.ring : A.expert_ring) + (size_t)fin[0][r] * 32)[c];
        if (bc) q = reinterpret_cast<const float4*>(A.bc_table + (size_t)fin[1][r] * 32)[c];
        if (A.rows) reinterpret_cast<float4*>(A.rows)[e] = m;
        if (bc) reinterpret_cast<float4*>(A.bc_rows)[e] = q;
    }
}

}  // namespace

extern "C" {

/* Minibatch assembly for one learn() call (replaces UniformMemory.sample buffer.py:38-48, the buffer/expert mixing and the
 * BC draw of HIRL.py:223-251, and the noise draw HIRL.py:265).  do_sample = 1: draw idx[batch] (rows < n_main index the main
 * ring, whose live length min(*total, cap) is read on the device; the rest the expert ring), idx_bc[batch], noise[4] =
 * sigma N(0,1) with Philox4x32-10(seed; row, call), without replacement inside each group.  do_sample = 0: idx / idx_bc are
 * inputs.  Either way the selected rows are then copied into the compact tiles rows[batch][32] / bc_rows[batch][32] that
 * the update stages read. */
int hx_sample_batch(const uint64_t* total, int64_t cap, const float* ring, const float* expert_ring, int64_t expert_len,
                    const float* bc_table, int64_t bc_len, int32_t batch, int32_t n_main, int32_t do_sample, uint64_t seed,
                    uint32_t call, float sigma, int32_t* idx, int32_t* idx_bc, float* noise, float* rows, float* bc_rows,
                    void* stream) {
    return hx_sample_batch_guarded(total, cap, ring, expert_ring, expert_len, bc_table, bc_len, batch, n_main, do_sample, seed, call, sigma, idx, idx_bc, noise,
                                   rows, bc_rows, 0u, stream);
}
/* The same with HxSample.guard: the `guard` ring slots behind the head *total are not drawn (hx_hirl_front's population, as a launch of its own). */
int hx_sample_batch_guarded(const uint64_t* total, int64_t cap, const float* ring, const float* expert_ring, int64_t expert_len,
                            const float* bc_table, int64_t bc_len, int32_t batch, int32_t n_main, int32_t do_sample, uint64_t seed,
                            uint32_t call, float sigma, int32_t* idx, int32_t* idx_bc, float* noise, float* rows, float* bc_rows,
                            uint32_t guard, void* stream) {
    HX_REQUIRE(idx && ring && rows && batch > 0 && batch <= 1024 && n_main >= 0 && n_main <= batch, "hx_sample_batch: bad arguments");
    HX_REQUIRE(guard == 0 || (do_sample && 2 * (int64_t)guard <= cap), "hx_sample_batch_guarded: a guard of more than half the ring leaves too little to draw from (cap >= 2 n)");
    HX_REQUIRE(!do_sample || (total && cap > 0), "hx_sample_batch: sampling needs total and cap");
    HX_REQUIRE(n_main == batch || expert_ring, "hx_sample_batch: expert rows requested without an expert ring");
    SampleArgs A{(const unsigned long long*)total, cap, expert_len, bc_len, batch, n_main, seed, call, sigma, idx, idx_bc, noise,
                 ring, expert_ring ? expert_ring : ring, bc_table, rows, bc_rows, do_sample, guard};
    hipLaunchKernelGGL(sample_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, A);
    HX_CHECK_LAUNCH("hx_sample_batch");
    return 0;
}

}  // extern "C"
