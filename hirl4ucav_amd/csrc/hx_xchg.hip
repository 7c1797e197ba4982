// hx_xchg.hip — the exchange step of a sharded update without a collective library: a ONE-SHOT all-reduce kernel over
// hipIpc peer mappings.  Every rank exposes its gradient message and one flag word; each rank's kernel announces "my message of this
// epoch is complete", waits for the peers' announcements, then READS the peers' messages directly over xGMI, sums them in rank order
// (every rank adds the same values in the same order -> bit-identical results everywhere, no second hop) and writes the sum locally.
//
// Why not a ring: the messages are 0.5-1.1 MB and the update is latency-bound; xGMI on MI355X is fully connected point to point (7
// links per GPU), so one hop over all links at once beats 2 (n - 1) ring steps.  The reference has no exchange at all (single process).
//
// Memory model.  Messages are plain device allocations written by earlier kernels on the same stream (a kernel's end publishes its
// stores system-wide); flags live in FINE-GRAINED device memory and are written / polled with system-scope atomics; before the peer
// reads, the kernel executes a system-scope acquire (L1 + non-local L2 lines are invalidated).  Messages are double-buffered by epoch
// parity: a rank overwrites buffer (e & 1) at epoch e + 2 only after it has seen every peer's flag reach e + 1, i.e. after every peer
// finished reading epoch e — one flag per rank is enough, no "done reading" round.
// Failure is STICKY and GLOBAL (fail-stop).  Every wait is bounded by the device's constant 100 MHz clock; a rank whose wait times out sets
// its status word, writes HX_XCHG_POISON into its OWN flag — every peer that polls that flag, now or at any later epoch, fails too instead
// of waiting or summing — and returns without touching dst.  Once the status word is set every later exchange on that rank returns at once
// (re-poisoning its flag), and the optimizer steps that would consume the sum are skipped (HxNets.xchg_status: hx_adam / hx_adam_mixed
// leave parameters, moments, targets and images alone).  The host raises at its next check (OneShotExchange.check: every replica-checksum
// cadence and at close).  EXPERIMENTAL: exercised with ranks sharing one GPU only; the cross-device visibility argument above has not
// run on two physical GPUs.
#include <cstring>

#include <cstdlib>

#include "hx_common.h"

namespace {

constexpr int kMaxWorld = 8;
constexpr int kThreads = 256;

struct OneShotArgs {
    float* dst;
    const float* buf[kMaxWorld];
    unsigned* flag[kMaxWorld];
    unsigned* status;  // [0] != 0: a wait timed out
    int world, rank;
    long long n;       // floats, multiple of 4
    unsigned epoch;
    unsigned long long timeout_ticks;  // s_memrealtime ticks (10 ns)
};
constexpr unsigned kPoison = 0xFFFFFFFFu;  // a flag value no epoch takes (hx_allreduce_oneshot refuses it)

__global__ __launch_bounds__(kThreads) void oneshot_allreduce_kernel(OneShotArgs A) {
    __shared__ int s_fail;
    if (threadIdx.x == 0) s_fail = __hip_atomic_load(A.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u ? 1 : 0;  // sticky
    __syncthreads();
    // 1. announce (one lane of the grid): the message was written by kernels that completed before this one started.  A rank that has
    //    already failed announces POISON instead: its peers must not wait for it, nor sum what it holds.
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(A.flag[A.rank], s_fail ? kPoison : A.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (s_fail) return;
    // 2. wait for every peer (one polling lane per peer and workgroup; epochs only grow: signed distance handles the wrap)
    if ((int)threadIdx.x < A.world && (int)threadIdx.x != A.rank) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            const unsigned v = __hip_atomic_load(A.flag[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (v == kPoison) { s_fail = 2; break; }           // the peer failed: so do we
            if ((int)(v - A.epoch) >= 0) break;
            __builtin_amdgcn_s_sleep(8);
            if (__builtin_amdgcn_s_memrealtime() - t0 > A.timeout_ticks) { s_fail = 1; break; }
        }
    }
    __syncthreads();
    if (s_fail) {
        if (threadIdx.x == 0) {
            __hip_atomic_store(A.status, (unsigned)s_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(A.flag[A.rank], kPoison, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    // system-scope acquire: peers' messages are read fresh.  ONE wave per workgroup executes it (the invalidate covers the CU's caches and
    // L2; the barrier orders the other waves' loads behind it) — executed by every wave it costs ~30 us per launch on MI355X
    // (tools/ubench/handoff_probe.hip: an acquire fence per wave 29 us, a release fence by one thread per workgroup 1.6 us).
    if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
    // 3. sum in rank order, 16 B per lane
    const long long n4 = A.n / 4;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (long long)gridDim.x * kThreads) {
        float4 s = reinterpret_cast<const float4*>(A.buf[0])[i];
        for (int r = 1; r < A.world; ++r) {
            const float4 v = reinterpret_cast<const float4*>(A.buf[r])[i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<float4*>(A.dst)[i] = s;
    }
}

// ---- two stages: reduce-scatter + all-gather ---------------------------------------------------------------------------
// The one-shot kernel reads world x n floats per rank (8.8 MB over xGMI for a 1.1 MB message at 8 ranks).  Here rank r first sums only ITS
// slice [n r / world, n (r + 1) / world) of every rank's message into its peer-mapped `red` buffer (stage 1: (world - 1) / world x n floats
// over xGMI), then copies the peers' reduced slices (stage 2: the same again): 2 (world - 1) / world x n per rank, every link busy in both
// stages.  Two launches on the stream (the boundary is the grid-wide synchronisation), two flag words per rank and message kind.  `red`
// needs no double buffer: a rank rewrites its slice at epoch e + 1 only after every peer has announced e + 1, which a peer does after its
// stage-2 kernel of epoch e has finished (stream order).  bf16 != 0: the reduced slices travel as bf16 (round to nearest even, the rank's own
// slice too: every replica sees the same bits) — half the bytes of stage 2, the sum itself is formed in fp32.
struct TwoStageArgs {
    float* dst;
    const float* buf[kMaxWorld];
    void* red[kMaxWorld];
    unsigned* flag[kMaxWorld];
    unsigned* status;
    int world, rank, bf16, stage;
    long long n;
    unsigned epoch;
    unsigned long long timeout_ticks;
};

// announce + bounded wait + acquire, shared by the two stages; returns false when the exchange has failed (status set, flag poisoned)
__device__ __forceinline__ bool xchg_handshake(unsigned* const* flag, unsigned* status, int world, int rank, unsigned epoch, unsigned long long timeout_ticks, int* s_fail) {
    if (threadIdx.x == 0) *s_fail = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u ? 1 : 0;  // sticky
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(flag[rank], *s_fail ? kPoison : epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (*s_fail) return false;
    if ((int)threadIdx.x < world && (int)threadIdx.x != rank) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            const unsigned v = __hip_atomic_load(flag[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (v == kPoison) { *s_fail = 2; break; }
            if ((int)(v - epoch) >= 0) break;
            __builtin_amdgcn_s_sleep(8);
            if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) { *s_fail = 1; break; }
        }
    }
    __syncthreads();
    if (*s_fail) {
        if (threadIdx.x == 0) {
            __hip_atomic_store(status, (unsigned)*s_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(flag[rank], kPoison, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return false;
    }
    if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // ONE wave per workgroup (see oneshot_allreduce_kernel)
    __syncthreads();
    return true;
}
__device__ __forceinline__ uint2 pack4_bf16(float4 v) {
    typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
    return make_uint2(__builtin_bit_cast(unsigned, v2bf{(__bf16)v.x, (__bf16)v.y}), __builtin_bit_cast(unsigned, v2bf{(__bf16)v.z, (__bf16)v.w}));
}
__device__ __forceinline__ float4 unpack4_bf16(uint2 q) {
    return make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xFFFF0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xFFFF0000u));
}

__global__ __launch_bounds__(kThreads) void twostage_kernel(TwoStageArgs A) {
    __shared__ int s_fail;
    if (!xchg_handshake(A.flag, A.status, A.world, A.rank, A.epoch, A.timeout_ticks, &s_fail)) return;
    const long long n4 = A.n / 4;
    if (A.stage == 0) {  // this rank's slice of every message, summed in rank order -> red[rank]
        const long long lo = n4 * A.rank / A.world, hi = n4 * (A.rank + 1) / A.world;
        for (long long i = lo + (long long)blockIdx.x * kThreads + threadIdx.x; i < hi; i += (long long)gridDim.x * kThreads) {
            float4 s = reinterpret_cast<const float4*>(A.buf[0])[i];
            for (int r = 1; r < A.world; ++r) {
                const float4 v = reinterpret_cast<const float4*>(A.buf[r])[i];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            if (A.bf16) reinterpret_cast<uint2*>(A.red[A.rank])[i] = pack4_bf16(s);
            else reinterpret_cast<float4*>(A.red[A.rank])[i] = s;
        }
    } else {  // every rank's reduced slice -> dst
        for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (long long)gridDim.x * kThreads) {
            int r = (int)((i * A.world + A.world - 1) / n4);  // the owner of float4 i: the r with n4 r / world <= i < n4 (r + 1) / world
            while (r > 0 && i < n4 * r / A.world) --r;
            while (r + 1 < A.world && i >= n4 * (r + 1) / A.world) ++r;
            reinterpret_cast<float4*>(A.dst)[i] = A.bf16 ? unpack4_bf16(reinterpret_cast<const uint2*>(A.red[r])[i]) : reinterpret_cast<const float4*>(A.red[r])[i];
        }
    }
}

}  // namespace

extern "C" {

/* Device memory that peers can map (hipIpc): finegrained != 0 for the flag / status words (system-scope atomics), 0 for messages. */
int hx_ipc_alloc(int64_t bytes, int32_t finegrained, void** dev_ptr) {
    HX_REQUIRE(bytes > 0 && dev_ptr, "hx_ipc_alloc: bad arguments");
    void* p = nullptr;
    if (finegrained) HX_CHECK_HIP(hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained));
    else HX_CHECK_HIP(hipMalloc(&p, (size_t)bytes));
    HX_CHECK_HIP(hipMemset(p, 0, (size_t)bytes));
    HX_CHECK_HIP(hipDeviceSynchronize());
    *dev_ptr = p;
    return 0;
}
int hx_ipc_free(void* dev_ptr) {
    HX_REQUIRE(dev_ptr, "hx_ipc_free: null");
    HX_CHECK_HIP(hipFree(dev_ptr));
    return 0;
}
/* handle64: 64 bytes of host memory (hipIpcMemHandle_t) to hand to the peers (any byte channel: torch.distributed.all_gather_object) */
int hx_ipc_export(void* dev_ptr, void* handle64) {
    HX_REQUIRE(dev_ptr && handle64, "hx_ipc_export: bad arguments");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    HX_CHECK_HIP(hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle64), dev_ptr));
    return 0;
}
int hx_ipc_import(const void* handle64, void** dev_ptr) {
    HX_REQUIRE(handle64 && dev_ptr, "hx_ipc_import: bad arguments");
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle64, sizeof h);
    HX_CHECK_HIP(hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess));
    return 0;
}
int hx_ipc_close(void* dev_ptr) {
    HX_REQUIRE(dev_ptr, "hx_ipc_close: null");
    HX_CHECK_HIP(hipIpcCloseMemHandle(dev_ptr));
    return 0;
}

/* dst[i] = sum over ranks r = 0..world-1 (in that order) of bufs[r][i], i < n (n a multiple of 4; all pointers 16-byte aligned).
 * bufs / flags: HOST arrays of `world` device pointers (own memory at index `rank`, peers' hipIpc mappings elsewhere); status: a device
 * word (fine-grained memory, starts at 0) that becomes 1 if a peer did not arrive within `timeout_ms`, 2 if a peer reported failure; it is
 * STICKY: from then on dst is left untouched, this rank's flag carries a poison value that fails every peer too, and every later call
 * returns at once (pass the word as HxNets.xchg_status and the optimizer steps are skipped as well).  epoch: this exchange's number,
 * increasing by 1 per call on every rank (each rank's flag word must start at 0, the first epoch is 1; 0xFFFFFFFF is reserved). */
int hx_allreduce_oneshot(float* dst, const float* const* bufs, uint32_t* const* flags, uint32_t* status, int32_t world, int32_t rank,
                         int64_t n, uint32_t epoch, int32_t timeout_ms, void* stream) {
    HX_REQUIRE(dst && bufs && flags && status && world >= 1 && world <= kMaxWorld && rank >= 0 && rank < world && n > 0 && n % 4 == 0,
               "hx_allreduce_oneshot: bad arguments (world <= 8, n a multiple of 4)");
    OneShotArgs A{};
    A.dst = dst; A.status = status; A.world = world; A.rank = rank; A.n = n; A.epoch = epoch;
    for (int r = 0; r < world; ++r) {
        HX_REQUIRE(bufs[r] && flags[r] && (reinterpret_cast<uintptr_t>(bufs[r]) & 15u) == 0, "hx_allreduce_oneshot: null or misaligned peer pointer");
        A.buf[r] = bufs[r];
        A.flag[r] = flags[r];
    }
    HX_REQUIRE(epoch != kPoison, "hx_allreduce_oneshot: epoch 0xFFFFFFFF is reserved");
    A.timeout_ticks = (unsigned long long)(timeout_ms > 0 ? timeout_ms : 2000) * 100000ull;  // the 100 MHz constant clock
    const long long n4 = n / 4;
    static const int env_blocks = getenv("HX_ONESHOT_BLOCKS") ? atoi(getenv("HX_ONESHOT_BLOCKS")) : 256;  // tuning knob
    const int max_blocks = env_blocks < 1 ? 1 : env_blocks;
    const int blocks = (int)((n4 + kThreads - 1) / kThreads < max_blocks ? (n4 + kThreads - 1) / kThreads : max_blocks);
    hipLaunchKernelGGL(oneshot_allreduce_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, A);
    HX_CHECK_LAUNCH("hx_allreduce_oneshot");
    return 0;
}

/* The same sum in two stages (reduce-scatter + all-gather: 2 (world - 1) / world x n floats per rank over xGMI instead of world x n), two
 * launches on `stream`.  reds: HOST array of `world` device pointers to the ranks' peer-mapped reduced-slice buffers (n floats each; with
 * bf16 != 0 they hold bf16 and the result is the fp32 sum rounded to bf16 — identical on every rank); flags2: a second flag word per rank
 * (stage 2), same rules as flags.  Every other argument as hx_allreduce_oneshot; the same sticky fail-stop behaviour.  EXPERIMENTAL. */
int hx_allreduce_twostage(float* dst, const float* const* bufs, void* const* reds, uint32_t* const* flags, uint32_t* const* flags2, uint32_t* status,
                          int32_t world, int32_t rank, int64_t n, uint32_t epoch, int32_t timeout_ms, int32_t bf16, void* stream) {
    HX_REQUIRE(dst && bufs && reds && flags && flags2 && status && world >= 1 && world <= kMaxWorld && rank >= 0 && rank < world && n > 0 && n % 4 == 0,
               "hx_allreduce_twostage: bad arguments (world <= 8, n a multiple of 4)");
    HX_REQUIRE(epoch != kPoison, "hx_allreduce_twostage: epoch 0xFFFFFFFF is reserved");
    TwoStageArgs A{};
    A.dst = dst; A.status = status; A.world = world; A.rank = rank; A.n = n; A.epoch = epoch; A.bf16 = bf16 ? 1 : 0;
    for (int r = 0; r < world; ++r) {
        HX_REQUIRE(bufs[r] && reds[r] && flags[r] && flags2[r] && (reinterpret_cast<uintptr_t>(bufs[r]) & 15u) == 0 && (reinterpret_cast<uintptr_t>(reds[r]) & 15u) == 0,
                   "hx_allreduce_twostage: null or misaligned peer pointer");
        A.buf[r] = bufs[r];
        A.red[r] = reds[r];
    }
    A.timeout_ticks = (unsigned long long)(timeout_ms > 0 ? timeout_ms : 2000) * 100000ull;
    const long long n4 = n / 4;
    static const int env_blocks = getenv("HX_ONESHOT_BLOCKS") ? atoi(getenv("HX_ONESHOT_BLOCKS")) : 256;
    const int max_blocks = env_blocks < 1 ? 1 : env_blocks;
    for (int stage = 0; stage < 2; ++stage) {
        A.stage = stage;
        for (int r = 0; r < world; ++r) A.flag[r] = stage == 0 ? flags[r] : flags2[r];
        const long long work = stage == 0 ? (n4 + world - 1) / world : n4;
        const int blocks = (int)((work + kThreads - 1) / kThreads < max_blocks ? (work + kThreads - 1) / kThreads : max_blocks);
        hipLaunchKernelGGL(twostage_kernel, dim3(blocks < 1 ? 1 : blocks), dim3(kThreads), 0, (hipStream_t)stream, A);
    }
    HX_CHECK_LAUNCH("hx_allreduce_twostage");
    return 0;
}

}  // extern "C"
