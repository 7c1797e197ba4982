// hx_rccl.hip — the gradient exchange of a sharded update straight on RCCL (gfx950, xGMI): ncclAllReduce enqueued on the engine's own stream
// through the C ABI, no torch.distributed call inside learn().  The reference has no counterpart (single process, hirl/agents/HIRL.py:52); the
// messages are SURVEY.md 8e's: the flat critic gradient and the merged actor message, one all-reduce per phase.
//
// Why not torch.distributed.all_reduce: measured at world size 1 (profiles/r03e_bench_staged_nccl_world1.json) every call costs ~8 us of host
// time that the stream waits for — work-object bookkeeping, stream-event hand-off to the process group's internal stream and back — 18 % of a
// 70 us step before a byte moves.  Here the collective is one more enqueue on the stream the update kernels run on.
//
// librccl.so is opened at run time (dlopen): the SAME copy the process already holds if torch has loaded one (RTLD_NOLOAD first), so that one
// process never runs two RCCL instances; the library's types are restated here (rccl.h: ncclUniqueId = 128 opaque bytes, ncclFloat32 = 7,
// ncclBfloat16 = 9, ncclSum = 0), which keeps the build free of the RCCL headers.
#include <dlfcn.h>

#include <cstring>

#include "hx_common.h"

namespace {

struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*CommDestroyFn)(Comm);
typedef const char* (*ErrorStringFn)(int);

struct Api {
    void* lib = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    AllReduceFn all_reduce = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    ErrorStringFn error_string = nullptr;
    const char* path = "";
};

Api* api() {
    static Api A;
    static bool tried = false;
    if (tried) return A.all_reduce ? &A : nullptr;
    tried = true;
    const char* env = getenv("HX_RCCL_LIBRARY");
    const char* names[] = {env ? env : "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (int pass = 0; pass < 2 && !A.lib; ++pass)  // pass 0: a copy already in the process (torch's); pass 1: load one
        for (const char* n : names) {
            A.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (A.lib) { A.path = n; break; }
        }
    if (!A.lib) return nullptr;
    A.get_unique_id = (GetUniqueIdFn)dlsym(A.lib, "ncclGetUniqueId");
    A.comm_init_rank = (CommInitRankFn)dlsym(A.lib, "ncclCommInitRank");
    A.all_reduce = (AllReduceFn)dlsym(A.lib, "ncclAllReduce");
    A.comm_destroy = (CommDestroyFn)dlsym(A.lib, "ncclCommDestroy");
    A.error_string = (ErrorStringFn)dlsym(A.lib, "ncclGetErrorString");
    if (!A.get_unique_id || !A.comm_init_rank || !A.all_reduce || !A.comm_destroy) A.all_reduce = nullptr;
    return A.all_reduce ? &A : nullptr;
}

#define HX_RCCL(expr, what)                                                                                              \
    do {                                                                                                                 \
        const int rc_ = (expr);                                                                                          \
        if (rc_ != 0) return ::hx::fail(HX_ERR_HIP, "%s: RCCL error %d (%s)", what, rc_, R->error_string ? R->error_string(rc_) : "?"); \
    } while (0)

// the bf16 wire format of hx_rccl_allreduce_bf16: four values per thread, round to nearest even (the compiler's __bf16 conversion)
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float4* __restrict__ src, uint2* __restrict__ dst, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = src[i];
    dst[i] = make_uint2(__builtin_bit_cast(unsigned, v2bf{(__bf16)v.x, (__bf16)v.y}), __builtin_bit_cast(unsigned, v2bf{(__bf16)v.z, (__bf16)v.w}));
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const uint2* __restrict__ src, float4* __restrict__ dst, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const uint2 q = src[i];
    dst[i] = make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xFFFF0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xFFFF0000u));
}

}  // namespace

extern "C" {

/* 0 when librccl.so and the entry points this file needs can be bound in this process; nothing collective, nothing allocated.  The all-rank
 * negotiation (hirl4ucav_amd/agents/exchange.py negotiate_rccl_direct) asks this on every rank BEFORE anyone enters ncclCommInitRank. */
int hx_rccl_available(void) {
    HX_REQUIRE(api(), "hx_rccl_available: librccl.so (ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy) not found "
                      "(HX_RCCL_LIBRARY names another path)");
    return 0;
}
/* 128 bytes that identify a new communicator: call on ONE rank, hand the bytes to every rank (any byte channel: the launcher's TCP store). */
int hx_rccl_unique_id(uint8_t* id128) {
    HX_REQUIRE(id128, "hx_rccl_unique_id: null");
    Api* R = api();
    HX_REQUIRE(R, "hx_rccl_unique_id: librccl.so not found (HX_RCCL_LIBRARY names another path)");
    UniqueId u;
    HX_RCCL(R->get_unique_id(&u), "hx_rccl_unique_id");
    std::memcpy(id128, u.internal, 128);
    return 0;
}
/* Collective over all `world` ranks (each on its own GPU, the current HIP device): *comm receives the communicator handle. */
int hx_rccl_init(const uint8_t* id128, int32_t world, int32_t rank, void** comm) {
    HX_REQUIRE(id128 && comm && world >= 1 && rank >= 0 && rank < world, "hx_rccl_init: bad arguments");
    Api* R = api();
    HX_REQUIRE(R, "hx_rccl_init: librccl.so not found (HX_RCCL_LIBRARY names another path)");
    UniqueId u;
    std::memcpy(u.internal, id128, 128);
    Comm c = nullptr;
    HX_RCCL(R->comm_init_rank(&c, world, u, rank), "hx_rccl_init");
    *comm = c;
    return 0;
}
/* buf[i] <- sum over the ranks of buf[i], i < n, in place, enqueued on `stream` (dtype 0: fp32, 1: bf16).  Every rank receives the same bits. */
int hx_rccl_allreduce(void* comm, void* buf, int64_t n, int32_t dtype, void* stream) {
    HX_REQUIRE(comm && buf && n > 0 && (dtype == 0 || dtype == 1), "hx_rccl_allreduce: bad arguments");
    Api* R = api();
    HX_REQUIRE(R, "hx_rccl_allreduce: librccl.so not loaded");
    HX_RCCL(R->all_reduce(buf, buf, (size_t)n, dtype == 0 ? 7 : 9, 0, (Comm)comm, (hipStream_t)stream), "hx_rccl_allreduce");
    return 0;
}
int hx_rccl_allreduce_bf16(void* comm, float* buf, uint16_t* scratch, int64_t n, void* stream) {
    HX_REQUIRE(comm && buf && scratch && n > 0 && (n & 3) == 0, "hx_rccl_allreduce_bf16: bad arguments (n must be a multiple of 4)");
    Api* R = api();
    HX_REQUIRE(R, "hx_rccl_allreduce_bf16: librccl.so not loaded");
    const long long n4 = n / 4;
    const dim3 grid((unsigned)((n4 + 255) / 256)), block(256);
    hipLaunchKernelGGL(cast_f32_bf16_kernel, grid, block, 0, (hipStream_t)stream, reinterpret_cast<const float4*>(buf), reinterpret_cast<uint2*>(scratch), n4);
    HX_CHECK_LAUNCH("hx_rccl_allreduce_bf16 (cast)");
    HX_RCCL(R->all_reduce(scratch, scratch, (size_t)n, 9, 0, (Comm)comm, (hipStream_t)stream), "hx_rccl_allreduce_bf16");
    hipLaunchKernelGGL(cast_bf16_f32_kernel, grid, block, 0, (hipStream_t)stream, reinterpret_cast<const uint2*>(scratch), reinterpret_cast<float4*>(buf), n4);
    HX_CHECK_LAUNCH("hx_rccl_allreduce_bf16 (uncast)");
    return 0;
}
int hx_rccl_destroy(void* comm) {
    HX_REQUIRE(comm, "hx_rccl_destroy: null");
    Api* R = api();
    HX_REQUIRE(R, "hx_rccl_destroy: librccl.so not loaded");
    HX_RCCL(R->comm_destroy((Comm)comm), "hx_rccl_destroy");
    return 0;
}

}  // extern "C"
