// sl_comm.hip - the collectives of the sharded sweep directly on RCCL, for callers
// of the C ABI that do not bring torch.distributed: one communicator per context, one rank per
// GPU, every exchange issued on the context's HIP stream.
//
//   sweep / finalize  ->  sl_allreduce_result : all ranks' 64-byte result records are all-gathered
//                         (ncclAllGather) and folded on the device - lexicographic min of the failing
//                         key, lexicographic max of the last-safe and largest keys, sums of the
//                         counters (RCCL has no 128-bit lexicographic reduction)
//   radix select      ->  sl_allreduce_sum_u64 (256-bin histograms)
//   value iteration   ->  sl_allgather (value-table shards), sl_allreduce_max_f64 (residual)
//
// RCCL is bound at run time (dlopen) so that libslhip.so neither needs it on single-GPU
// installations nor collides with the copy a host framework has already loaded.
#include <dlfcn.h>

#include "sl_common.h"

namespace {

typedef struct { char internal[128]; } rccl_unique_id;
typedef void* rccl_comm;
enum { RCCL_SUM = 0, RCCL_MAX = 2, RCCL_UINT8 = 1, RCCL_UINT64 = 5, RCCL_FLOAT64 = 8 };

struct RcclApi {
    void* handle = nullptr;
    int (*get_unique_id)(rccl_unique_id*) = nullptr;
    int (*comm_init_rank)(rccl_comm*, int, rccl_unique_id, int) = nullptr;
    int (*comm_destroy)(rccl_comm) = nullptr;
    int (*all_gather)(const void*, void*, size_t, int, rccl_comm, hipStream_t) = nullptr;
    int (*all_reduce)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t) = nullptr;
    const char* (*error_string)(int) = nullptr;
};

RcclApi g_rccl;

bool rccl_bind() {
    if (g_rccl.handle) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {                        // a copy that is already loaded wins
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (h) break;
    }
    for (size_t k = 0; !h && k < sizeof(names) / sizeof(names[0]); ++k) h = dlopen(names[k], RTLD_NOW);
    if (!h) return false;
    RcclApi api;
    api.handle = h;
    api.get_unique_id = reinterpret_cast<decltype(api.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
    api.comm_init_rank = reinterpret_cast<decltype(api.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
    api.comm_destroy = reinterpret_cast<decltype(api.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
    api.all_gather = reinterpret_cast<decltype(api.all_gather)>(dlsym(h, "ncclAllGather"));
    api.all_reduce = reinterpret_cast<decltype(api.all_reduce)>(dlsym(h, "ncclAllReduce"));
    api.error_string = reinterpret_cast<decltype(api.error_string)>(dlsym(h, "ncclGetErrorString"));
    if (!api.get_unique_id || !api.comm_init_rank || !api.comm_destroy || !api.all_gather ||
        !api.all_reduce)
        return false;
    g_rccl = api;
    return true;
}

int rccl_fail(sl_ctx* ctx, const char* what, int rc) {
    return sl_fail(ctx, SL_ERR_HIP, "%s failed: %s", what,
                   g_rccl.error_string ? g_rccl.error_string(rc) : "RCCL error");
}

}  // namespace

extern "C" int sl_comm_unique_id(unsigned char* id_out) {
    if (!id_out) return sl_fail(nullptr, SL_ERR_INVALID, "sl_comm_unique_id: NULL");
    if (!rccl_bind()) return sl_fail(nullptr, SL_ERR_UNSUPPORTED, "RCCL (librccl.so) not found");
    rccl_unique_id id;
    const int rc = g_rccl.get_unique_id(&id);
    if (rc) return rccl_fail(nullptr, "ncclGetUniqueId", rc);
    memcpy(id_out, id.internal, SL_COMM_ID_BYTES);
    return SL_OK;
}

extern "C" int sl_comm_init(sl_ctx* ctx, const unsigned char* id_in, int rank, int world) {
    if (!ctx || !id_in || world < 1 || rank < 0 || rank >= world)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_comm_init: bad argument");
    if (ctx->comm) return sl_fail(ctx, SL_ERR_INVALID, "sl_comm_init: communicator already set");
    if (!rccl_bind()) return sl_fail(ctx, SL_ERR_UNSUPPORTED, "RCCL (librccl.so) not found");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    rccl_unique_id id;
    memcpy(id.internal, id_in, SL_COMM_ID_BYTES);
    rccl_comm comm = nullptr;
    const int rc = g_rccl.comm_init_rank(&comm, world, id, rank);
    if (rc) return rccl_fail(ctx, "ncclCommInitRank", rc);
    SL_HIP_CHECK(ctx, hipMalloc(&ctx->d_comm_records, sizeof(sl_sweep_result) * (size_t)world));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return SL_OK;
}

extern "C" int sl_comm_destroy(sl_ctx* ctx) {
    if (!ctx) return SL_OK;
    if (ctx->comm) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        g_rccl.comm_destroy(static_cast<rccl_comm>(ctx->comm));
        ctx->comm = nullptr;
    }
    if (ctx->d_comm_records) (void)hipFree(ctx->d_comm_records);
    ctx->d_comm_records = nullptr;
    ctx->comm_world = 1;
    ctx->comm_rank = 0;
    return SL_OK;
}

static int need_comm(sl_ctx* ctx, const char* who) {
    if (!ctx) return sl_fail(nullptr, SL_ERR_INVALID, "%s: NULL context", who);
    if (!ctx->comm) return sl_fail(ctx, SL_ERR_INVALID, "%s: call sl_comm_init first", who);
    return SL_OK;
}

extern "C" int sl_allreduce_result(sl_ctx* ctx, sl_sweep_result* d_result) {
    int rc = need_comm(ctx, "sl_allreduce_result");
    if (rc) return rc;
    if (!d_result) return sl_fail(ctx, SL_ERR_INVALID, "sl_allreduce_result: NULL record");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    rc = g_rccl.all_gather(d_result, ctx->d_comm_records, sizeof(sl_sweep_result), RCCL_UINT8,
                           static_cast<rccl_comm>(ctx->comm), ctx->stream);
    if (rc) return rccl_fail(ctx, "ncclAllGather", rc);
    return sl_fold_results(ctx, static_cast<const sl_sweep_result*>(ctx->d_comm_records),
                           ctx->comm_world, d_result);
}

extern "C" int sl_allgather(sl_ctx* ctx, const void* d_send, void* d_recv, int64_t bytes_per_rank) {
    int rc = need_comm(ctx, "sl_allgather");
    if (rc) return rc;
    if (!d_send || !d_recv || bytes_per_rank < 0)
        return sl_fail(ctx, SL_ERR_INVALID, "sl_allgather: bad argument");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    rc = g_rccl.all_gather(d_send, d_recv, (size_t)bytes_per_rank, RCCL_UINT8,
                           static_cast<rccl_comm>(ctx->comm), ctx->stream);
    return rc ? rccl_fail(ctx, "ncclAllGather", rc) : SL_OK;
}

extern "C" int sl_allreduce_sum_u64(sl_ctx* ctx, uint64_t* d_values, int64_t count) {
    int rc = need_comm(ctx, "sl_allreduce_sum_u64");
    if (rc) return rc;
    if (!d_values || count < 0) return sl_fail(ctx, SL_ERR_INVALID, "sl_allreduce_sum_u64: bad argument");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    rc = g_rccl.all_reduce(d_values, d_values, (size_t)count, RCCL_UINT64, RCCL_SUM,
                           static_cast<rccl_comm>(ctx->comm), ctx->stream);
    return rc ? rccl_fail(ctx, "ncclAllReduce", rc) : SL_OK;
}

extern "C" int sl_allreduce_max_f64(sl_ctx* ctx, double* d_values, int64_t count) {
    int rc = need_comm(ctx, "sl_allreduce_max_f64");
    if (rc) return rc;
    if (!d_values || count < 0) return sl_fail(ctx, SL_ERR_INVALID, "sl_allreduce_max_f64: bad argument");
    SL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    rc = g_rccl.all_reduce(d_values, d_values, (size_t)count, RCCL_FLOAT64, RCCL_MAX,
                           static_cast<rccl_comm>(ctx->comm), ctx->stream);
    return rc ? rccl_fail(ctx, "ncclAllReduce", rc) : SL_OK;
}
