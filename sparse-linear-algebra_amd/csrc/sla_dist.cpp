// sla_dist.cpp -- RCCL plumbing for the row-sharded path (one process per GPU, 1-D row blocks).
//
// The reference has no communication layer at all (pure single-threaded Haskell); this is the
// exchange step the row-sharded (#>) needs: an all-gather of the SpMV input vector over xGMI per
// SpMV, plus all-gathers of one or two per-rank partial sums per inner-product group (summed in
// rank order by every rank, so all ranks take bit-identical decisions).
//
// librccl is resolved lazily with dlopen so that the single-GPU path carries no RCCL dependency
// (and so that a process that already imported torch shares torch's copy of librccl.so.1).
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "sla_internal.hpp"

namespace sla {

namespace {
typedef struct { char internal[128]; } NcclUniqueId;
typedef void *NcclComm;
typedef int (*fn_get_unique_id)(NcclUniqueId *);
typedef int (*fn_comm_init_rank)(NcclComm *, int, NcclUniqueId, int);
typedef int (*fn_comm_destroy)(NcclComm);
typedef int (*fn_all_gather)(const void *, void *, size_t, int, NcclComm, hipStream_t);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, NcclComm, hipStream_t);
typedef const char *(*fn_err_string)(int);
constexpr int kNcclFloat64 = 8, kNcclInt32 = 2, kNcclMax = 2;

struct Rccl {
    void *handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_err_string err_string = nullptr;
    std::string load_error;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) {
            r.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            r.load_error = std::string("dlopen(librccl) failed: ") + dlerror();
            return;
        }
        r.get_unique_id = (fn_get_unique_id)dlsym(r.handle, "ncclGetUniqueId");
        r.comm_init_rank = (fn_comm_init_rank)dlsym(r.handle, "ncclCommInitRank");
        r.comm_destroy = (fn_comm_destroy)dlsym(r.handle, "ncclCommDestroy");
        r.all_gather = (fn_all_gather)dlsym(r.handle, "ncclAllGather");
        r.all_reduce = (fn_all_reduce)dlsym(r.handle, "ncclAllReduce");
        r.err_string = (fn_err_string)dlsym(r.handle, "ncclGetErrorString");
        if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather || !r.all_reduce)
            r.load_error = "librccl is missing a required symbol";
    });
    return r;
}

int rccl_fail(const char *what, int rc) {
    Rccl &r = rccl();
    std::string msg = std::string(what) + " failed";
    if (r.err_string) msg += std::string(": ") + r.err_string(rc);
    return fail(SLA_ERR_RCCL, msg);
}
}  // namespace

int dist_unique_id(void *out128) {
    Rccl &r = rccl();
    if (!r.load_error.empty()) return fail(SLA_ERR_RCCL, r.load_error);
    NcclUniqueId id;
    int rc = r.get_unique_id(&id);
    if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(out128, id.internal, 128);
    return SLA_OK;
}

int dist_comm_init(sla_ctx *ctx, const void *unique_id) {
    Rccl &r = rccl();
    if (!r.load_error.empty()) return fail(SLA_ERR_RCCL, r.load_error);
    NcclUniqueId id;
    memcpy(id.internal, unique_id, 128);
    NcclComm comm = nullptr;
    int rc = r.comm_init_rank(&comm, ctx->nranks, id, ctx->rank);
    if (rc != 0) return rccl_fail("ncclCommInitRank", rc);
    ctx->comm = comm;
    return SLA_OK;
}

int dist_comm_destroy(sla_ctx *ctx) {
    if (ctx->comm) {
        rccl().comm_destroy((NcclComm)ctx->comm);
        ctx->comm = nullptr;
    }
    return SLA_OK;
}

int dist_allgather_f64(sla_ctx *ctx, const double *send, double *recv, int64_t count) {
    if (!ctx->comm) return fail(SLA_ERR_RCCL, "all-gather requested on a context without a communicator");
    int rc = rccl().all_gather(send, recv, (size_t)count, kNcclFloat64, (NcclComm)ctx->comm, ctx->stream);
    if (rc != 0) return rccl_fail("ncclAllGather", rc);
    return SLA_OK;
}

// max over ranks of a host int (used for the global isDiagonalSM / method agreement); synchronises
int dist_allreduce_max_i32(sla_ctx *ctx, int *value_host) {
    if (!ctx->collectives || !ctx->comm) return SLA_OK;
    int *d = (int *)ctx->d_result;
    SLA_HIP_TRY(hipMemcpyAsync(d, value_host, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    int rc = rccl().all_reduce(d, d, 1, kNcclInt32, kNcclMax, (NcclComm)ctx->comm, ctx->stream);
    if (rc != 0) return rccl_fail("ncclAllReduce", rc);
    SLA_HIP_TRY(hipMemcpyAsync(value_host, d, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SLA_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SLA_OK;
}

}  // namespace sla
