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

#include <algorithm>
#include <chrono>
#include <thread>
#include <condition_variable>
#include <map>
#include <mutex>
#include <cmath>
#include <vector>

#include "sla_internal.hpp"

namespace sla {

namespace {
typedef struct { char internal[128]; } NcclUniqueId;
typedef void *NcclComm;
typedef int (*fn_get_unique_id)(NcclUniqueId *);
typedef int (*fn_comm_init_rank)(NcclComm *, int, NcclUniqueId, int);
typedef int (*fn_comm_destroy)(NcclComm);
typedef int (*fn_comm_count)(NcclComm, int *);
typedef int (*fn_all_gather)(const void *, void *, size_t, int, NcclComm, hipStream_t);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, NcclComm, hipStream_t);
typedef int (*fn_reduce_scatter)(const void *, void *, size_t, int, int, NcclComm, hipStream_t);
typedef int (*fn_send)(const void *, size_t, int, int, NcclComm, hipStream_t);
typedef int (*fn_recv)(void *, size_t, int, int, NcclComm, hipStream_t);
typedef int (*fn_group)(void);
typedef const char *(*fn_err_string)(int);
constexpr int kNcclFloat64 = 8, kNcclInt32 = 2, kNcclMax = 2;

struct Rccl {
    void *handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_comm_count comm_count = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_reduce_scatter reduce_scatter = nullptr;
    fn_send send = nullptr;
    fn_recv recv = nullptr;
    fn_group group_start = nullptr, group_end = nullptr;
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
        r.comm_count = (fn_comm_count)dlsym(r.handle, "ncclCommCount");
        r.all_gather = (fn_all_gather)dlsym(r.handle, "ncclAllGather");
        r.all_reduce = (fn_all_reduce)dlsym(r.handle, "ncclAllReduce");
        r.reduce_scatter = (fn_reduce_scatter)dlsym(r.handle, "ncclReduceScatter");
        r.send = (fn_send)dlsym(r.handle, "ncclSend");
        r.recv = (fn_recv)dlsym(r.handle, "ncclRecv");
        r.group_start = (fn_group)dlsym(r.handle, "ncclGroupStart");
        r.group_end = (fn_group)dlsym(r.handle, "ncclGroupEnd");
        r.err_string = (fn_err_string)dlsym(r.handle, "ncclGetErrorString");
        if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather || !r.all_reduce)
            r.load_error = "librccl is missing a required symbol";
    });
    return r;
}

// ---------------------------------------------------------------------------------------------------------
// Loopback "communicator" (TEST BACKEND): all ranks are host threads of ONE process driving contexts on ONE
// GPU; a collective = publish my pointer, barrier, copy what I need out of my peers' buffers, barrier.  Slow,
// but it runs the real sharded code path (row offsets, window exchange plan, rank-ordered sums,
// reduce-scatter layout) with P > 1 on a single-GPU box, where RCCL refuses to place two ranks.
// ---------------------------------------------------------------------------------------------------------
struct LoopGroup {
    int nranks = 0, joined = 0, left = 0;
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    uint64_t generation = 0;
    std::vector<const void *> ptr;
    std::vector<int64_t> aux;
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t gen = generation;
        if (++waiting == nranks) {
            waiting = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen; });
        }
    }
};
std::mutex g_loop_mu;
std::map<int, LoopGroup *> g_loop_groups;

LoopGroup *loop_of(sla_ctx *c) { return (LoopGroup *)c->loop; }

// SLA_FAULT_INJECT (test hook, read once): "p2p" makes every grouped send / recv flow of this file fail with SLA_ERR_RCCL before it
// touches the communicator, "p2p_hang" makes the pre-flight's send / recv phase sleep on the host forever (no GPU work in flight):
// what bench.py's fallback ladder and staged watchdog are rehearsed with on a one-GPU box.  "p2p_data_rank1": the pre-flight's send / recv
// phase completes everywhere but RANK 1 ALONE reports that the data arrived wrong -- an asymmetric failure: the ranks must still end
// up on the same flow (bench.py agrees on the fallback over its control plane).
int fault_inject() {
    static const int f = [] {
        const char *e = getenv("SLA_FAULT_INJECT");
        return !e ? 0 : strcmp(e, "p2p") == 0 ? 1 : strcmp(e, "p2p_hang") == 0 ? 2 : strcmp(e, "p2p_data_rank1") == 0 ? 3 : 0;
    }();
    return f;
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

int dist_loopback_join(sla_ctx *ctx, int group_key) {
    std::lock_guard<std::mutex> lk(g_loop_mu);
    LoopGroup *&g = g_loop_groups[group_key];
    if (!g) {
        g = new LoopGroup();
        g->nranks = ctx->nranks;
        g->ptr.assign((size_t)ctx->nranks, nullptr);
        g->aux.assign((size_t)ctx->nranks, 0);
    }
    if (g->nranks != ctx->nranks) return fail(SLA_ERR_INVALID, "loopback group joined with a different nranks");
    g->joined++;
    ctx->loop = g;
    return SLA_OK;
}

int dist_comm_destroy(sla_ctx *ctx) {
    if (ctx->loop) {
        std::lock_guard<std::mutex> lk(g_loop_mu);
        LoopGroup *g = (LoopGroup *)ctx->loop;
        if (++g->left == g->nranks) {
            for (auto it = g_loop_groups.begin(); it != g_loop_groups.end(); ++it)
                if (it->second == g) { g_loop_groups.erase(it); break; }
            delete g;
        }
        ctx->loop = nullptr;
    }
    if (ctx->comm) {
        rccl().comm_destroy((NcclComm)ctx->comm);
        ctx->comm = nullptr;
    }
    return SLA_OK;
}

// how many ranks does the communicator behind this context span (what RCCL itself reports)
int dist_comm_count(sla_ctx *ctx, int *nranks) {
    *nranks = 1;
    if (LoopGroup *g = loop_of(ctx)) {
        *nranks = g->nranks;
    } else if (ctx->comm) {
        Rccl &r = rccl();
        if (!r.comm_count) return fail(SLA_ERR_RCCL, "librccl has no ncclCommCount");
        const int rc = r.comm_count((NcclComm)ctx->comm, nranks);
        if (rc != 0) return rccl_fail("ncclCommCount", rc);
    }
    return SLA_OK;
}

int dist_allgather_f64(sla_ctx *ctx, const double *send, double *recv, int64_t count) {
    if (LoopGroup *g = loop_of(ctx)) {
        SLA_HIP_TRY(hipStreamSynchronize(stream_of(ctx)));
        g->ptr[(size_t)ctx->rank] = send;
        g->barrier();
        for (int q = 0; q < ctx->nranks; ++q)
            if (recv + (size_t)q * (size_t)count != g->ptr[(size_t)q])   // (in place: the own slot is already there)
                SLA_HIP_TRY(hipMemcpyAsync(recv + (size_t)q * (size_t)count, g->ptr[(size_t)q], sizeof(double) * (size_t)count,
                                           hipMemcpyDeviceToDevice, stream_of(ctx)));
        SLA_HIP_TRY(hipStreamSynchronize(stream_of(ctx)));
        g->barrier();
        return SLA_OK;
    }
    if (!ctx->comm) return fail(SLA_ERR_RCCL, "all-gather requested on a context without a communicator");
    int rc = rccl().all_gather(send, recv, (size_t)count, kNcclFloat64, (NcclComm)ctx->comm, stream_of(ctx));
    if (rc != 0) return rccl_fail("ncclAllGather", rc);
    return SLA_OK;
}

// sum over ranks of full-length partial vectors, each rank keeping its shard (sharded transpose SpMV)
int dist_reduce_scatter_f64(sla_ctx *ctx, const double *send, double *recv, int64_t recvcount) {
    if (LoopGroup *g = loop_of(ctx)) {
        SLA_HIP_TRY(hipStreamSynchronize(stream_of(ctx)));
        g->ptr[(size_t)ctx->rank] = send;
        g->barrier();
        std::vector<double> acc((size_t)recvcount, 0.0), tmp((size_t)recvcount);
        for (int q = 0; q < ctx->nranks; ++q) {  // rank-ordered sum of the peers' segments for my shard
            SLA_HIP_TRY(hipMemcpy(tmp.data(), (const double *)g->ptr[(size_t)q] + (size_t)ctx->rank * (size_t)recvcount,
                                  sizeof(double) * (size_t)recvcount, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < acc.size(); ++i) acc[i] += tmp[i];
        }
        SLA_HIP_TRY(hipMemcpy(recv, acc.data(), sizeof(double) * (size_t)recvcount, hipMemcpyHostToDevice));
        g->barrier();
        return SLA_OK;
    }
    Rccl &r = rccl();
    if (!ctx->comm || !r.reduce_scatter) return fail(SLA_ERR_RCCL, "reduce-scatter requested without a communicator");
    int rc = r.reduce_scatter(send, recv, (size_t)recvcount, kNcclFloat64, 0 /* ncclSum */, (NcclComm)ctx->comm, stream_of(ctx));
    if (rc != 0) return rccl_fail("ncclReduceScatter", rc);
    return SLA_OK;
}

// Halo / window exchange: every rank receives only the x entries its rows reference and sends only what
// its peers' rows reference (contiguous ranges, one grouped ncclSend/ncclRecv per peer pair), then drops
// its own shard in place.  For a slab-partitioned stencil this is two plane-sized messages per SpMV
// instead of an all-gather of the whole vector.
int dist_exchange_window(sla_ctx *ctx, const XPlan &plan, const double *xlocal, int64_t my_begin, int64_t n_local, double *xfull) {
    if (LoopGroup *g = loop_of(ctx)) {
        SLA_HIP_TRY(hipStreamSynchronize(stream_of(ctx)));
        g->ptr[(size_t)ctx->rank] = xlocal;
        g->aux[(size_t)ctx->rank] = my_begin;
        g->barrier();
        if (n_local > 0 && xfull + my_begin != xlocal)
            SLA_HIP_TRY(hipMemcpyAsync(xfull + my_begin, xlocal, sizeof(double) * (size_t)n_local, hipMemcpyDeviceToDevice, stream_of(ctx)));
        for (int q = 0; q < ctx->nranks; ++q) {
            if (q == ctx->rank || plan.recv_len[(size_t)q] <= 0) continue;
            const double *src = (const double *)g->ptr[(size_t)q] + (plan.recv_begin[(size_t)q] - g->aux[(size_t)q]);
            SLA_HIP_TRY(hipMemcpyAsync(xfull + plan.recv_begin[(size_t)q], src, sizeof(double) * (size_t)plan.recv_len[(size_t)q],
                                       hipMemcpyDeviceToDevice, stream_of(ctx)));
        }
        SLA_HIP_TRY(hipStreamSynchronize(stream_of(ctx)));
        g->barrier();
        return SLA_OK;
    }
    Rccl &r = rccl();
    if (!ctx->comm) return fail(SLA_ERR_RCCL, "window exchange requested on a context without a communicator");
    if (fault_inject() == 1) return fail(SLA_ERR_RCCL, "fault injected: grouped ncclSend / ncclRecv (SLA_FAULT_INJECT=p2p)");
    if (!r.send || !r.recv || !r.group_start || !r.group_end) return fail(SLA_ERR_RCCL, "librccl lacks ncclSend/ncclRecv");
    if (n_local > 0 && xfull + my_begin != xlocal)
        SLA_HIP_TRY(hipMemcpyAsync(xfull + my_begin, xlocal, sizeof(double) * (size_t)n_local, hipMemcpyDeviceToDevice, stream_of(ctx)));
    int rc = r.group_start();
    if (rc != 0) return rccl_fail("ncclGroupStart", rc);
    for (int q = 0; q < ctx->nranks; ++q) {
        if (q == ctx->rank) continue;
        if (plan.recv_len[(size_t)q] > 0) {
            rc = r.recv(xfull + plan.recv_begin[(size_t)q], (size_t)plan.recv_len[(size_t)q], kNcclFloat64, q, (NcclComm)ctx->comm, stream_of(ctx));
            if (rc != 0) { r.group_end(); return rccl_fail("ncclRecv", rc); }
        }
        if (plan.send_len[(size_t)q] > 0) {
            rc = r.send(xlocal + (plan.send_begin[(size_t)q] - my_begin), (size_t)plan.send_len[(size_t)q], kNcclFloat64, q, (NcclComm)ctx->comm, stream_of(ctx));
            if (rc != 0) { r.group_end(); return rccl_fail("ncclSend", rc); }
        }
    }
    rc = r.group_end();
    if (rc != 0) return rccl_fail("ncclGroupEnd", rc);
    return SLA_OK;
}

// The same all-gather written as grouped point-to-point transfers: inside a dist_group_begin/end pair together with a
// window exchange, everything is ONE pure send/recv group (the all-to-all pattern), i.e. one RCCL launch.
int dist_allgather_p2p_f64(sla_ctx *ctx, const double *send, double *recv, int64_t count) {
    if (fault_inject() == 1) return fail(SLA_ERR_RCCL, "fault injected: grouped ncclSend / ncclRecv (SLA_FAULT_INJECT=p2p)");
    if (loop_of(ctx) || !ctx->comm || ctx->nranks == 1) return dist_allgather_f64(ctx, send, recv, count);
    Rccl &r = rccl();
    if (!r.send || !r.recv || !r.group_start || !r.group_end) return dist_allgather_f64(ctx, send, recv, count);
    if (send != recv + (size_t)ctx->rank * (size_t)count)   // (in-place callers already hold their own slot)
        SLA_HIP_TRY(hipMemcpyAsync(recv + (size_t)ctx->rank * (size_t)count, send, sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, stream_of(ctx)));
    int rc = r.group_start();
    if (rc != 0) return rccl_fail("ncclGroupStart", rc);
    for (int q = 0; q < ctx->nranks; ++q) {   // every rank posts its transfers in the same (peer-ascending, recv-then-send) order
        if (q == ctx->rank) continue;
        rc = r.recv(recv + (size_t)q * (size_t)count, (size_t)count, kNcclFloat64, q, (NcclComm)ctx->comm, stream_of(ctx));
        if (rc != 0) { r.group_end(); return rccl_fail("ncclRecv", rc); }
        rc = r.send(send, (size_t)count, kNcclFloat64, q, (NcclComm)ctx->comm, stream_of(ctx));
        if (rc != 0) { r.group_end(); return rccl_fail("ncclSend", rc); }
    }
    rc = r.group_end();
    if (rc != 0) return rccl_fail("ncclGroupEnd", rc);
    return SLA_OK;
}

int dist_group_begin(sla_ctx *ctx) {
    if (loop_of(ctx) || !ctx->comm) return SLA_OK;   // (the loopback backend completes every collective before it returns)
    Rccl &r = rccl();
    if (!r.group_start) return fail(SLA_ERR_RCCL, "librccl lacks ncclGroupStart");
    const int rc = r.group_start();
    return rc != 0 ? rccl_fail("ncclGroupStart", rc) : SLA_OK;
}
int dist_group_end(sla_ctx *ctx) {
    if (loop_of(ctx) || !ctx->comm) return SLA_OK;
    Rccl &r = rccl();
    if (!r.group_end) return fail(SLA_ERR_RCCL, "librccl lacks ncclGroupEnd");
    const int rc = r.group_end();
    return rc != 0 ? rccl_fail("ncclGroupEnd", rc) : SLA_OK;
}

// Pure host planning, identical on every rank given the same `windows` table.
void plan_window_exchange(int nranks, int rank, int64_t n, const int64_t *windows, XPlan &plan) {
    const int64_t S = (n + nranks - 1) / nranks;
    auto own = [&](int q, int64_t &b, int64_t &e) { b = std::min<int64_t>(n, S * q); e = std::min<int64_t>(n, S * (q + 1)); };
    auto isect = [](int64_t a0, int64_t a1, int64_t b0, int64_t b1, int64_t &o0, int64_t &len) {
        const int64_t lo = std::max(a0, b0), hi = std::min(a1, b1);
        o0 = lo; len = hi > lo ? hi - lo : 0;
    };
    plan.send_begin.assign((size_t)nranks, 0); plan.send_len.assign((size_t)nranks, 0);
    plan.recv_begin.assign((size_t)nranks, 0); plan.recv_len.assign((size_t)nranks, 0);
    int64_t mb, me;
    own(rank, mb, me);
    int64_t worst_recv = 0;
    for (int p = 0; p < nranks; ++p) {       // received entries of the busiest rank decide the mode
        int64_t tot = 0;
        const int64_t w0 = windows[2 * p], w1 = windows[2 * p + 1] + 1;   // half-open
        for (int q = 0; q < nranks; ++q) {
            if (q == p) continue;
            int64_t qb, qe, o, len;
            own(q, qb, qe);
            isect(w0, w1, qb, qe, o, len);
            tot += len;
            if (p == rank) { plan.recv_begin[(size_t)q] = o; plan.recv_len[(size_t)q] = len; }
            if (q == rank) { plan.send_begin[(size_t)p] = o; plan.send_len[(size_t)p] = len; }
        }
        worst_recv = std::max(worst_recv, tot);
    }
    // a window exchange pays when the busiest rank receives well under what an all-gather delivers
    plan.use_window = nranks > 1 && worst_recv * 2 < (n - S > 0 ? n - S : 1);
}

// ---------------------------------------------------------------------------------------------------------
// Overlapped all-gather of x for all-gather-mode matrices on the tile form (see AgPlan in sla_internal.hpp).
// Pure host planning, identical on every rank (and what sla_plan_allgather_passes exports).
// ---------------------------------------------------------------------------------------------------------
void plan_allgather_passes(int nranks, int rank, int64_t n, int shift, int groups, int order, AgPlan &plan) {
    const int64_t W = (int64_t)1 << shift, S = (n + nranks - 1) / nranks;
    const int P = (int)((n + W - 1) / W);
    auto own = [&](int q, int64_t &b, int64_t &e) { b = std::min<int64_t>(n, S * q); e = std::min<int64_t>(n, S * (q + 1)); };
    plan.order = order;
    plan.P = P;
    plan.shift = shift;
    plan.nranks = nranks;
    plan.rank = rank;
    plan.groups.clear();
    if (order == 1) {   // whole shards, in source-rank order
        for (int q = 0; q < nranks; ++q) {
            int64_t b, e;
            own(q, b, e);
            plan.groups.push_back({AgPiece{q, b, e}});
        }
    } else {            // G column chunks of every shard, cut at panel boundaries
        const int G = std::max(1, groups);
        plan.groups.assign((size_t)G, {});
        for (int q = 0; q < nranks; ++q) {
            int64_t b, e;
            own(q, b, e);
            if (e <= b) continue;
            // the panels lying wholly inside the shard are dealt out evenly over the groups; the partial panels at its two ends --
            // each shared with a neighbouring shard -- both travel in group 0, so that a panel straddling two shards is complete
            // after the first group instead of after the last
            const int64_t jf = (b + W - 1) / W, je = e / W;
            if (je <= jf) {
                plan.groups[0].push_back(AgPiece{q, b, e});
                continue;
            }
            if (jf * W > b) plan.groups[0].push_back(AgPiece{q, b, jf * W});
            for (int g = 0; g < G; ++g) {
                const int64_t cb = (jf + (je - jf) * g / G) * W, ce = (jf + (je - jf) * (g + 1) / G) * W;
                if (ce > cb) plan.groups[(size_t)g].push_back(AgPiece{q, cb, ce});
            }
            if (e > je * W) plan.groups[0].push_back(AgPiece{q, je * W, e});
        }
    }
    plan.G = (int)plan.groups.size();
    // groups a panel needs on THIS rank: the last group that brings one of its columns (own columns need none)
    std::vector<int32_t> need((size_t)P, 0);
    for (int g = 0; g < plan.G; ++g)
        for (const AgPiece &pc : plan.groups[(size_t)g]) {
            if (pc.src == rank || pc.e <= pc.b) continue;
            for (int64_t j = pc.b / W; j <= (pc.e - 1) / W; ++j) need[(size_t)j] = std::max<int32_t>(need[(size_t)j], g + 1);
        }
    plan.vis.resize((size_t)P);
    for (int j = 0; j < P; ++j) plan.vis[(size_t)j] = j;
    if (order == 1) {
        for (int j = 1; j < P; ++j) need[(size_t)j] = std::max(need[(size_t)j], need[(size_t)j - 1]);   // ascending walk: what has been waited for stays waited for
    } else {
        std::stable_sort(plan.vis.begin(), plan.vis.end(), [&](int32_t a, int32_t b) { return need[(size_t)a] < need[(size_t)b]; });
    }
    plan.pass_ptr.clear();
    plan.pass_need.clear();
    for (int t = 0; t < P; ++t) {
        const int32_t nd = need[(size_t)plan.vis[(size_t)t]];
        if (t == 0 || nd != plan.pass_need.back()) {
            plan.pass_ptr.push_back(t);
            plan.pass_need.push_back(nd);
        }
    }
    plan.pass_ptr.push_back(P);
}

int dist_exchange_group(sla_ctx *ctx, const std::vector<AgPiece> &pieces, const double *xlocal, int64_t my_begin, double *xfull) {
    if (LoopGroup *g = loop_of(ctx)) {
        SLA_HIP_TRY(hipStreamSynchronize(stream_of(ctx)));
        g->ptr[(size_t)ctx->rank] = xlocal;
        g->aux[(size_t)ctx->rank] = my_begin;
        g->barrier();
        for (const AgPiece &pc : pieces) {
            if (pc.src == ctx->rank || pc.e <= pc.b) continue;
            const double *src = (const double *)g->ptr[(size_t)pc.src] + (pc.b - g->aux[(size_t)pc.src]);
            SLA_HIP_TRY(hipMemcpyAsync(xfull + pc.b, src, sizeof(double) * (size_t)(pc.e - pc.b), hipMemcpyDeviceToDevice, stream_of(ctx)));
        }
        SLA_HIP_TRY(hipStreamSynchronize(stream_of(ctx)));
        g->barrier();
        return SLA_OK;
    }
    if (ctx->nranks == 1) return SLA_OK;   // (a 1-rank communicator has no peers: nothing moves)
    {   // an empty group (shards narrower than a panel travel whole in group 0) is no launch at all -- on every rank alike
        bool any = false;
        for (const AgPiece &pc : pieces) any = any || pc.e > pc.b;
        if (!any) return SLA_OK;
    }
    Rccl &r = rccl();
    if (!ctx->comm) return fail(SLA_ERR_RCCL, "grouped all-gather requested on a context without a communicator");
    if (fault_inject() == 1) return fail(SLA_ERR_RCCL, "fault injected: grouped ncclSend / ncclRecv (SLA_FAULT_INJECT=p2p)");
    if (!r.send || !r.recv || !r.group_start || !r.group_end) return fail(SLA_ERR_RCCL, "librccl lacks ncclSend/ncclRecv");
    int rc = r.group_start();
    if (rc != 0) return rccl_fail("ncclGroupStart", rc);
    // every rank walks the same piece list: a pair of ranks posts its transfers in the same order on both sides
    for (const AgPiece &pc : pieces) {
        if (pc.e <= pc.b) continue;
        const size_t len = (size_t)(pc.e - pc.b);
        if (pc.src == ctx->rank) {
            for (int q = 0; q < ctx->nranks; ++q) {
                if (q == ctx->rank) continue;
                rc = r.send(xlocal + (pc.b - my_begin), len, kNcclFloat64, q, (NcclComm)ctx->comm, stream_of(ctx));
                if (rc != 0) { r.group_end(); return rccl_fail("ncclSend", rc); }
            }
        } else {
            rc = r.recv(xfull + pc.b, len, kNcclFloat64, pc.src, (NcclComm)ctx->comm, stream_of(ctx));
            if (rc != 0) { r.group_end(); return rccl_fail("ncclRecv", rc); }
        }
    }
    rc = r.group_end();
    if (rc != 0) return rccl_fail("ncclGroupEnd", rc);
    return SLA_OK;
}

// Rehearsal of the point-to-point transfers on ONE rank (sla_dist_p2p_selftest; tests/test_gpu_parity.py).  No node with more than
// one GPU was ever available to the author, and on a 1-rank communicator the exchanges have no peer: ncclSend / ncclRecv had never
// been CALLED on hardware.  Here the rank is its own peer: `pieces` grouped recv / send pairs on the context's stream -- the same
// dlsym'd entry points, argument order, datatype constant, group calls and stream the window exchange and the grouped all-gather
// use -- moving `count` doubles; the result is compared on the host.
int dist_p2p_selftest(sla_ctx *ctx, int64_t count, int pieces, double *max_abs_err) {
    if (loop_of(ctx) || !ctx->comm) return fail(SLA_ERR_INVALID, "sla_dist_p2p_selftest: needs a context with an RCCL communicator");
    if (count < 1 || pieces < 1 || pieces > count) return fail(SLA_ERR_INVALID, "sla_dist_p2p_selftest: bad sizes");
    Rccl &r = rccl();
    if (fault_inject() == 1) return fail(SLA_ERR_RCCL, "fault injected: grouped ncclSend / ncclRecv (SLA_FAULT_INJECT=p2p)");
    if (!r.send || !r.recv || !r.group_start || !r.group_end) return fail(SLA_ERR_RCCL, "librccl lacks ncclSend/ncclRecv");
    std::vector<double> h((size_t)count), back((size_t)count, 0.0);
    for (int64_t i = 0; i < count; ++i) h[(size_t)i] = 1.0 + (double)i * 0.5;
    double *src = nullptr, *dst = nullptr;
    SLA_HIP_TRY(dev_malloc(ctx, (void **)&src, sizeof(double) * (size_t)count));
    hipError_t e = dev_malloc(ctx, (void **)&dst, sizeof(double) * (size_t)count);
    if (e == hipSuccess) e = hipMemcpy(src, h.data(), sizeof(double) * (size_t)count, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemsetAsync(dst, 0, sizeof(double) * (size_t)count, stream_of(ctx));
    int rc = 0;
    const char *what = nullptr;
    if (e == hipSuccess) {
        rc = r.group_start();
        if (rc != 0) what = "ncclGroupStart";
        for (int p = 0; p < pieces && rc == 0; ++p) {
            const int64_t b = count * p / pieces, len = count * (p + 1) / pieces - b;
            rc = r.recv(dst + b, (size_t)len, kNcclFloat64, ctx->rank, (NcclComm)ctx->comm, stream_of(ctx));
            if (rc != 0) { what = "ncclRecv"; break; }
            rc = r.send(src + b, (size_t)len, kNcclFloat64, ctx->rank, (NcclComm)ctx->comm, stream_of(ctx));
            if (rc != 0) what = "ncclSend";
        }
        const int rc2 = r.group_end();
        if (rc == 0 && rc2 != 0) { rc = rc2; what = "ncclGroupEnd"; }
        if (rc == 0) e = hipStreamSynchronize(stream_of(ctx));
        if (rc == 0 && e == hipSuccess) e = hipMemcpy(back.data(), dst, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost);
    }
    (void)hipFree(src);
    if (dst) (void)hipFree(dst);
    if (rc != 0) return rccl_fail(what, rc);
    SLA_HIP_TRY(e);
    double err = 0.0;
    for (int64_t i = 0; i < count; ++i) err = std::max(err, std::fabs(back[(size_t)i] - h[(size_t)i]));
    if (max_abs_err) *max_abs_err = err;
    return SLA_OK;
}

// First contact of a multi-rank job (sla_dist_preflight): one collective at a time across the REAL ranks, checked on the host, so that a
// hang or an error names its collective before anything is timed.  phase 0: ncclAllGather; 1: the all-gather as ONE group of
// ncclSend / ncclRecv pairs between all ranks (the pattern of the halo exchange and of the overlapped all-gather); 2: the integer
// all-reduce (max) the lowering uses for its cross-rank decisions.  Every rank must call it with the same arguments.
int dist_preflight(sla_ctx *ctx, int phase, int64_t count, double *max_abs_err, double *ms) {
    if (count < 1 || phase < 0 || phase > 2) return fail(SLA_ERR_INVALID, "sla_dist_preflight: bad arguments");
    const auto t0 = std::chrono::steady_clock::now();
    double err = 0.0;
    if (phase == 2) {
        int v = 100 + ctx->rank;
        SLA_TRY(dist_allreduce_max_i32(ctx, &v));
        err = std::fabs((double)v - (double)(100 + ctx->nranks - 1));
    } else {
        if (phase == 1 && fault_inject() == 1) return fail(SLA_ERR_RCCL, "fault injected: grouped ncclSend / ncclRecv (SLA_FAULT_INJECT=p2p)");
        if (phase == 1 && fault_inject() == 2)
            for (;;) std::this_thread::sleep_for(std::chrono::seconds(1));
        const size_t n = (size_t)count, P = (size_t)ctx->nranks;
        std::vector<double> h(n), back(n * P, -1.0);
        for (size_t i = 0; i < n; ++i) h[i] = 1.0e6 * (double)(ctx->rank + 1) + (double)i;
        double *src = nullptr, *dst = nullptr;
        SLA_HIP_TRY(dev_malloc(ctx, (void **)&src, sizeof(double) * n));
        hipError_t e = dev_malloc(ctx, (void **)&dst, sizeof(double) * n * P);
        // (on the context's stream: a memset on the legacy stream is not ordered against a collective on this non-blocking one -- the first
        // 1-rank rehearsal of this function saw its own NaN fill land AFTER the gather)
        if (e == hipSuccess) e = hipMemcpyAsync(src, h.data(), sizeof(double) * n, hipMemcpyHostToDevice, stream_of(ctx));
        if (e == hipSuccess) e = hipMemsetAsync(dst, 0xff, sizeof(double) * n * P, stream_of(ctx));
        if (e == hipSuccess) e = hipStreamSynchronize(stream_of(ctx));
        int rc = SLA_OK;
        if (e == hipSuccess) rc = phase == 0 ? dist_allgather_f64(ctx, src, dst, count) : dist_allgather_p2p_f64(ctx, src, dst, count);
        if (rc == SLA_OK && e == hipSuccess) e = hipStreamSynchronize(stream_of(ctx));
        if (rc == SLA_OK && e == hipSuccess) e = hipMemcpy(back.data(), dst, sizeof(double) * n * P, hipMemcpyDeviceToHost);
        (void)hipFree(src);
        if (dst) (void)hipFree(dst);
        SLA_TRY(rc);
        SLA_HIP_TRY(e);
        for (size_t q = 0; q < P; ++q)
            for (size_t i = 0; i < n; ++i) {
                const double d = std::fabs(back[q * n + i] - (1.0e6 * (double)(q + 1) + (double)i));
                err = d == d ? std::max(err, d) : 1.0e300;
            }
    }
    if (phase == 1 && fault_inject() == 3 && ctx->rank == 1) err = 1.0;
    if (max_abs_err) *max_abs_err = err;
    if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return SLA_OK;
}

void ag_plan_free(AgPlan *p) {
    if (!p) return;
    for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
    if (p->d_vis) (void)hipFree(p->d_vis);
    if (p->d_yrun) (void)hipFree(p->d_yrun);
    delete p;
}

// After build_tiles: all-gather-mode tile matrices of a sharded context (or of a single-rank context rehearsing one rank's pass
// structure, option ag_sim_ranks) get the plan, its visit list on the device, the running-sum buffer and one event per group.
int build_ag_plan(sla_csr *A, bool failed) {
    sla_ctx *c = A->ctx;
    if (c->ag_groups <= 0 || c->overlap < 0) return SLA_OK;     // (options are per job: every rank sets the same)
    const bool sim = !c->collectives && c->ag_sim_ranks > 1;
    if (!sim) {
        if (!c->collectives) return SLA_OK;
        if (A->xplan && c->x_exchange != 1 && (A->xplan->use_window || c->x_exchange == 2)) return SLA_OK;   // window-mode matrices exchange halos instead (the mode is the same on every rank)
        // The exchange pattern must be the same on every rank, but the tile form is chosen per slab: if one rank's slab did not take it,
        // every rank keeps the plain ncclAllGather.
        int bad = (A->use_tiles && !failed) ? 0 : 1;
        SLA_TRY(dist_allreduce_max_i32(c, &bad));
        if (bad) return SLA_OK;
    } else if (!A->use_tiles || failed) {
        return SLA_OK;
    }
    AgPlan *pl = new AgPlan();
    pl->sim = sim;
    plan_allgather_passes(sim ? c->ag_sim_ranks : c->nranks, sim ? std::min(c->ag_sim_rank, c->ag_sim_ranks - 1) : c->rank, A->n, A->tl_shift,
                          c->ag_groups, c->ag_order, *pl);
    if (pl->P != A->tl_P) {   // (cannot happen: both are ceil(n / 2^shift))
        delete pl;
        return fail(SLA_ERR_INVALID, "all-gather pass plan and tile form disagree on the panel count");
    }
    hipError_t e = dev_malloc(c, (void **)&pl->d_vis, sizeof(int32_t) * (size_t)std::max(pl->P, 1));
    if (e == hipSuccess) e = hipMemcpy(pl->d_vis, pl->vis.data(), sizeof(int32_t) * (size_t)pl->P, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = dev_malloc(c, (void **)&pl->d_yrun, sizeof(double) * (size_t)std::max<int64_t>(A->rows, 1));
    for (int g = 0; g < pl->G && e == hipSuccess; ++g) {
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) pl->ev.push_back(ev);
    }
    if (e != hipSuccess) {
        ag_plan_free(pl);
        return fail(SLA_ERR_ALLOC, std::string("all-gather pass plan: ") + hipGetErrorString(e));
    }
    A->ag = pl;
    return SLA_OK;
}

bool ag_split(const sla_csr *A) {
    const sla_ctx *c = A->ctx;
    return A->ag && tiles_on(A) && c->overlap >= 0 && c->ag_groups > 0 && (A->ag->sim ? !c->collectives : c->collectives);
}

// max over ranks of a host int (used for the global isDiagonalSM / method agreement); synchronises
int dist_allreduce_max_i32(sla_ctx *ctx, int *value_host) {
    if (LoopGroup *g = loop_of(ctx)) {
        g->aux[(size_t)ctx->rank] = *value_host;
        g->barrier();
        int mx = *value_host;
        for (int q = 0; q < ctx->nranks; ++q) mx = std::max<int>(mx, (int)g->aux[(size_t)q]);
        g->barrier();
        *value_host = mx;
        return SLA_OK;
    }
    if (!ctx->collectives || !ctx->comm) return SLA_OK;
    int *d = (int *)ctx->d_result;
    SLA_HIP_TRY(hipMemcpyAsync(d, value_host, sizeof(int), hipMemcpyHostToDevice, stream_of(ctx)));
    int rc = rccl().all_reduce(d, d, 1, kNcclInt32, kNcclMax, (NcclComm)ctx->comm, stream_of(ctx));
    if (rc != 0) return rccl_fail("ncclAllReduce", rc);
    SLA_HIP_TRY(hipMemcpyAsync(value_host, d, sizeof(int), hipMemcpyDeviceToHost, stream_of(ctx)));
    SLA_HIP_TRY(hipStreamSynchronize(stream_of(ctx)));
    return SLA_OK;
}

}  // namespace sla
