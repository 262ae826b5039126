// sla_api.cpp -- C ABI: context, SpMatrix lowering, SpVector, (#>) (<#) (<.>) norm2 axpby.
// Reference citations per entry point are in include/sla_hip.h.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <limits>

#include "sla_internal.hpp"

namespace sla {

static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }

// ---- device binding (sla_internal.hpp) -------------------------------------------------------------------------
static thread_local const sla_ctx *t_bound = nullptr;
int g_debug_binding = [] { const char *e = getenv("SLA_DEBUG_BINDING"); return e ? atoi(e) : 0; }();
static std::atomic<long> g_binding_violations{0};
const sla_ctx *bound_ctx() { return t_bound; }
Bind::Bind(const sla_ctx *c) : prev(t_bound) {
    if (!c || !c->kids.empty()) return;   // (a parent context owns no device: its rank contexts are bound on their worker threads)
    if (!prev || prev->device != c->device) (void)hipSetDevice(c->device);
    t_bound = c;
}
Bind::~Bind() {
    if (prev && t_bound && prev->device != t_bound->device) (void)hipSetDevice(prev->device);
    t_bound = prev;
}
void binding_violation(const sla_ctx *c, const char *what) {
    const long k = ++g_binding_violations;
    if (k <= 20)
        fprintf(stderr, "[sla] BINDING VIOLATION #%ld: %s for context %p (rank %d, device %d) on a thread bound to %p\n", k, what, (const void *)c,
                c ? c->rank : -1, c ? c->device : -1, (const void *)t_bound);
    if (g_debug_binding >= 2) abort();
}
long binding_violations() { return g_binding_violations.load(); }
int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

ProfScope::ProfScope(sla_ctx *ctx, int kernel_id) : c(ctx), on(false) {
    if (kernel_id >= 0 && (c->prof_kernel == kernel_id || c->prof_kernel == SLA_KERNEL_ALL) && c->prof_count < c->prof_max) {
        on = true;
        c->prof_ids[(size_t)c->prof_count] = kernel_id;
        (void)hipEventRecord(c->prof_ev[2 * (size_t)c->prof_count], stream_of(c));
    }
}
ProfScope::~ProfScope() {
    if (on) {
        (void)hipEventRecord(c->prof_ev[2 * (size_t)c->prof_count + 1], stream_of(c));
        c->prof_count++;
    }
}

// A single-device handle combined with a bundle of a multi-device context (or with a handle of another context) passes the
// dimension checks -- a bundle carries n but no device pointer -- and would fault on the device: refuse it here.
int mixed_handles(const char *what) {
    return fail(SLA_ERR_INVALID, std::string(what) + ": operands must all come from the same context (a multi-device bundle cannot be combined with a single-device handle)");
}

static int64_t shard_of(const sla_ctx *c, int64_t n) { return (n + c->nranks - 1) / c->nranks; }

static void row_range(const sla_ctx *c, int64_t m, int64_t *b, int64_t *e) {
    const int64_t s = shard_of(c, m);
    *b = std::min<int64_t>(m, s * c->rank);
    *e = std::min<int64_t>(m, s * (c->rank + 1));
}

static int ensure_xfull(sla_ctx *c, int64_t count) {
    if (c->xfull_cap >= count) return SLA_OK;
    if (c->d_xfull) (void)guard_free(c->d_xfull);
    c->d_xfull = nullptr;
    c->xfull_cap = 0;
    SLA_HIP_TRY(guard_malloc(c, (void **)&c->d_xfull, sizeof(double) * (size_t)std::max<int64_t>(count, 1)));
    c->xfull_cap = count;
    return SLA_OK;
}

// full-length gather base for an SpMV whose input is `x` (all-gather over xGMI when sharded)
int gather_raw(sla_ctx *c, const sla_csr *A, const double *local, int64_t shard, const double **base) {
    if (!c->collectives) {
        *base = local;
        return SLA_OK;
    }
    SLA_TRY(ensure_xfull(c, shard * c->nranks));
    if (A && A->xplan && c->x_exchange != 1 && (A->xplan->use_window || c->x_exchange == 2)) {
        const int64_t b = std::min<int64_t>(A->n, shard * c->rank), e = std::min<int64_t>(A->n, shard * (c->rank + 1));
        ProfScope prof(c, SLA_KERNEL_EXCHANGE);
        SLA_TRY(dist_exchange_window(c, *A->xplan, local, b, e - b, c->d_xfull));
    } else {
        ProfScope prof(c, SLA_KERNEL_EXCHANGE);
        SLA_TRY(dist_allgather_f64(c, local, c->d_xfull, shard));
    }
    *base = c->d_xfull;
    return SLA_OK;
}
bool halo_inplace_extents(const sla_csr *A, const sla_vec *x, int64_t *left, int64_t *right) {
    const sla_ctx *c = x->ctx;
    if (!(c->collectives && c->halo_inplace && A && A->xplan && c->x_exchange != 1 && (A->xplan->use_window || c->x_exchange == 2)))
        return false;
    const XPlan &pl = *A->xplan;
    const int64_t b = x->begin, cap = (int64_t)(c->vec_guard / sizeof(double)) - 8;   // (8: the row-pair gathers' own slack)
    int64_t lo = b, hi = b + x->n_local;
    for (int q = 0; q < c->nranks; ++q)
        if (q != c->rank && pl.recv_len[(size_t)q] > 0) {
            if (pl.recv_begin[(size_t)q] < b - cap || pl.recv_begin[(size_t)q] + pl.recv_len[(size_t)q] > b + x->shard + cap) return false;
            lo = std::min(lo, pl.recv_begin[(size_t)q]);
            hi = std::max(hi, pl.recv_begin[(size_t)q] + pl.recv_len[(size_t)q]);
        }
    if (left) *left = b - lo;
    if (right) *right = hi - (b + x->n_local);
    return true;
}

int gather_x(const sla_csr *A, sla_vec *x, const double **base) {
    sla_ctx *c = x->ctx;
    if (c->collectives) {
        // the neighbours' planes fit the slack around this vector: receive them in place, gather from x - first_row
        const int64_t b = x->begin;
        if (halo_inplace_extents(A, x, nullptr, nullptr)) {
            const XPlan &pl = *A->xplan;
            static const bool dbg = getenv("SLA_DEBUG_EXCHANGE") != nullptr;
            if (dbg) fprintf(stderr, "[sla] rank %d: in-place halo exchange (own rows %lld..%lld)\n", c->rank, (long long)b, (long long)(b + x->n_local));
            ProfScope prof(c, SLA_KERNEL_EXCHANGE);
            SLA_TRY(dist_exchange_window(c, pl, x->d, b, x->n_local, x->d - b));
            *base = x->d - b;
            return SLA_OK;
        }
    }
    return gather_raw(c, A, x->d, x->shard, base);
}

// sharded only: all-gather every rank's referenced column window and derive the exchange plan
static int build_xplan(sla_csr *A, int64_t rows, const int64_t *rowptr, const int64_t *col) {
    sla_ctx *c = A->ctx;
    if (!c->collectives) return SLA_OK;
    int64_t w[2] = {1, 0};  // empty
    const int64_t nnz = rowptr[rows];
    if (nnz > 0) {
        w[0] = col[0];
        w[1] = col[0];
        for (int64_t i = 0; i < rows; ++i)
            if (rowptr[i + 1] > rowptr[i]) {  // canonical CSR: first / last entry of a row are its min / max
                w[0] = std::min(w[0], col[rowptr[i]]);
                w[1] = std::max(w[1], col[rowptr[i + 1] - 1]);
            }
    }
    // ship the two int64 as raw 8-byte words through the f64 all-gather (no arithmetic touches them)
    double *d = c->d_result + 64;
    SLA_HIP_TRY(hipMemcpyAsync(d, w, sizeof(w), hipMemcpyHostToDevice, stream_of(c)));
    SLA_TRY(dist_allgather_f64(c, d, d + 8, 2));
    std::vector<int64_t> all((size_t)2 * c->nranks);
    SLA_HIP_TRY(hipMemcpyAsync(all.data(), d + 8, sizeof(int64_t) * all.size(), hipMemcpyDeviceToHost, stream_of(c)));
    SLA_HIP_TRY(hipStreamSynchronize(stream_of(c)));
    A->xplan = new XPlan();
    plan_window_exchange(c->nranks, c->rank, A->n, all.data(), *A->xplan);
    return SLA_OK;
}

// Row-sharded comm / compute overlap.  For the wave-sliced forms the 512-row steps are split into the
// INTERIOR ones -- every row references columns of this rank's own block only: they can run while the halo exchange is in
// flight -- and the BOUNDARY ones, each list in the visiting order of the full walk (the plane-tiled `sched` when there is
// one).  Taken when the window exchange is in use and most steps are interior.
static int build_overlap_lists(sla_csr *A, int64_t m, int64_t n, int64_t row_begin, int64_t rows, const int64_t *rowptr, const int64_t *col) {
    sla_ctx *c = A->ctx;
    if (!c->collectives || c->overlap < 0 || !A->use_wdia || !A->xplan || m != n || rows == 0) return SLA_OK;
    if (!(A->xplan->use_window || c->x_exchange == 2) || c->x_exchange == 1) return SLA_OK;
    const int64_t lo = row_begin, hi = row_begin + rows;
    std::vector<char> bnd((size_t)A->nblk_wd, 0);
    for (int64_t i = 0; i < rows; ++i)
        if (rowptr[i + 1] > rowptr[i] && (col[rowptr[i]] < lo || col[rowptr[i + 1] - 1] >= hi)) bnd[(size_t)(i / 512)] = 1;   // canonical CSR: min / max column
    std::vector<int32_t> li, lb;
    for (int32_t t = 0; t < A->nblk_wd; ++t) {
        const int32_t s = A->h_wsched.empty() ? t : A->h_wsched[(size_t)t];
        (bnd[(size_t)s] ? lb : li).push_back(s);
    }
    if (lb.empty() || li.size() < lb.size()) return SLA_OK;   // nothing to exchange for / too little to hide it behind
    SLA_HIP_TRY(dev_malloc(c, (void **)&A->d_ov_int, sizeof(int32_t) * li.size()));
    SLA_HIP_TRY(dev_malloc(c, (void **)&A->d_ov_bnd, sizeof(int32_t) * lb.size()));
    SLA_HIP_TRY(hipMemcpy(A->d_ov_int, li.data(), sizeof(int32_t) * li.size(), hipMemcpyHostToDevice));
    SLA_HIP_TRY(hipMemcpy(A->d_ov_bnd, lb.data(), sizeof(int32_t) * lb.size(), hipMemcpyHostToDevice));
    A->ov_nint = (int32_t)li.size();
    A->ov_nbnd = (int32_t)lb.size();
    return SLA_OK;
}

// (#>) with its input exchange.  When the matrix has interior / boundary step lists and the halo lands in place around x:
//   compute stream:  ... producers of x | record(x ready) | interior launch .................. | wait(halo done) | boundary launch
//   comm stream:                         wait(x ready) | halo send / recv (RCCL) | record(halo done)
// so the exchange costs nothing beyond the boundary launch.  The fused partial sums of the two launches occupy consecutive
// slots (interior first): *np = their total.  SLA_OVERLAP=0 runs the very same launches with the exchange serialised on
// the compute stream (bit-identical results: same kernels, same partial layout).
int spmv_exchanged(sla_csr *A, sla_vec *x, SpmvLaunch l, int *np) {
    sla_ctx *c = A->ctx;
    if (np) *np = spmv_grid(A);
    if (!(overlap_split(A) && !l.x2 && !l.yinit && x->ctx == c && halo_inplace_extents(A, x, nullptr, nullptr))) {
        SLA_TRY(gather_x(A, x, &l.x));
        return launch_spmv(A, l);
    }
    const int64_t b = x->begin;
    if (c->overlap > 0) {
        if (!c->comm_stream) {
            SLA_HIP_TRY(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
            SLA_HIP_TRY(hipEventCreateWithFlags(&c->ev_x_ready, hipEventDisableTiming));
            SLA_HIP_TRY(hipEventCreateWithFlags(&c->ev_x_done, hipEventDisableTiming));
        }
        SLA_HIP_TRY(hipEventRecord(c->ev_x_ready, stream_of(c)));
        SLA_HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->ev_x_ready, 0));
        hipStream_t compute = c->stream;
        c->stream = c->comm_stream;   // (the exchange enqueues on "the context stream")
        int rc;
        {
            ProfScope prof(c, SLA_KERNEL_EXCHANGE);   // (events on the comm stream)
            rc = dist_exchange_window(c, *A->xplan, x->d, b, x->n_local, x->d - b);
        }
        c->stream = compute;
        SLA_TRY(rc);
        SLA_HIP_TRY(hipEventRecord(c->ev_x_done, c->comm_stream));
    } else {
        ProfScope prof(c, SLA_KERNEL_EXCHANGE);
        SLA_TRY(dist_exchange_window(c, *A->xplan, x->d, b, x->n_local, x->d - b));
    }
    l.x = x->d - b;
    SpmvLaunch li = l;
    li.part = 1;
    SLA_TRY(launch_spmv(A, li));
    if (c->overlap > 0) SLA_HIP_TRY(hipStreamWaitEvent(stream_of(c), c->ev_x_done, 0));
    const int gi = overlap_grid(A, 1);
    SpmvLaunch lb = l;
    lb.part = 2;
    lb.pres = nullptr;         // the residual test and the step count belong to the first launch
    lb.step_begin &= ~1;
    if (lb.p1) lb.p1 += gi;
    if (lb.p2) lb.p2 += gi;
    if (lb.p3) lb.p3 += gi;
    if (lb.p4) lb.p4 += gi;
    SLA_TRY(launch_spmv(A, lb));
    if (np) *np = gi + overlap_grid(A, 2);
    return SLA_OK;
}

// sums of one or two partial arrays -> host (global over ranks); synchronises the stream
int reduce_to_host(sla_ctx *c, const double *p1, const double *p2, int np, double *out) {
    SLA_TRY(launch_finalize(c, p1, p2, np, c->d_result));
    const double *src = c->d_result;
    if (c->collectives) {
        SLA_TRY(dist_allgather_f64(c, c->d_result, c->d_result + 16, 2));
        SLA_TRY(launch_finalize_cols(c, c->d_result + 16, c->nranks, 1, 2, 2, c->d_result + 8));
        src = c->d_result + 8;
    }
    SLA_HIP_TRY(hipMemcpyAsync(c->h_result, src, 2 * sizeof(double), hipMemcpyDeviceToHost, stream_of(c)));
    SLA_HIP_TRY(hipStreamSynchronize(stream_of(c)));
    out[0] = c->h_result[0];
    if (p2) out[1] = c->h_result[1];
    return SLA_OK;
}

static constexpr size_t kVecPoolMaxBytes = (size_t)16 << 30;  // keep at most 16 GiB of idle vector buffers

static hipError_t pool_alloc(sla_ctx *c, size_t bytes, void **p) {
    auto it = c->vec_pool.find(bytes);
    if (it != c->vec_pool.end()) {
        *p = it->second;
        c->vec_pool.erase(it);
        c->vec_pool_bytes -= bytes;
        return hipSuccess;
    }
    return guard_malloc(c, p, bytes, c->vec_guard);
}

static void pool_free(sla_ctx *c, void *p, size_t bytes) {
    if (!p) return;
    if (c && c->vec_pool_bytes + bytes <= kVecPoolMaxBytes && c->vec_pool.size() < 64) {
        c->vec_pool.emplace(bytes, p);
        c->vec_pool_bytes += bytes;
    } else {
        (void)guard_free(p, c ? c->vec_guard : kGuardBytes);
    }
}

int vec_alloc(sla_ctx *c, int64_t n, sla_vec **out) {
    sla_vec *v = new sla_vec();
    v->ctx = c;
    v->n = n;
    v->shard = shard_of(c, n);
    int64_t b, e;
    row_range(c, n, &b, &e);
    v->begin = b;
    v->n_local = e - b;
    hipError_t err = pool_alloc(c, sizeof(double) * (size_t)std::max<int64_t>(v->shard, 1), (void **)&v->d);
    if (err != hipSuccess) {
        delete v;
        return fail(SLA_ERR_ALLOC, std::string("hipMalloc(vector): ") + hipGetErrorString(err));
    }
    err = hipMemsetAsync(v->d, 0, sizeof(double) * (size_t)std::max<int64_t>(v->shard, 1), stream_of(c));
    if (err != hipSuccess) {
        pool_free(c, v->d, sizeof(double) * (size_t)std::max<int64_t>(v->shard, 1));
        delete v;
        return fail(SLA_ERR_HIP, std::string("hipMemsetAsync: ") + hipGetErrorString(err));
    }
    *out = v;
    return SLA_OK;
}

static int csr_upload(sla_ctx *c, int64_t m, int64_t n, int64_t row_begin, int64_t rows, const int64_t *rowptr,
                      const int64_t *col, const double *val, sla_csr **out, bool panel_view = false);

// A rank whose input fails validation still owes its peers the agreement collective of csr_upload (they would block in it
// forever): contribute "failed", then report the local error.
static int csr_reject(sla_ctx *c, int rc) {
    if (c && c->collectives) {
        const std::string msg = g_last_error;
        int agree = 2;
        (void)dist_allreduce_max_i32(c, &agree);
        set_error(msg);
    }
    return rc;
}

// Column panels for irregular matrices (see launch_spmv_panels): panel p = the entries with column in
// [p W, (p+1) W), as a CSR view over the same rows.  Worth it when x does not fit the XCD-private L2 and
// every row still has about one entry per panel.
static int build_panels(sla_csr *A, int64_t m, int64_t n, int64_t row_begin, int64_t rows, const int64_t *rowptr,
                        const int64_t *col, const double *val) {
    sla_ctx *c = A->ctx;
    const int64_t nnz = rowptr[rows];
    if (!c->panels || A->use_diag || A->xwin_fraction >= 0.5 || rows == 0) return SLA_OK;
    const int64_t W = std::max<int64_t>(c->panel_cols, 1);
    if (n <= 2 * W) return SLA_OK;                                   // x (nearly) fits the L2 already
    int64_t P = std::min<int64_t>((n + W - 1) / W, nnz / rows);      // >= ~1 entry per row per panel
    if (P < 2) return SLA_OK;
    const int64_t Wp = (n + P - 1) / P;
    std::vector<int64_t> cur(rowptr, rowptr + rows), prp((size_t)rows + 1), pcol;
    std::vector<double> pval;
    for (int64_t p = 0; p < P; ++p) {
        const int64_t chi = std::min<int64_t>(n, (p + 1) * Wp);
        pcol.clear();
        pval.clear();
        prp[0] = 0;
        for (int64_t i = 0; i < rows; ++i) {
            int64_t k = cur[(size_t)i];
            const int64_t e = rowptr[i + 1];
            while (k < e && col[k] < chi) {
                pcol.push_back(col[k]);
                pval.push_back(val[k]);
                ++k;
            }
            cur[(size_t)i] = k;
            prp[(size_t)i + 1] = (int64_t)pcol.size();
        }
        sla_csr *V = nullptr;
        SLA_TRY(csr_upload(c, m, n, row_begin, rows, prp.data(), pcol.data(), pval.data(), &V, true));
        V->is_panel_view = true;
        A->panels.push_back(V);
    }
    SLA_HIP_TRY(dev_malloc(c, (void **)&A->d_panel_y, sizeof(double) * (size_t)std::max<int64_t>(rows, 1)));
    return SLA_OK;
}

// The lowering analyses (dictionaries, codes, slice records) are row-parallel: run fn(t, lo, hi) over T contiguous row
// ranges whose boundaries are multiples of `align` rows, on T host threads (SLA_HOST_THREADS, default <= 16).
static int host_threads() {
    static const int t = [] {
        const char *s = getenv("SLA_HOST_THREADS");
        int v = s ? atoi(s) : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
        return std::max(1, std::min(v, 64));
    }();
    return t;
}
template <class F>
static int par_rows(int64_t rows, int64_t align, F fn, int64_t serial_below = 200000) {
    int T = host_threads();
    const int64_t units = (rows + align - 1) / align;
    if (units < 64 || rows < serial_below) T = 1;
    T = (int)std::min<int64_t>(T, std::max<int64_t>(units, 1));
    auto range = [&](int t, int64_t &lo, int64_t &hi) {
        lo = std::min<int64_t>(rows, units * t / T * align);
        hi = std::min<int64_t>(rows, units * (t + 1) / T * align);
    };
    if (T == 1) { fn(0, (int64_t)0, rows); return 1; }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) {
        int64_t lo, hi;
        range(t, lo, hi);
        th.emplace_back([=, &fn] { fn(t, lo, hi); });
    }
    for (auto &x : th) x.join();
    return T;
}

// ---------------------------------------------------------------------------------------------------------------
// csr_upload = one lowering analysis per function.  What they share travels in `Low`; SLA_LOW_LOCALS re-opens it under the
// names the analyses use.
// ---------------------------------------------------------------------------------------------------------------
struct Low {
    sla_ctx *c;
    sla_csr *A;
    int64_t m, n, row_begin, rows, nnz;
    const int64_t *rowptr, *col;
    const double *val;
    bool panel_view, dbg_lower;
    hipError_t err = hipSuccess;
    std::vector<int32_t> rb;            // row-block starts
    std::vector<int64_t> offs;          // sorted distinct diagonal offsets (<= 256) when the matrix has that structure
    std::vector<uint8_t> dcodes;        // per entry: index into offs
    // (every lowered array carries kArraySlack zeroed bytes behind its end: the pipelined stream kernel reads whole row blocks
    // with clamped, unconditional loads, and an empty row block at the very end of the matrix reads "its" first entry there)
    void upload(void **dst, const void *src, size_t bytes) {
        if (err != hipSuccess) return;
        err = dev_malloc(c, dst, bytes + kArraySlack);
        if (err == hipSuccess && bytes) err = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
        if (err == hipSuccess) err = hipMemset((char *)*dst + bytes, 0, kArraySlack);
    }
};
#define SLA_LOW_LOCALS(L)                                                                                                  \
    [[maybe_unused]] sla_ctx *c = (L).c;                                                                                   \
    [[maybe_unused]] sla_csr *A = (L).A;                                                                                   \
    [[maybe_unused]] const int64_t m = (L).m, n = (L).n, row_begin = (L).row_begin, rows = (L).rows, nnz = (L).nnz;        \
    [[maybe_unused]] const int64_t *rowptr = (L).rowptr, *col = (L).col;                                                   \
    [[maybe_unused]] const double *val = (L).val;                                                                          \
    [[maybe_unused]] const bool panel_view = (L).panel_view, dbg_lower = (L).dbg_lower;                                    \
    [[maybe_unused]] hipError_t &err = (L).err;                                                                            \
    [[maybe_unused]] std::vector<int32_t> &rb = (L).rb;                                                                    \
    [[maybe_unused]] std::vector<int64_t> &offs = (L).offs;                                                                \
    [[maybe_unused]] std::vector<uint8_t> &dcodes = (L).dcodes;                                                            \
    [[maybe_unused]] auto upload = [&](void **dst_, const void *src_, size_t bytes_) { (L).upload(dst_, src_, bytes_); }

// canonical CSR arrays (i32 columns, i32 / i64 row pointers) + the row-block tables of the general kernels
static void low_csr_arrays(Low &L) {
    SLA_LOW_LOCALS(L);
    // the values go up on a second host thread while this one narrows and uploads the indices (pageable copies are
    // bound by the staging memcpy of the calling thread, not by the link)
    hipError_t err_val = hipSuccess;
    struct Joiner {   // (a host allocation failing below must not unwind past a joinable thread)
        std::thread &t;
        ~Joiner() { if (t.joinable()) t.join(); }
    };
    std::thread val_up([&] {
        Bind bind(c);   // (a new thread starts on device 0)
        if (err_val == hipSuccess) err_val = dev_malloc(c, (void **)&A->d_val, sizeof(double) * (size_t)nnz + kArraySlack);
        if (err_val == hipSuccess && nnz) err_val = hipMemcpy(A->d_val, val, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice);
        if (err_val == hipSuccess) err_val = hipMemset((char *)A->d_val + sizeof(double) * (size_t)nnz, 0, kArraySlack);
    });
    Joiner val_up_joiner{val_up};
    {
        std::vector<int32_t> col32((size_t)nnz);
        par_rows(rows, 1, [&](int, int64_t lo, int64_t hi) {
            for (int64_t k = rowptr[lo]; k < rowptr[hi]; ++k) col32[(size_t)k] = (int32_t)col[k];
        });
        upload((void **)&A->d_col, col32.data(), sizeof(int32_t) * (size_t)nnz);
    }
    if (A->rp64) {
        upload(&A->d_rowptr, rowptr, sizeof(int64_t) * (size_t)(rows + 1));
        std::vector<int64_t> rbk(rb.size());
        for (size_t b = 0; b < rb.size(); ++b) rbk[b] = rowptr[rb[b]];
        upload(&A->d_rbk, rbk.data(), sizeof(int64_t) * rbk.size());
    } else {
        std::vector<int32_t> rbk(rb.size());
        for (size_t b = 0; b < rb.size(); ++b) rbk[b] = (int32_t)rowptr[rb[b]];
        upload(&A->d_rbk, rbk.data(), sizeof(int32_t) * rbk.size());
        std::vector<int32_t> rp32((size_t)rows + 1);
        for (int64_t i = 0; i <= rows; ++i) rp32[(size_t)i] = (int32_t)rowptr[i];
        upload(&A->d_rowptr, rp32.data(), sizeof(int32_t) * (size_t)(rows + 1));
    }
    upload((void **)&A->d_rb, rb.data(), sizeof(int32_t) * rb.size());
    val_up.join();
    if (err == hipSuccess) err = err_val;
}

// LDS x window of every row block (spmv_xwin_kernel) and the share of the entries that fall inside
static void low_xwin_statistics(Low &L) {
    SLA_LOW_LOCALS(L);
    if (!panel_view) {
        // LDS x window of each row block: kXWin columns starting kXWinHalo left of its first diagonal column
        std::vector<int32_t> rbw(rb.size(), 0);
        int64_t inside = 0, total = 0;
        const int64_t wmax = std::max<int64_t>(0, n - kXWin);
        std::vector<int64_t> part_in((size_t)host_threads(), 0), part_tot((size_t)host_threads(), 0);
        par_rows((int64_t)rb.size() - 1, 1, [&](int t, int64_t blo, int64_t bhi) {
            int64_t in = 0, tot = 0;
            for (int64_t b = blo; b < bhi; ++b) {
                const int64_t w = std::min<int64_t>(wmax, std::max<int64_t>(0, row_begin + rb[(size_t)b] - kXWinHalo));
                rbw[(size_t)b] = (int32_t)w;
                const int64_t k0 = rowptr[rb[(size_t)b]], k1 = rowptr[rb[(size_t)b + 1]];
                if (k1 - k0 > kNnzPerRowBlock) continue;  // long-row blocks gather from global memory
                tot += k1 - k0;
                for (int64_t k = k0; k < k1; ++k) in += (col[k] >= w && col[k] < w + kXWin) ? 1 : 0;
            }
            part_in[(size_t)t] = in;
            part_tot[(size_t)t] = tot;
        }, 4096);
        for (size_t t = 0; t < part_in.size(); ++t) { inside += part_in[t]; total += part_tot[t]; }
        A->xwin_fraction = total ? (double)inside / (double)total : 0.0;
        A->use_xwin = A->xwin_fraction >= 0.5;
        upload((void **)&A->d_rbw, rbw.data(), sizeof(int32_t) * rbw.size());
    }
}

// dictionary of diagonal offsets + 1-byte column codes (spmv_diag_kernel): <= 256 distinct col - row values
static void low_diagonal_dictionary(Low &L) {
    SLA_LOW_LOCALS(L);
    if (!panel_view) {
        // dictionary of diagonal offsets: worthwhile (and representable in a byte) when col - row takes at
        // most 256 distinct values, i.e. for stencil / banded structure
        bool ok = nnz > 0;
        {   // distinct offsets: per-thread sets, merged
            std::vector<std::vector<int64_t>> loc((size_t)host_threads());
            std::vector<char> bad((size_t)host_threads(), 0);
            par_rows(rows, 1, [&](int t, int64_t lo, int64_t hi) {
                std::vector<int64_t> &mine = loc[(size_t)t];
                for (int64_t i = lo; i < hi && !bad[(size_t)t]; ++i) {
                    const int64_t gr = row_begin + i;
                    for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
                        const int64_t d = col[k] - gr;
                        bool found = false;
                        for (int64_t o : mine) if (o == d) { found = true; break; }   // <= 256 entries: a linear scan is fine
                        if (!found) {
                            if (mine.size() == 256) { bad[(size_t)t] = 1; break; }
                            mine.push_back(d);
                        }
                    }
                }
            });
            for (size_t t = 0; t < loc.size() && ok; ++t) {
                if (bad[t]) ok = false;
                for (int64_t d : loc[t]) {
                    if (std::find(offs.begin(), offs.end(), d) != offs.end()) continue;
                    if (offs.size() == 256) { ok = false; break; }
                    offs.push_back(d);
                }
            }
            if (!ok) offs.clear();
        }
        if (ok) {
            std::sort(offs.begin(), offs.end());
            std::vector<int32_t> dict(256, (int32_t)offs.back());
            for (size_t t = 0; t < offs.size(); ++t) dict[t] = (int32_t)offs[t];
            std::vector<uint8_t> &codes = dcodes;
            codes.resize((size_t)nnz);
            par_rows(rows, 1, [&](int, int64_t lo, int64_t hi) {
                for (int64_t i = lo; i < hi; ++i) {
                    const int64_t gr = row_begin + i;
                    for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
                        codes[(size_t)k] = (uint8_t)(std::lower_bound(offs.begin(), offs.end(), col[k] - gr) - offs.begin());
                }
            });
            upload((void **)&A->d_code, codes.data(), codes.size());
            upload((void **)&A->d_dict, dict.data(), sizeof(int32_t) * dict.size());
            A->use_diag = true;
            A->ndiag = (int)offs.size();
        }
    }
}

// value-indexed forms for constant-coefficient stencils: (offset, value) pair dictionary + byte codes (spmv_vdict_kernel), the
// wave-sliced records and the plane-tiled visiting order (spmv_wdia_kernel)
static void low_value_indexed(Low &L) {
    SLA_LOW_LOCALS(L);
    if (!panel_view && !A->rp64 && nnz > 0 && A->max_row_nnz <= kVdMaxRowNnz) {
        // value-indexed form: dictionary of (col - row, value bit pattern) pairs, one byte per entry
        struct Pair { int64_t off; uint64_t bits; };
        auto bits_of = [](double v) { uint64_t u; memcpy(&u, &v, 8); return u; };
        constexpr int kSlots = 1024;                       // open addressing, <= 256 live keys
        struct PairTable {
            std::vector<int> slot = std::vector<int>(kSlots, -1);
            std::vector<Pair> pairs;
            int find(int64_t off, uint64_t bits, bool insert) {
                uint64_t h = ((uint64_t)off * 0x9E3779B97F4A7C15ull) ^ (bits * 0xC2B2AE3D27D4EB4Full);
                h ^= h >> 29;
                for (int i = (int)(h & (kSlots - 1));; i = (i + 1) & (kSlots - 1)) {
                    const int id = slot[(size_t)i];
                    if (id < 0) {
                        if (!insert || pairs.size() == 256) return -1;
                        slot[(size_t)i] = (int)pairs.size();
                        pairs.push_back({off, bits});
                        return (int)pairs.size() - 1;
                    }
                    if (pairs[(size_t)id].off == off && pairs[(size_t)id].bits == bits) return id;
                }
            }
        };
        PairTable tab;                                     // the matrix's table: per-thread tables, merged
        std::vector<Pair> &pairs = tab.pairs;
        auto find = [&](int64_t off, uint64_t bits, bool insert) -> int { return tab.find(off, bits, insert); };
        bool ok = true;
        {
            std::vector<PairTable> loc((size_t)host_threads());
            std::vector<char> bad((size_t)host_threads(), 0);
            par_rows(rows, 1, [&](int t, int64_t lo, int64_t hi) {
                PairTable &mine = loc[(size_t)t];
                for (int64_t i = lo; i < hi && !bad[(size_t)t]; ++i) {
                    const int64_t gr = row_begin + i;
                    for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
                        if (mine.find(col[k] - gr, bits_of(val[k]), true) < 0) { bad[(size_t)t] = 1; break; }
                }
            });
            for (size_t t = 0; t < loc.size() && ok; ++t) {
                if (bad[t]) ok = false;
                for (const Pair &pr : loc[t].pairs)
                    if (ok && find(pr.off, pr.bits, true) < 0) ok = false;
            }
        }
        if (ok) {
            // canonical table order: by offset, then by value bits (independent of the input order)
            std::vector<int> order(pairs.size()), rank(pairs.size());
            for (size_t t = 0; t < order.size(); ++t) order[t] = (int)t;
            std::sort(order.begin(), order.end(), [&](int x, int y) {
                return pairs[(size_t)x].off != pairs[(size_t)y].off ? pairs[(size_t)x].off < pairs[(size_t)y].off
                                                                      : pairs[(size_t)x].bits < pairs[(size_t)y].bits;
            });
            std::vector<int32_t> doff(256, 0);
            std::vector<double> dval(256, 0.0);
            for (size_t t = 0; t < order.size(); ++t) {
                rank[(size_t)order[t]] = (int)t;
                doff[t] = (int32_t)pairs[(size_t)order[t]].off;
                memcpy(&dval[t], &pairs[(size_t)order[t]].bits, 8);
            }
            std::vector<uint8_t> codes(((size_t)nnz + 3) / 4 * 4 + 16, 0);
            par_rows(rows, 1, [&](int, int64_t lo, int64_t hi) {   // (lookups only: the table is read-only now)
                for (int64_t i = lo; i < hi; ++i) {
                    const int64_t gr = row_begin + i;
                    for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
                        codes[(size_t)k] = (uint8_t)rank[(size_t)find(col[k] - gr, bits_of(val[k]), false)];
                }
            });
            upload((void **)&A->d_vcode, codes.data(), codes.size());
            upload((void **)&A->d_vdoff, doff.data(), sizeof(int32_t) * doff.size());
            upload((void **)&A->d_vdval, dval.data(), sizeof(double) * dval.size());
            A->use_vdict = true;
            A->npairs = (int)pairs.size();
            A->nblk_vd = (int32_t)((rows + kVdRows - 1) / kVdRows);
            // wave-sliced form: per 128 rows (two per lane) the sorted union of the pair codes with an even-row and
            // an odd-row lane mask each.  Taken when the slices are reasonably full (>= 1/4 of the row slots busy
            // on average) and x is addressable with a 32-bit byte offset.
            const int64_t nsl = (rows + 127) / 128;
            std::vector<int32_t> wptr((size_t)nsl + 1, 0);
            std::vector<uint64_t> wme, wmo;
            std::vector<double> wval;
            std::vector<int32_t> woff;
            std::vector<uint8_t> wcode;
            bool wok = n < ((int64_t)1 << 28);
            if (wok) {   // slices are independent: each thread builds the records of a contiguous range of slices
                struct Part { std::vector<uint64_t> me, mo; std::vector<double> v; std::vector<int32_t> o, cnt; std::vector<uint8_t> cd; };
                std::vector<Part> part((size_t)host_threads());
                const int T = par_rows(rows, 128, [&](int t, int64_t lo, int64_t hi) {
                    Part &P = part[(size_t)t];
                    uint64_t lane_mask[2][256];
                    for (int64_t rlo = lo; rlo < hi; rlo += 128) {
                        uint64_t present[4] = {0, 0, 0, 0};
                        const int64_t rhi = std::min<int64_t>(hi, rlo + 128);
                        for (int64_t i = rlo; i < rhi; ++i)
                            for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
                                const int cd = codes[(size_t)k];
                                if (!((present[cd >> 6] >> (cd & 63)) & 1)) {
                                    present[cd >> 6] |= 1ull << (cd & 63);
                                    lane_mask[0][cd] = lane_mask[1][cd] = 0;
                                }
                                lane_mask[(i - rlo) & 1][cd] |= 1ull << ((i - rlo) >> 1);
                            }
                        const size_t first = P.me.size();
                        for (int cd = 0; cd < 256; ++cd)   // ascending code = ascending (offset, value bits)
                            if ((present[cd >> 6] >> (cd & 63)) & 1) {
                                P.me.push_back(lane_mask[0][cd]);
                                P.mo.push_back(lane_mask[1][cd]);
                                P.v.push_back(dval[(size_t)cd]);
                                P.o.push_back(doff[(size_t)cd]);
                                P.cd.push_back((uint8_t)cd);
                            }
                        P.cnt.push_back((int32_t)(P.me.size() - first));
                    }
                });
                int64_t sl = 0;
                for (int t = 0; t < T && wok; ++t) {
                    const Part &P = part[(size_t)t];
                    for (int32_t cnt : P.cnt) {
                        if (cnt > kWdMaxSliceRecords) wok = false;   // the records past the 8 pipelined ones go chunk by chunk
                        wptr[(size_t)sl + 1] = wptr[(size_t)sl] + cnt;
                        ++sl;
                    }
                    wme.insert(wme.end(), P.me.begin(), P.me.end());
                    wmo.insert(wmo.end(), P.mo.begin(), P.mo.end());
                    wval.insert(wval.end(), P.v.begin(), P.v.end());
                    woff.insert(woff.end(), P.o.begin(), P.o.end());
                    wcode.insert(wcode.end(), P.cd.begin(), P.cd.end());
                }
                if ((int64_t)wme.size() * 32 > nnz + 2048) wok = false;   // < 1/4 full: the byte-code kernel is the better form
            }
            if (wok) {
                A->nwent = (int64_t)wme.size();
                for (int t = 0; t < 8; ++t) { wme.push_back(0); wmo.push_back(0); wval.push_back(0.0); woff.push_back(0); }
                upload((void **)&A->d_wptr, wptr.data(), sizeof(int32_t) * wptr.size());
                upload((void **)&A->d_wme, wme.data(), sizeof(uint64_t) * wme.size());
                upload((void **)&A->d_wmo, wmo.data(), sizeof(uint64_t) * wmo.size());
                upload((void **)&A->d_wval, wval.data(), sizeof(double) * wval.size());
                upload((void **)&A->d_woff, woff.data(), sizeof(int32_t) * woff.size());
                A->use_wdia = true;
                A->nslices = (int32_t)nsl;
                A->nblk_wd = (int32_t)((nsl + 3) / 4);
                {   // LDS windows: cluster the offsets (doff[] is ascending); a record then names an element of the staged buffer
                    WdWin W;
                    bool lok = true;
                    int32_t wmax[kWdWinMax] = {};
                    for (int t = 0; t < A->npairs && lok; ++t) {
                        const int32_t o = doff[(size_t)t];
                        if (W.n > 0 && o - wmax[W.n - 1] < kWdWinMerge) { wmax[W.n - 1] = o; continue; }
                        if (W.n == kWdWinMax) { lok = false; break; }
                        W.omin[W.n] = o & ~1;        // (floor to even: staged as aligned 16-byte pairs)
                        wmax[W.n] = o;
                        ++W.n;
                    }
                    for (int k = 0; k < W.n && lok; ++k) {
                        const int32_t elems = wmax[k] - W.omin[k] + 512 + 2;   // + 1: odd first row of a slab, + 1: second row of the last pair
                        W.pb[k + 1] = W.pb[k] + (elems + 1) / 2;
                    }
                    W.pairs = W.pb[W.n];
                    if (lok && W.n > 0 && W.pairs <= kWdWinMaxPairs && A->npairs <= 8) {
                        // uniform records: every slice carries all pairs in table order; a pair it does not use has empty masks
                        WdUni U;
                        U.n = A->npairs;
                        for (int t = 0; t < A->npairs; ++t) {
                            int k = 0;
                            while (k + 1 < W.n && doff[(size_t)t] >= W.omin[k + 1]) ++k;
                            U.lpos[t] = 2 * W.pb[k] + (doff[(size_t)t] - W.omin[k]);
                            U.val[t] = dval[(size_t)t];
                        }

                        std::vector<uint64_t> wum((size_t)nsl * 16, 0);
                        for (int64_t sl2 = 0; sl2 < nsl; ++sl2)
                            for (int32_t e = wptr[(size_t)sl2]; e < wptr[(size_t)sl2 + 1]; ++e) {
                                wum[(size_t)sl2 * 16 + wcode[(size_t)e]] = wme[(size_t)e];
                                wum[(size_t)sl2 * 16 + 8 + wcode[(size_t)e]] = wmo[(size_t)e];
                            }
                        upload((void **)&A->d_wum, wum.data(), sizeof(uint64_t) * wum.size());
                        int64_t clo = n, chi = -1;                // (columns ascend inside a row)
                        for (int64_t i = 0; i < rows; ++i)
                            if (rowptr[i + 1] > rowptr[i]) {
                                clo = std::min<int64_t>(clo, col[rowptr[i]]);
                                chi = std::max<int64_t>(chi, col[rowptr[i + 1] - 1]);
                            }
                        A->wd_col_lo = (int32_t)clo;
                        A->wd_col_hi = (int32_t)chi;
                        // x[own row] from the staged buffer (an epilogue operand that is the gathered vector): offset 0 inside a
                        // window, and every own row inside the column range the windows are filled for
                        if (clo <= row_begin && row_begin + rows - 1 <= chi)
                            for (int k = 0; k < W.n; ++k)
                                if (W.omin[k] <= 0 && 0 <= wmax[k]) U.lpos0 = 2 * W.pb[k] - W.omin[k];
                        A->wd_win = W;
                        A->wd_uni = U;
                        A->wd_lds = true;
                        // Plane march (spmv_wdia_march_kernel): three windows {-D}, {in-plane}, {+D} with ONE pair in each far
                        // window, D even, the in-plane window <= 512 pairs and around offset 0, an unsharded matrix.  The masks
                        // are laid out per (tile, plane, wavefront) for rows plane * D + tile * 512 + wavefront * 128 + [0, 128)
                        // -- a plane is not a whole number of 128-row slices (216^2 = 364.5 of them).
                        const int np = A->npairs;
                        const int64_t D = np >= 3 ? (int64_t)doff[(size_t)np - 1] : 0;
                        if (W.n == 3 && (np == 5 || np == 7) && row_begin == 0 && rows == m && m == n && D >= 1024 && (D & 1) == 0 &&
                            doff[0] == -D && doff[1] >= W.omin[1] && doff[(size_t)np - 2] <= wmax[1] && W.omin[1] <= 0 && wmax[1] >= 0 &&
                            W.pb[2] - W.pb[1] <= 512 && rows >= 2 * D) {
                            WdMarch G;
                            G.D = (int32_t)D;
                            G.T = (int32_t)((D + 511) / 512);
                            G.planes = (int32_t)((rows + D - 1) / D);
                            G.omin = W.omin[1];
                            G.pairs = W.pb[2] - W.pb[1];
                            const int slots = std::max(1, c->wd_march_occ) * c->n_cu;
                            G.S = std::max(1, std::min(G.planes, slots / G.T));
                            G.PS = (G.planes + G.S - 1) / G.S;
                            G.S = (G.planes + G.PS - 1) / G.PS;
                            G.ntasks = G.T * G.S;
                            WdUni M;
                            M.n = np;
                            for (int t = 0; t < np; ++t) {
                                const int64_t o = t == 0 ? 0 : t == np - 1 ? 0 : (int64_t)doff[(size_t)t];   // (relative to the pair's own plane)
                                M.lpos[t] = (int32_t)(o - G.omin);
                                M.val[t] = dval[(size_t)t];
                            }
                            M.lpos0 = -G.omin;
                            std::vector<uint64_t> wm((size_t)G.T * (size_t)G.planes * 4 * 16, 0);
                            par_rows(G.planes, 1, [&](int, int64_t plo, int64_t phi) {   // (the slices of a plane belong to one thread)
                                for (int64_t kk = plo; kk < phi; ++kk) {
                                    const int64_t r0 = kk * D, r1 = std::min<int64_t>(rows, r0 + D);
                                    for (int64_t i = r0; i < r1; ++i) {
                                        const int64_t pos = i - r0, tile = pos >> 9;
                                        const size_t sl2 = ((size_t)(tile * G.planes + kk) * 4 + (size_t)((pos >> 7) & 3)) * 16;
                                        const uint64_t bit = 1ull << ((pos & 127) >> 1);
                                        for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) wm[sl2 + (size_t)(pos & 1) * 8 + codes[(size_t)k]] |= bit;
                                    }
                                }
                            }, 4);
                            upload((void **)&A->d_wum_m, wm.data(), sizeof(uint64_t) * wm.size());
                            A->wd_mg = G;
                            A->wd_muni = M;
                            A->wd_march = true;
                        }
                    }
                }
                // Visiting order of the 512-row steps.  A 3-D stencil row touches x one PLANE (the far diagonal, D rows)
                // behind and ahead; swept in row order, a line of x is needed again 2 D rows later, by which time
                // the vectors streaming through the 4 MiB L2 have evicted it (216^3: D = 46656, 1.35 extra reads
                // of x measured).  So the sweep is tiled: the steps are grouped by their position inside the plane
                // (tiles of `tile` steps) and each tile is walked plane after plane, which makes the three touches
                // of a line neighbours in time.  Only the order changes; every step is still done exactly once.
                int64_t far = 0;
                for (int t = 0; t < A->npairs; ++t) far = std::max<int64_t>(far, std::llabs((long long)doff[(size_t)t]));
                const double bpp = (double)far / 512.0;   // steps per plane
                int tile = c->wd_tile;
                if (tile < 0) tile = (far * 8 >= (256 << 10) && far * 4 <= rows) ? (int)std::max(8.0, bpp / 6.0 + 0.5) : 0;
                if (tile > 0 && bpp > 2.0 * tile) {
                    std::vector<int32_t> sched((size_t)A->nblk_wd);
                    std::vector<int32_t> key((size_t)A->nblk_wd);
                    for (int32_t bb = 0; bb < A->nblk_wd; ++bb) {
                        sched[(size_t)bb] = bb;
                        const double pos = fmod((double)bb, bpp);   // position of the step inside its plane, in steps
                        key[(size_t)bb] = (int32_t)(pos / tile);
                    }
                    std::stable_sort(sched.begin(), sched.end(), [&](int32_t x, int32_t y) { return key[(size_t)x] < key[(size_t)y]; });
                    upload((void **)&A->d_wsched, sched.data(), sizeof(int32_t) * sched.size());
                    A->h_wsched = sched;
                }
            }
        }
    }
}

// wave-sliced form for variable coefficients (wdia-vv): diagonal records with per-row value blocks
static void low_wave_sliced_variable(Low &L) {
    SLA_LOW_LOCALS(L);
    if (!panel_view && A->use_diag && !A->use_wdia && !A->rp64 && c->wdia_vv && n < ((int64_t)1 << 28) && nnz > 0) {
        // Wave-sliced form for VARIABLE coefficients (banded / stencil structure, arbitrary values): per 128-row slice the
        // sorted union of its diagonal offsets with the two row masks each, and per record a block of 128 values laid out
        // like the rows (lane l holds rows 2l, 2l+1: one 16-byte load).  8 B per stored slot instead of 8 + 1 B per entry
        // plus rowptr, no codes, no LDS.  Taken when at least half of the slots hold an entry.
        const int64_t nsl = (rows + 127) / 128;
        std::vector<int32_t> wptr((size_t)nsl + 1, 0);
        std::vector<uint64_t> wme, wmo;
        std::vector<int32_t> woff;
        std::vector<double> wvb;
        uint64_t lane_mask[2][256];
        int slot_of[256];
        bool wok = true;
        for (int64_t sl = 0; sl < nsl && wok; ++sl) {
            uint64_t present[4] = {0, 0, 0, 0};
            const int64_t rlo = sl * 128, rhi = std::min<int64_t>(rows, rlo + 128);
            for (int64_t i = rlo; i < rhi; ++i)
                for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
                    const int cd = dcodes[(size_t)k];
                    if (!((present[cd >> 6] >> (cd & 63)) & 1)) {
                        present[cd >> 6] |= 1ull << (cd & 63);
                        lane_mask[0][cd] = lane_mask[1][cd] = 0;
                    }
                    lane_mask[(i - rlo) & 1][cd] |= 1ull << ((i - rlo) >> 1);
                }
            const size_t first = wme.size();
            for (int cd = 0; cd < 256; ++cd)   // ascending code = ascending offset
                if ((present[cd >> 6] >> (cd & 63)) & 1) {
                    slot_of[cd] = (int)(wme.size() - first);
                    wme.push_back(lane_mask[0][cd]);
                    wmo.push_back(lane_mask[1][cd]);
                    woff.push_back((int32_t)offs[(size_t)cd]);
                }
            wvb.resize(wme.size() * 128, 0.0);
            for (int64_t i = rlo; i < rhi; ++i)
                for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
                    wvb[(first + (size_t)slot_of[dcodes[(size_t)k]]) * 128 + (size_t)(i - rlo)] = val[k];
            wptr[(size_t)sl + 1] = (int32_t)wme.size();
            if ((int64_t)wme.size() * 64 > nnz + 4096) wok = false;   // less than half of the slots used
            if (wptr[(size_t)sl + 1] - wptr[(size_t)sl] > kWdMaxSliceRecords) wok = false;
        }
        if (wok) {
            A->nwent = (int64_t)wme.size();
            for (int t = 0; t < 8; ++t) { wme.push_back(0); wmo.push_back(0); woff.push_back(0); }
            wvb.resize(wme.size() * 128, 0.0);
            upload((void **)&A->d_wptr, wptr.data(), sizeof(int32_t) * wptr.size());
            upload((void **)&A->d_wme, wme.data(), sizeof(uint64_t) * wme.size());
            upload((void **)&A->d_wmo, wmo.data(), sizeof(uint64_t) * wmo.size());
            upload((void **)&A->d_woff, woff.data(), sizeof(int32_t) * woff.size());
            upload((void **)&A->d_wvblk, wvb.data(), sizeof(double) * wvb.size());
            A->use_wdia = true;
            A->wd_vv = true;
            A->nslices = (int32_t)nsl;
            A->nblk_wd = (int32_t)((nsl + 3) / 4);
        }
    }
}

// LDS-panel form for dense rows (spmv_lpanel_kernel): per (panel, row) entry ranges + the task runs of the persistent grid
static void low_lds_panels(Low &L) {
    SLA_LOW_LOCALS(L);
    if (!panel_view && !A->use_wdia && !A->use_vdict && rows > 0 && err == hipSuccess) {
        // LDS-panel form (spmv_lpanel_kernel): worthwhile when a row has enough entries per kLpW-column panel to
        // keep a wavefront's lanes busy, affordable when the (panel, row) pointer table stays a fraction of the matrix
        const int64_t P = (n + kLpW - 1) / kLpW;
        const int64_t W = ((n + P - 1) / P + 63) / 64 * 64;   // equal panels (a narrow last panel would be all short segments)
        const size_t rpsz = A->rp64 ? sizeof(int64_t) : sizeof(int32_t);
        const int64_t min_seg = c->lp_min_seg;
        if (P <= 4096 && nnz >= min_seg * rows * P && (P + 1) * rows * (int64_t)rpsz <= nnz * 12 / 4) {
            std::vector<int64_t> pp((size_t)((P + 1) * rows));
            par_rows(rows, 1, [&](int, int64_t lo, int64_t hi) {
                for (int64_t i = lo; i < hi; ++i) {
                    const int64_t *cb = col + rowptr[i], *ce = col + rowptr[i + 1];
                    const int64_t *cur = cb;
                    for (int64_t p = 0; p <= P; ++p) {
                        cur = std::lower_bound(cur, ce, p * W);
                        pp[(size_t)(p * rows + i)] = rowptr[i] + (cur - cb);
                    }
                }
            }, 4096);
            if (A->rp64) {
                upload(&A->d_lpp, pp.data(), sizeof(int64_t) * pp.size());
            } else {
                std::vector<int32_t> pp32(pp.begin(), pp.end());
                upload(&A->d_lpp, pp32.data(), sizeof(int32_t) * pp32.size());
            }
            if (err == hipSuccess) err = dev_malloc(c, (void **)&A->d_lpy, sizeof(double) * (size_t)(P * rows));
            // panel-major second copy of the entries (sla_spmv_lpanel.hip: lp_reorder_kernel), built on the device from the arrays just
            // uploaded; when it exists d_lpp is replaced by the P x rows + 1 segment starts into it.  12 B per entry: taken while it
            // (10 B: 16-bit panel offsets) stays below 48 GB and the allocation succeeds (a failure here is not an error: the row-major arrays serve)
            if (err == hipSuccess && c->lp_copy && nnz * 12 <= ((int64_t)48 << 30) && A->d_col && A->d_val) {
                std::vector<int64_t> q((size_t)(P * rows) + 1, 0);
                for (int64_t p = 0; p < P; ++p)
                    for (int64_t i = 0; i < rows; ++i)
                        q[(size_t)(p * rows + i) + 1] = q[(size_t)(p * rows + i)] + (pp[(size_t)((p + 1) * rows + i)] - pp[(size_t)(p * rows + i)]);
                void *dq = nullptr;
                uint16_t *c2 = nullptr;
                double *v2 = nullptr;
                hipError_t e2 = W <= 65536 ? dev_malloc(c, (void **)&c2, sizeof(uint16_t) * (size_t)nnz + kArraySlack) : hipErrorInvalidValue;
                if (e2 == hipSuccess) e2 = dev_malloc(c, (void **)&v2, sizeof(double) * (size_t)nnz + kArraySlack);
                if (e2 == hipSuccess) e2 = dev_malloc(c, &dq, rpsz * q.size() + kArraySlack);
                if (e2 == hipSuccess) {
                    if (A->rp64) {
                        e2 = hipMemcpy(dq, q.data(), sizeof(int64_t) * q.size(), hipMemcpyHostToDevice);
                    } else {
                        std::vector<int32_t> q32(q.begin(), q.end());
                        e2 = hipMemcpy(dq, q32.data(), sizeof(int32_t) * q32.size(), hipMemcpyHostToDevice);
                    }
                }
                if (e2 == hipSuccess && launch_lp_reorder(c, A->rp64, A->d_lpp, dq, A->d_col, A->d_val, c2, v2, rows, P, (int32_t)W) != SLA_OK) e2 = hipErrorUnknown;
                if (e2 == hipSuccess) {
                    (void)hipFree(A->d_lpp);
                    A->d_lpp = dq;
                    A->d_lpcol = c2;
                    A->d_lpval = v2;
                } else {
                    (void)hipGetLastError();
                    if (c2) (void)hipFree(c2);
                    if (v2) (void)hipFree(v2);
                    if (dq) (void)hipFree(dq);
                }
            }
            // row chunks: ~32 tasks per workgroup of the persistent grid (measured: 8 -> 0.936 ms, 32 -> 0.900 ms, 64 ->
            // 0.902 ms on the 200k-row 1 % matrix), at least 64 rows (4 per wavefront) each
            const int tasks_per_cu = std::max(1, c->lp_tasks);
            const int64_t want = std::max<int64_t>(1, (tasks_per_cu * (int64_t)c->n_cu + P - 1) / P);
            const int64_t chunk = std::max<int64_t>(64, (rows + want - 1) / want);
            const int64_t C = (rows + chunk - 1) / chunk;
            A->lp_P = (int32_t)P;
            A->lp_W = (int32_t)W;
            {   // lanes per segment ~ half the mean segment length (so that two strided loads cover a typical segment)
                const int64_t seg = nnz / (rows * P);
                // measured, 200 k rows x 13 panels, ms per (#>) [64 / 32 / 16 / 8 lanes]: segment 153: 0.90 / 1.15 / 1.33 / 1.80;
                // 92: 0.68 / 0.73 / 0.87 / 1.15; 61: 0.65 / 0.52 / 0.59 / 0.77; 30: 0.61 / 0.36 / 0.36 / 0.43; 15: 0.58 / 0.32 /
                // 0.26 / 0.27 (stream kernel: 2.21 / 1.39 / 1.07 / 0.56 / 0.28)
                A->lp_cfg = seg >= 80 ? 0 : seg >= 40 ? 1 : 2;
                if (c->lp_cfg >= 0) A->lp_cfg = std::min(3, c->lp_cfg);
            }
            int64_t clo = n, chi = -1;
            for (int64_t i = 0; i < rows; ++i)
                if (rowptr[i + 1] > rowptr[i]) {   // canonical CSR: first / last entry of a row are its min / max column
                    clo = std::min(clo, col[rowptr[i]]);
                    chi = std::max(chi, col[rowptr[i + 1] - 1]);
                }
            A->lp_col_lo = (int32_t)clo;
            A->lp_col_hi = (int32_t)chi;
            A->lp_chunk = (int32_t)chunk;
            A->lp_C = (int32_t)C;
            // tasks (panel-major) are dealt out in contiguous runs of equal ENTRY counts, one run per workgroup
            const int64_t ntasks = P * C;
            const int G = (int)std::min<int64_t>(ntasks, c->n_cu);
            // (a segment costs a memory round trip however short it is -- with entries alone balanced, workgroups holding
            // 34-entry segments took 2.4x as long as those with 154-entry ones: weigh a row like row_cost entries)
            const int64_t row_cost = c->lp_rowcost;
            std::vector<int64_t> upto((size_t)ntasks + 1, 0);   // weight before task t
            for (int64_t t = 0; t < ntasks; ++t) {   // (panel-major; row-chunk-major was tried: 0.906 -> 0.971 ms, x reloaded per task)
                const int64_t p = t / C, cc = t % C;
                const int64_t lo = cc * chunk, hi = std::min<int64_t>(rows, lo + chunk);
                int64_t w = 0;
                for (int64_t i = lo; i < hi; ++i) w += pp[(size_t)((p + 1) * rows + i)] - pp[(size_t)(p * rows + i)];
                upto[(size_t)t + 1] = upto[(size_t)t] + w + row_cost * (hi - lo);
            }
            std::vector<int32_t> tb((size_t)G + 1, 0);
            for (int g = 1; g < G; ++g) {
                const int64_t target = upto[(size_t)ntasks] / G * g;
                tb[(size_t)g] = (int32_t)(std::lower_bound(upto.begin(), upto.end(), target) - upto.begin());
                tb[(size_t)g] = std::max(tb[(size_t)g], tb[(size_t)g - 1]);
            }
            tb[(size_t)G] = (int32_t)ntasks;
            if (dbg_lower) {
                fprintf(stderr, "[sla] lpanel: P=%lld C=%lld chunk=%lld G=%d total=%lld tb:", (long long)P, (long long)C, (long long)chunk, G, (long long)upto[(size_t)ntasks]);
                for (int g = 0; g <= G; g += std::max(1, G / 16)) fprintf(stderr, " %d", tb[(size_t)g]);
                fprintf(stderr, "\n");
            }
            A->lp_G = G;
            upload((void **)&A->d_lpt, tb.data(), sizeof(int32_t) * tb.size());
            A->use_lpanel = err == hipSuccess;
        }
    }
}

static int csr_upload(sla_ctx *c, int64_t m, int64_t n, int64_t row_begin, int64_t rows, const int64_t *rowptr,
                      const int64_t *col, const double *val, sla_csr **out, bool panel_view) {
    const int64_t nnz = rowptr[rows];
    if (n > (int64_t)std::numeric_limits<int32_t>::max() || rows >= (int64_t)std::numeric_limits<int32_t>::max())
        return fail(SLA_ERR_INVALID, "matrix dimension exceeds the 32-bit device index width");
    static const bool dbg_lower = getenv("SLA_DEBUG_LOWER") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg_lower || panel_view) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[sla] lowering: %-28s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
        t_last = t;
    };
    sla_csr *A = new sla_csr();
    A->ctx = c;
    A->m = m;
    A->n = n;
    A->row_begin = row_begin;
    A->rows = rows;
    A->nnz = nnz;
    // (SLA_FORCE_RP64=1: test hook -- run the 64-bit row-pointer instantiations of the kernels on small matrices)
    // (SLA_FORCE_RP64=2: the parent only -- its column-panel views keep their natural 32-bit width, the mixed case of a > 2^31-entry matrix)
    A->rp64 = nnz > (int64_t)std::numeric_limits<int32_t>::max() || c->force_rp64 == 1 || (c->force_rp64 == 2 && !panel_view);
    Low L{c, A, m, n, row_begin, rows, nnz, rowptr, col, val, panel_view, dbg_lower};
    hipError_t &err = L.err;
    build_row_blocks(rows, rowptr, L.rb, A->max_row_nnz, c->row_align, c->rb_nnz);
    A->nrb = (int32_t)L.rb.size() - 1;
    int diag_not = host_is_diagonal(rows, row_begin, rowptr, col) ? 0 : 1;
    if (m != n && rows > 0) { /* isDiagonalSM only counts (i,i) entries; nothing extra to do */ }
    low_csr_arrays(L);
    lap("row blocks + CSR upload");
    low_xwin_statistics(L);
    lap("x-window statistics");
    low_diagonal_dictionary(L);
    low_value_indexed(L);
    lap("pair dictionary + wave slices");
    low_wave_sliced_variable(L);
    lap("variable-coefficient slices");
    low_lds_panels(L);
    lap("LDS panel table");
    if (panel_view) {
        if (err != hipSuccess) {
            sla_csr_destroy(A);
            return fail(SLA_ERR_ALLOC, std::string("CSR upload: ") + hipGetErrorString(err));
        }
        *out = A;
        return SLA_OK;
    }
    // Every rank enters the agreement collective, failed or not (a rank returning early would leave its peers blocked in
    // it): the all-reduced maximum carries isDiagonalSM's verdict in bit 0 and "some rank failed" as a value >= 2.
    int agree = err != hipSuccess ? 2 : diag_not;
    int rc = dist_allreduce_max_i32(c, &agree);
    if (err != hipSuccess) {
        sla_csr_destroy(A);
        return fail(SLA_ERR_ALLOC, std::string("CSR upload: ") + hipGetErrorString(err));
    }
    if (rc == SLA_OK && agree >= 2) rc = fail(SLA_ERR_INVALID, "matrix creation failed on another rank of the row-sharded job");
    if (rc != SLA_OK) {
        sla_csr_destroy(A);
        return rc;
    }
    A->is_diagonal = agree == 0;
    rc = build_xplan(A, rows, rowptr, col);
    if (rc == SLA_OK) rc = build_overlap_lists(A, m, n, row_begin, rows, rowptr, col);
    if (rc == SLA_OK) rc = build_tiles(A, n, rows, rowptr, col, val);
    lap("tile form");
    if (rc == SLA_OK && !(A->use_lpanel && c->lpanel) && !A->use_tiles) rc = build_panels(A, m, n, row_begin, rows, rowptr, col, val);
    if (rc != SLA_OK) {
        sla_csr_destroy(A);
        return rc;
    }
    *out = A;
    return SLA_OK;
}

// lazily built transpose (transposeSM, SpMatrix.hs:717).  Row-sharded: the transpose of this rank's row block
// only -- all n rows of A^T, but just the columns this rank owns; its SpMV yields a full-length PARTIAL result
// that spmv_transposed reduce-scatters over the ranks.
int csr_transposed(sla_csr *A, sla_csr **out) {
    if (A->transposed) {
        *out = A->transposed;
        return SLA_OK;
    }
    HostCsr h, t;
    h.m = A->rows;
    h.n = A->n;
    h.rowptr.resize((size_t)A->rows + 1);
    h.col.resize((size_t)A->nnz);
    h.val.resize((size_t)A->nnz);
    SLA_TRY(sla_csr_export(A, h.rowptr.data(), h.col.data(), h.val.data()));
    transpose_csr(h, t);  // t: n rows, columns = LOCAL row ids 0..rows-1
    sla_csr *T = nullptr;
    if (!A->ctx->collectives) {
        SLA_TRY(csr_upload(A->ctx, t.m, t.n, 0, t.m, t.rowptr.data(), t.col.data(), t.val.data(), &T));
    } else {
        // a private view (no exchange plan, no collectives at creation): columns stay local row ids, the
        // gather base is this rank's shard
        SLA_TRY(csr_upload(A->ctx, t.m, A->m, 0, t.m, t.rowptr.data(), t.col.data(), t.val.data(), &T, true));
    }
    A->transposed = T;
    *out = T;
    return SLA_OK;
}

// y = transpose A #> x (vecMatSD, Common.hs:253-256) on local shards; row-sharded: partial + reduce-scatter
int spmv_transposed(sla_csr *A, const double *x_local, double *y_local, int64_t y_shard) {
    sla_ctx *c = A->ctx;
    sla_csr *T = nullptr;
    SLA_TRY(csr_transposed(A, &T));
    SpmvLaunch l;
    l.x = x_local;
    if (!c->collectives) {
        l.y = y_local;
        return launch_spmv(T, l);
    }
    const int64_t full = y_shard * c->nranks;
    if (c->tfull_cap < full) {
        if (c->d_tfull) (void)hipFree(c->d_tfull);
        c->d_tfull = nullptr;
        c->tfull_cap = 0;
        SLA_HIP_TRY(dev_malloc(c, (void **)&c->d_tfull, sizeof(double) * (size_t)std::max<int64_t>(full, 1)));
        SLA_HIP_TRY(hipMemsetAsync(c->d_tfull, 0, sizeof(double) * (size_t)std::max<int64_t>(full, 1), stream_of(c)));
        c->tfull_cap = full;
    }
    // the padding rows [n, shard * nranks) must read as zero in the reduce-scatter (an earlier, larger matrix may have left
    // its partials there: the buffer is per context, not per matrix)
    if (full > T->m) SLA_HIP_TRY(hipMemsetAsync(c->d_tfull + T->m, 0, sizeof(double) * (size_t)(full - T->m), stream_of(c)));
    l.y = c->d_tfull;
    SLA_TRY(launch_spmv(T, l));
    return dist_reduce_scatter_f64(c, c->d_tfull, y_local, y_shard);
}

}  // namespace sla

using namespace sla;

extern "C" {

const char *sla_last_error(void) { return g_last_error.c_str(); }
long sla_debug_binding_violations(void) { return binding_violations(); }
const char *sla_version(void) { return "sla_hip 0.1 (gfx950)"; }

// The A/B and test knobs ("Knobs"): every one defaults to the measured-best setting.  ONE table serves both
// ways of setting them: sla_ctx_set_option(ctx, "wdia", "0") -- the typed, per-context entry of the ABI -- and the environment
// (SLA_WDIA=0, read once when a context is created).  A knob that steers the lowering takes effect for matrices created afterwards.
namespace {
struct IntKnob { const char *name; int sla_ctx::*field; int lo, hi; };
const IntKnob kIntKnobs[] = {
    {"step_graph", &sla_ctx::step_graph, -1, 1},
    {"xcd_remap", &sla_ctx::xcd_remap, 0, 1},
    {"dual_spmv", &sla_ctx::dual_spmv, 0, 1},
    {"xwin", &sla_ctx::xwin, 0, 1},
    {"stream_pipe", &sla_ctx::stream_pipe, 0, 1},
    {"stream_wide", &sla_ctx::stream_wide, 0, 1},
    {"diag", &sla_ctx::diag, 0, 1},
    {"vdict", &sla_ctx::vdict, 0, 1},
    {"wdia", &sla_ctx::wdia, 0, 1},
    {"lpanel", &sla_ctx::lpanel, 0, 1},
    {"lp_tasks", &sla_ctx::lp_tasks, 1, 1 << 20},
    {"lp_copy", &sla_ctx::lp_copy, 0, 1},
    {"lp_cfg", &sla_ctx::lp_cfg, -1, 3},
    {"lp_minseg", &sla_ctx::lp_min_seg, 1, 1 << 20},
    {"lp_rowcost", &sla_ctx::lp_rowcost, 0, 1 << 20},
    {"force_rp64", &sla_ctx::force_rp64, 0, 2},
    {"bicg_ghost", &sla_ctx::bicg_ghost, 0, 1},
    {"wd_tile", &sla_ctx::wd_tile, -1, 1 << 20},
    {"bicg_fuse45", &sla_ctx::bicg_fuse45, 0, 1},
    {"wd_lds", &sla_ctx::wd_lds, 0, 2},
    {"wd_lds_occ", &sla_ctx::wd_lds_occ, 0, 4},
    {"wd_march", &sla_ctx::wd_march, 0, 2},
    {"wd_march_occ", &sla_ctx::wd_march_occ, 1, 4},
    {"wd_nt_store", &sla_ctx::wd_nt_store, 0, 1},
    {"wdia_vv", &sla_ctx::wdia_vv, 0, 1},
    {"vec_nt", &sla_ctx::vec_nt, -1, 1},
    {"vec_policy", &sla_ctx::vec_policy, 0, 0x7fff},
    {"halo_inplace", &sla_ctx::halo_inplace, 0, 1},
    {"panels", &sla_ctx::panels, 0, 1},
    {"overlap", &sla_ctx::overlap, -1, 1},
    {"tiles", &sla_ctx::tiles, 0, 1},
    {"tile_shift", &sla_ctx::tile_shift, 0, 20},
    {"tile_slack", &sla_ctx::tile_slack, 0, 64},
    {"row_align", &sla_ctx::row_align, 0, 256},
    {"rb_nnz", &sla_ctx::rb_nnz, 0, 1024},
    {"spmv_grid", &sla_ctx::spmv_grid_max, 1, kMaxParts},
};
const char *const kOtherKnobs[] = {"wd_grid", "spmv_algo", "panel_cols", "device_coo_min", "x_exchange"};

// one option, by its lower-case name; false: unknown name or value out of range
bool ctx_apply_option(sla_ctx *c, const std::string &name, const char *value) {
    for (const IntKnob &k : kIntKnobs)
        if (name == k.name) {
            char *end = nullptr;
            const long v = strtol(value, &end, 10);
            if (end == value || v < k.lo || v > k.hi) return false;
            c->*(k.field) = (int)v;
            return true;
        }
    if (name == "wd_grid") {        // persistent grid of the wave-sliced kernels (a multiple of 8: one share per XCD)
        const int g = atoi(value);
        if (g < 8 || g > kMaxParts) return false;
        c->wd_grid_max = g & ~7;
        return true;
    }
    if (name == "spmv_algo") {
        if (strcmp(value, "scalar") != 0 && strcmp(value, "stream") != 0) return false;
        c->spmv_algo = strcmp(value, "scalar") == 0 ? 1 : 0;
        return true;
    }
    if (name == "panel_cols") { c->panel_cols = atoll(value); return c->panel_cols > 0; }
    if (name == "device_coo_min") { c->device_coo_min = atoll(value); return true; }
    if (name == "x_exchange") {
        if (strcmp(value, "allgather") == 0) c->x_exchange = 1;
        else if (strcmp(value, "window") == 0) c->x_exchange = 2;
        else if (strcmp(value, "auto") == 0) c->x_exchange = 0;
        else return false;
        return true;
    }
    return false;
}
std::string ctx_option_value(const sla_ctx *c, const std::string &name, bool *known) {
    *known = true;
    for (const IntKnob &k : kIntKnobs)
        if (name == k.name) return std::to_string(c->*(k.field));
    if (name == "wd_grid") return std::to_string(c->wd_grid_max);
    if (name == "spmv_algo") return c->spmv_algo == 1 ? "scalar" : "stream";
    if (name == "panel_cols") return std::to_string(c->panel_cols);
    if (name == "device_coo_min") return std::to_string(c->device_coo_min);
    if (name == "x_exchange") return c->x_exchange == 1 ? "allgather" : c->x_exchange == 2 ? "window" : "auto";
    *known = false;
    return "";
}
std::string env_name(const char *knob) {
    std::string e = "SLA_";
    for (const char *p = knob; *p; ++p) e += (char)toupper((unsigned char)*p);
    return e;
}
}  // namespace

static void ctx_read_knobs(sla_ctx *c) {
    auto from_env = [&](const char *knob) {
        if (const char *s = getenv(env_name(knob).c_str()))
            if (!ctx_apply_option(c, knob, s)) fprintf(stderr, "[sla] ignoring %s=%s (unknown value)\n", env_name(knob).c_str(), s);
    };
    for (const IntKnob &k : kIntKnobs) from_env(k.name);
    for (const char *k : kOtherKnobs) from_env(k);
}

static int ctx_create_common(int device_id, int rank, int nranks, const void *uid, sla_ctx_t *out) {
    if (!out || nranks < 1 || rank < 0 || rank >= nranks) return fail(SLA_ERR_INVALID, "sla_ctx_create: bad arguments");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(SLA_ERR_NO_DEVICE, "no HIP device visible: libsla_hip has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail(SLA_ERR_INVALID, "sla_ctx_create: device id out of range");
    SLA_HIP_TRY(hipSetDevice(device_id));
    sla_ctx *c = new sla_ctx();
    c->device = device_id;
    c->rank = rank;
    c->nranks = nranks;
    Bind bind(c);
    {   // (a failing allocation must not leak the half-built context)
        hipError_t he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (he == hipSuccess) he = dev_malloc(c, (void **)&c->d_parts, sizeof(double) * 4 * kMaxParts);
        if (he == hipSuccess) he = dev_malloc(c, (void **)&c->d_result, sizeof(double) * 4096);
        if (he == hipSuccess) he = hipHostMalloc((void **)&c->h_result, sizeof(double) * 64, hipHostMallocDefault);
        if (he != hipSuccess) {
            sla_ctx_destroy(c);
            return fail(SLA_ERR_HIP, std::string("sla_ctx_create: ") + hipGetErrorString(he));
        }
    }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && cus > 0) {
            c->n_cu = cus;
            c->wd_grid_max = std::min<int>(kMaxParts, std::max(8, (kWdBlocksPerCu * cus) & ~7));
            c->wd_grid_max_vv = std::min<int>(kMaxParts, std::max(8, (kWdBlocksPerCuVV * cus) & ~7));
        }
    }
    ctx_read_knobs(c);   // (after the device-derived defaults: SLA_WD_GRID overrides the per-CU grid)
    if (uid) {
        int rc = dist_comm_init(c, uid);
        if (rc != SLA_OK) {
            sla_ctx_destroy(c);
            return rc;
        }
        const char *f = getenv("SLA_FORCE_COLLECTIVES");
        c->collectives = nranks > 1 || (f && atoi(f) != 0);
        if (c->collectives) c->vec_guard = kHaloBytes;
    }
    *out = c;
    return SLA_OK;
}

int sla_ctx_create(int device_id, sla_ctx_t *out) { return ctx_create_common(device_id, 0, 1, nullptr, out); }

int sla_dist_unique_id(void *unique_id_128) {
    if (!unique_id_128) return fail(SLA_ERR_INVALID, "null unique id buffer");
    return dist_unique_id(unique_id_128);
}

int sla_ctx_create_dist(int device_id, int rank, int nranks, const void *unique_id_128, sla_ctx_t *out) {
    if (!unique_id_128) return fail(SLA_ERR_INVALID, "null unique id");
    return ctx_create_common(device_id, rank, nranks, unique_id_128, out);
}

int sla_ctx_create_loopback(int device_id, int rank, int nranks, int group_key, sla_ctx_t *out) {
    SLA_TRY(ctx_create_common(device_id, rank, nranks, nullptr, out));
    Bind bind(*out);
    int rc = dist_loopback_join(*out, group_key);
    if (rc != SLA_OK) {
        sla_ctx_destroy(*out);
        *out = nullptr;
        return rc;
    }
    (*out)->collectives = true;
    (*out)->vec_guard = kHaloBytes;
    return SLA_OK;
}

int sla_ctx_destroy(sla_ctx_t c) {
    if (c && !c->kids.empty()) return m_ctx_destroy(c);
    if (!c) return SLA_OK;
    Bind bind(c);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    dist_comm_destroy(c);
    for (auto &kv : c->vec_pool) (void)guard_free(kv.second, c->vec_guard);
    c->vec_pool.clear();
    for (hipEvent_t ev : c->prof_ev) (void)hipEventDestroy(ev);
    if (c->d_parts) (void)hipFree(c->d_parts);
    if (c->d_result) (void)hipFree(c->d_result);
    if (c->h_result) (void)hipHostFree(c->h_result);
    if (c->d_xfull) (void)guard_free(c->d_xfull);
    if (c->d_tfull) (void)hipFree(c->d_tfull);
    if (c->ev_x_ready) (void)hipEventDestroy(c->ev_x_ready);
    if (c->ev_x_done) (void)hipEventDestroy(c->ev_x_done);
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SLA_OK;
}

int sla_ctx_sync(sla_ctx_t c) {
    if (c && !c->kids.empty()) return m_ctx_sync(c);
    if (!c) return fail(SLA_ERR_INVALID, "null context");
    Bind bind(c);
    SLA_HIP_TRY(hipStreamSynchronize(stream_of(c)));
    return SLA_OK;
}

int sla_ctx_set_option(sla_ctx_t c, const char *name, const char *value) {
    if (!c || !name || !value) return fail(SLA_ERR_INVALID, "sla_ctx_set_option: null argument");
    if (!c->kids.empty()) {
        for (sla_ctx *k : c->kids) SLA_TRY(sla_ctx_set_option(k, name, value));
        return SLA_OK;
    }
    if (!ctx_apply_option(c, name, value))
        return fail(SLA_ERR_INVALID, std::string("sla_ctx_set_option: unknown option or value out of range: ") + name + "=" + value);
    return SLA_OK;
}

int sla_ctx_get_option(sla_ctx_t c, const char *name, char *buf, int buflen) {
    if (c && !c->kids.empty()) return sla_ctx_get_option(c->kids[0], name, buf, buflen);
    if (!c || !name || !buf || buflen <= 0) return fail(SLA_ERR_INVALID, "sla_ctx_get_option: null argument");
    bool known = false;
    const std::string v = ctx_option_value(c, name, &known);
    if (!known) return fail(SLA_ERR_INVALID, std::string("sla_ctx_get_option: unknown option ") + name);
    snprintf(buf, (size_t)buflen, "%s", v.c_str());
    return SLA_OK;
}

int sla_ctx_rank(sla_ctx_t c, int *rank, int *nranks) {
    if (!c) return fail(SLA_ERR_INVALID, "null context");
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    return SLA_OK;
}

int sla_ctx_row_range(sla_ctx_t c, int64_t m, int64_t *begin, int64_t *end) {
    if (!c || m < 0) return fail(SLA_ERR_INVALID, "sla_ctx_row_range: bad arguments");
    int64_t b, e;
    row_range(c, m, &b, &e);
    if (begin) *begin = b;
    if (end) *end = e;
    return SLA_OK;
}

// ---- CSR ------------------------------------------------------------------------------------------

int sla_csr_from_coo(sla_ctx_t c, int64_t m, int64_t n, int64_t nnz, const int64_t *row, const int64_t *col,
                     const double *val, int dup_policy, sla_csr_t *out) {
    if (c && !c->kids.empty()) return m_csr_from_coo(c, m, n, nnz, row, col, val, dup_policy, out);
    return no_throw("sla_csr_from_coo", [&]() -> int {
        if (!c || !out || (nnz > 0 && (!row || !col || !val))) return fail(SLA_ERR_INVALID, "sla_csr_from_coo: null argument");
        Bind bind(c);
        HostCsr h;
        if (!c->collectives && nnz >= c->device_coo_min && device_coo_supported(m, n, nnz)) {
            if (m < 0 || n < 0) return fail(SLA_ERR_INVALID, "negative dimension");
            std::vector<char> oob((size_t)host_threads(), 0);
            par_rows(nnz, 1, [&](int t, int64_t lo, int64_t hi) {
                char bad = 0;
                for (int64_t k = lo; k < hi; ++k) bad |= (row[k] < 0) | (row[k] >= m) | (col[k] < 0) | (col[k] >= n);
                oob[(size_t)t] = bad;
            });
            if (std::find(oob.begin(), oob.end(), (char)1) != oob.end())
                return fail(SLA_ERR_OOB, "insertSpMatrix : index out of bounds");
            SLA_TRY(device_coo_to_csr(c, m, n, nnz, row, col, val, dup_policy, h));
            // already canonical by construction: skip the validation pass of sla_csr_from_csr
            return csr_upload(c, m, n, 0, m, h.rowptr.data(), h.col.data(), h.val.data(), out);
        }
        SLA_TRY(build_csr_from_coo(m, n, nnz, row, col, val, dup_policy, h));
        return sla_csr_from_csr(c, m, n, h.rowptr.data(), h.col.data(), h.val.data(), out);
    });
}

int sla_csr_from_csr(sla_ctx_t c, int64_t m, int64_t n, const int64_t *rowptr, const int64_t *colidx,
                     const double *val, sla_csr_t *out) {
    if (c && !c->kids.empty()) return m_csr_from_csr(c, m, n, rowptr, colidx, val, out);
    return no_throw("sla_csr_from_csr", [&]() -> int {
        if (!c || !out || !rowptr || m < 0 || n < 0) return fail(SLA_ERR_INVALID, "sla_csr_from_csr: bad argument");
        Bind bind(c);
        int64_t b, e;
        row_range(c, m, &b, &e);
        if (c->nranks == 1) return sla_csr_from_csr_rows(c, m, n, 0, m, rowptr, colidx, val, out);
        std::vector<int64_t> rp((size_t)(e - b) + 1);
        for (int64_t i = b; i <= e; ++i) rp[(size_t)(i - b)] = rowptr[i] - rowptr[b];
        return sla_csr_from_csr_rows(c, m, n, b, e - b, rp.data(), colidx + rowptr[b], val + rowptr[b], out);
    });
}

int sla_csr_from_csr_rows(sla_ctx_t c, int64_t m, int64_t n, int64_t row_begin, int64_t row_count,
                          const int64_t *rowptr_local, const int64_t *colidx, const double *val, sla_csr_t *out) {
    if (c && !c->kids.empty()) return multi_unsupported("sla_csr_from_csr_rows (pre-sharded input)");
    return no_throw("sla_csr_from_csr_rows", [&]() -> int {
        if (!c || !out || !rowptr_local || m < 0 || n < 0 || row_count < 0)
            return fail(SLA_ERR_INVALID, "sla_csr_from_csr_rows: bad argument");
        Bind bind(c);
        int64_t b, e;
        row_range(c, m, &b, &e);
        if (row_begin != b || row_count != e - b)
            return csr_reject(c, fail(SLA_ERR_INVALID, "sla_csr_from_csr_rows: rows do not match sla_ctx_row_range"));
        if (rowptr_local[0] != 0) return csr_reject(c, fail(SLA_ERR_INVALID, "rowptr_local[0] must be 0"));
        const int64_t nnz = rowptr_local[row_count];
        if (nnz > 0 && (!colidx || !val)) return csr_reject(c, fail(SLA_ERR_INVALID, "null colidx/val"));
        // monotone row pointers first (the column checks below index through them), then rows in parallel; the
        // lowest-numbered kind of violation wins so that the result does not depend on the thread count
        for (int64_t i = 0; i < row_count; ++i)
            if (rowptr_local[i + 1] < rowptr_local[i]) return csr_reject(c, fail(SLA_ERR_INVALID, "rowptr not monotone"));
        std::vector<int> bad((size_t)host_threads(), 0);   // 1: out of bounds, 2: not strictly ascending
        par_rows(row_count, 1, [&](int t, int64_t lo, int64_t hi) {
            int b_ = 0;
            for (int64_t i = lo; i < hi && b_ != 1; ++i)
                for (int64_t k = rowptr_local[i]; k < rowptr_local[i + 1]; ++k) {
                    if (colidx[k] < 0 || colidx[k] >= n) { b_ = 1; break; }
                    if (k > rowptr_local[i] && colidx[k] <= colidx[k - 1]) b_ = 2;
                }
            bad[(size_t)t] = b_;
        });
        if (std::find(bad.begin(), bad.end(), 1) != bad.end()) return csr_reject(c, fail(SLA_ERR_OOB, "insertSpMatrix : index out of bounds"));
        if (std::find(bad.begin(), bad.end(), 2) != bad.end())
            return csr_reject(c, fail(SLA_ERR_INVALID, "columns must be strictly ascending inside a row (canonical CSR)"));
        return csr_upload(c, m, n, row_begin, row_count, rowptr_local, colidx, val, out);
    });
}

static int export_host(sla_csr_t A, HostCsr &h) {
    h.m = A->m;
    h.n = A->n;
    h.rowptr.resize((size_t)A->rows + 1);
    h.col.resize((size_t)A->nnz);
    h.val.resize((size_t)A->nnz);
    return sla_csr_export(A, h.rowptr.data(), h.col.data(), h.val.data());
}

int sla_jacobi_pre(sla_csr_t A, sla_csr_t *out) {
    if (A && !A->kids.empty()) return multi_unsupported("sla_jacobi_pre");
    return no_throw("sla_jacobi_pre", [&]() -> int {
        if (!A || !out) return fail(SLA_ERR_INVALID, "null argument");
        if (A->ctx->collectives) return fail(SLA_ERR_INVALID, "sla_jacobi_pre: single-rank contexts only");
        Bind bind(A->ctx);
        HostCsr h, d;
        SLA_TRY(export_host(A, h));
        d.m = A->m;
        d.n = A->n;
        d.rowptr.assign((size_t)A->m + 1, 0);
        for (int64_t i = 0; i < A->m; ++i) {  // extractDiag keeps (i,i) where stored; fmap recip
            for (int64_t k = h.rowptr[(size_t)i]; k < h.rowptr[(size_t)i + 1]; ++k)
                if (h.col[(size_t)k] == i) {
                    d.col.push_back(i);
                    d.val.push_back(1.0 / h.val[(size_t)k]);
                }
            d.rowptr[(size_t)i + 1] = (int64_t)d.col.size();
        }
        return csr_upload(A->ctx, d.m, d.n, 0, d.m, d.rowptr.data(), d.col.data(), d.val.data(), out);
    });
}

// ---- triangular solves, SSOR factors -------------------------------------------------------
static void tri_plan_free(sla_tri_plan *p) {
    if (!p) return;
    if (p->graph) (void)hipGraphExecDestroy(p->graph);
    if (p->d_order) (void)hipFree(p->d_order);
    if (p->d_tptr) (void)hipFree(p->d_tptr);
    if (p->d_tcol) (void)hipFree(p->d_tcol);
    if (p->d_tval) (void)hipFree(p->d_tval);
    if (p->d_tdiag) (void)hipFree(p->d_tdiag);
    delete p;
}

// level(i) = 1 + max level of the rows i depends on (the triangle's side of row i); rows of a level are independent
static int tri_plan_build(sla_csr *T, int upper, int64_t *bad_row) {
    if (T->tri[upper]) return SLA_OK;
    HostCsr h;
    SLA_TRY(export_host(T, h));
    const int64_t n = T->m;
    std::vector<int32_t> level((size_t)n, 0);
    int64_t nlev = 0;
    auto row_level = [&](int64_t i) -> int {   // returns 0 when the diagonal entry is unusable
        int lv = 0;
        bool diag_ok = false;
        for (int64_t k = h.rowptr[(size_t)i]; k < h.rowptr[(size_t)i + 1]; ++k) {
            const int64_t j = h.col[(size_t)k];
            if (j == i) diag_ok = !(fabs(h.val[(size_t)k]) <= 1e-12);          // isNz (Eps.hs:41-42)
            else if (upper ? j > i : j < i) lv = std::max(lv, (int)level[(size_t)j]);
        }
        return diag_ok ? lv + 1 : 0;
    };
    for (int64_t t = 0; t < n; ++t) {
        const int64_t i = upper ? n - 1 - t : t;                                  // the reference's sweep order
        const int lv = row_level(i);
        if (lv == 0) {
            if (bad_row) *bad_row = i;
            return fail(SLA_ERR_NEEDS_PIVOTING, std::string(upper ? "triUpperSolve : U (" : "triLowerSolve : L (") + std::to_string(i) + "," +
                                                    std::to_string(i) + ") is close to 0. Permute the rows to obtain a nonzero diagonal");
        }
        level[(size_t)i] = lv;
        nlev = std::max<int64_t>(nlev, lv);
    }
    sla_tri_plan *p = new sla_tri_plan();
    p->nlevels = nlev;
    p->level_ptr.assign((size_t)nlev + 1, 0);
    for (int64_t i = 0; i < n; ++i) p->level_ptr[(size_t)level[(size_t)i]]++;     // counts at [lv], lv >= 1
    for (int64_t l = 1; l <= nlev; ++l) p->widest = std::max(p->widest, p->level_ptr[(size_t)l]);
    {   // exclusive prefix: rows of level lv (1-based) occupy [ptr[lv - 1], ptr[lv])
        int64_t run = 0;
        for (int64_t l = 1; l <= nlev; ++l) { const int64_t cnt = p->level_ptr[(size_t)l]; p->level_ptr[(size_t)l - 1] = run; run += cnt; }
        p->level_ptr[(size_t)nlev] = run;
    }
    std::vector<int32_t> order((size_t)std::max<int64_t>(n, 1));
    std::vector<int64_t> fill(p->level_ptr.begin(), p->level_ptr.end());
    for (int64_t i = 0; i < n; ++i) order[(size_t)fill[(size_t)level[(size_t)i] - 1]++] = (int32_t)i;   // ascending rows inside a level
    // the triangle in schedule order
    std::vector<int64_t> tptr((size_t)n + 1, 0);
    std::vector<int32_t> tcol;
    std::vector<double> tval, tdiag((size_t)std::max<int64_t>(n, 1), 1.0);
    tcol.reserve((size_t)h.col.size() / 2 + 16);
    tval.reserve((size_t)h.col.size() / 2 + 16);
    for (int64_t t = 0; t < n; ++t) {
        const int64_t i = order[(size_t)t];
        for (int64_t k = h.rowptr[(size_t)i]; k < h.rowptr[(size_t)i + 1]; ++k) {
            const int64_t j = h.col[(size_t)k];
            if (j == i) tdiag[(size_t)t] = h.val[(size_t)k];
            else if (upper ? j > i : j < i) { tcol.push_back((int32_t)j); tval.push_back(h.val[(size_t)k]); }
        }
        tptr[(size_t)t + 1] = (int64_t)tcol.size();
    }
    hipError_t e = hipSuccess;
    auto up = [&](void **dst, const void *src, size_t bytes) {
        if (e != hipSuccess) return;
        e = dev_malloc(T->ctx, dst, std::max<size_t>(bytes, 8));
        if (e == hipSuccess && bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    };
    up((void **)&p->d_order, order.data(), sizeof(int32_t) * order.size());
    up((void **)&p->d_tptr, tptr.data(), sizeof(int64_t) * tptr.size());
    up((void **)&p->d_tcol, tcol.data(), sizeof(int32_t) * tcol.size());
    up((void **)&p->d_tval, tval.data(), sizeof(double) * tval.size());
    up((void **)&p->d_tdiag, tdiag.data(), sizeof(double) * tdiag.size());
    if (e != hipSuccess) {
        tri_plan_free(p);
        return fail(SLA_ERR_ALLOC, std::string("triangular schedule: ") + hipGetErrorString(e));
    }
    T->tri[upper] = p;
    return SLA_OK;
}

int sla_tri_solve_info(sla_csr_t T, int upper, int64_t *levels, int64_t *widest_level) {
    if (T && !T->kids.empty()) return multi_unsupported("sla_tri_solve_info");
    if (!T) return fail(SLA_ERR_INVALID, "null matrix");
    if (T->ctx->collectives || T->m != T->n) return fail(SLA_ERR_INVALID, "sla_tri_solve: square matrices on single-rank contexts only");
    upper = upper ? 1 : 0;
    Bind bind(T->ctx);
    SLA_TRY(tri_plan_build(T, upper, nullptr));
    if (levels) *levels = T->tri[upper]->nlevels;
    if (widest_level) *widest_level = T->tri[upper]->widest;
    return SLA_OK;
}

int sla_tri_solve(sla_csr_t T, int upper, sla_vec_t b, sla_vec_t x, int64_t *bad_row) {
    if (T && !T->kids.empty()) return multi_unsupported("sla_tri_solve");
    return no_throw("sla_tri_solve", [&]() -> int {
        if (!T || !b || !x) return fail(SLA_ERR_INVALID, "null argument");
        sla_ctx *c = T->ctx;
        Bind bind(c);
        if (c->collectives || T->m != T->n) return fail(SLA_ERR_INVALID, "sla_tri_solve: square matrices on single-rank contexts only");
        if (b->ctx != c || x->ctx != c || b->d == x->d) return fail(SLA_ERR_INVALID, "sla_tri_solve: b and x must be distinct vectors of the matrix's context");
        if (b->n != T->m || x->n != T->m) return fail(SLA_ERR_DIM_MISMATCH, "triangular solve : mismatched dimensions");
        upper = upper ? 1 : 0;
        SLA_TRY(tri_plan_build(T, upper, bad_row));
        sla_tri_plan *p = T->tri[upper];
        if (T->m == 0) return SLA_OK;
        if (!p->graph || p->gb != b->d || p->gx != x->d) {
            // capture the level launches once per (b, x) buffer pair; later solves with the same buffers replay the graph
            if (p->graph) { (void)hipGraphExecDestroy(p->graph); p->graph = nullptr; }
            hipGraph_t g = nullptr;
            SLA_HIP_TRY(hipStreamBeginCapture(stream_of(c), hipStreamCaptureModeThreadLocal));
            int rc = SLA_OK;
            for (int64_t l = 0; l < p->nlevels && rc == SLA_OK; ++l)
                rc = launch_tri_level(T, p, p->level_ptr[(size_t)l], p->level_ptr[(size_t)l + 1] - p->level_ptr[(size_t)l], b->d, x->d);
            if (rc == SLA_OK) rc = launch_tri_sparsify(c, T->m, x->d);
            const hipError_t e = hipStreamEndCapture(stream_of(c), &g);
            if (rc != SLA_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
            SLA_HIP_TRY(e);
            const hipError_t ei = hipGraphInstantiate(&p->graph, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            SLA_HIP_TRY(ei);
            p->gb = b->d;
            p->gx = x->d;
        }
        SLA_HIP_TRY(hipGraphLaunch(p->graph, stream_of(c)));
        return SLA_OK;
    });
}

int sla_ssor_pre(sla_csr_t A, double omega, sla_csr_t *l, sla_csr_t *r) {
    if (A && !A->kids.empty()) return multi_unsupported("sla_ssor_pre");
    return no_throw("sla_ssor_pre", [&]() -> int {
        if (!A || !l || !r) return fail(SLA_ERR_INVALID, "null argument");
        if (A->ctx->collectives) return fail(SLA_ERR_INVALID, "sla_ssor_pre: single-rank contexts only");
        if (A->m != A->n) return fail(SLA_ERR_DIM_MISMATCH, "mSsorPre : square matrices only");
        Bind bind(A->ctx);
        HostCsr h, L, R;
        SLA_TRY(export_host(A, h));
        const int64_t n = A->m;
        std::vector<double> rd((size_t)n, 0.0);      // reciprocal d: recip of the stored diagonal entries
        std::vector<char> has((size_t)n, 0);
        for (int64_t i = 0; i < n; ++i)
            for (int64_t k = h.rowptr[(size_t)i]; k < h.rowptr[(size_t)i + 1]; ++k)
                if (h.col[(size_t)k] == i) { rd[(size_t)i] = 1.0 / h.val[(size_t)k]; has[(size_t)i] = 1; }
        L.m = R.m = L.n = R.n = n;
        L.rowptr.assign((size_t)n + 1, 0);
        R.rowptr.assign((size_t)n + 1, 0);
        for (int64_t i = 0; i < n; ++i) {
            for (int64_t k = h.rowptr[(size_t)i]; k < h.rowptr[(size_t)i + 1]; ++k) {
                const int64_t j = h.col[(size_t)k];
                if (j < i && has[(size_t)j]) {                       // (eye ^-^ omega e)_ij = -(omega e_ij); times (1 / d_jj): one-term sum
                    const double m = -(omega * h.val[(size_t)k]);
                    L.col.push_back(j);
                    L.val.push_back(0.0 + rd[(size_t)j] * m);
                }
            }
            if (has[(size_t)i]) { L.col.push_back(i); L.val.push_back(0.0 + rd[(size_t)i] * 1.0); }
            L.rowptr[(size_t)i + 1] = (int64_t)L.col.size();
            for (int64_t k = h.rowptr[(size_t)i]; k < h.rowptr[(size_t)i + 1]; ++k) {
                const int64_t j = h.col[(size_t)k];
                if (j == i) { R.col.push_back(j); R.val.push_back(h.val[(size_t)k]); }
                else if (j > i) { R.col.push_back(j); R.val.push_back(-(omega * h.val[(size_t)k])); }
            }
            R.rowptr[(size_t)i + 1] = (int64_t)R.col.size();
        }
        sla_csr *lo = nullptr, *ro = nullptr;
        SLA_TRY(csr_upload(A->ctx, n, n, 0, n, L.rowptr.data(), L.col.data(), L.val.data(), &lo));
        const int rc = csr_upload(A->ctx, n, n, 0, n, R.rowptr.data(), R.col.data(), R.val.data(), &ro);
        if (rc != SLA_OK) { sla_csr_destroy(lo); return rc; }
        *l = lo;
        *r = ro;
        return SLA_OK;
    });
}

int sla_csr_diag_mul(sla_csr_t D, sla_csr_t A, sla_csr_t *out) {
    if ((D && !D->kids.empty()) || (A && !A->kids.empty())) return multi_unsupported("sla_csr_diag_mul");
    return no_throw("sla_csr_diag_mul", [&]() -> int {
        if (!D || !A || !out) return fail(SLA_ERR_INVALID, "null argument");
        if (D->ctx != A->ctx || A->ctx->collectives) return fail(SLA_ERR_INVALID, "sla_csr_diag_mul: one single-rank context");
        if (D->n != A->m) return fail(SLA_ERR_DIM_MISMATCH, "matMat : incompatible matrix sizes");  // SpMatrix.hs:795
        Bind bind(A->ctx);
        HostCsr hd, ha, r;
        SLA_TRY(export_host(D, hd));
        SLA_TRY(export_host(A, ha));
        for (int64_t i = 0; i < D->m; ++i) {
            const int64_t len = hd.rowptr[(size_t)i + 1] - hd.rowptr[(size_t)i];
            if (len > 1 || (len == 1 && hd.col[(size_t)hd.rowptr[(size_t)i]] != i))
                return fail(SLA_ERR_INVALID, "sla_csr_diag_mul: the left factor must be diagonal");
        }
        r.m = D->m;
        r.n = A->n;
        r.rowptr.assign((size_t)D->m + 1, 0);
        for (int64_t i = 0; i < D->m; ++i) {
            if (hd.rowptr[(size_t)i + 1] > hd.rowptr[(size_t)i]) {
                const double dii = hd.val[(size_t)hd.rowptr[(size_t)i]];
                for (int64_t k = ha.rowptr[(size_t)i]; k < ha.rowptr[(size_t)i + 1]; ++k) {
                    const double x = 0.0 + ha.val[(size_t)k] * dii;  // dott: sum (liftI2 (*) colA rowD), one term
                    if (fabs(x) > 1e-12) {                           // sparsifySM (Eps.hs:41-42)
                        r.col.push_back(ha.col[(size_t)k]);
                        r.val.push_back(x);
                    }
                }
            }
            r.rowptr[(size_t)i + 1] = (int64_t)r.col.size();
        }
        return csr_upload(A->ctx, r.m, r.n, 0, r.m, r.rowptr.data(), r.col.data(), r.val.data(), out);
    });
}

int sla_csr_destroy(sla_csr_t A) {
    if (A && !A->kids.empty()) return m_csr_destroy(A);
    if (!A) return SLA_OK;
    Bind bind(A->ctx);
    if (A->transposed) sla_csr_destroy(A->transposed);
    for (sla_csr *V : A->panels) sla_csr_destroy(V);
    if (A->d_panel_y) (void)hipFree(A->d_panel_y);
    if (A->d_tlrow) (void)hipFree(A->d_tlrow);
    if (A->d_tloff) (void)hipFree(A->d_tloff);
    if (A->d_tlidx) (void)hipFree(A->d_tlidx);
    if (A->d_tlval) (void)hipFree(A->d_tlval);
    if (A->d_tlprog) (void)hipFree(A->d_tlprog);
    if (A->d_ov_int) (void)hipFree(A->d_ov_int);
    if (A->d_ov_bnd) (void)hipFree(A->d_ov_bnd);
    delete A->xplan;
    if (A->d_rowptr) (void)hipFree(A->d_rowptr);
    if (A->d_col) (void)hipFree(A->d_col);
    if (A->d_val) (void)hipFree(A->d_val);
    if (A->d_rb) (void)hipFree(A->d_rb);
    if (A->d_rbk) (void)hipFree(A->d_rbk);
    if (A->d_rbw) (void)hipFree(A->d_rbw);
    if (A->d_code) (void)hipFree(A->d_code);
    if (A->d_dict) (void)hipFree(A->d_dict);
    tri_plan_free(A->tri[0]);
    tri_plan_free(A->tri[1]);
    if (A->d_wptr) (void)hipFree(A->d_wptr);
    if (A->d_wsched) (void)hipFree(A->d_wsched);
    if (A->d_lpp) (void)hipFree(A->d_lpp);
    if (A->d_lpy) (void)hipFree(A->d_lpy);
    if (A->d_lpcol) (void)hipFree(A->d_lpcol);
    if (A->d_lpval) (void)hipFree(A->d_lpval);
    if (A->d_lpt) (void)hipFree(A->d_lpt);
    if (A->d_wvblk) (void)hipFree(A->d_wvblk);
    if (A->d_wme) (void)hipFree(A->d_wme);
    if (A->d_wmo) (void)hipFree(A->d_wmo);
    if (A->d_wval) (void)hipFree(A->d_wval);
    if (A->d_woff) (void)hipFree(A->d_woff);
    if (A->d_wum) (void)hipFree(A->d_wum);
    if (A->d_wum_m) (void)hipFree(A->d_wum_m);
    if (A->d_vcode) (void)hipFree(A->d_vcode);
    if (A->d_vdoff) (void)hipFree(A->d_vdoff);
    if (A->d_vdval) (void)hipFree(A->d_vdval);
    delete A;
    return SLA_OK;
}

int sla_csr_dims(sla_csr_t A, int64_t *m, int64_t *n, int64_t *nnz_local, int64_t *rows_local) {
    if (!A) return fail(SLA_ERR_INVALID, "null matrix");
    if (m) *m = A->m;
    if (n) *n = A->n;
    if (nnz_local) *nnz_local = A->nnz;
    if (rows_local) *rows_local = A->rows;
    return SLA_OK;
}

int sla_csr_export(sla_csr_t A, int64_t *rowptr, int64_t *colidx, double *val) {
    if (A && !A->kids.empty()) return m_csr_export(A, rowptr, colidx, val);
    return no_throw("sla_csr_export", [&]() -> int {
        if (!A) return fail(SLA_ERR_INVALID, "null matrix");
        sla_ctx *c = A->ctx;
        Bind bind(c);
        SLA_HIP_TRY(hipStreamSynchronize(stream_of(c)));
        if (rowptr) {
            if (A->rp64) {
                SLA_HIP_TRY(hipMemcpy(rowptr, A->d_rowptr, sizeof(int64_t) * (size_t)(A->rows + 1), hipMemcpyDeviceToHost));
            } else {
                std::vector<int32_t> t((size_t)A->rows + 1);
                SLA_HIP_TRY(hipMemcpy(t.data(), A->d_rowptr, sizeof(int32_t) * t.size(), hipMemcpyDeviceToHost));
                for (size_t i = 0; i < t.size(); ++i) rowptr[i] = t[i];
            }
        }
        if (colidx && A->nnz) {
            std::vector<int32_t> t((size_t)A->nnz);
            SLA_HIP_TRY(hipMemcpy(t.data(), A->d_col, sizeof(int32_t) * t.size(), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < t.size(); ++i) colidx[i] = t[i];
        }
        if (val && A->nnz) SLA_HIP_TRY(hipMemcpy(val, A->d_val, sizeof(double) * (size_t)A->nnz, hipMemcpyDeviceToHost));
        return SLA_OK;
    });
}

int sla_csr_is_diagonal(sla_csr_t A, int *out) {
    if (!A || !out) return fail(SLA_ERR_INVALID, "null argument");
    *out = A->is_diagonal ? 1 : 0;
    return SLA_OK;
}

int sla_csr_kernel_info(sla_csr_t A, char *buf, int buflen) {
    if (A && !A->kids.empty()) return sla_csr_kernel_info(A->kids[0], buf, buflen);
    if (!A || !buf || buflen <= 0) return fail(SLA_ERR_INVALID, "null argument");
    snprintf(buf, (size_t)buflen, "algo=%s grid=%d block=%d row_blocks=%d nnz_per_row_block=%d max_row_nnz=%lld rowptr=%s xcd_remap=%d",
             A->ctx->spmv_algo == 1 ? "scalar" : (A->use_wdia && wd_on(A)) ? (A->wd_vv ? "wdia-vv" : wd_march_on(A) ? "wdia+march" : wd_lds_on(A) ? "wdia+ldswin" : "wdia") : (A->use_vdict && A->ctx->vdict) ? (A->use_xwin && A->ctx->xwin ? "vdict+xwin" : "vdict") : (A->use_lpanel && A->ctx->lpanel) ? "stream+ldspanels" : tiles_on(A) ? "tiles" : (!A->panels.empty() && A->ctx->panels) ? "stream+colpanels" : (A->use_diag && A->ctx->diag ? (A->use_xwin && A->ctx->xwin ? "stream+diagdict+xwin" : "stream+diagdict") : (pipe_on(A) ? "stream+pipe" : stream_xwin_on(A) ? "stream+xwin" : "stream")), spmv_grid(A), kBlock, A->nrb, kNnzPerRowBlock,
             (long long)A->max_row_nnz, A->rp64 ? "i64" : "i32", A->ctx->xcd_remap);
    if (A->use_lpanel && A->ctx->lpanel && A->ctx->spmv_algo == 0) {   // LDS-panel geometry
        const size_t used = strlen(buf);
        if (used + 1 < (size_t)buflen)
            snprintf(buf + used, (size_t)buflen - used, " lds_panels=%d panel_cols=%d lanes_per_segment=%d tasks=%d entries=%s", A->lp_P, A->lp_W,
                     64 >> A->lp_cfg, A->lp_P * A->lp_C, A->d_lpcol ? "panel-major-copy" : "row-major");
    }
    {   // bytes of matrix data the chosen form streams per (#>) (what K1's HBM roofline is priced against in bench.py)
        const sla_ctx *c = A->ctx;
        const int64_t rps = A->rp64 ? 8 : 4;
        int64_t mb;
        if (c->spmv_algo == 1) mb = 12 * A->nnz + rps * (A->rows + 1);
        else if (A->use_wdia && wd_on(A) && wd_march_on(A)) mb = 128 * 4 * (int64_t)A->wd_mg.T * A->wd_mg.planes;   // 16 lane masks per (tile, plane, wavefront)
        else if (A->use_wdia && wd_on(A) && wd_lds_on(A)) mb = 128 * (int64_t)A->nslices;   // 16 lane masks per slice
        else if (A->use_wdia && wd_on(A)) mb = A->nwent * (A->wd_vv ? 20 + 128 * 8 : 28) + 4 * ((int64_t)A->nslices + 1);
        else if (A->use_vdict && c->vdict) mb = A->nnz + 4 * (A->rows + 1);
        else if (A->use_lpanel && c->lpanel) mb = (A->d_lpcol ? 10 : 12) * A->nnz + (int64_t)(A->lp_P + 1) * A->rows * rps + 16 * (int64_t)A->lp_P * A->rows;
        else if (tiles_on(A)) mb = 12 * A->nnz + 4 * (int64_t)A->tl_S * (A->tl_P + 1) + 4 * ((int64_t)A->tl_S + 1) + rps * A->tl_S;
        else if (!A->panels.empty() && c->panels) mb = 12 * A->nnz + (int64_t)A->panels.size() * (rps * A->rows + 8 * (int64_t)A->nrb) + 16 * ((int64_t)A->panels.size() - 1) * A->rows;
        else mb = (A->use_diag && c->diag ? 9 : 12) * A->nnz + rps * (A->rows + 1) + (4 + rps) * (int64_t)A->nrb;
        const size_t used = strlen(buf);
        if (used + 1 < (size_t)buflen) snprintf(buf + used, (size_t)buflen - used, " matrix_bytes=%lld", (long long)mb);
        if (!A->panels.empty() && c->panels && !tiles_on(A) && c->spmv_algo == 0) {
            const size_t u2 = strlen(buf);
            if (u2 + 1 < (size_t)buflen) snprintf(buf + u2, (size_t)buflen - u2, " col_panels=%d", (int)A->panels.size());
        }
    }
    if (tiles_on(A)) {   // tile geometry; exact_fold: every row is folded entry by entry in ascending column order
        const size_t used = strlen(buf);
        if (used + 1 < (size_t)buflen)
            snprintf(buf + used, (size_t)buflen - used, " slices=%d panels=%d panel_cols=%d max_segment=%lld exact_fold=1 pacing=%s", A->tl_S, A->tl_P,
                     1 << A->tl_shift, (long long)A->tl_maxseg, A->ctx->xcd8 == 1 && A->ctx->tile_slack > 0 ? "on" : "off");
    }
    if (A->use_wdia && wd_on(A) && wd_march_on(A)) {   // plane-march geometry
        const size_t used = strlen(buf);
        if (used + 1 < (size_t)buflen)
            snprintf(buf + used, (size_t)buflen - used, " plane=%d planes=%d tiles=%d runs=%d(x%d planes) win_pairs=%d pairs=%d", A->wd_mg.D, A->wd_mg.planes,
                     A->wd_mg.T, A->wd_mg.S, A->wd_mg.PS, A->wd_mg.pairs, A->wd_muni.n);
    } else if (A->use_wdia && wd_on(A) && wd_lds_on(A)) {   // LDS-window geometry: windows, staged 16-byte pairs per buffer (> 1024: the 6-load instantiation), pairs folded
        const size_t used = strlen(buf);
        if (used + 1 < (size_t)buflen)
            snprintf(buf + used, (size_t)buflen - used, " windows=%d win_pairs=%d pairs=%d", A->wd_win.n, A->wd_win.pairs, A->wd_uni.n);
    }
    if (overlap_split(A)) {
        const size_t used = strlen(buf);
        if (used + 1 < (size_t)buflen)
            snprintf(buf + used, (size_t)buflen - used, " overlap=%s interior_steps=%d boundary_steps=%d", A->ctx->overlap > 0 ? "streams" : "serial", A->ov_nint, A->ov_nbnd);
    }
    if (A->xplan) {   // row-sharded: how the SpMV input is exchanged
        const size_t used = strlen(buf);
        if (used + 1 < (size_t)buflen)
            snprintf(buf + used, (size_t)buflen - used, " x_exchange=%s",
                     A->ctx->x_exchange != 1 && (A->xplan->use_window || A->ctx->x_exchange == 2) ? "window" : "allgather");
    }
    return SLA_OK;
}

// ---- vectors ----------------------------------------------------------------------------------------

int sla_vec_create(sla_ctx_t c, int64_t n, const double *host, sla_vec_t *out) {
    if (c && !c->kids.empty()) return m_vec_create(c, n, host, out);
    if (!c || !out || n < 0) return fail(SLA_ERR_INVALID, "sla_vec_create: bad argument");
    Bind bind(c);
    sla_vec *v = nullptr;
    SLA_TRY(vec_alloc(c, n, &v));
    if (host && v->n_local > 0) {
        hipError_t e = hipMemcpyAsync(v->d, host + v->begin, sizeof(double) * (size_t)v->n_local, hipMemcpyHostToDevice, stream_of(c));
        if (e == hipSuccess) e = hipStreamSynchronize(stream_of(c));
        if (e != hipSuccess) {
            sla_vec_destroy(v);
            return fail(SLA_ERR_HIP, std::string("vector upload: ") + hipGetErrorString(e));
        }
    }
    *out = v;
    return SLA_OK;
}

int sla_vec_create_local(sla_ctx_t c, int64_t n, const double *host_local, sla_vec_t *out) {
    if (c && !c->kids.empty()) return m_vec_create(c, n, host_local, out);   // the caller of a multi-device context owns every row
    if (!c || !out || n < 0) return fail(SLA_ERR_INVALID, "sla_vec_create_local: bad argument");
    Bind bind(c);
    sla_vec *v = nullptr;
    SLA_TRY(vec_alloc(c, n, &v));
    if (host_local && v->n_local > 0) {
        hipError_t e = hipMemcpyAsync(v->d, host_local, sizeof(double) * (size_t)v->n_local, hipMemcpyHostToDevice, stream_of(c));
        if (e == hipSuccess) e = hipStreamSynchronize(stream_of(c));
        if (e != hipSuccess) {
            sla_vec_destroy(v);
            return fail(SLA_ERR_HIP, std::string("vector upload: ") + hipGetErrorString(e));
        }
    }
    *out = v;
    return SLA_OK;
}

int sla_vec_destroy(sla_vec_t v) {
    if (v && !v->kids.empty()) return m_vec_destroy(v);
    if (!v) return SLA_OK;
    Bind bind(v->ctx);
    pool_free(v->ctx, v->d, sizeof(double) * (size_t)std::max<int64_t>(v->shard, 1));
    delete v;
    return SLA_OK;
}

int sla_vec_dim(sla_vec_t v, int64_t *n, int64_t *n_local) {
    if (!v) return fail(SLA_ERR_INVALID, "null vector");
    if (n) *n = v->n;
    if (n_local) *n_local = v->n_local;
    return SLA_OK;
}

int sla_vec_to_host_local(sla_vec_t v, double *host_local) {
    if (v && !v->kids.empty()) return m_vec_to_host(v, host_local);
    if (!v || !host_local) return fail(SLA_ERR_INVALID, "null argument");
    sla_ctx *c = v->ctx;
    Bind bind(c);
    if (v->n_local > 0)
        SLA_HIP_TRY(hipMemcpyAsync(host_local, v->d, sizeof(double) * (size_t)v->n_local, hipMemcpyDeviceToHost, stream_of(c)));
    SLA_HIP_TRY(hipStreamSynchronize(stream_of(c)));
    return SLA_OK;
}

int sla_vec_to_host(sla_vec_t v, double *host) {
    if (v && !v->kids.empty()) return m_vec_to_host(v, host);
    if (!v || !host) return fail(SLA_ERR_INVALID, "null argument");
    sla_ctx *c = v->ctx;
    Bind bind(c);
    if (!c->collectives) return sla_vec_to_host_local(v, host);
    const double *base = nullptr;
    SLA_TRY(gather_x(nullptr, v, &base));
    SLA_HIP_TRY(hipMemcpyAsync(host, base, sizeof(double) * (size_t)v->n, hipMemcpyDeviceToHost, stream_of(c)));
    SLA_HIP_TRY(hipStreamSynchronize(stream_of(c)));
    return SLA_OK;
}

int sla_vec_copy(sla_vec_t src, sla_vec_t dst) {
    if (src && !src->kids.empty()) return m_vec_copy(src, dst);
    if (!src || !dst) return fail(SLA_ERR_INVALID, "null vector");
    if (!dst->kids.empty()) return mixed_handles("sla_vec_copy");
    if (src->n != dst->n || src->ctx != dst->ctx) return fail(SLA_ERR_DIM_MISMATCH, "sla_vec_copy: mismatched dimensions");
    Bind bind(src->ctx);
    if (src->shard > 0)
        SLA_HIP_TRY(hipMemcpyAsync(dst->d, src->d, sizeof(double) * (size_t)src->shard, hipMemcpyDeviceToDevice, stream_of(src->ctx)));
    return SLA_OK;
}

// ---- (#>) (<#) (<.>) norm2 axpby ---------------------------------------------------------------------

int sla_spmv(sla_csr_t A, sla_vec_t x, sla_vec_t y) {
    if (A && !A->kids.empty()) return m_spmv(A, x, y, false);
    if (!A || !x || !y) return fail(SLA_ERR_INVALID, "null argument");
    if (!x->kids.empty() || !y->kids.empty() || x->ctx != A->ctx || y->ctx != A->ctx) return mixed_handles("sla_spmv");
    if (A->n != x->n) return fail(SLA_ERR_DIM_MISMATCH, "matVec : mismatched dimensions");  // Common.hs:250
    if (A->m != y->n) return fail(SLA_ERR_DIM_MISMATCH, "matVec : result vector has the wrong dimension");
    if (x == y) return fail(SLA_ERR_INVALID, "sla_spmv: x and y must be distinct");
    Bind bind(A->ctx);
    SpmvLaunch l;
    l.y = y->d;
    l.kernel_id = SLA_KERNEL_SPMV;
    return spmv_exchanged(A, x, l, nullptr);
}

int sla_spmv_t(sla_csr_t A, sla_vec_t x, sla_vec_t y) {
    if (A && !A->kids.empty()) return m_spmv(A, x, y, true);
    return no_throw("sla_spmv_t", [&]() -> int {
        if (!A || !x || !y) return fail(SLA_ERR_INVALID, "null argument");
        if (!x->kids.empty() || !y->kids.empty() || x->ctx != A->ctx || y->ctx != A->ctx) return mixed_handles("sla_spmv_t");
        if (A->m != x->n) return fail(SLA_ERR_DIM_MISMATCH, "vecMat : mismatching dimensions");  // Common.hs:256
        if (A->n != y->n) return fail(SLA_ERR_DIM_MISMATCH, "vecMat : result vector has the wrong dimension");
        if (x == y) return fail(SLA_ERR_INVALID, "sla_spmv_t: x and y must be distinct");
        Bind bind(A->ctx);
        return spmv_transposed(A, x->d, y->d, y->shard);
    });
}

int sla_dot(sla_vec_t x, sla_vec_t y, double *out) {
    if (x && !x->kids.empty()) return out ? m_dot(x, y, out) : fail(SLA_ERR_INVALID, "null argument");
    if (!x || !y || !out) return fail(SLA_ERR_INVALID, "null argument");
    if (!y->kids.empty()) return mixed_handles("sla_dot");
    if (x->ctx != y->ctx) return fail(SLA_ERR_INVALID, "vectors from different contexts");
    // the reference's liftI2 takes max of the dims and never checks (SpVector.hs:64); dense device
    // vectors need equal length
    if (x->n != y->n) return fail(SLA_ERR_DIM_MISMATCH, "<.> : mismatched dimensions");
    sla_ctx *c = x->ctx;
    Bind bind(c);
    SLA_TRY(launch_dot(c, x->n_local, x->d, y->d, c->d_parts));
    return reduce_to_host(c, c->d_parts, nullptr, vec_grid(x->n_local), out);
}

int sla_nrm2(sla_vec_t x, double *out) {
    if (x && !x->kids.empty()) return out ? m_nrm2(x, out) : fail(SLA_ERR_INVALID, "null argument");
    double ss = 0.0;
    SLA_TRY(sla_dot(x, x, &ss));
    *out = sqrt(ss);  // norm2 = sqrt . norm2Sq
    return SLA_OK;
}

int sla_axpby(double a, sla_vec_t x, double b, sla_vec_t y) {
    if (x && !x->kids.empty()) return m_axpby(a, x, b, y);
    if (!x || !y) return fail(SLA_ERR_INVALID, "null argument");
    if (!y->kids.empty()) return mixed_handles("sla_axpby");
    if (x->n != y->n || x->ctx != y->ctx) return fail(SLA_ERR_DIM_MISMATCH, "^+^ : mismatched dimensions");
    Bind bind(x->ctx);
    return launch_axpby(x->ctx, x->n_local, a, x->d, b, y->d);
}

int sla_scal(double a, sla_vec_t x) {
    if (x && !x->kids.empty()) return m_scal(a, x);
    if (!x) return fail(SLA_ERR_INVALID, "null argument");
    Bind bind(x->ctx);
    return launch_scal(x->ctx, x->n_local, a, x->d);
}

int sla_plan_window_exchange(int nranks, int rank, int64_t n, const int64_t *windows, int64_t *send_begin,
                             int64_t *send_len, int64_t *recv_begin, int64_t *recv_len, int *use_window) {
    if (nranks < 1 || rank < 0 || rank >= nranks || n < 0 || !windows) return fail(SLA_ERR_INVALID, "sla_plan_window_exchange: bad argument");
    XPlan plan;
    plan_window_exchange(nranks, rank, n, windows, plan);
    for (int q = 0; q < nranks; ++q) {
        if (send_begin) send_begin[q] = plan.send_begin[(size_t)q];
        if (send_len) send_len[q] = plan.send_len[(size_t)q];
        if (recv_begin) recv_begin[q] = plan.recv_begin[(size_t)q];
        if (recv_len) recv_len[q] = plan.recv_len[(size_t)q];
    }
    if (use_window) *use_window = plan.use_window ? 1 : 0;
    return SLA_OK;
}

// ---- measurement hooks ---------------------------------------------------------------------------------

int sla_prof_start(sla_ctx_t c, int kernel_id, int max_launches) {
    if (c && !c->kids.empty()) { for (sla_ctx *k : c->kids) SLA_TRY(sla_prof_start(k, kernel_id, max_launches)); return SLA_OK; }
    if (!c || max_launches < 0 || kernel_id < SLA_KERNEL_ALL || kernel_id >= SLA_KERNEL_COUNT) return fail(SLA_ERR_INVALID, "sla_prof_start: bad argument");
    Bind bind(c);
    while ((int)c->prof_ev.size() < 2 * max_launches) {
        hipEvent_t ev;
        SLA_HIP_TRY(hipEventCreate(&ev));
        c->prof_ev.push_back(ev);
    }
    c->prof_ids.assign((size_t)max_launches, -2);
    c->prof_ms.clear();
    c->prof_kernel = kernel_id;
    c->prof_max = max_launches;
    c->prof_count = 0;
    return SLA_OK;
}

static void prof_stats(const sla_ctx *c, int kernel_id, int *launches, double *mean_ms, double *min_ms) {
    double sum = 0.0, mn = 1e300;
    int cnt = 0;
    for (size_t i = 0; i < c->prof_ms.size(); ++i) {
        if (kernel_id != SLA_KERNEL_ALL && c->prof_ids[i] != kernel_id) continue;
        sum += c->prof_ms[i];
        mn = std::min<double>(mn, c->prof_ms[i]);
        ++cnt;
    }
    if (launches) *launches = cnt;
    if (mean_ms) *mean_ms = cnt ? sum / cnt : 0.0;
    if (min_ms) *min_ms = cnt ? mn : 0.0;
}

int sla_prof_query(sla_ctx_t c, int kernel_id, int *launches, double *mean_ms, double *min_ms) {
    if (c && !c->kids.empty()) return sla_prof_query(c->kids[0], kernel_id, launches, mean_ms, min_ms);
    if (!c || kernel_id < SLA_KERNEL_ALL || kernel_id >= SLA_KERNEL_COUNT) return fail(SLA_ERR_INVALID, "sla_prof_query: bad argument");
    prof_stats(c, kernel_id, launches, mean_ms, min_ms);
    return SLA_OK;
}

int sla_device_count(int *count) {
    if (!count) return fail(SLA_ERR_INVALID, "null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return SLA_OK;
}

int sla_ctx_comm_ranks(sla_ctx_t c, int *nranks) {
    if (c && !c->kids.empty()) return sla_ctx_comm_ranks(c->kids[0], nranks);
    if (!c || !nranks) return fail(SLA_ERR_INVALID, "null argument");
    return dist_comm_count(c, nranks);
}

int sla_prof_stop(sla_ctx_t c, int *launches, double *mean_ms, double *min_ms) {
    if (c && !c->kids.empty()) { for (size_t r = c->kids.size(); r-- > 1;) SLA_TRY(sla_prof_stop(c->kids[r], nullptr, nullptr, nullptr)); return sla_prof_stop(c->kids[0], launches, mean_ms, min_ms); }
    if (!c) return fail(SLA_ERR_INVALID, "null context");
    Bind bind(c);
    SLA_HIP_TRY(hipStreamSynchronize(stream_of(c)));
    c->prof_ms.assign((size_t)c->prof_count, 0.f);
    for (int i = 0; i < c->prof_count; ++i)
        SLA_HIP_TRY(hipEventElapsedTime(&c->prof_ms[(size_t)i], c->prof_ev[2 * (size_t)i], c->prof_ev[2 * (size_t)i + 1]));
    prof_stats(c, SLA_KERNEL_ALL, launches, mean_ms, min_ms);
    c->prof_kernel = -2;
    c->prof_max = 0;
    return SLA_OK;
}

}  // extern "C"
