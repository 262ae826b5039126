// sla_multi.cpp -- single-process multi-device contexts (`sla_ctx_create(n_gpus, device_ids)`).
//
// The reference's caller is ONE Haskell program calling `linSolve0 BICGSTAB_ aa b x0` (Sparse.hs:1016-1021): for it to use
// the 8 GPUs of a node the fan-out has to live behind the C ABI.  sla_ctx_create_multi builds one rank context per device
// (the same contexts the one-process-per-GPU mode creates: RCCL communicator from a shared unique id, or the in-process
// loopback communicator when a device id repeats -- the test backend for 1-GPU boxes) and returns a PARENT context.  Every
// handle created from the parent (matrix, vector, solver) is a bundle of the per-rank handles; every call on a bundle runs
// the ordinary per-rank entry point on all ranks at once, one host thread per rank -- the ranks' calls contain collectives
// and host synchronisation, so they must be issued concurrently, exactly as N processes would.  Results that are global by
// construction (inner products, solver info, H of the Arnoldi relation) are taken from rank 0; vectors and matrices are
// given / returned whole (each rank keeps or contributes its row block).
//
// What is not sharded (`##`, factorizations, triangular solves, pre-sharded input) returns SLA_ERR_INVALID on a
// multi-device context.
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "sla_internal.hpp"

namespace sla {

// One persistent host thread per rank >= 1 of a parent context (rank 0's share runs on the caller's thread).  The per-rank entry
// points contain collectives and host synchronisation, so they must be issued concurrently, exactly as N processes would;
// spawning N threads per API call (every sla_dot on a multi context) cost a thread creation + join each time and gave every
// call a brand-new thread whose current HIP device is 0.  A worker binds its device once when it starts and every entry point
// it runs binds again (Bind, sla_internal.hpp).
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = false, quit = false;
    void loop(int device) {
        (void)hipSetDevice(device);
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return has_job || quit; });
            if (quit) return;
            lk.unlock();
            job();
            lk.lock();
            has_job = false;
            done = true;
            cv.notify_all();
        }
    }
    void post(std::function<void()> j) {
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(j);
        has_job = true;
        done = false;
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
    }
};
struct Pool {
    std::vector<Worker *> w;   // w[r - 1] serves rank r
    explicit Pool(const std::vector<int> &devices) {
        for (size_t r = 1; r < devices.size(); ++r) {
            Worker *k = new Worker();
            const int dev = devices[r];
            k->th = std::thread([k, dev] { k->loop(dev); });
            w.push_back(k);
        }
    }
    ~Pool() {
        for (Worker *k : w) {
            {
                std::lock_guard<std::mutex> lk(k->mu);
                k->quit = true;
                k->cv.notify_all();
            }
            k->th.join();
            delete k;
        }
    }
};

// f(rank) on every rank concurrently; the first failing rank's status and message are the call's
static int fanout(Pool *pool, int n, const std::function<int(int)> &f) {
    std::vector<int> rc((size_t)n, SLA_OK);
    std::vector<std::string> msg((size_t)n);
    for (int r = 1; r < n; ++r)
        pool->w[(size_t)r - 1]->post([&, r] {
            rc[(size_t)r] = f(r);
            if (rc[(size_t)r] != SLA_OK) msg[(size_t)r] = sla_last_error();
        });
    rc[0] = f(0);
    if (rc[0] != SLA_OK) msg[0] = sla_last_error();
    for (int r = 1; r < n; ++r) pool->w[(size_t)r - 1]->wait();
    for (int r = 0; r < n; ++r)
        if (rc[(size_t)r] != SLA_OK) {
            set_error(msg[(size_t)r]);
            return rc[(size_t)r];
        }
    return SLA_OK;
}
static int fanout(const sla_ctx *parent, const std::function<int(int)> &f) { return fanout((Pool *)parent->pool, (int)parent->kids.size(), f); }

int multi_unsupported(const char *what) {
    return fail(SLA_ERR_INVALID, std::string(what) + ": not available on a multi-device context (not sharded, SURVEY 8(e)); use a single-device context");
}

template <class H>
static H *bundle(sla_ctx *parent, std::vector<H *> &kids) {
    H *b = new H();
    b->ctx = parent;
    b->kids = kids;
    return b;
}

// ---- context ------------------------------------------------------------------------------------------------
int m_ctx_destroy(sla_ctx *p) {
    (void)fanout(p, [&](int r) { return sla_ctx_destroy(p->kids[(size_t)r]); });
    p->kids.clear();
    delete (Pool *)p->pool;
    delete p;
    return SLA_OK;
}
int m_ctx_sync(sla_ctx *p) {
    return fanout(p, [&](int r) { return sla_ctx_sync(p->kids[(size_t)r]); });
}

// ---- matrices -----------------------------------------------------------------------------------------------
template <class F>
static int make_csr(sla_ctx *p, sla_csr_t *out, F create) {
    const int n = (int)p->kids.size();
    std::vector<sla_csr *> kids((size_t)n, nullptr);
    const int rc = fanout(p, [&](int r) { return create(p->kids[(size_t)r], &kids[(size_t)r]); });
    if (rc != SLA_OK) {
        for (sla_csr *k : kids) sla_csr_destroy(k);
        return rc;
    }
    sla_csr *b = bundle(p, kids);
    b->m = kids[0]->m;
    b->n = kids[0]->n;
    b->rows = b->m;
    b->is_diagonal = kids[0]->is_diagonal;
    for (sla_csr *k : kids) b->nnz += k->nnz;
    *out = b;
    return SLA_OK;
}
int m_csr_from_coo(sla_ctx *p, int64_t m, int64_t n, int64_t nnz, const int64_t *row, const int64_t *col, const double *val, int dup, sla_csr_t *out) {
    return make_csr(p, out, [&](sla_ctx *c, sla_csr **o) { return sla_csr_from_coo(c, m, n, nnz, row, col, val, dup, o); });
}
int m_csr_from_csr(sla_ctx *p, int64_t m, int64_t n, const int64_t *rp, const int64_t *ci, const double *va, sla_csr_t *out) {
    return make_csr(p, out, [&](sla_ctx *c, sla_csr **o) { return sla_csr_from_csr(c, m, n, rp, ci, va, o); });
}
int m_csr_from_matrix_market(sla_ctx *p, const char *path, int dup, sla_csr_t *out) {
    return make_csr(p, out, [&](sla_ctx *c, sla_csr **o) { return sla_csr_from_matrix_market(c, path, dup, o); });
}
int m_csr_destroy(sla_csr *A) {
    (void)fanout(A->ctx, [&](int r) { return sla_csr_destroy(A->kids[(size_t)r]); });
    A->kids.clear();
    delete A;
    return SLA_OK;
}
int m_csr_export(sla_csr *A, int64_t *rowptr, int64_t *colidx, double *val) {
    // the rank blocks are consecutive row ranges: concatenate them (row pointers re-based)
    int64_t row0 = 0, nz0 = 0;
    rowptr[0] = 0;
    for (sla_csr *k : A->kids) {
        std::vector<int64_t> rp((size_t)k->rows + 1);
        SLA_TRY(sla_csr_export(k, rp.data(), colidx + nz0, val + nz0));
        for (int64_t i = 1; i <= k->rows; ++i) rowptr[row0 + i] = nz0 + rp[(size_t)i];
        row0 += k->rows;
        nz0 += k->nnz;
    }
    return SLA_OK;
}

// ---- vectors ------------------------------------------------------------------------------------------------
template <class F>
static int make_vec(sla_ctx *p, sla_vec_t *out, F create) {
    const int n = (int)p->kids.size();
    std::vector<sla_vec *> kids((size_t)n, nullptr);
    const int rc = fanout(p, [&](int r) { return create(p->kids[(size_t)r], &kids[(size_t)r]); });
    if (rc != SLA_OK) {
        for (sla_vec *k : kids) sla_vec_destroy(k);
        return rc;
    }
    sla_vec *b = bundle(p, kids);
    b->n = kids[0]->n;
    b->n_local = b->n;
    b->shard = b->n;
    *out = b;
    return SLA_OK;
}
int m_vec_create(sla_ctx *p, int64_t n, const double *host, sla_vec_t *out) {
    return make_vec(p, out, [&](sla_ctx *c, sla_vec **o) { return sla_vec_create(c, n, host, o); });
}
int m_vec_from_matrix_market(sla_ctx *p, const char *path, sla_vec_t *out) {
    return make_vec(p, out, [&](sla_ctx *c, sla_vec **o) { return sla_vec_from_matrix_market(c, path, o); });
}
int m_vec_destroy(sla_vec *v) {
    (void)fanout(v->ctx, [&](int r) { return sla_vec_destroy(v->kids[(size_t)r]); });
    v->kids.clear();
    delete v;
    return SLA_OK;
}
int m_vec_to_host(sla_vec *v, double *host) {   // every rank downloads its own block into its slice: no collective
    return fanout(v->ctx, [&](int r) {
        sla_vec *k = v->kids[(size_t)r];
        return k->n_local > 0 ? sla_vec_to_host_local(k, host + k->begin) : SLA_OK;
    });
}

template <class H>
static bool same_shape(const H *a, const H *b) { return a && b && a->kids.size() == b->kids.size() && a->ctx == b->ctx; }
#define SLA_NEED_BUNDLE(cond, what) \
    if (!(cond)) return fail(SLA_ERR_INVALID, std::string(what) + ": operands must all come from the same multi-device context")

int m_vec_copy(sla_vec *s, sla_vec *d) {
    SLA_NEED_BUNDLE(same_shape(s, d), "sla_vec_copy");
    return fanout(s->ctx, [&](int r) { return sla_vec_copy(s->kids[(size_t)r], d->kids[(size_t)r]); });
}
int m_spmv(sla_csr *A, sla_vec *x, sla_vec *y, bool transposed) {
    SLA_NEED_BUNDLE(x && y && A->kids.size() == x->kids.size() && same_shape(x, y) && A->ctx == x->ctx, "sla_spmv");
    return fanout(A->ctx, [&](int r) {
        return transposed ? sla_spmv_t(A->kids[(size_t)r], x->kids[(size_t)r], y->kids[(size_t)r])
                          : sla_spmv(A->kids[(size_t)r], x->kids[(size_t)r], y->kids[(size_t)r]);
    });
}
int m_dot(sla_vec *x, sla_vec *y, double *out) {   // every rank ends up with the same rank-ordered sum: take rank 0's
    SLA_NEED_BUNDLE(same_shape(x, y), "sla_dot");
    std::vector<double> v(x->kids.size(), 0.0);
    SLA_TRY(fanout(x->ctx, [&](int r) { return sla_dot(x->kids[(size_t)r], y->kids[(size_t)r], &v[(size_t)r]); }));
    *out = v[0];
    return SLA_OK;
}
int m_nrm2(sla_vec *x, double *out) {
    std::vector<double> v(x->kids.size(), 0.0);
    SLA_TRY(fanout(x->ctx, [&](int r) { return sla_nrm2(x->kids[(size_t)r], &v[(size_t)r]); }));
    *out = v[0];
    return SLA_OK;
}
int m_axpby(double a, sla_vec *x, double b, sla_vec *y) {
    SLA_NEED_BUNDLE(same_shape(x, y), "sla_axpby");
    return fanout(x->ctx, [&](int r) { return sla_axpby(a, x->kids[(size_t)r], b, y->kids[(size_t)r]); });
}
int m_scal(double a, sla_vec *x) {
    return fanout(x->ctx, [&](int r) { return sla_scal(a, x->kids[(size_t)r]); });
}

// ---- solver state records -----------------------------------------------------------------------------------
int m_solver_init(int method, sla_csr *A, sla_vec *b, sla_vec *x0, sla_solver_t *out) {
    SLA_NEED_BUNDLE(b && x0 && A->kids.size() == b->kids.size() && same_shape(b, x0) && A->ctx == b->ctx, "sla_solver_init");
    const int n = (int)A->kids.size();
    std::vector<sla_solver *> kids((size_t)n, nullptr);
    const int rc = fanout(A->ctx, [&](int r) { return sla_solver_init(method, A->kids[(size_t)r], b->kids[(size_t)r], x0->kids[(size_t)r], &kids[(size_t)r]); });
    if (rc != SLA_OK) {
        for (sla_solver *k : kids) sla_solver_destroy(k);
        return rc;
    }
    sla_solver *s = bundle(A->ctx, kids);
    s->method = method;
    *out = s;
    return SLA_OK;
}
int m_solver_step(sla_solver *S, int k) {
    return fanout(S->ctx, [&](int r) { return sla_solver_step(S->kids[(size_t)r], k); });
}
int m_solver_get(sla_solver *S, int field, sla_vec *out) {
    SLA_NEED_BUNDLE(out && S->kids.size() == out->kids.size() && S->ctx == out->ctx, "sla_solver_get");
    return fanout(S->ctx, [&](int r) { return sla_solver_get(S->kids[(size_t)r], field, out->kids[(size_t)r]); });
}
int m_solver_clone(sla_solver *S, sla_solver_t *out) {
    const int n = (int)S->kids.size();
    std::vector<sla_solver *> kids((size_t)n, nullptr);
    const int rc = fanout(S->ctx, [&](int r) { return sla_solver_clone(S->kids[(size_t)r], &kids[(size_t)r]); });
    if (rc != SLA_OK) {
        for (sla_solver *k : kids) sla_solver_destroy(k);
        return rc;
    }
    sla_solver *s = bundle(S->ctx, kids);
    s->method = S->method;
    *out = s;
    return SLA_OK;
}
int m_solver_set_shadow(sla_solver *S, sla_vec *r0hat) {
    SLA_NEED_BUNDLE(r0hat && S->kids.size() == r0hat->kids.size() && S->ctx == r0hat->ctx, "sla_solver_set_shadow");
    return fanout(S->ctx, [&](int r) { return sla_solver_set_shadow(S->kids[(size_t)r], r0hat->kids[(size_t)r]); });
}
int m_solver_destroy(sla_solver *S) {
    (void)fanout(S->ctx, [&](int r) { return sla_solver_destroy(S->kids[(size_t)r]); });
    S->kids.clear();
    delete S;
    return SLA_OK;
}

// ---- linSolve0 / GMRES / (<\>) / arnoldi -----------------------------------------------------------------------
int m_linsolve0(int method, sla_csr *A, sla_vec *b, sla_vec *x0, const sla_solve_opts *o, sla_vec *xo, sla_solve_info *info) {
    SLA_NEED_BUNDLE(b && x0 && xo && A->kids.size() == b->kids.size() && same_shape(b, x0) && same_shape(b, xo) && A->ctx == b->ctx, "sla_linsolve0");
    sla_solve_opts own, rest;   // rank 0 alone fills the caller's residual trace (the values are identical on every rank: one download, no race)
    SLA_TRY(read_solve_opts(o, &own, "sla_linsolve0"));
    sla_solve_info probe;
    SLA_TRY(info_begin(info, &probe, "sla_linsolve0"));
    rest = own;
    rest.history = nullptr;
    rest.history_cap = 0;
    std::vector<sla_solve_info> infos(A->kids.size(), sla_solve_info SLA_SOLVE_INFO_INIT);
    const int rc = fanout(A->ctx, [&](int r) {
        return sla_linsolve0(method, A->kids[(size_t)r], b->kids[(size_t)r], x0->kids[(size_t)r], r == 0 ? &own : &rest, xo->kids[(size_t)r], &infos[(size_t)r]);
    });
    info_commit(info, infos[0]);   // (every rank takes the same decisions from the same rank-ordered sums)
    return rc;
}
int m_gmres(sla_csr *A, sla_vec *b, sla_vec *x0, int restart, const sla_solve_opts *o, sla_vec *xo, sla_solve_info *info) {
    SLA_NEED_BUNDLE(b && x0 && xo && A->kids.size() == b->kids.size() && same_shape(b, x0) && same_shape(b, xo) && A->ctx == b->ctx, "sla_gmres");
    sla_solve_info probe;
    SLA_TRY(info_begin(info, &probe, "sla_gmres"));
    std::vector<sla_solve_info> infos(A->kids.size(), sla_solve_info SLA_SOLVE_INFO_INIT);
    const int rc = fanout(A->ctx, [&](int r) {
        return sla_gmres(A->kids[(size_t)r], b->kids[(size_t)r], x0->kids[(size_t)r], restart, o, xo->kids[(size_t)r], &infos[(size_t)r]);
    });
    info_commit(info, infos[0]);
    return rc;
}
int m_linsolve(sla_csr *A, sla_vec *b, sla_vec *xo, sla_solve_info *info) {
    SLA_NEED_BUNDLE(b && xo && A->kids.size() == b->kids.size() && same_shape(b, xo) && A->ctx == b->ctx, "sla_linsolve");
    sla_solve_info probe;
    SLA_TRY(info_begin(info, &probe, "sla_linsolve"));
    std::vector<sla_solve_info> infos(A->kids.size(), sla_solve_info SLA_SOLVE_INFO_INIT);
    const int rc = fanout(A->ctx, [&](int r) {
        return sla_linsolve(A->kids[(size_t)r], b->kids[(size_t)r], xo->kids[(size_t)r], &infos[(size_t)r]);
    });
    info_commit(info, infos[0]);
    return rc;
}
int m_arnoldi(sla_csr *A, sla_vec *b, int kn, double *Q, double *H, int *k_done) {
    SLA_NEED_BUNDLE(b && A->kids.size() == b->kids.size() && A->ctx == b->ctx, "sla_arnoldi");
    const int n = (int)A->kids.size();
    const int64_t nglob = b->n;
    std::vector<std::vector<double>> ql((size_t)n), hl((size_t)n);
    std::vector<int> kd((size_t)n, 0);
    SLA_TRY(fanout(A->ctx, [&](int r) {
        sla_vec *bk = b->kids[(size_t)r];
        if (Q) ql[(size_t)r].assign((size_t)std::max<int64_t>(bk->n_local, 1) * (size_t)(kn + 1), 0.0);
        hl[(size_t)r].assign((size_t)(kn + 1) * (size_t)kn, 0.0);
        return sla_arnoldi(A->kids[(size_t)r], bk, kn, Q ? ql[(size_t)r].data() : nullptr, hl[(size_t)r].data(), &kd[(size_t)r]);
    }));
    ::memcpy(H, hl[0].data(), sizeof(double) * (size_t)(kn + 1) * (size_t)kn);
    *k_done = kd[0];
    if (Q)   // the ranks' row blocks (leading dimension n_local) -> the n x (k+1) column-major matrix
        for (int r = 0; r < n; ++r) {
            const sla_vec *bk = b->kids[(size_t)r];
            for (int j = 0; j <= kd[0]; ++j)
                if (bk->n_local > 0)
                    ::memcpy(Q + (size_t)j * (size_t)nglob + (size_t)bk->begin, ql[(size_t)r].data() + (size_t)j * (size_t)bk->n_local, sizeof(double) * (size_t)bk->n_local);
        }
    return SLA_OK;
}

}  // namespace sla

using namespace sla;

extern "C" int sla_ctx_create_multi(int n_gpus, const int *device_ids, sla_ctx_t *out) {
    return no_throw("sla_ctx_create_multi", [&]() -> int {
        if (!out || n_gpus < 1 || n_gpus > 64) return fail(SLA_ERR_INVALID, "sla_ctx_create_multi: bad argument");
        if (n_gpus == 1) return sla_ctx_create(device_ids ? device_ids[0] : 0, out);
        std::vector<int> ids((size_t)n_gpus);
        for (int r = 0; r < n_gpus; ++r) ids[(size_t)r] = device_ids ? device_ids[r] : r;
        const bool distinct = std::set<int>(ids.begin(), ids.end()).size() == ids.size();
        std::vector<sla_ctx *> kids((size_t)n_gpus, nullptr);
        Pool *pool = new Pool(ids);
        int rc;
        if (distinct) {   // one RCCL rank per device, all created concurrently from one unique id (like ncclCommInitAll)
            char uid[128];
            rc = sla_dist_unique_id(uid);
            if (rc == SLA_OK) rc = fanout(pool, n_gpus, [&](int r) { return sla_ctx_create_dist(ids[(size_t)r], r, n_gpus, uid, &kids[(size_t)r]); });
        } else {          // repeated device id: RCCL refuses two ranks on one GPU -> the in-process loopback communicator (test backend)
            static std::atomic<int> next_key{0x4d554c54};
            const int key = next_key.fetch_add(1);
            rc = fanout(pool, n_gpus, [&](int r) { return sla_ctx_create_loopback(ids[(size_t)r], r, n_gpus, key, &kids[(size_t)r]); });
        }
        if (rc != SLA_OK) {
            const std::string msg = sla_last_error();
            for (sla_ctx *k : kids) sla_ctx_destroy(k);
            delete pool;
            set_error(msg);
            return rc;
        }
        sla_ctx *p = new sla_ctx();
        p->device = ids[0];
        p->kids = kids;
        p->pool = pool;
        *out = p;
        return SLA_OK;
    });
}
