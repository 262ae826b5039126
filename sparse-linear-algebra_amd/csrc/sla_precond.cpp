// sla_precond.cpp -- C ABI of the preconditioner builders and triangular solves on the host-exported CSR (Numeric/LinearAlgebra/Sparse.hs:
// jacobiPre :678-683, ssorPre :697-709, triLowerSolve / triUpperSolve :731-811; the level-scheduled substitution kernels are in
// sla_tri.hip, ilu0Pre in sla_ilu0.cpp).  Host code: set-up is off the solver path.  Reference citations per entry point are in
// include/sla_hip.h.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cstring>
#include <limits>

#include "sla_internal.hpp"

using namespace sla;

namespace sla {
void tri_plan_free(sla_tri_plan *p) {
    if (!p) return;
    if (p->graph) (void)hipGraphExecDestroy(p->graph);
    if (p->d_order) (void)hipFree(p->d_order);
    if (p->d_tptr) (void)hipFree(p->d_tptr);
    if (p->d_tcol) (void)hipFree(p->d_tcol);
    if (p->d_tval) (void)hipFree(p->d_tval);
    if (p->d_tdiag) (void)hipFree(p->d_tdiag);
    delete p;
}
}  // namespace sla

extern "C" {

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
        if (c->tri_syncfree) {   // one persistent launch; a lane that waited too long says so and the level schedule below runs instead
            int *d_fail = (int *)(c->d_result + 1600);
            SLA_TRY(launch_tri_syncfree(T, p, b->d, x->d, d_fail));
            SLA_TRY(launch_tri_sparsify(c, T->m, x->d));
            int h_fail = 0;
            SLA_HIP_TRY(hipMemcpyAsync(&h_fail, d_fail, sizeof(int), hipMemcpyDeviceToHost, stream_of(c)));
            SLA_HIP_TRY(hipStreamSynchronize(stream_of(c)));
            if (!h_fail) return SLA_OK;
            T->ctx->tri_fallbacks++;
        }
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

}  // extern "C"
