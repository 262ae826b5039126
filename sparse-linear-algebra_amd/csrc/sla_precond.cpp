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
static void tri_blocks_free(sla_tri_plan *p) {
    if (p->d_bl_slots) (void)hipFree(p->d_bl_slots);
    if (p->d_bl_row) (void)hipFree(p->d_bl_row);
    if (p->d_bl_col) (void)hipFree(p->d_bl_col);
    if (p->d_bl_ptr) (void)hipFree(p->d_bl_ptr);
    if (p->d_bl_val) (void)hipFree(p->d_bl_val);
    if (p->d_bl_diag) (void)hipFree(p->d_bl_diag);
    p->d_bl_slots = nullptr;
    p->d_bl_row = p->d_bl_col = nullptr;
    p->d_bl_ptr = nullptr;
    p->d_bl_val = p->d_bl_diag = nullptr;
    p->nb = 0;
    p->brows = 0;
}
void tri_plan_free(sla_tri_plan *p) {
    if (!p) return;
    if (p->graph) (void)hipGraphExecDestroy(p->graph);
    if (p->d_order) (void)hipFree(p->d_order);
    if (p->d_tptr) (void)hipFree(p->d_tptr);
    if (p->d_tcol) (void)hipFree(p->d_tcol);
    if (p->d_tval) (void)hipFree(p->d_tval);
    if (p->d_tdiag) (void)hipFree(p->d_tdiag);
    tri_blocks_free(p);
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

// The block-local form of the schedule (option tri_syncfree = 2; kernel and rationale: sla_tri.hip).  q = position in the reference's
// sweep (row q of a lower, row n - 1 - q of an upper triangle): every row reads rows of smaller q only.  The rows are laid out in a
// BLOCK ORDER -- any order in which every row still comes after the rows it reads (checked here, row by row) -- and cut into blocks of
// <= R consecutive positions; inside a block the rows are ordered by their level counted over the block's OWN rows, so a workgroup that
// walks its slots in order never waits for a later slot; a block's level counts blocks the same way, and the workgroups take the blocks
// in (level, block) order -- everything a block reads from another block sits earlier in that order.
//
// Block order.  What a solve costs is the longest chain of blocks times the ~5 us a value takes from one CU to another through memory.
// In the sweep order itself a block of a 3-D stencil is a strip of ONE grid plane, and the chain crosses a block boundary per plane (216
// at 216^3).  When the triangle's rows reach back by exactly three distances 1 < s2 < s3 with s2 | s3 (a 7-point-like stencil on an
// s2 x s3 / s2 x . grid), the positions are re-ordered into bricks of a whole line x b lines x c planes (b ~ c, b c s2 <= R), brick after
// brick: the chain then crosses ~ (lines / b + planes / c) boundaries.  The guess is only a guess about speed: an order that puts a row
// before one it reads is thrown away for the sweep order.
static int tri_blocks_build(sla_csr *T, int upper) {
    sla_tri_plan *p = T->tri[upper];
    const int64_t n = T->m;
    const int64_t R = std::max<int64_t>(8, std::min<int64_t>(T->ctx->tri_block_rows, kTriBlockRows));
    if (p->nb && p->brows == (int32_t)R) return SLA_OK;
    tri_blocks_free(p);
    HostCsr h;
    SLA_TRY(export_host(T, h));
    auto row_of = [&](int64_t q) { return upper ? n - 1 - q : q; };   // (its own inverse)
    auto for_deps = [&](int64_t q, auto &&f) {                          // f(sweep position of a row that row_of(q) reads)
        const int64_t i = row_of(q);
        for (int64_t k = h.rowptr[(size_t)i]; k < h.rowptr[(size_t)i + 1]; ++k) {
            const int64_t j = h.col[(size_t)k];
            if (j != i && (upper ? j > i : j < i)) f(row_of(j));
        }
    };
    // the distances the rows reach back by (at most a handful are kept: more than three and the order stays the sweep's)
    int64_t dist[4] = {0, 0, 0, 0};
    int ndist = 0;
    for (int64_t q = 0; q < n && ndist <= 3; ++q)
        for_deps(q, [&](int64_t qq) {
            const int64_t d = q - qq;
            bool known = false;
            for (int t = 0; t < std::min(ndist, 4); ++t) known = known || dist[t] == d;
            if (!known) { if (ndist < 4) dist[ndist] = d; ++ndist; }
        });
    std::vector<int32_t> ord((size_t)n), pos((size_t)n);   // block-order position -> sweep position, and back
    std::vector<int64_t> bq{0};                             // block b = positions [bq[b], bq[b + 1])
    bool bricks = false;
    if (ndist == 3) {
        std::sort(dist, dist + 3);
        const int64_t s2 = dist[1], s3 = dist[2];
        if (dist[0] == 1 && s2 > 1 && s3 % s2 == 0 && s3 / s2 > 1 && s2 <= R / 4) {
            const int64_t nl = s3 / s2, per = R / s2;                       // lines per plane; lines a block may hold
            int64_t c = std::max<int64_t>(1, (int64_t)floor(sqrt((double)per)));
            int64_t b = std::max<int64_t>(1, per / c);
            b = std::min(b, nl);
            const int64_t npl = (n + s3 - 1) / s3;
            c = std::min(c, npl);
            int64_t at = 0;
            for (int64_t K = 0; K < npl; K += c)
                for (int64_t J = 0; J < nl; J += b) {
                    for (int64_t k = K; k < std::min(npl, K + c); ++k)
                        for (int64_t j = J; j < std::min(nl, J + b); ++j) {
                            const int64_t q0 = k * s3 + j * s2;
                            for (int64_t q = q0; q < std::min(n, q0 + s2); ++q) ord[(size_t)at++] = (int32_t)q;
                        }
                    if (at > bq.back()) bq.push_back(at);
                }
            bricks = at == n;
            if (bricks) {
                for (int64_t t = 0; t < n; ++t) pos[(size_t)ord[(size_t)t]] = (int32_t)t;
                for (int64_t q = 0; q < n && bricks; ++q) for_deps(q, [&](int64_t qq) { bricks = bricks && pos[(size_t)qq] < pos[(size_t)q]; });
            }
        }
    }
    if (!bricks) {
        for (int64_t q = 0; q < n; ++q) ord[(size_t)q] = pos[(size_t)q] = (int32_t)q;
        // Cuts in the sweep order.  A block can only start once its FIRST row's dependencies are there, and the rows behind it usually
        // hang on that row: a cut in the middle of a chain (row q reads row q - 1: a stencil line) makes the block wait for the END of
        // the block before it -- the blocks would run one after the other.  So a cut goes where the first row of the new block reaches
        // back FARTHEST (the start of a line, better still of a plane), searched over the second half of the positions a block may hold.
        std::vector<int32_t> reach((size_t)n);   // q - (largest position row q reads), saturated; n for a row that reads nothing
        for (int64_t q = 0; q < n; ++q) {
            int64_t far = -1;
            for_deps(q, [&](int64_t qq) { far = std::max(far, qq); });
            reach[(size_t)q] = (int32_t)std::min<int64_t>(far < 0 ? n : q - far, std::numeric_limits<int32_t>::max());
        }
        bq.assign(1, 0);
        while (bq.back() < n) {
            const int64_t q0 = bq.back(), hi = std::min(n, q0 + R);
            int64_t cut = hi;
            if (hi < n) {
                int32_t best = -1;
                for (int64_t q = q0 + (R + 1) / 2; q <= hi; ++q)
                    if (reach[(size_t)q] >= best) { best = reach[(size_t)q]; cut = q; }
            }
            bq.push_back(cut);
        }
    }
    const int64_t nb = (int64_t)bq.size() - 1;
    std::vector<int32_t> blk((size_t)n), lv((size_t)n), cell((size_t)n);   // per position: block, level inside it, slot inside it
    for (int64_t b = 0; b < nb; ++b) std::fill(blk.begin() + bq[(size_t)b], blk.begin() + bq[(size_t)b + 1], (int32_t)b);
    std::vector<int32_t> blevel((size_t)nb, 1), slot_pos((size_t)n);
    std::vector<int32_t> cnt;
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t t0 = bq[(size_t)b], t1 = bq[(size_t)b + 1];
        int32_t maxlv = 0;
        for (int64_t t = t0; t < t1; ++t) {
            int32_t l = 0;
            for_deps(ord[(size_t)t], [&](int64_t qq) {
                const int64_t tt = pos[(size_t)qq];
                if (tt >= t0) l = std::max(l, lv[(size_t)tt]);
                else blevel[(size_t)b] = std::max(blevel[(size_t)b], blevel[(size_t)blk[(size_t)tt]] + 1);
            });
            lv[(size_t)t] = l + 1;
            maxlv = std::max(maxlv, l + 1);
        }
        cnt.assign((size_t)maxlv + 2, 0);   // the block's rows by (level, position): counting sort
        for (int64_t t = t0; t < t1; ++t) cnt[(size_t)lv[(size_t)t] + 1]++;
        for (int32_t l = 1; l <= maxlv + 1; ++l) cnt[(size_t)l] += cnt[(size_t)l - 1];
        for (int64_t t = t0; t < t1; ++t) {
            const int32_t sl = cnt[(size_t)lv[(size_t)t]]++;
            cell[(size_t)t] = sl;
            slot_pos[(size_t)(t0 + sl)] = (int32_t)t;
        }
    }
    std::vector<int64_t> taken((size_t)nb);
    for (int64_t b = 0; b < nb; ++b) taken[(size_t)b] = b;
    std::stable_sort(taken.begin(), taken.end(), [&](int64_t a, int64_t b) { return blevel[(size_t)a] < blevel[(size_t)b]; });
    std::vector<int64_t> slots((size_t)(2 * nb + 2), 0), tptr((size_t)n + 1, 0);
    std::vector<int32_t> rows((size_t)std::max<int64_t>(n, 1)), tcol;
    std::vector<double> tval, tdiag((size_t)std::max<int64_t>(n, 1), 1.0);
    tcol.reserve((size_t)h.col.size() / 2 + 16);
    tval.reserve((size_t)h.col.size() / 2 + 16);
    int64_t s = 0, ncross = 0;
    for (int64_t pi = 0; pi < nb; ++pi) {
        const int64_t b = taken[(size_t)pi], t0 = bq[(size_t)b], t1 = bq[(size_t)b + 1];
        slots[(size_t)(2 * pi)] = s;
        slots[(size_t)(2 * pi + 1)] = ord[(size_t)t0];
        for (int64_t t = t0; t < t1; ++t, ++s) {
            const int64_t i = row_of(ord[(size_t)slot_pos[(size_t)t]]);
            rows[(size_t)s] = (int32_t)i;
            for (int64_t k = h.rowptr[(size_t)i]; k < h.rowptr[(size_t)i + 1]; ++k) {   // ascending columns: the reference's fold order
                const int64_t j = h.col[(size_t)k];
                if (j == i) { tdiag[(size_t)s] = h.val[(size_t)k]; continue; }
                if (!(upper ? j > i : j < i)) continue;
                const int64_t tt = pos[(size_t)row_of(j)];
                const bool own = tt >= t0 && tt < t1;
                ncross += own ? 0 : 1;
                tcol.push_back(own ? ~cell[(size_t)tt] : (int32_t)j);
                tval.push_back(h.val[(size_t)k]);
            }
            tptr[(size_t)s + 1] = (int64_t)tcol.size();
        }
    }
    slots[(size_t)(2 * nb)] = s;
    hipError_t e = hipSuccess;
    auto up = [&](void **dst, const void *src, size_t bytes) {
        if (e != hipSuccess) return;
        e = dev_malloc(T->ctx, dst, std::max<size_t>(bytes, 8));
        if (e == hipSuccess && bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    };
    up((void **)&p->d_bl_slots, slots.data(), sizeof(int64_t) * slots.size());
    up((void **)&p->d_bl_row, rows.data(), sizeof(int32_t) * rows.size());
    up((void **)&p->d_bl_ptr, tptr.data(), sizeof(int64_t) * tptr.size());
    up((void **)&p->d_bl_col, tcol.data(), sizeof(int32_t) * tcol.size());
    up((void **)&p->d_bl_val, tval.data(), sizeof(double) * tval.size());
    up((void **)&p->d_bl_diag, tdiag.data(), sizeof(double) * tdiag.size());
    if (e != hipSuccess) {
        tri_blocks_free(p);
        return fail(SLA_ERR_ALLOC, std::string("triangular schedule (blocks): ") + hipGetErrorString(e));
    }
    p->nb = nb;
    p->brows = (int32_t)R;
    p->bricks = bricks;
    p->bl_cross = tcol.empty() ? 0.0 : (double)ncross / (double)tcol.size();
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
        // 0: one launch per dependency level (a HIP graph), 1: one persistent launch whose rows poll x in memory, 2: the block-local
        // persistent launch, 3 (default): 2 where the schedule is deep and narrow (from 64 levels of <= 65536 rows on the average the
        // levels' launch latency, 5 us each, is what the solve costs) AND at most a quarter of the entries read another block's row -- else 0
        int mode = c->tri_syncfree;
        if (mode == 3) {
            mode = 0;
            if (p->nlevels >= 64 && T->m / p->nlevels <= 65536 && !p->bl_rejected) {
                SLA_TRY(tri_blocks_build(T, upper));
                if (p->bl_cross <= 0.25) mode = 2;   // (else most values would be polled in memory: a random matrix' triangle ran 38 ms that way, 1.1 ms by levels)
                else {   // rejected: keep the decision, not the second copy of the triangle (it stayed allocated for the matrix's lifetime)
                    const double cross = p->bl_cross;
                    tri_blocks_free(p);
                    p->bl_cross = cross;
                    p->bl_rejected = true;
                }
            }
        }
        c->tri_mode_used = mode;
        c->tri_plan_note = mode == 2 ? "blocks=" + std::to_string(p->nb) + " block_rows=" + std::to_string(p->brows) + " bricks=" + std::to_string(p->bricks ? 1 : 0) +
                                           " outside_share=" + std::to_string(p->bl_cross)
                                     : std::string("levels=") + std::to_string(p->nlevels);
        if (mode) {   // a lane that waited too long says so and the level schedule below runs instead
            int *d_fail = (int *)(c->d_result + 1600);
            int h_fail = 0;
            if (mode == 2) {
                SLA_TRY(tri_blocks_build(T, upper));
                // (a device whose CUs cannot hold the block kernel -- less LDS, a failing attribute or occupancy query -- is not an error of
                // the solve: the level schedule below runs instead, counted like a run-time give-up)
                const int rc = launch_tri_blocks(T, p, upper, b->d, x->d, d_fail);
                if (rc == SLA_TRI_NO_FIT) h_fail = 1;
                else SLA_TRY(rc);
            } else {
                SLA_TRY(launch_tri_syncfree(T, p, b->d, x->d, d_fail));
            }
            if (!h_fail) {
                SLA_TRY(launch_tri_sparsify(c, T->m, x->d));
                SLA_HIP_TRY(hipMemcpyAsync(&h_fail, d_fail, sizeof(int), hipMemcpyDeviceToHost, stream_of(c)));
                SLA_HIP_TRY(hipStreamSynchronize(stream_of(c)));
                if (!h_fail) return SLA_OK;
            }
            T->ctx->tri_fallbacks++;
        }
        if (p->nlevels > 65536) {   // (a graph of that many nodes is not worth its memory: plain launches)
            for (int64_t l = 0; l < p->nlevels; ++l)
                SLA_TRY(launch_tri_level(T, p, p->level_ptr[(size_t)l], p->level_ptr[(size_t)l + 1] - p->level_ptr[(size_t)l], b->d, x->d));
            return launch_tri_sparsify(c, T->m, x->d);
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
