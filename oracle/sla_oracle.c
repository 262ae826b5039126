/*
 * sla_oracle.c -- CPU restatement of the reference hot path.  See sla_oracle.h for status.
 * TEST INFRASTRUCTURE ONLY: never linked into or called from the product (libsla_hip.so).
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 * Arithmetic conventions restated from the reference:
 *   - `sum` over an IntMap = strict left fold from 0 in ascending key order
 *     (src/Data/Sparse/Internal/IntM.hs:17 derived Foldable, base-4.18 Foldable.sum);
 *   - a*x then + : two roundings, never an FMA (build with -ffp-contract=off);
 *   - x ^-^ y = x ^+^ negateV y  (Class.hs:68-69)  ->  x + (-(y));
 *   - scalar norm2Sq = (**2) -> libm pow(x, 2.0)  (Class.hs:405-408).
 */
#include "sla_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- A0: construction */

static int key_less_eq(const int64_t *row, const int64_t *col, int64_t a, int64_t b) {
    if (row[a] != row[b]) return row[a] < row[b];
    return col[a] <= col[b];
}

/* stable bottom-up merge sort of a permutation by (row, col) */
static int sort_perm(int64_t nnz, const int64_t *row, const int64_t *col, int64_t *perm) {
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz > 0 ? nnz : 1));
    if (!tmp) return ORC_ERR_ALLOC;
    int64_t *src = perm, *dst = tmp;
    for (int64_t w = 1; w < nnz; w *= 2) {
        for (int64_t lo = 0; lo < nnz; lo += 2 * w) {
            int64_t mid = lo + w < nnz ? lo + w : nnz;
            int64_t hi = lo + 2 * w < nnz ? lo + 2 * w : nnz;
            int64_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                if (key_less_eq(row, col, src[i], src[j])) dst[k++] = src[i++];
                else dst[k++] = src[j++];
            }
            while (i < mid) dst[k++] = src[i++];
            while (j < hi) dst[k++] = src[j++];
        }
        int64_t *t = src; src = dst; dst = t;
    }
    if (src != perm) memcpy(perm, src, sizeof(int64_t) * (size_t)nnz);
    free(tmp);
    return ORC_OK;
}

/* fromListSM (SpMatrix.hs:218-224): foldl' of insertSpMatrix (:205-210) = IntMap.insert, so a
 * later (i,j) replaces an earlier one (insertIM2, IntMap2.hs:24-28).  The IntMap-of-IntMap
 * traversal order (ascending row, ascending col) is the CSR order of CSR.hs:74-78. */
int orc_coo_to_csr(int64_t m, int64_t n, int64_t nnz, const int64_t *row, const int64_t *col,
                   const double *val, int64_t *rowptr, int64_t *colidx, double *valout,
                   int64_t *nnz_out) {
    for (int64_t k = 0; k < nnz; ++k)
        if (row[k] < 0 || row[k] >= m || col[k] < 0 || col[k] >= n) return ORC_ERR_OOB;
    int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz > 0 ? nnz : 1));
    if (!perm) return ORC_ERR_ALLOC;
    for (int64_t k = 0; k < nnz; ++k) perm[k] = k;
    int rc = sort_perm(nnz, row, col, perm);
    if (rc) { free(perm); return rc; }
    int64_t out = 0;
    int64_t *rows_sorted = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz > 0 ? nnz : 1));
    if (!rows_sorted) { free(perm); return ORC_ERR_ALLOC; }
    for (int64_t k = 0; k < nnz; ++k) {
        int64_t a = perm[k];
        /* the stable sort keeps input order inside a (row,col) group: keep its LAST member */
        if (k + 1 < nnz && row[perm[k + 1]] == row[a] && col[perm[k + 1]] == col[a]) continue;
        rows_sorted[out] = row[a];
        colidx[out] = col[a];
        valout[out] = val[a];
        ++out;
    }
    orc_cs_ptr(m, rows_sorted, out, rowptr);
    *nnz_out = out;
    free(rows_sorted);
    free(perm);
    return ORC_OK;
}

/* csPtrV (==) n xs (vector/.../Vector/Utils.hs:12-26): ptr[0]=0, ptr[i+1]=ptr[i]+#{x==i};
 * doc example [1,1,2,3], n=4 -> [0,0,2,3,4]. */
void orc_cs_ptr(int64_t n, const int64_t *sorted_ix, int64_t len, int64_t *ptr) {
    int64_t pos = 0, count = 0;
    ptr[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        while (pos < len && sorted_ix[pos] == i) { ++pos; ++count; }
        ptr[i + 1] = count;
    }
}

/* transposeIM2 = ifoldlIM2 (flip insertIM2) (IntMap2.hs:88-89): row j of A^T holds a_ij keyed by
 * i, traversed ascending in i. */
int orc_csr_transpose(int64_t m, int64_t n, const int64_t *rowptr, const int64_t *colidx,
                      const double *val, int64_t *t_rowptr, int64_t *t_colidx, double *t_val) {
    int64_t nnz = rowptr[m];
    for (int64_t j = 0; j <= n; ++j) t_rowptr[j] = 0;
    for (int64_t k = 0; k < nnz; ++k) t_rowptr[colidx[k] + 1]++;
    for (int64_t j = 0; j < n; ++j) t_rowptr[j + 1] += t_rowptr[j];
    int64_t *cursor = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    if (!cursor) return ORC_ERR_ALLOC;
    for (int64_t j = 0; j < n; ++j) cursor[j] = t_rowptr[j];
    for (int64_t i = 0; i < m; ++i)
        for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            int64_t d = cursor[colidx[k]]++;
            t_colidx[d] = i;
            t_val[d] = val[k];
        }
    free(cursor);
    return ORC_OK;
}

/* isDiagonalSM (SpMatrix.hs:411-415): #rows whose map has exactly one entry, on the diagonal,
 * must equal nrows. */
int orc_is_diagonal(int64_t m, const int64_t *rowptr, const int64_t *colidx) {
    int64_t d = 0;
    for (int64_t i = 0; i < m; ++i)
        if (rowptr[i + 1] - rowptr[i] == 1 && colidx[rowptr[i]] == i) ++d;
    return d == m;
}

/* ---------------------------------------------------------------- A1..A4: BLAS-1 / SpMV */

/* matVecSD (Common.hs:247-250): fmap (`dotu` x) rows; dotu u v = sum (liftI2 (*) u v)
 * (:259-260): acc = 0; for ascending j: acc = acc + a_ij * x_j. */
void orc_spmv(int64_t m, const int64_t *rowptr, const int64_t *colidx, const double *val,
              const double *x, double *y) {
#ifdef ORC_OMP /* row-parallel: per-row order unchanged (liboracle_omp.so, "fair CPU" timing only) */
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < m; ++i) {
        double acc = 0.0;
        for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            double prod = val[k] * x[colidx[k]];
            acc = acc + prod;
        }
        y[i] = acc;
    }
}

/* NOT the reference's order: y = A x with every row folded over column PANELS of 2^shift columns in a given visiting order
 * (pos[j] = position of panel j in the walk), ascending columns inside a panel, separately rounded multiply and add from 0.0.
 * This restates what the product's overlapped all-gather does on a sharded tile-form matrix in "arrival" order (DESIGN.md section
 * 6: own panels first, then by exchange group), so that its rows can be checked bit for bit; with pos = identity it is orc_spmv. */
void orc_spmv_panel_order(int64_t m, const int64_t *rowptr, const int64_t *colidx, const double *val, const double *x, double *y,
                          int shift, int64_t npanels, const int32_t *pos) {
    for (int64_t i = 0; i < m; ++i) {
        /* the row's (panel) segments are contiguous (columns ascend): visit them by ascending pos -- repeated minimum search, rows are short */
        double acc = 0.0;
        int64_t done_pos = -1;
        for (;;) {
            int64_t best = -1, best_pos = npanels;            /* next segment in the walk */
            for (int64_t k = rowptr[i]; k < rowptr[i + 1];) {
                const int64_t j = colidx[k] >> shift;
                if ((int64_t)pos[j] > done_pos && (int64_t)pos[j] < best_pos) { best = k; best_pos = pos[j]; }
                while (k < rowptr[i + 1] && (colidx[k] >> shift) == j) ++k;
            }
            if (best < 0) break;
            const int64_t j = colidx[best] >> shift;
            for (int64_t k = best; k < rowptr[i + 1] && (colidx[k] >> shift) == j; ++k) {
                double prod = val[k] * x[colidx[k]];
                acc = acc + prod;
            }
            done_pos = best_pos;
        }
        y[i] = acc;
    }
}

/* v <.> w = sum (liftI2 (<.>) v w)  (SpVector.hs:116-117; Double: (<.>) = (*), Class.hs:400) */
double orc_dot(int64_t n, const double *x, const double *y) {
    double acc = 0.0;
#ifdef ORC_OMP /* parallel reduction: NOT the reference's summation order; timing build only */
#pragma omp parallel for reduction(+ : acc) schedule(static)
#endif
    for (int64_t i = 0; i < n; ++i) {
        double prod = x[i] * y[i];
        acc = acc + prod;
    }
    return acc;
}

/* norm2Sq = sum . fmap norm2Sq (SpVector.hs:122); scalar norm2Sq = (**2) (Class.hs:407) */
double orc_norm2sq(int64_t n, const double *x) {
    double acc = 0.0;
#ifdef ORC_OMP
#pragma omp parallel for reduction(+ : acc) schedule(static)
#endif
    for (int64_t i = 0; i < n; ++i) {
        double sq = pow(x[i], 2.0);
        acc = acc + sq;
    }
    return acc;
}

/* norm2 c = sqrt (norm2Sq c)  (SpVector.hs:128) */
double orc_norm2(int64_t n, const double *x) { return sqrt(orc_norm2sq(n, x)); }

/* (^+^) = liftU2 (+)  (SpVector.hs:107-108) on dense operands */
void orc_add(int64_t n, const double *x, const double *y, double *out) {
#ifdef ORC_OMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < n; ++i) out[i] = x[i] + y[i];
}
/* x ^-^ y = x ^+^ negateV y  (Class.hs:68-69) */
void orc_sub(int64_t n, const double *x, const double *y, double *out) {
#ifdef ORC_OMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < n; ++i) out[i] = x[i] + (-y[i]);
}
/* n .* v = fmap (n *) v  (SpVector.hs:112-114) */
void orc_scale(int64_t n, double a, const double *x, double *out) {
#ifdef ORC_OMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < n; ++i) out[i] = a * x[i];
}

/* TIMING BUILD ONLY (liboracle_omp.so): copies whose pages are first touched by the thread that will read them in the
 * row-parallel loops above (schedule(static) over rows / elements), so that a multi-socket host serves every thread from
 * its own NUMA node.  Plain copies in the serial build. */
void orc_par_copy_csr(int64_t m, const int64_t *rowptr, const int64_t *colidx, const double *val, int64_t *rowptr2,
                      int64_t *colidx2, double *val2) {
#ifdef ORC_OMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < m; ++i) {
        rowptr2[i] = rowptr[i];
        for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            colidx2[k] = colidx[k];
            val2[k] = val[k];
        }
    }
    rowptr2[m] = rowptr[m];
}
void orc_par_copy_f64(int64_t n, const double *src, double *dst) {
#ifdef ORC_OMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < n; ++i) dst[i] = src[i];
}

static double *vnew(int64_t n) { return (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1)); }
/* Temporaries of the step functions.  The faithful single-thread build allocates and frees them per step, as it always did.
 * The OpenMP TIMING build (liboracle_omp.so, "fair CPU" leg of bench.py, never used for parity) keeps them: five 80 MB
 * malloc / free pairs per step meant mmap + first-touch page faults inside every timed iteration.  New buffers
 * are first-touched row-parallel so that their pages sit with the threads that work on them. */
#ifdef ORC_OMP
#define ORC_TMP_SLOTS 8
static double *tmp_pool[ORC_TMP_SLOTS];
static int64_t tmp_pool_n[ORC_TMP_SLOTS];
static double *tnew(int64_t n) {
    for (int i = 0; i < ORC_TMP_SLOTS; ++i)
        if (tmp_pool[i] && tmp_pool_n[i] == n) { double *p = tmp_pool[i]; tmp_pool[i] = NULL; return p; }
    double *p = vnew(n);
    if (p) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) p[i] = 0.0;
    }
    return p;
}
static void tfree(double *p, int64_t n) {
    if (!p) return;
    for (int i = 0; i < ORC_TMP_SLOTS; ++i)
        if (!tmp_pool[i]) { tmp_pool[i] = p; tmp_pool_n[i] = n; return; }
    free(p);
}
#else
static double *tnew(int64_t n) { return vnew(n); }
static void tfree(double *p, int64_t n) { (void)n; free(p); }
#endif

/* ---------------------------------------------------------------- A5: BiCGSTAB */

/* bicgsInit aa b x0 = BICGSTAB x0 r0 r0, r0 = b ^-^ (aa #> x0)   (Sparse.hs:962-965) */
void orc_bicgstab_init(const orc_csr *A, const double *b, const double *x0, double *x, double *r,
                       double *p) {
    int64_t n = A->m;
    double *ax = vnew(n);
    orc_spmv(A->m, A->rowptr, A->colidx, A->val, x0, ax);
    orc_sub(n, b, ax, r);
    memcpy(p, r, sizeof(double) * (size_t)n);
    if (x != x0) memcpy(x, x0, sizeof(double) * (size_t)n);
    free(ax);
}

/* bicgstabStep (Sparse.hs:972-981) */
int orc_bicgstab_step(const orc_csr *A, const double *r0hat, double *x, double *r, double *p) {
    return orc_bicgstab_step_ex(A, r0hat, x, r, p, 0);
}

/* rho_identity = 0: the reference's step, term by term.  rho_identity = 1 is NOT a reference formula: the numerator of beta
 * is evaluated as (s . r0hat) - omega (aas . r0hat) -- what the product's fused K4+K5 sweep does --
 * with the same left-fold sums.  The tests run both to measure where the two formulas part on a given system. */
int orc_bicgstab_step_ex(const orc_csr *A, const double *r0hat, double *x, double *r, double *p, int rho_identity) {
    int64_t n = A->m;
    double *aap = tnew(n), *s = tnew(n), *aas = tnew(n), *t = tnew(n), *t2 = tnew(n);
    if (!aap || !s || !aas || !t || !t2) return ORC_ERR_ALLOC;
    orc_spmv(A->m, A->rowptr, A->colidx, A->val, p, aap);          /* aap = aa #> p          */
    double rr0 = orc_dot(n, r, r0hat);
    double alpha = rr0 / orc_dot(n, aap, r0hat);                    /* alphaj                 */
    orc_scale(n, alpha, aap, t);
    orc_sub(n, r, t, s);                                            /* sj = r ^-^ (alpha.*aap)*/
    orc_spmv(A->m, A->rowptr, A->colidx, A->val, s, aas);           /* aasj = aa #> sj        */
    double omega = orc_dot(n, aas, s) / orc_dot(n, aas, aas);       /* omegaj                 */
    orc_scale(n, alpha, p, t);
    orc_add(n, x, t, x);                                            /* (x ^+^ alpha.*p)       */
    orc_scale(n, omega, s, t);
    orc_add(n, x, t, x);                                            /*   ^+^ omega.*sj        */
    double *rnew = t2;
    orc_scale(n, omega, aas, t);
    orc_sub(n, s, t, rnew);                                         /* rj1 = sj ^-^ omega.*aas*/
    double rho1 = rho_identity ? orc_dot(n, s, r0hat) - omega * orc_dot(n, aas, r0hat) : orc_dot(n, rnew, r0hat);
    double beta = rho1 / rr0 * alpha / omega;                       /* ((a/b)*alpha)/omega    */
    orc_scale(n, omega, aap, t);
    orc_sub(n, p, t, t);                                            /* p ^-^ omega.*aap       */
    orc_scale(n, beta, t, t);
    orc_add(n, rnew, t, p);                                         /* pj1                    */
    orc_par_copy_f64(n, rnew, r);
    tfree(aap, n); tfree(s, n); tfree(aas, n); tfree(t, n); tfree(t2, n);
    return ORC_OK;
}

/* ---------------------------------------------------------------- A6: CGS */

/* cgsInit aa b x0 = CGS x0 r0 r0 r0   (Sparse.hs:921-924) */
void orc_cgs_init(const orc_csr *A, const double *b, const double *x0, double *x, double *r,
                  double *p, double *u) {
    int64_t n = A->m;
    orc_bicgstab_init(A, b, x0, x, r, p);
    memcpy(u, r, sizeof(double) * (size_t)n);
}

/* cgsStep (Sparse.hs:928-939) */
int orc_cgs_step(const orc_csr *A, const double *rhat, double *x, double *r, double *p, double *u) {
    int64_t n = A->m;
    double *aap = tnew(n), *q = tnew(n), *uq = tnew(n), *t = tnew(n), *auq = tnew(n);
    if (!aap || !q || !uq || !t || !auq) return ORC_ERR_ALLOC;
    orc_spmv(A->m, A->rowptr, A->colidx, A->val, p, aap);           /* aap = aa #> p          */
    double rr = orc_dot(n, r, rhat);
    double alpha = rr / orc_dot(n, aap, rhat);                      /* alphaj                 */
    orc_scale(n, alpha, aap, t);
    orc_sub(n, u, t, q);                                            /* q = u ^-^ alpha.*aap   */
    orc_add(n, u, q, uq);                                           /* u ^+^ q                */
    orc_scale(n, alpha, uq, t);
    orc_add(n, x, t, x);                                            /* xj1                    */
    orc_spmv(A->m, A->rowptr, A->colidx, A->val, uq, auq);          /* aa #> (u ^+^ q)        */
    orc_scale(n, alpha, auq, t);
    orc_sub(n, r, t, r);                                            /* rj1 (in place)         */
    double beta = orc_dot(n, r, rhat) / rr;                         /* betaj                  */
    orc_scale(n, beta, q, t);
    orc_add(n, r, t, u);                                            /* uj1 = rj1 ^+^ beta.*q  */
    orc_scale(n, beta, p, t);
    orc_add(n, q, t, t);                                            /* q ^+^ beta.*p          */
    orc_scale(n, beta, t, t);
    orc_add(n, u, t, p);                                            /* pj1                    */
    tfree(aap, n); tfree(q, n); tfree(uq, n); tfree(t, n); tfree(auq, n);
    return ORC_OK;
}

/* ---------------------------------------------------------------- A7: CGNE */

/* cgneInit (Sparse.hs:864-868): r0 = b ^-^ (aa #> x0); p0 = transposeSM aa #> r0 */
void orc_cgne_init(const orc_csr *A, const orc_csr *At, const double *b, const double *x0,
                   double *x, double *r, double *p) {
    int64_t n = A->m;
    double *ax = vnew(n);
    orc_spmv(A->m, A->rowptr, A->colidx, A->val, x0, ax);
    orc_sub(n, b, ax, r);
    orc_spmv(At->m, At->rowptr, At->colidx, At->val, r, p);
    if (x != x0) memcpy(x, x0, sizeof(double) * (size_t)A->n);
    free(ax);
}

/* cgneStep (Sparse.hs:870-878) */
int orc_cgne_step(const orc_csr *A, const orc_csr *At, double *x, double *r, double *p) {
    int64_t m = A->m, n = A->n;
    double *t = vnew(n > m ? n : m), *ap = vnew(m), *atr = vnew(n);
    if (!t || !ap || !atr) return ORC_ERR_ALLOC;
    double rr = orc_dot(m, r, r);
    double alpha = rr / orc_dot(n, p, p);                           /* alphai                 */
    orc_scale(n, alpha, p, t);
    orc_add(n, x, t, x);                                            /* x1                     */
    orc_spmv(A->m, A->rowptr, A->colidx, A->val, p, ap);
    orc_scale(m, alpha, ap, t);
    orc_sub(m, r, t, r);                                            /* r1                     */
    double beta = orc_dot(m, r, r) / rr;                            /* beta                   */
    orc_spmv(At->m, At->rowptr, At->colidx, At->val, r, atr);       /* transpose aa #> r1     */
    orc_scale(n, beta, p, t);
    orc_add(n, atr, t, p);                                          /* p1                     */
    free(t); free(ap); free(atr);
    return ORC_OK;
}

/* ---------------------------------------------------------------- (f).3: BCG (extension) */

/* bcgInit / bcgStep restate the COMMENTED code of Sparse.hs:886-909 (dead in the reference: `linSolve0 BCG_` throws, :1031,
 * and orc_linsolve0 keeps doing so).  PARITY UNPINNED BY THE REFERENCE: no test, golden vector or call site of it exists there;
 * this restatement is pinned only to the commented formulas, with the two initialisers the comment leaves out completed the
 * only way the step can use them (p0 = r0, p0hat = r0hat).
 * bcgInit aa b x0 = BCG x0 r0 r0hat p0 p0hat where r0 = b ^-^ (aa #> x0) ; r0hat = r0   (:891-897) */
void orc_bcg_init(const orc_csr *A, const double *b, const double *x0, double *x, double *r, double *rhat, double *p, double *phat) {
    int64_t n = A->m;
    orc_bicgstab_init(A, b, x0, x, r, p);          /* x = x0, r = b ^-^ (aa #> x0), p = r  */
    memcpy(rhat, r, sizeof(double) * (size_t)n);
    memcpy(phat, r, sizeof(double) * (size_t)n);
}

/* bcgStep aa (BCG x r rhat p phat) (Sparse.hs:899-909); At = transpose aa, built once by the caller */
int orc_bcg_step(const orc_csr *A, const orc_csr *At, double *x, double *r, double *rhat, double *p, double *phat) {
    int64_t n = A->m;
    double *aap = tnew(n), *atp = tnew(n), *t = tnew(n);
    if (!aap || !atp || !t) return ORC_ERR_ALLOC;
    orc_spmv(A->m, A->rowptr, A->colidx, A->val, p, aap);           /* aap = aa #> p                     */
    double rr = orc_dot(n, r, rhat);
    double alpha = rr / orc_dot(n, aap, phat);                      /* alpha                             */
    orc_scale(n, alpha, p, t);
    orc_add(n, x, t, x);                                            /* x1 = x ^+^ alpha.*p               */
    orc_scale(n, alpha, aap, t);
    orc_sub(n, r, t, r);                                            /* r1 = r ^-^ alpha.*aap             */
    orc_spmv(At->m, At->rowptr, At->colidx, At->val, phat, atp);    /* transpose aa #> phat              */
    orc_scale(n, alpha, atp, t);
    orc_sub(n, rhat, t, rhat);                                      /* rhat1 = rhat ^-^ alpha.*(...)     */
    double beta = orc_dot(n, r, rhat) / rr;                         /* beta = (r1.rhat1) / (r.rhat)      */
    orc_scale(n, beta, p, t);
    orc_add(n, r, t, p);                                            /* p1 = r1 ^+^ beta.*p               */
    orc_scale(n, beta, phat, t);
    orc_add(n, rhat, t, phat);                                      /* phat1 = rhat1 ^+^ beta.*phat      */
    tfree(aap, n); tfree(atp, n); tfree(t, n);
    return ORC_OK;
}

/* ---------------------------------------------------------------- A8: linSolve0 */

/* trueResidualNorm x = norm2 ((aa #> x) ^-^ b)   (Sparse.hs:1041) */
static double true_resnorm(const orc_csr *A, const double *x, const double *b, double *w) {
    orc_spmv(A->m, A->rowptr, A->colidx, A->val, x, w);
    orc_sub(A->m, w, b, w);
    return orc_norm2(A->m, w);
}

/* linSolve0 (Sparse.hs:1016-1072) */
int orc_linsolve0(int method, const orc_csr *A, int64_t nb, const double *b, const double *x0,
                  double *x_out, int64_t *iters_out, double *resnorm_out, double *r0norm_out) {
    int64_t m = A->m, n = A->n;
    *iters_out = 0; *resnorm_out = NAN; *r0norm_out = NAN;
    if (m != nb) return ORC_ERR_DIM;                                /* :1022                  */
    if (orc_is_diagonal(m, A->rowptr, A->colidx)) {                 /* :1024-1025             */
        /* reciprocal aa #> b: row i = 0 + recip(a_ii) * b_i (Class.hs:174, Common.hs:247) */
        for (int64_t i = 0; i < m; ++i) {
            double prod = (1.0 / A->val[A->rowptr[i]]) * b[i];
            x_out[i] = 0.0 + prod;
        }
        return ORC_OK;
    }
    if (method != ORC_BICGSTAB && method != ORC_CGS && method != ORC_CGNE)
        return ORC_ERR_UNSUPPORTED;                                 /* :1031                  */
    int64_t nv = n > m ? n : m;
    double *r0hat = vnew(nv), *x = vnew(nv), *r = vnew(nv), *p = vnew(nv), *u = vnew(nv), *w = vnew(nv);
    int64_t *tp = NULL, *tc = NULL; double *tv = NULL;
    orc_csr At = {0};
    if (!r0hat || !x || !r || !p || !u || !w) return ORC_ERR_ALLOC;
    orc_spmv(m, A->rowptr, A->colidx, A->val, x0, w);
    orc_sub(m, b, w, r0hat);                                        /* r0hat (:1032)          */
    double r0norm = orc_norm2(m, r0hat);                            /* :1033                  */
    double tol = fmax(1e-6, 1e-4 * r0norm);                         /* :1034-1037             */
    *r0norm_out = r0norm;
    const int64_t nits = 200;
    if (method == ORC_BICGSTAB) orc_bicgstab_init(A, b, x0, x, r, p);
    else if (method == ORC_CGS) orc_cgs_init(A, b, x0, x, r, p, u);
    else {
        int64_t nnz = A->rowptr[m];
        tp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
        tc = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz > 0 ? nnz : 1));
        tv = vnew(nnz);
        if (!tp || !tc || !tv) return ORC_ERR_ALLOC;
        orc_csr_transpose(m, n, A->rowptr, A->colidx, A->val, tp, tc, tv);
        At.m = n; At.n = m; At.rowptr = tp; At.colidx = tc; At.val = tv;
        orc_cgne_init(A, &At, b, x0, x, r, p);
    }
    int64_t it = 0;
    for (;;) {                                                      /* runIter (:1043-1052)   */
        if (it >= nits) break;                                      /* silent return at 200   */
        if (method == ORC_BICGSTAB) orc_bicgstab_step(A, r0hat, x, r, p);
        else if (method == ORC_CGS) orc_cgs_step(A, r0hat, x, r, p, u);
        else orc_cgne_step(A, &At, x, r, p);
        double res = true_resnorm(A, x, b, w);
        *resnorm_out = res;
        ++it;
        if (res <= tol) break;                                      /* NaN compares false     */
    }
    *iters_out = it;
    memcpy(x_out, x, sizeof(double) * (size_t)n);
    free(r0hat); free(x); free(r); free(p); free(u); free(w); free(tp); free(tc); free(tv);
    return ORC_OK;
}

/* ---------------------------------------------------------------- A9: Arnoldi */

/* normalize2 v = v ./ norm2 v = (recip (norm2 v)) .* v   (Class.hs:94-95, SpVector.hs:126) */
static void normalize2(int64_t n, const double *v, double *out) {
    double s = 1.0 / orc_norm2(n, v);
    orc_scale(n, s, v, out);
}

/* arnoldi (Sparse.hs:630-667).  Loop state (qv, hh, i, fbreak) starts at i = 1 with two basis
 * vectors; stops when i == kn || fbreak.  Returned k_done = final i (= nmax). */
int orc_arnoldi(const orc_csr *A, int64_t nb, const double *b, int64_t kn, double *Q, double *H,
                int64_t *k_done) {
    int64_t m = A->m, n = A->n;
    if (n != nb) return ORC_ERR_DIM;                                /* :636-637               */
    int64_t ldh = kn + 1;
    for (int64_t k = 0; k < ldh * kn; ++k) H[k] = 0.0;
    double *aq = vnew(m), *acc = vnew(m), *t = vnew(m), *hcol = vnew(kn + 2);
    if (!aq || !acc || !t || !hcol) return ORC_ERR_ALLOC;
    double *q0 = Q, *q1 = Q + n;
    normalize2(n, b, q0);                                           /* q0 = normalize2 b      */
    orc_spmv(m, A->rowptr, A->colidx, A->val, q0, aq);              /* aq0 = aa #> q0         */
    double h11 = orc_dot(n, q0, aq);                                /* h11 = q0 `dot` aq0     */
    orc_scale(n, h11, q0, t);
    orc_sub(n, aq, t, acc);                                         /* q1nn                   */
    double h21 = orc_norm2(n, acc);                                 /* norm2' q1nn            */
    normalize2(n, acc, q1);
    H[0] = h11;
    if (kn >= 1) H[1] = h21;
    int64_t i = 1;
    int fbreak = 0;
    while (!(i == kn || fbreak)) {                                  /* modifyUntil tf (:638)  */
        const double *qi = Q + i * n;                               /* V.last qv              */
        orc_spmv(m, A->rowptr, A->colidx, A->val, qi, aq);          /* aqi                    */
        for (int64_t k = 0; k <= i; ++k) hcol[k] = orc_dot(n, Q + k * n, aq);   /* :655       */
        /* V.foldl' (^+^) zv (zipWith (.*) hhcoli qv): zv is the EMPTY map, so the first
         * union passes h0*q0 through unchanged. */
        orc_scale(n, hcol[0], Q, acc);
        for (int64_t k = 1; k <= i; ++k) {
            orc_scale(n, hcol[k], Q + k * n, t);
            orc_add(n, acc, t, acc);
        }
        orc_sub(n, aq, acc, acc);                                   /* qipnn (:657-658)       */
        double nrm = orc_norm2(n, acc);                             /* qipnorm                */
        normalize2(n, acc, Q + (i + 1) * n);                        /* qip                    */
        for (int64_t k = 0; k <= i; ++k) H[i * ldh + k] = hcol[k];
        H[i * ldh + i + 1] = nrm;
        fbreak = fabs(nrm) <= 1e-12;                                /* nearZero (Eps.hs:41-42)*/
        ++i;
    }
    *k_done = i;
    free(aq); free(acc); free(t); free(hcol);
    return ORC_OK;
}

/* ---------------------------------------------------------------- A10: GMRES(m) */

static int is_nz_val(double v) { return !(fabs(v) <= 1e-12); }      /* isNz = not . nearZero (Eps.hs:41-42) */
int orc_tri_upper_solve(const orc_csr *T, const double *b, double *x, int64_t *bad_row);

/* qr (Sparse.hs:306-331) with givens (:253-283) and givensCoef / hypot (:286-295), on a dense m x n array (m >= n, column-major,
 * leading dimension m) that carries the IntMap's STRUCTURE next to the values: st[] marks the stored entries.
 *   gminit = (eye m, mm, subdiagIndicesSM mm): the (i, j), i > j, stored in the INPUT, rows ascending, columns ascending inside a
 *            row (IntMap2.hs:124-131) -- fixed up front, fill-in is never revisited;
 *   per (i, j): nearZero (m @@ (i, j)) -> skip (:258); i' = the smallest row /= i whose FIRST stored column is j (:270-279), none ->
 *            skip; (c, s, r) = givensCoef (m @@ (i', j)) (m @@ (i, j)) = (a / r, b / r, r), r = sqrt (a * a + b * b) -- sqrt, not libm hypot;
 *            G = eye with (i, i) = c, (i, j) = - s, (j, i) = s, (j, j) = c (:263-267: rows i and j -- the COLUMN index j, not i': the
 *            reference's own choice; on a Hessenberg matrix i' = j anyway);
 *            qmatt' = g #~# qmatt, m' = g #~# m with #~# = sparsifySM . (##) (SpMatrix.hs:820-824): every entry of the two touched rows is
 *            the ascending-k left fold from 0 of separately rounded products, then entries with |x| <= 1e-12 are dropped (also from
 *            the untouched rows: sparsifySM filters the whole product);
 *   result (transpose qt, r): Qt is returned as is (m x m, column-major), R in place of the input copy.
 * The structure matters because candidateRows' looks at stored keys, not values. */
static void qr_apply(int64_t m, int64_t ncols, int64_t i, int64_t j, double c, double s, double *M, char *st) {
    for (int64_t col = 0; col < ncols; ++col) {
        double *ci = M + col * m + i, *cj = M + col * m + j;
        char *si = st + col * m + i, *sj = st + col * m + j;
        /* row i of G: keys {j, i} ascending; row j of G: keys {j, i} ascending (j < i); intersections with the stored keys of column `col` */
        double ri = 0.0, rj = 0.0;
        if (*sj) { double p1 = (-s) * *cj; ri = ri + p1; double p2 = c * *cj; rj = rj + p2; }
        if (*si) { double p1 = c * *ci; ri = ri + p1; double p2 = s * *ci; rj = rj + p2; }
        *ci = ri;
        *cj = rj;
        *si = 1;   /* (##) yields an explicit entry for every (row key of G) x (column key of M) pair ... */
        *sj = 1;
    }
    for (int64_t k = 0; k < m * ncols; ++k) {   /* ... and sparsifySM drops what fails isNz, everywhere */
        if (st[k] && !is_nz_val(M[k])) { st[k] = 0; M[k] = 0.0; }
    }
}
int orc_qr_dense(int64_t m, int64_t n, const double *A, const char *stored, double *Qt, double *R) {
    if (m < n) return ORC_ERR_DIM;                                     /* "Givens: matrix must have rows >= cols" (:256) */
    char *sr = (char *)malloc((size_t)(m * n > 0 ? m * n : 1)), *sq = (char *)malloc((size_t)(m * m > 0 ? m * m : 1));
    int64_t *ii = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m * n > 0 ? m * n : 1)), *jj = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m * n > 0 ? m * n : 1));
    if (!sr || !sq || !ii || !jj) return ORC_ERR_ALLOC;
    for (int64_t k = 0; k < m * n; ++k) { R[k] = A[k]; sr[k] = stored ? stored[k] : (A[k] != 0.0); if (!sr[k]) R[k] = 0.0; }
    for (int64_t k = 0; k < m * m; ++k) { Qt[k] = 0.0; sq[k] = 0; }
    for (int64_t d = 0; d < m; ++d) { Qt[d * m + d] = 1.0; sq[d * m + d] = 1; }
    int64_t cnt = 0;
    for (int64_t i = 0; i < m; ++i)                                    /* subdiagIndices: rows ascending, columns ascending */
        for (int64_t j = 0; j < n && j < i; ++j)
            if (sr[j * m + i]) { ii[cnt] = i; jj[cnt] = j; ++cnt; }
    for (int64_t t = 0; t < cnt; ++t) {
        const int64_t i = ii[t], j = jj[t];
        const double bval = sr[j * m + i] ? R[j * m + i] : 0.0;        /* aa @@ (i, j): 0 when absent */
        if (!is_nz_val(bval)) continue;
        int64_t ip = -1;
        for (int64_t r = 0; r < m && ip < 0; ++r) {                    /* candidateRows': head of the ascending keys */
            if (r == i || !sr[j * m + r]) continue;
            int first = 1;
            for (int64_t cc = 0; cc < j && first; ++cc) first = !sr[cc * m + r];
            if (first) ip = r;
        }
        if (ip < 0) continue;
        const double a = R[j * m + ip];
        const double a2 = a * a, b2 = bval * bval;                     /* mag2 i = i * conj i */
        const double rr = sqrt(a2 + b2);
        const double c = a / rr, s = bval / rr;
        qr_apply(m, m, i, j, c, s, Qt, sq);
        qr_apply(m, n, i, j, c, s, R, sr);
    }
    free(sr); free(sq); free(ii); free(jj);
    return ORC_OK;
}

/* The least-squares step of the commented gmres (Sparse.hs:837-848), in the reference's own terms:
 *   b' = norm2' b .* ei mp1 1 ; (qh, rh) <- qr ha ; rhs' = takeSV (dim b' - 1) (transpose qh #> b') ;
 *   rh' = takeRows (nrows rh - 1) rh ; yhat <- triUpperSolve rh' rhs'
 * with ha the (k+1) x k Hessenberg matrix of arnoldi (stored entries: rows 0 .. j+1 of column j, as fromCols of the hhcoli builds
 * it).  Independent of the product's Givens sweep (csrc/sla_solvers.cpp: hessenberg_lsq), which it is compared with. */
static int hessenberg_lsq(int64_t k, int64_t ldh, const double *H, double beta, double *y) {
    const int64_t m = k + 1;
    double *Hk = vnew(m * k), *Qt = vnew(m * m), *R = vnew(m * k), *rhs = vnew(k), *tv = vnew(k * k + 1);
    char *st = (char *)malloc((size_t)(m * k > 0 ? m * k : 1));
    int64_t *tp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(k + 1)), *tc = (int64_t *)malloc(sizeof(int64_t) * (size_t)(k * k + 1));
    if (!Hk || !Qt || !R || !rhs || !tv || !st || !tp || !tc) return ORC_ERR_ALLOC;
    for (int64_t j = 0; j < k; ++j)
        for (int64_t i = 0; i < m; ++i) { Hk[j * m + i] = H[j * ldh + i]; st[j * m + i] = i <= j + 1; }
    int rc = orc_qr_dense(m, k, Hk, st, Qt, R);
    if (rc == ORC_OK) {
        /* transpose qh = Qt ; (Qt #> b')_i = 0 + Qt(i, 0) * beta where Qt(i, 0) is stored (b' holds the single key 0) */
        for (int64_t i = 0; i < k; ++i) { double p = Qt[0 * m + i] * beta; rhs[i] = 0.0 + p; }
        orc_csr T;                                                   /* rh' = the first k rows of rh, its isNz entries */
        int64_t nz = 0;
        tp[0] = 0;
        for (int64_t i = 0; i < k; ++i) {
            for (int64_t j = 0; j < k; ++j)
                if (is_nz_val(R[j * m + i])) { tc[nz] = j; tv[nz] = R[j * m + i]; ++nz; }
            tp[i + 1] = nz;
        }
        T.m = k; T.n = k; T.rowptr = tp; T.colidx = tc; T.val = tv;
        int64_t bad = -1;
        rc = orc_tri_upper_solve(&T, rhs, y, &bad);
    }
    free(Hk); free(Qt); free(R); free(rhs); free(tv); free(st); free(tp); free(tc);
    return rc;
}

int orc_gmres(const orc_csr *A, int64_t nb, const double *b, const double *x0, int64_t restart,
              int64_t max_restarts, double tol_abs, double tol_rel, double *x_out,
              int64_t *iters_out, double *resnorm_out, double *r0norm_out) {
    int64_t m = A->m, n = A->n;
    *iters_out = 0; *resnorm_out = NAN; *r0norm_out = NAN;
    if (m != nb || m != n) return ORC_ERR_DIM;
    double *x = vnew(n), *r = vnew(n), *w = vnew(n), *Q = vnew(n * (restart + 1)),
           *H = vnew((restart + 1) * restart), *y = vnew(restart), *t = vnew(n);
    if (!x || !r || !w || !Q || !H || !y || !t) return ORC_ERR_ALLOC;
    memcpy(x, x0, sizeof(double) * (size_t)n);
    double tol = 0.0;
    for (int64_t cyc = 0; cyc <= max_restarts; ++cyc) {
        orc_spmv(m, A->rowptr, A->colidx, A->val, x, w);
        orc_sub(n, b, w, r);
        double beta = orc_norm2(n, r);
        if (cyc == 0) { *r0norm_out = beta; tol = fmax(tol_abs, tol_rel * beta); }
        *resnorm_out = beta;
        if (beta <= tol || cyc == max_restarts) break;
        int64_t k = 0;
        orc_arnoldi(A, n, r, restart, Q, H, &k);
        if (hessenberg_lsq(k, restart + 1, H, beta, y) != ORC_OK) break;   /* NeedsPivoting in triUpperSolve: the reference would throw */
        for (int64_t j = 0; j < k; ++j) {                           /* x += Q[:, :k] y        */
            orc_scale(n, y[j], Q + j * n, t);
            orc_add(n, x, t, x);
        }
        *iters_out += k;
    }
    memcpy(x_out, x, sizeof(double) * (size_t)n);
    free(x); free(r); free(w); free(Q); free(H); free(y); free(t);
    return ORC_OK;
}

/* ---------------------------------------------------------------- A11: (##) */

/* matMatUnsafeWith transposeIM2 (SpMatrix.hs:808-811): for every row key of A and every column
 * key of B: sum (liftI2 (*) colB rowA), ascending k; explicit zeros are kept. */
int orc_matmat(const orc_csr *A, const orc_csr *B, int64_t *c_rowptr, int64_t *c_colidx,
               double *c_val, int64_t cap, int64_t *c_nnz) {
    if (A->n != B->m) return ORC_ERR_DIM;                           /* matMatCheck (:795)     */
    int64_t bnnz = B->rowptr[B->m];
    int64_t *tp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(B->n + 1));
    int64_t *tc = (int64_t *)malloc(sizeof(int64_t) * (size_t)(bnnz > 0 ? bnnz : 1));
    double *tv = vnew(bnnz);
    if (!tp || !tc || !tv) return ORC_ERR_ALLOC;
    orc_csr_transpose(B->m, B->n, B->rowptr, B->colidx, B->val, tp, tc, tv);
    int64_t out = 0;
    c_rowptr[0] = 0;
    for (int64_t i = 0; i < A->m; ++i) {
        int64_t a0 = A->rowptr[i], a1 = A->rowptr[i + 1];
        if (a1 > a0) {                                              /* row key present in A   */
            for (int64_t j = 0; j < B->n; ++j) {
                int64_t b0 = tp[j], b1 = tp[j + 1];
                if (b1 == b0) continue;                             /* column key absent in B */
                double acc = 0.0;
                int64_t ka = a0, kb = b0;
                while (ka < a1 && kb < b1) {                        /* intersectionWith (*)   */
                    if (A->colidx[ka] < tc[kb]) ++ka;
                    else if (A->colidx[ka] > tc[kb]) ++kb;
                    else { double prod = tv[kb] * A->val[ka]; acc = acc + prod; ++ka; ++kb; }
                }
                if (out >= cap) { free(tp); free(tc); free(tv); return ORC_ERR_ALLOC; }
                c_colidx[out] = j; c_val[out] = acc; ++out;
            }
        }
        c_rowptr[i + 1] = out;
    }
    *c_nnz = out;
    free(tp); free(tc); free(tv);
    return ORC_OK;
}

/* ---------------------------------------------------------------------------------------------
 * triangular solves and the SSOR factors
 * ------------------------------------------------------------------------------------------- */
static double tri_diag(const orc_csr *T, int64_t i) {                /* ll @@ (i, i): 0 when absent */
    for (int64_t k = T->rowptr[i]; k < T->rowptr[i + 1]; ++k)
        if (T->colidx[k] == i) return T->val[k];
    return 0.0;
}
static int is_nz(double v) { return !(fabs(v) <= 1e-12); }          /* isNz = not . nearZero        */

/* triLowerSolve, Sparse.hs:750-776: w_i = (b_i - subrow(i, 0..i-1) `dot` w) / l_ii, i ascending */
int orc_tri_lower_solve(const orc_csr *T, const double *b, double *x, int64_t *bad_row) {
    if (T->m != T->n) return ORC_ERR_DIM;
    for (int64_t i = 0; i < T->m; ++i) {
        double lii = tri_diag(T, i);
        if (!is_nz(lii)) { if (bad_row) *bad_row = i; return ORC_ERR_PIVOT; }   /* oops i (:757)   */
        double r = 0.0;
        for (int64_t k = T->rowptr[i]; k < T->rowptr[i + 1]; ++k) {
            int64_t j = T->colidx[k];
            if (j < i) { double prod = T->val[k] * x[j]; r = r + prod; }        /* ascending fold  */
        }
        x[i] = (b[i] - r) / lii;                                                  /* (:760)          */
    }
    for (int64_t i = 0; i < T->m; ++i) if (!is_nz(x[i])) x[i] = 0.0;             /* sparsifySV (:776) */
    return ORC_OK;
}

/* triUpperSolve, Sparse.hs:784-811: x_i = (w_i - subrow(i, i+1..n-1) `dot` x) / u_ii, i descending */
int orc_tri_upper_solve(const orc_csr *T, const double *b, double *x, int64_t *bad_row) {
    if (T->m != T->n) return ORC_ERR_DIM;
    for (int64_t i = T->m - 1; i >= 0; --i) {
        double uii = tri_diag(T, i);
        if (!is_nz(uii)) { if (bad_row) *bad_row = i; return ORC_ERR_PIVOT; }
        double r = 0.0;
        for (int64_t k = T->rowptr[i]; k < T->rowptr[i + 1]; ++k) {
            int64_t j = T->colidx[k];
            if (j > i) { double prod = T->val[k] * x[j]; r = r + prod; }
        }
        x[i] = (b[i] - r) / uii;
    }
    for (int64_t i = 0; i < T->m; ++i) if (!is_nz(x[i])) x[i] = 0.0;
    return ORC_OK;
}

/* mSsorPre, Sparse.hs:712-720 */
int orc_ssor_pre(const orc_csr *A, double omega, int64_t *l_rowptr, int64_t *l_colidx, double *l_val,
                 int64_t *r_rowptr, int64_t *r_colidx, double *r_val) {
    if (A->m != A->n) return ORC_ERR_DIM;
    int64_t n = A->m, lo = 0, ro = 0;
    double *rd = vnew(n);                       /* reciprocal d: recip of the STORED diagonal entries */
    char *has = (char *)calloc((size_t)(n > 0 ? n : 1), 1);
    if (!rd || !has) return ORC_ERR_ALLOC;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k)
            if (A->colidx[k] == i) { rd[i] = 1.0 / A->val[k]; has[i] = 1; }
    l_rowptr[0] = 0;
    r_rowptr[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        /* row i of (eye n ^-^ scale omega e): -(omega * e_ij) for j < i (x ^-^ y = x ^+^ negated y), then 1 at j = i;
         * times column j of reciprocal d = a single entry rd[j] at (j, j) when the diagonal entry is stored */
        for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k) {
            int64_t j = A->colidx[k];
            if (j < i && has[j]) {
                double m = -(omega * A->val[k]);
                double prod = rd[j] * m, acc = 0.0;                 /* sum of the one-term intersection */
                acc = acc + prod;
                l_colidx[lo] = j; l_val[lo] = acc; ++lo;
            }
        }
        if (has[i]) { double prod = rd[i] * 1.0, acc = 0.0; acc = acc + prod; l_colidx[lo] = i; l_val[lo] = acc; ++lo; }
        l_rowptr[i + 1] = lo;
        /* r = d ^-^ scale omega f: d_ii at j = i, -(omega * f_ij) for j > i */
        for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k) {
            int64_t j = A->colidx[k];
            if (j == i) { r_colidx[ro] = j; r_val[ro] = A->val[k]; ++ro; }
            else if (j > i) { r_colidx[ro] = j; r_val[ro] = -(omega * A->val[k]); ++ro; }
        }
        r_rowptr[i + 1] = ro;
    }
    free(rd);
    free(has);
    return ORC_OK;
}

/* ---------------------------------------------------------------------------------------------
 * lu (Doolittle, Sparse.hs:488-527) and ilu0Pre (:696-706) -- the reference's definition of "ILU(0)" is
 * the COMPLETE factorisation filtered to A's pattern afterwards.  Dense restatement (n x n value + presence arrays; toy
 * sizes only, like the reference's own IntMap version), following the reference's loop order literally:
 *   luInit : L = eye n with column 0 = a_i0 / u00 for the STORED a_i0, i >= 1 (unfiltered); U = row 0 of A (unfiltered)
 *   i = 1 .. n-1:  row i of U: u_ij = a_ij - contractSub l u i j (i-1), j = i .. n-1, kept when isNz        (uUpd)
 *                  col i of L: l_ri = (a_ri - contractSub l u r i (r-1)) / u_ii, r = i+1 .. n-1, kept when isNz (lUpd);
 *                  not (isNz u_ii) with rows left to solve => NeedsPivoting "solveForLij" "U(i,i)"
 *   contractSub a b i j n = foldlWithKey' (\acc k x -> if k > n then acc else acc + x * b @@! (k, j)) 0 (row i of a)
 *                           (SpMatrix.hs:857-864): ascending k over the STORED entries of the row, absent b entries read 0.
 * ------------------------------------------------------------------------------------------- */
static double contract_sub(int64_t n, const char *lp, const double *lv, const char *up, const double *uv, int64_t i,
                           int64_t j, int64_t kmax) {
    double acc = 0.0;
    for (int64_t k = 0; k < n; ++k) {
        if (!lp[i * n + k] || k > kmax) continue;
        double b = up[k * n + j] ? uv[k * n + j] : 0.0;
        double prod = lv[i * n + k] * b;
        acc = acc + prod;
    }
    return acc;
}

/* L and U as dense n x n (row-major) value / presence arrays.  Returns ORC_OK, ORC_ERR_DIM (not square) or ORC_ERR_PIVOT. */
int orc_lu_dense(const orc_csr *A, double *lv, char *lp, double *uv, char *up, int64_t *bad) {
    if (A->m != A->n) return ORC_ERR_DIM;
    const int64_t n = A->m;
    double *av = vnew(n * n);
    char *ap = (char *)calloc((size_t)(n * n > 0 ? n * n : 1), 1);
    if (!av || !ap) return ORC_ERR_ALLOC;
    memset(lp, 0, (size_t)(n * n)); memset(up, 0, (size_t)(n * n));
    for (int64_t i = 0; i < n; ++i)
        for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k) { av[i * n + A->colidx[k]] = A->val[k]; ap[i * n + A->colidx[k]] = 1; }
    int rc = ORC_OK;
    /* luInit */
    for (int64_t i = 0; i < n; ++i) { lp[i * n + i] = 1; lv[i * n + i] = 1.0; }
    for (int64_t j = 0; j < n; ++j) if (ap[j]) { up[j] = 1; uv[j] = av[j]; }
    const double u00 = up[0] ? uv[0] : 0.0;
    if (!is_nz(u00)) { if (bad) *bad = 0; rc = ORC_ERR_PIVOT; }
    if (rc == ORC_OK) {
        for (int64_t i = 1; i < n; ++i) if (ap[i * n]) { lp[i * n] = 1; lv[i * n] = av[i * n] / u00; }
        for (int64_t i = 1; i < n && rc == ORC_OK; ++i) {
            for (int64_t j = i; j < n; ++j) {                               /* uUpd */
                double a = ap[i * n + j] ? av[i * n + j] : 0.0;
                double u = a - contract_sub(n, lp, lv, up, uv, i, j, i - 1);
                if (is_nz(u)) { up[i * n + j] = 1; uv[i * n + j] = u; }
            }
            for (int64_t r = i + 1; r < n; ++r) {                           /* lUpd */
                double ujj = up[i * n + i] ? uv[i * n + i] : 0.0;
                if (!is_nz(ujj)) { if (bad) *bad = i; rc = ORC_ERR_PIVOT; break; }
                double a = ap[r * n + i] ? av[r * n + i] : 0.0;
                double l = (a - contract_sub(n, lp, lv, up, uv, r, i, r - 1)) / ujj;
                if (is_nz(l)) { lp[r * n + i] = 1; lv[r * n + i] = l; }
            }
        }
    }
    free(av); free(ap);
    return rc;
}

/* lu (filter == 0) / ilu0Pre (filter != 0: entries kept only where A stores one, `ifilterSM (isJust . lookupSM aa)`) as CSR */
int orc_lu(const orc_csr *A, int filter, int64_t *l_rowptr, int64_t *l_colidx, double *l_val, int64_t *u_rowptr,
           int64_t *u_colidx, double *u_val, int64_t *bad) {
    if (A->m != A->n) return ORC_ERR_DIM;
    const int64_t n = A->m, nn = n * n > 0 ? n * n : 1;
    double *lv = vnew(nn), *uv = vnew(nn);
    char *lp = (char *)malloc((size_t)nn), *up = (char *)malloc((size_t)nn), *ap = (char *)calloc((size_t)nn, 1);
    if (!lv || !uv || !lp || !up || !ap) return ORC_ERR_ALLOC;
    int rc = orc_lu_dense(A, lv, lp, uv, up, bad);
    for (int64_t i = 0; i < n; ++i)
        for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k) ap[i * n + A->colidx[k]] = 1;
    int64_t lo = 0, uo = 0;
    l_rowptr[0] = u_rowptr[0] = 0;
    for (int64_t i = 0; i < n && rc == ORC_OK; ++i) {
        for (int64_t j = 0; j < n; ++j) {
            if (filter && !ap[i * n + j]) continue;
            if (lp[i * n + j]) { l_colidx[lo] = j; l_val[lo] = lv[i * n + j]; ++lo; }
            if (up[i * n + j]) { u_colidx[uo] = j; u_val[uo] = uv[i * n + j]; ++uo; }
        }
        l_rowptr[i + 1] = lo;
        u_rowptr[i + 1] = uo;
    }
    free(lv); free(uv); free(lp); free(up); free(ap);
    return rc;
}
