/*
 * sla_oracle.h -- CPU restatement ("oracle") of the SpMV / CGS / BiCGSTAB / CGNE /
 * Arnoldi hot path of ocramz/sparse-linear-algebra.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (libsla_hip.so) never
 * links, loads or calls anything in oracle/.
 *
 * Parity status: the reference is Haskell and cannot be compiled in this image (no
 * ghc/cabal/stack), so this restatement is pinned against the reference's own
 * known-answer tests (tests/golden/reference_vectors.json, values copied as data from
 * test/LibSpec.hs, README.md and vector/.../Vector/Utils.hs).  Last-bit summation order is
 * NOT pinned by any reference test (all its exact assertions are on small integers); the
 * order used here (ascending index, strict left fold from 0.0, separate mul/add roundings)
 * follows containers' IntMap traversal order and base-4.18 Foldable.sum.  GMRES has no live
 * reference implementation ("parity unpinned" for orc_gmres as a whole); its least-squares step is the
 * reference's qr + triUpperSolve, each pinned to the reference's own test cases.
 *
 * All indices are int64 (Haskell Int), all values IEEE f64 (Haskell Double).
 * Compile with -O2 -ffp-contract=off (GHC's x86-64 NCG never fuses mul+add).
 */
#ifndef SLA_ORACLE_H
#define SLA_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_OK = 0, ORC_ERR_DIM = 1, ORC_ERR_UNSUPPORTED = 2, ORC_ERR_OOB = 3, ORC_ERR_ALLOC = 6 };
/* LinSolveMethod constructor order, src/Numeric/LinearAlgebra/Sparse.hs:1007-1011 */
enum { ORC_GMRES = 0, ORC_CGNE = 1, ORC_BCG = 2, ORC_CGS = 3, ORC_BICGSTAB = 4 };

/* A0: fromListSM (SpMatrix.hs:205-224, IntMap2.hs:24-28) lowered to the CSR layout of
 * vector/src/Data/Sparse/Internal/CSR.hs:43-50,74-78.  Duplicates: last wins.  Out of bounds
 * index -> ORC_ERR_OOB (reference: `error "insertSpMatrix : index out of bounds"`).
 * rowptr has m+1 entries, colidx/val have capacity nnz; *nnz_out = entries after dedupe. */
int orc_coo_to_csr(int64_t m, int64_t n, int64_t nnz, const int64_t *row, const int64_t *col,
                   const double *val, int64_t *rowptr, int64_t *colidx, double *valout,
                   int64_t *nnz_out);
/* csPtrV (==) n xs, vector/src/Data/Sparse/Internal/Vector/Utils.hs:12-26 */
void orc_cs_ptr(int64_t n, const int64_t *sorted_ix, int64_t len, int64_t *ptr);
/* transposeIM2 (IntMap2.hs:88-89) on CSR: rows of A^T with ascending column order */
int orc_csr_transpose(int64_t m, int64_t n, const int64_t *rowptr, const int64_t *colidx,
                      const double *val, int64_t *t_rowptr, int64_t *t_colidx, double *t_val);
/* isDiagonalSM, SpMatrix.hs:411-415 */
int orc_is_diagonal(int64_t m, const int64_t *rowptr, const int64_t *colidx);

/* A1: (#>) = matVecSD, Common.hs:247-250 + dotu :259-260 */
/* timing build helpers (first-touch copies for the OpenMP leg of bench.py's cpu_baseline; plain copies otherwise) */
void orc_par_copy_csr(int64_t m, const int64_t *rowptr, const int64_t *colidx, const double *val, int64_t *rowptr2,
                      int64_t *colidx2, double *val2);
void orc_par_copy_f64(int64_t n, const double *src, double *dst);
void orc_spmv(int64_t m, const int64_t *rowptr, const int64_t *colidx, const double *val,
              const double *x, double *y);
/* y = A x, every row folded over column panels of 2^shift columns in the visiting order pos[] (NOT the reference's order unless pos is
 * the identity): the product's overlapped all-gather in "arrival" order, restated for bit-exact row checks */
void orc_spmv_panel_order(int64_t m, const int64_t *rowptr, const int64_t *colidx, const double *val, const double *x, double *y,
                          int shift, int64_t npanels, const int32_t *pos);
/* A2: (<.>), SpVector.hs:116-117 */
double orc_dot(int64_t n, const double *x, const double *y);
/* A3: norm2Sq / norm2, SpVector.hs:119-129, scalar norm2Sq = (**2) Class.hs:405-408 */
double orc_norm2sq(int64_t n, const double *x);
double orc_norm2(int64_t n, const double *x);
/* A4: ^+^ / ^-^ / .* (SpVector.hs:107-114): out may alias x or y */
void orc_add(int64_t n, const double *x, const double *y, double *out);
void orc_sub(int64_t n, const double *x, const double *y, double *out);
void orc_scale(int64_t n, double a, const double *x, double *out);

/* CSR matrix view used by the solver restatements */
typedef struct {
    int64_t m, n;
    const int64_t *rowptr, *colidx;
    const double *val;
} orc_csr;

/* A5: bicgsInit / bicgstabStep, Sparse.hs:962-981.  State vectors are caller-owned, length n. */
void orc_bicgstab_init(const orc_csr *A, const double *b, const double *x0, double *x, double *r,
                       double *p);
int orc_bicgstab_step(const orc_csr *A, const double *r0hat, double *x, double *r, double *p);
/* rho_identity = 1: beta's numerator through (s . r0hat) - omega (aas . r0hat) -- the PRODUCT's default formula, not the reference's */
int orc_bicgstab_step_ex(const orc_csr *A, const double *r0hat, double *x, double *r, double *p, int rho_identity);
/* A6: cgsInit / cgsStep, Sparse.hs:921-939 */
void orc_cgs_init(const orc_csr *A, const double *b, const double *x0, double *x, double *r,
                  double *p, double *u);
int orc_cgs_step(const orc_csr *A, const double *rhat, double *x, double *r, double *p, double *u);
/* A7: cgneInit / cgneStep, Sparse.hs:855-878 (At = transpose of A, built once by the caller) */
void orc_cgne_init(const orc_csr *A, const orc_csr *At, const double *b, const double *x0,
                   double *x, double *r, double *p);
int orc_cgne_step(const orc_csr *A, const orc_csr *At, double *x, double *r, double *p);
/* (f).3 extension: bcgInit / bcgStep, the commented code of Sparse.hs:886-909 (parity unpinned by the reference: dead code there) */
void orc_bcg_init(const orc_csr *A, const double *b, const double *x0, double *x, double *r, double *rhat, double *p, double *phat);
int orc_bcg_step(const orc_csr *A, const orc_csr *At, double *x, double *r, double *rhat, double *p, double *phat);

/* A8: linSolve0, Sparse.hs:1016-1072.  Returns ORC_OK / ORC_ERR_DIM / ORC_ERR_UNSUPPORTED.
 * nb = dim b.  iters_out = number of steps taken (200 = silent return), resnorm_out = last true
 * residual norm computed (NaN if none), r0norm_out = ||b - A x0||.  diag shortcut: iters = 0. */
int orc_linsolve0(int method, const orc_csr *A, int64_t nb, const double *b, const double *x0,
                  double *x_out, int64_t *iters_out, double *resnorm_out, double *r0norm_out);

/* A9: arnoldi, Sparse.hs:630-667.  Q column-major n x (kn+1), H column-major (kn+1) x kn with
 * leading dimension kn+1 (zero-filled); *k_done = nmax (number of H columns actually produced). */
int orc_arnoldi(const orc_csr *A, int64_t nb, const double *b, int64_t kn, double *Q, double *H,
                int64_t *k_done);

/* qr (Sparse.hs:306-331; givens :253-283, givensCoef / hypot :286-295) on a dense column-major m x n array (m >= n) with its stored-entry
 * structure: stored[k] != 0 marks the IntMap's keys (NULL: the non-zero values).  Qt (m x m) = the accumulated rotations (the
 * reference returns transpose Qt), R (m x n) = the rotated matrix; both sparsified like the reference's #~# (|x| <= 1e-12 -> absent = 0).
 * Pinned to the reference's own QR cases tm2 / tm4 / tm6 / issueMatrix through checkQr0 (test/MatrixFactorizationsSpec.hs:46-74). */
int orc_qr_dense(int64_t m, int64_t n, const double *A, const char *stored, double *Qt, double *R);

/* A10: restarted GMRES(m) built on orc_arnoldi, each cycle solved as the commented sketch does (Sparse.hs:837-848): qr of the
 * Hessenberg matrix (orc_qr_dense), rhs' = the leading entries of transpose qh #> (norm2 r .* e1), triUpperSolve on the leading rows of
 * rh (orc_tri_upper_solve), x = Q y.  The two building blocks are pinned by the reference's tests; their composition is not
 * (the reference's gmres is commented out): PARITY UNPINNED for the composition. */
int orc_gmres(const orc_csr *A, int64_t nb, const double *b, const double *x0, int64_t restart,
              int64_t max_restarts, double tol_abs, double tol_rel, double *x_out,
              int64_t *iters_out, double *resnorm_out, double *r0norm_out);

/* A11: (##) for SpMatrix Double, SpMatrix.hs:787-811: C = A B on the structurally dense
 * index set rows(A) x cols(B) (explicit zeros kept).  Output as CSR; c_colidx/c_val capacity
 * must be >= nrows_with_entries(A) * ncols_with_entries(B). */
int orc_matmat(const orc_csr *A, const orc_csr *B, int64_t *c_rowptr, int64_t *c_colidx,
               double *c_val, int64_t cap, int64_t *c_nnz);

/* triLowerSolve / triUpperSolve (Sparse.hs:750-811).  Forward / backward substitution with
 * the diagonal and the strictly lower (upper) part of T; entries on the other side are ignored
 * (extractSubRow takes columns 0..i-1 / i+1..n-1 only).  A diagonal entry that is missing or |.| <= 1e-12
 * (isNz, Eps.hs:41-42) => ORC_ERR_PIVOT with *bad_row set (the reference throws NeedsPivoting).  The result
 * goes through sparsifySV: entries with |x_i| <= 1e-12 read back as 0.0. */
#define ORC_ERR_PIVOT 5
int orc_tri_lower_solve(const orc_csr *T, const double *b, double *x, int64_t *bad_row);
int orc_tri_upper_solve(const orc_csr *T, const double *b, double *x, int64_t *bad_row);

/* mSsorPre (Sparse.hs:712-720): L = (I - omega E) ## reciprocal D, R = D - omega F with E / D / F the strictly
 * lower / diagonal / strictly upper parts of A (diagPartitions, :673-678).  Both as CSR holding the structurally
 * non-zero entries only ((##) itself keeps explicit zeros over the whole index set; the values at the stored
 * positions are the same).  Capacities: nnz(A) + n each. */
int orc_ssor_pre(const orc_csr *A, double omega, int64_t *l_rowptr, int64_t *l_colidx, double *l_val,
                 int64_t *r_rowptr, int64_t *r_colidx, double *r_val);

#ifdef __cplusplus
}
#endif
/* lu (Sparse.hs:488-527) and ilu0Pre (:696-706): the complete Doolittle factorisation (filter == 0) and its
 * restriction to A's stored positions (filter != 0), restated on dense n x n arrays (toy sizes, like the reference's).  CSR
 * outputs need capacity n * n each.  ORC_ERR_PIVOT + *bad = j when u_jj fails isNz while rows remain to be solved. */
int orc_lu(const orc_csr *A, int filter, int64_t *l_rowptr, int64_t *l_colidx, double *l_val, int64_t *u_rowptr,
           int64_t *u_colidx, double *u_val, int64_t *bad);
int orc_lu_dense(const orc_csr *A, double *lv, char *lp, double *uv, char *up, int64_t *bad);

#endif
