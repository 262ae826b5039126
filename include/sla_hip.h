/*
 * sla_hip.h -- C ABI of libsla_hip.so: the MI355X (gfx950) backend for the SpMV-dominated
 * hot path of ocramz/sparse-linear-algebra (Numeric.LinearAlgebra.Sparse).
 *
 * The reference has no FFI of its own (it is pure Haskell); its extension surface is the
 * typeclass set of src/Numeric/LinearAlgebra/Class.hs with the instances in
 * src/Data/Sparse/{Common,SpVector,SpMatrix}.hs.  Each entry point below names the reference
 * method (file:line, relative to the reference repo) it stands in for; INTEGRATION.md shows the
 * `foreign import ccall` module a maintainer adds on the Haskell side.
 *
 * Conventions
 *   - plain pointers and sizes only; int64_t = Haskell Int, double = Haskell Double;
 *   - every call returns an sla_status; sla_last_error() gives a thread-local message;
 *   - host arrays are borrowed for the duration of the call; handles are owned by the library
 *     and released by the matching *_destroy;
 *   - a context and everything created from it is single-threaded (one caller at a time);
 *   - all device work is enqueued on the context's HIP stream; calls that return values to the
 *     host synchronise, the others may return after enqueue;
 *   - non-convergence is NOT an error (Sparse.hs:1045 returns silently after 200 iterations): it
 *     is reported in sla_solve_info.flags.
 */
#ifndef SLA_HIP_H
#define SLA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    SLA_OK = 0,
    SLA_ERR_DIM_MISMATCH = 1,       /* MatVecSizeMismatchException (Control/Exception/Common.hs:44-51),
                                       `error "matVec : mismatched dimensions"` (Common.hs:250) */
    SLA_ERR_UNSUPPORTED_METHOD = 2, /* IterE "linSolve0" "Only BICGSTAB_, CGS_, and CGNE_ ..." (Sparse.hs:1031) */
    SLA_ERR_OOB = 3,                /* `error "insertSpMatrix : index out of bounds"` (SpMatrix.hs:208) */
    SLA_ERR_HIP = 4,
    SLA_ERR_RCCL = 5,
    SLA_ERR_ALLOC = 6,
    SLA_ERR_INVALID = 7,            /* NULL handle, negative size, size beyond the device index width */
    SLA_ERR_NO_DEVICE = 8,          /* no HIP device: the library has no CPU fallback */
    SLA_ERR_NEEDS_PIVOTING = 9      /* NeedsPivoting "triLowerSolve" "L (i, i)" (Control/Exception/Common.hs:58; Sparse.hs:757, :792) */
} sla_status;

/* LinSolveMethod, constructor order of Sparse.hs:1007-1011 */
typedef enum { SLA_GMRES_ = 0, SLA_CGNE_ = 1, SLA_BCG_ = 2, SLA_CGS_ = 3, SLA_BICGSTAB_ = 4 } sla_method;

/* duplicate policy of sla_csr_from_coo */
typedef enum {
    SLA_DUP_LAST_WINS = 0, /* fromListSM (SpMatrix.hs:218-224, IntMap2.hs:24-28) */
    SLA_DUP_SUM = 1        /* MatrixMarket-style assembly (extension; not a reference behaviour) */
} sla_dup_policy;

typedef struct sla_ctx *sla_ctx_t;       /* one GPU (one rank of a row-sharded job): stream, scratch, RCCL comm */
typedef struct sla_csr *sla_csr_t;       /* device CSR (this rank's row block); immutable after creation */
typedef struct sla_vec *sla_vec_t;       /* device dense f64 vector (this rank's row block) */
typedef struct sla_solver *sla_solver_t; /* CGS / BiCGSTAB / CGNE state record on the device */

/* solver options; NULL = the reference's hard-coded values (Sparse.hs:1034-1037).
 * ABI versioning: both structs open with `struct_size` = sizeof of the struct in the CALLER's header (SLA_SOLVE_OPTS_INIT /
 * SLA_SOLVE_INFO_INIT set it).  The library reads / writes only the members that lie inside it, so a caller built against an
 * older, shorter layout keeps working when members are appended; a struct_size smaller than the first version's layout (or 0:
 * a caller from before the field existed) is refused with SLA_ERR_INVALID instead of being read past its end. */
#define SLA_ABI_VERSION 2
typedef struct {
    int32_t struct_size;   /* sizeof(sla_solve_opts) */
    int32_t max_iters;     /* nits   = 200  */
    double tol_abs;        /* tolAbs = 1e-6 */
    double tol_rel;        /* tolRel = 1e-4 */
    int32_t check_every;   /* host polls the device convergence flag every this many iterations
                              (default 16); the device itself tests the TRUE residual after EVERY
                              iteration exactly as runIter does (Sparse.hs:1043-1052), so the
                              returned iterate is the reference's regardless of this value */
    int32_t true_residual; /* 1 (default) = recompute ||A x - b|| each iteration like the reference;
                              0 = extension: skip the check SpMV and run max_iters steps */
    double *history;       /* sla_linsolve0: HOST buffer (may be NULL) receiving the residual trace -- history[j - 1] = the true residual
                              norm ||A x_j - b|| the device evaluated after iteration j, j = 1 .. info->history_len: what cgsStepDebug
                              (Sparse.hs:942-948) prints per iteration, kept in a device buffer during the solve and downloaded once */
    int32_t history_cap;   /* its capacity in doubles (the trace stops there).  The trace needs true_residual = 1: without the
                              per-iteration residual there is nothing to record and history_len stays 0 */
} sla_solve_opts;
#define SLA_SOLVE_OPTS_INIT {(int32_t)sizeof(sla_solve_opts), 200, 1e-6, 1e-4, 16, 1, 0, 0}

enum { /* sla_solve_info.flags */
    SLA_FLAG_CONVERGED = 1,   /* resnorm <= tol reached */
    SLA_FLAG_MAX_ITERS = 2,   /* returned after max_iters without meeting tol (reference: silent) */
    SLA_FLAG_DIAGONAL = 4,    /* isDiagonalSM shortcut taken (Sparse.hs:1024-1025) */
    SLA_FLAG_BREAKDOWN = 8,   /* Arnoldi: nearZero h_{i+1,i} (Sparse.hs:665-667) */
    SLA_FLAG_NONFINITE = 16,  /* residual became NaN/Inf (reference propagates NaN, no guard) */
    SLA_FLAG_SYNC_TIMEOUT = 32, /* a persistent on-chip step launch gave up waiting for its other workgroups (another job holding CUs): the state record is
                                  unchanged by that launch's unfinished steps only up to the step it stopped in -- treat it as lost */
    SLA_FLAG_RELAXED_ORDER = 64 /* the matrix's (#>) adds a row's products in an order that is not fixed from run to run (sla_csr_props.fold ==
                                  SLA_FOLD_RELAXED: the CU-wide tile form in relaxed order, an OPT-IN -- option tile_relaxed = 1 before the matrix is
                                  created; no default form is order-relaxed): x, iters and resnorm of this solve are within the rounding bound of the
                                  reference's fold but may differ in the last bits next time.  Set by sla_linsolve0 / sla_gmres / sla_linsolve */
};

typedef struct {
    int32_t struct_size; /* IN: sizeof(sla_solve_info) of the caller's header (see above) */
    int32_t iters;   /* solver steps taken */
    int32_t flags;
    double resnorm;  /* last true residual norm ||A x - b||_2 evaluated (NaN if none) */
    double r0norm;   /* ||b - A x0||_2 */
    double tol;      /* max tol_abs (tol_rel * r0norm) */
    int32_t history_len; /* entries written to sla_solve_opts.history (= min (iters, history_cap); 0 without a trace) */
} sla_solve_info;
#define SLA_SOLVE_INFO_INIT {(int32_t)sizeof(sla_solve_info), 0, 0, 0.0, 0.0, 0.0, 0}

/* which state vector sla_solver_get returns: record fields _x/_r/_p/_u (Sparse.hs:919),
 * _xBicgstab/_rBicgstab/_pBicgstab (:959-960), _xCgne/_rCgne/_pCgne (:855-856) */
typedef enum { SLA_STATE_X = 0, SLA_STATE_R = 1, SLA_STATE_P = 2, SLA_STATE_U = 3,
               SLA_STATE_RHAT = 4, SLA_STATE_PHAT = 5 /* BCG records only: _rHatBcg, _pHatBcg (Sparse.hs:886-887) */ } sla_state_field;

/* ---- context ---------------------------------------------------------------------------------- */

/* Single-GPU context on `device_id`.  SLA_ERR_NO_DEVICE when no GPU is visible. */
int sla_ctx_create(int device_id, sla_ctx_t *out);
/* ONE caller, n_gpus devices (the reference's caller is a single Haskell program): the library creates one rank
 * context per device -- RCCL communicator from one unique id, like ncclCommInitAll -- and returns a PARENT context.  Matrices
 * and vectors created from it are given / returned WHOLE (rows split in contiguous blocks over the devices); every call on
 * them runs on all devices at once, one internal host thread per device (the per-rank calls contain collectives, so they are
 * issued concurrently exactly as n processes would).  (#>), (<.>), norm2, the vector algebra, the solver state records,
 * linSolve0, arnoldi, GMRES and (<\>) are available; what is not sharded (##, preconditioner set-up, triangular solves,
 * pre-sharded input) returns SLA_ERR_INVALID.  device_ids == NULL means 0 .. n_gpus-1; a REPEATED device id selects the
 * in-process loopback communicator (test backend: RCCL refuses two ranks on one GPU); n_gpus == 1 is sla_ctx_create. */
int sla_ctx_create_multi(int n_gpus, const int *device_ids, sla_ctx_t *out);
/* One rank of a row-sharded job (one process per GPU).  unique_id = the 128 bytes produced by
 * sla_dist_unique_id on rank 0 and distributed out of band (bench.py uses torch.distributed). */
int sla_dist_unique_id(void *unique_id_128);
int sla_ctx_create_dist(int device_id, int rank, int nranks, const void *unique_id_128, sla_ctx_t *out);
/* TEST BACKEND: rank `rank` of an in-process loopback group -- every rank is a host thread of one process,
 * all on `device_id`; collectives are device copies behind host barriers.  It runs the real row-sharded code
 * path (row offsets, exchange plan, rank-ordered sums, reduce-scatter layout) with nranks > 1 on a 1-GPU box,
 * where RCCL refuses two ranks per GPU.  Contexts that pass the same group_key form one group. */
int sla_ctx_create_loopback(int device_id, int rank, int nranks, int group_key, sla_ctx_t *out);
int sla_ctx_destroy(sla_ctx_t);
int sla_ctx_sync(sla_ctx_t);  /* hipStreamSynchronize of the context stream */
/* Typed entry for the tuning / A-B knobs of DESIGN.md section 4 (the same table the SLA_* environment variables feed when a
 * context is created): name = the lower-case knob name without the SLA_ prefix ("wdia", "tile_shift", "x_exchange", ...),
 * value = its textual value ("0", "17", "window").  Knobs that steer the lowering apply to matrices created AFTERWARDS.  On a
 * multi-device parent the option is set on every rank context.  Unknown name / value out of range => SLA_ERR_INVALID. */
int sla_ctx_set_option(sla_ctx_t, const char *name, const char *value);
int sla_ctx_get_option(sla_ctx_t, const char *name, char *buf, int buflen);
int sla_ctx_rank(sla_ctx_t, int *rank, int *nranks);
/* rows [begin, end) of an m-row matrix / m-vector owned by this rank (contiguous 1-D row blocks) */
int sla_ctx_row_range(sla_ctx_t, int64_t m, int64_t *begin, int64_t *end);
const char *sla_last_error(void);
const char *sla_version(void);
int sla_abi_version(void);   /* SLA_ABI_VERSION the library was built with */

/* ---- A0: SpMatrix -> device CSR ("lower once") ------------------------------------------------ */

/* fromListSM (m,n) triples (SpMatrix.hs:218-224): sort by (row, col), duplicates resolved by
 * `dup_policy`, canonical CSR of vector/src/Data/Sparse/Internal/CSR.hs:43-50,74-78.  Every rank
 * passes the full triple list and keeps its own row block. */
int sla_csr_from_coo(sla_ctx_t, int64_t m, int64_t n, int64_t nnz, const int64_t *row, const int64_t *col,
                     const double *val, int dup_policy, sla_csr_t *out);
/* Already-canonical CSR (ascending columns inside each row, no duplicates).  `rowptr` has m+1 entries. */
int sla_csr_from_csr(sla_ctx_t, int64_t m, int64_t n, const int64_t *rowptr, const int64_t *colidx,
                     const double *val, sla_csr_t *out);
/* Pre-sharded input: this rank's rows [row_begin, row_begin+row_count) of an m x n matrix, with
 * rowptr_local[0] == 0 and GLOBAL column indices.  row range must equal sla_ctx_row_range(m). */
int sla_csr_from_csr_rows(sla_ctx_t, int64_t m, int64_t n, int64_t row_begin, int64_t row_count,
                          const int64_t *rowptr_local, const int64_t *colidx, const double *val, sla_csr_t *out);
/* MatrixMarket ingestion with the loader semantics of the reference's test/Perf.hs:20-45: `matrix
 * coordinate real general`, 1-based -> 0-based, entries fed to fromListSM in file order, no symmetric
 * expansion; `matrix array` files give dense vectors (right-hand sides). */
int sla_csr_from_matrix_market(sla_ctx_t, const char *path, int dup_policy, sla_csr_t *out);
int sla_vec_from_matrix_market(sla_ctx_t, const char *path, sla_vec_t *out);
/* The array layouts of the reference's `vector/` package, single-device contexts unless noted.
 * CSC (vector/src/Data/Sparse/Internal/CSC.hs:17-24: cscColPtr has n+1 entries, cscRowIx / cscVal one per stored entry; toCSC
 * :51-55): the arrays are the canonical CSR of the transpose and are validated as such (row indices in [0, m), ascending and
 * unrepeated inside a column; otherwise SLA_ERR_INVALID / SLA_ERR_OOB -- unsorted triplets go through sla_csr_from_coo).  The
 * result is the same lowered matrix sla_csr_from_csr gives for the CSR arrays of A (device sort); the CSC side stays attached as
 * A's cached transpose, so (<#), CGNE_ and bcgStep need no second sort. */
int sla_csr_from_csc(sla_ctx_t, int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowidx, const double *val,
                     sla_csr_t *out);
/* The lowered matrix as CSC arrays (fromCSC0, CSC.hs:61-77, gives the triplets back): colptr n+1 entries, rowidx / val nnz;
 * any pointer may be NULL. */
int sla_csr_export_csc(sla_csr_t, int64_t *colptr, int64_t *rowidx, double *val);
/* transposeSM (src/Data/Sparse/SpMatrix.hs:717) / transposeCSR (vector/.../CSR.hs:138-141) / transposeCSC (CSC.hs:104-108):
 * a new lowered matrix owned by the caller (sort by (column, row) on the device). */
int sla_csr_transpose(sla_csr_t A, sla_csr_t *out);
/* CSB (vector/src/Data/Sparse/Internal/CSB.hs:38-70, Buluc et al.): square blocks of edge `beta`; blkptr has nbx * nby + 1
 * entries with nbx = ceil(m / beta), nby = ceil(n / beta) (csbParams :72-77); block f(i, j) = (i div beta) + (j div beta) * nbx
 * (blockIx :88-92); rowix / colix are RELATIVE to the block (0 .. beta-1), unordered inside it.  Expanded to coordinates and
 * lowered through fromListSM's path (a repeated (i, j): the later one wins); an index outside its block or the matrix =>
 * SLA_ERR_OOB.  Works on every kind of context (each rank passes all the arrays, like sla_csr_from_coo). */
int sla_csr_from_csb(sla_ctx_t, int64_t m, int64_t n, int64_t beta, const int64_t *blkptr, const int64_t *rowix,
                     const int64_t *colix, const double *val, sla_csr_t *out);
/* jacobiPre x = recip <$> extractDiag x (Sparse.hs:689-690): the diagonal matrix of reciprocal diagonal
 * entries (rows without a stored diagonal entry stay empty).  Single-rank contexts. */
int sla_jacobi_pre(sla_csr_t A, sla_csr_t *out);
/* D #~# A for a DIAGONAL left factor D (matMatSparsified, SpMatrix.hs:816-824, restricted to the case the
 * preconditioners need): row i of the result is d_ii * row i of A, entries with |x| <= 1e-12 dropped, rows
 * without a D entry dropped.  `jacobiPre aa #~# aa` is the left-Jacobi-preconditioned operator. */
int sla_csr_diag_mul(sla_csr_t D, sla_csr_t A, sla_csr_t *out);
/* triLowerSolve (upper == 0, Sparse.hs:750-776) / triUpperSolve (upper != 0, :784-811): forward / backward
 * substitution x_i = (b_i - subrow_i . x) / t_ii with the diagonal and the strictly lower (upper) part of T; entries
 * on the other side are ignored like the reference's extractSubRow does.  Every row is the ascending left fold of
 * separately rounded products followed by one subtraction and one division (bit-identical to the reference's order);
 * the result goes through sparsifySV (|x_i| <= 1e-12 reads back as 0).  A diagonal entry that is missing or
 * |t_ii| <= 1e-12 => SLA_ERR_NEEDS_PIVOTING (the reference throws NeedsPivoting), `bad_row` (may be NULL) receives the row.
 * Rows are level-scheduled at the first call (one launch per dependency level, replayed as a HIP graph while b / x
 * stay the same buffers).  Single-rank contexts; T square, b and x of its dimension, x != b. */
int sla_tri_solve(sla_csr_t T, int upper, sla_vec_t b, sla_vec_t x, int64_t *bad_row);
/* number of dependency levels / the widest level of the schedule sla_tri_solve uses (builds it if needed) */
int sla_tri_solve_info(sla_csr_t T, int upper, int64_t *levels, int64_t *widest_level);
/* mSsorPre aa omega = (l, r), l = (eye n ^-^ scale omega e) ## reciprocal d, r = d ^-^ scale omega f with e / d / f the
 * strictly lower / diagonal / strictly upper parts of aa (Sparse.hs:712-720, diagPartitions :673-678).  Both factors
 * hold the structurally non-zero entries ((##) itself would keep explicit zeros over the whole index set; the
 * values at the stored positions are identical).  Single-rank contexts. */
int sla_ssor_pre(sla_csr_t A, double omega, sla_csr_t *l, sla_csr_t *r);
/* ilu0Pre aa = (L, U) (Sparse.hs:696-706): the reference defines it as the COMPLETE Doolittle factorisation `lu aa`
 * (:488-527; unit diagonal in L) with every entry outside aa's stored positions removed afterwards.
 *   exact_lu != 0: that definition, entry for entry (complete LU with fill-in, then the filter); at most 4096 rows.
 *   exact_lu == 0: EXTENSION -- the incomplete factorisation proper: the same recurrences on aa's stored positions only, any
 *                  size; bit-identical to the exact mode whenever `lu aa` creates no fill outside aa's pattern.
 * Values failing isNz (|x| <= 1e-12) are not stored (row 0 of U and column 0 of L excepted, as in luInit); a pivot u_jj failing
 * isNz while rows below remain => SLA_ERR_NEEDS_PIVOTING ("NeedsPivoting solveForLij U(j,j)"), *bad_row = j (may be NULL).
 * Host-side set-up (like sla_ssor_pre); apply the factors with sla_tri_solve.  Single-rank contexts, square matrices. */
int sla_ilu0_pre(sla_csr_t A, int exact_lu, sla_csr_t *l, sla_csr_t *u, int64_t *bad_row);
/* m1 ## m2 (transpose_b == 0) and m1 ##^ m2 = m1 ## transpose m2 (transpose_b != 0): matMat_ (SpMatrix.hs:768-811).  The result is
 * STRUCTURALLY DENSE over (rows of m1 holding an entry) x (columns of m2 holding an entry) -- an empty intersection still
 * yields an explicit 0.0, like the reference's `sum (liftI2 (*) col row)`; every entry is the ascending left fold, from 0,
 * of separately rounded products.  Incompatible sizes => SLA_ERR_DIM_MISMATCH with the reference's message
 * ("matMat : incompatible matrix sizes", :795).  Single-rank contexts; at most 2^28 result entries. */
int sla_csr_matmat(sla_csr_t A, sla_csr_t B, int transpose_b, sla_csr_t *out);
int sla_csr_destroy(sla_csr_t);
/* dim / nnz of SpMatrix (local_rows/local_nnz = this rank's block) */
int sla_csr_dims(sla_csr_t, int64_t *m, int64_t *n, int64_t *nnz_local, int64_t *rows_local);
/* Device arrays copied back and widened to int64 for bit-exact index parity checks (local block;
 * rowptr has rows_local+1 entries starting at 0). */
int sla_csr_export(sla_csr_t, int64_t *rowptr, int64_t *colidx, double *val);
/* isDiagonalSM (SpMatrix.hs:411-415) evaluated at creation (global answer on every rank) */
int sla_csr_is_diagonal(sla_csr_t, int *out);

/* ---- SpVector (dense on the device) ------------------------------------------------------------ */

/* fromListDenseSV / mkSpVR / fromVector (SpVector.hs:183,194,240): `host` holds all n entries; each
 * rank keeps its block.  host == NULL gives zeroV. */
int sla_vec_create(sla_ctx_t, int64_t n, const double *host, sla_vec_t *out);
/* this rank's block only: `host_local` holds the entries of sla_ctx_row_range(n) */
int sla_vec_create_local(sla_ctx_t, int64_t n, const double *host_local, sla_vec_t *out);
int sla_vec_destroy(sla_vec_t);
int sla_vec_dim(sla_vec_t, int64_t *n, int64_t *n_local);
/* toVectorDense (SpVector.hs:250): all n entries (all-gathered when sharded) */
int sla_vec_to_host(sla_vec_t, double *host);
int sla_vec_to_host_local(sla_vec_t, double *host_local);
int sla_vec_copy(sla_vec_t src, sla_vec_t dst);

/* ---- A1..A4: (#>), (<#), (<.>), norm2, (^+^) (^-^) (.*) ----------------------------------------- */

int sla_spmv(sla_csr_t A, sla_vec_t x, sla_vec_t y);   /* y = A #> x   (Common.hs:242-250) */
int sla_spmv_t(sla_csr_t A, sla_vec_t x, sla_vec_t y); /* y = x <# A = transpose A #> x (Common.hs:253-256) */
int sla_dot(sla_vec_t x, sla_vec_t y, double *out);    /* x <.> y      (SpVector.hs:116-117) */
int sla_nrm2(sla_vec_t x, double *out);                /* norm2 x      (SpVector.hs:119-129) */
int sla_axpby(double a, sla_vec_t x, double b, sla_vec_t y); /* y := a .* x ^+^ b .* y (SpVector.hs:107-114) */
int sla_scal(double a, sla_vec_t x);                   /* x := a .* x; normalize2 = scal (1/norm2 x) */

/* ---- A5..A7: solver state records ---------------------------------------------------------------- */

/* cgsInit / bicgsInit / cgneInit (Sparse.hs:921-924, 962-965, 864-868).  `method` is SLA_CGS_,
 * SLA_BICGSTAB_ or SLA_CGNE_.  The shadow residual r0hat = b - A x0 of the README usage
 * (README.md:205-226) is kept inside the state.
 * Extension: SLA_BCG_ builds the record of the reference's COMMENTED bcgInit / bcgStep (Sparse.hs:886-909: x r rhat p phat with
 * rhat0 = p0 = phat0 = r0; one (#>) and one (<#) per step).  sla_linsolve0 still rejects SLA_BCG_ like linSolve0 does (:1031). */
int sla_solver_init(int method, sla_csr_t A, sla_vec_t b, sla_vec_t x0, sla_solver_t *out);
/* k applications of cgsStep / bicgstabStep / cgneStep (Sparse.hs:928-939, 972-981, 870-878):
 * `iterate step s !! k`.  Enqueues and returns; no convergence test. */
int sla_solver_step(sla_solver_t, int k_steps);
/* copy a state field (_x, _r, _p, _u; BCG records: _x, _r, _p, _rHat, _pHat) into `out` */
int sla_solver_get(sla_solver_t, int field, sla_vec_t out);
/* A deep copy of a state record.  The reference's step functions are pure (`bicgstabStep aa r0hat s` returns a new record,
 * Sparse.hs:972-981): a binding that must keep s while stepping on -- `iterate (bicgstabStep aa r0hat) s0 !! k`,
 * README.md:222-226 -- clones first, then steps the clone (the shim's bicgstabStep / cgsStep do exactly that). */
int sla_solver_clone(sla_solver_t, sla_solver_t *out);
/* Replace the shadow residual of a CGS / BiCGSTAB state -- the explicit `r0hat` / `rhat` argument of bicgstabStep / cgsStep
 * (sla_solver_init stores r0 = b - A x0, the README's choice) -- and re-evaluate the carried rho = r . r0hat with it. */
int sla_solver_set_shadow(sla_solver_t, sla_vec_t r0hat);
int sla_solver_destroy(sla_solver_t);
/* convenience spellings used by the Haskell shim */
int sla_bicgstab_init(sla_csr_t A, sla_vec_t b, sla_vec_t x0, sla_solver_t *out);
int sla_bicgstab_step(sla_solver_t, int k_steps);
int sla_cgs_init(sla_csr_t A, sla_vec_t b, sla_vec_t x0, sla_solver_t *out);
int sla_cgs_step(sla_solver_t, int k_steps);

/* ---- A8: linSolve0 ----------------------------------------------------------------------------- */

/* linSolve0 method aa b x0 (Sparse.hs:1016-1072): size guard -> diagonal shortcut -> loop of
 * (step; true residual; test) entirely on the device.  SLA_GMRES_ / SLA_BCG_ return
 * SLA_ERR_UNSUPPORTED_METHOD exactly like the reference (use sla_gmres for the extension). */
int sla_linsolve0(int method, sla_csr_t A, sla_vec_t b, sla_vec_t x0, const sla_solve_opts *opts,
                  sla_vec_t x_out, sla_solve_info *info);

/* ---- A9/A10: Arnoldi, GMRES(m), (<\>) ------------------------------------------------------------ */

/* arnoldi aa b kn (Sparse.hs:630-667).  Q_colmajor: n x (kn+1) host buffer (may be NULL to skip the
 * copy), H_colmajor: (kn+1) x kn host buffer with leading dimension kn+1, *k_done = number of H
 * columns produced (< kn after a breakdown).  Q holds this rank's rows when sharded (ld = n_local). */
int sla_arnoldi(sla_csr_t A, sla_vec_t b, int kn, double *Q_colmajor, double *H_colmajor, int *k_done);
/* Restarted GMRES(m) on the device Arnoldi (the reference's gmres is commented out, Sparse.hs:828-848:
 * parity is pinned at the arnoldi level only).  opts->max_iters bounds the total Arnoldi steps. */
int sla_gmres(sla_csr_t A, sla_vec_t b, sla_vec_t x0, int restart, const sla_solve_opts *opts,
              sla_vec_t x_out, sla_solve_info *info);
/* aa <\> b (Class.hs:244-249) as the dead instance defined it (Sparse.hs:1080-1084): GMRES from
 * x0 = 0.1 * ones. */
int sla_linsolve(sla_csr_t A, sla_vec_t b, sla_vec_t x_out, sla_solve_info *info);

/* ---- measurement hooks (bench.py) ---------------------------------------------------------------- */

/* kernels whose launches can be bracketed by HIP events on the context stream */
typedef enum {
    SLA_KERNEL_ALL = -1,      /* sla_prof_start: record every kernel below, each launch tagged with its id */
    SLA_KERNEL_SPMV = 0,      /* plain y = A x */
    SLA_KERNEL_SPMV_DOT = 1,  /* K1: Ap = A p fused with Ap . r0hat */
    SLA_KERNEL_SPMV_DOT2 = 2, /* K3: As = A s fused with As . s, As . As */
    SLA_KERNEL_SPMV_RES = 3,  /* true-residual SpMV fused with ||A x - b||^2 */
    SLA_KERNEL_SPMV_DUAL = 4, /* K1 + true residual of the previous iterate from ONE matrix sweep (linSolve0) */
    SLA_KERNEL_BICG_K2 = 5,   /* alpha ; s = r - alpha Ap */
    SLA_KERNEL_BICG_K4 = 6,   /* omega ; x += alpha p + omega s ; r = s - omega As ; r . r0hat */
    SLA_KERNEL_BICG_K5 = 7,   /* beta ; p = r + beta (p - omega Ap) */
    SLA_KERNEL_CGS_C2 = 8,    /* alpha ; q = u - alpha A p ; u + q ; x += alpha (u + q) */
    SLA_KERNEL_CGS_C4 = 9,    /* beta ; u = r + beta q ; p = u + beta (q + beta p) */
    SLA_KERNEL_BICG_K45 = 10, /* K4 + K5 in one sweep (single-rank flow; K3 then also sums As . r0hat and s . r0hat) */
    SLA_KERNEL_EXCHANGE = 11, /* row-sharded: the exchange of an SpMV's input vector (ncclAllGather, or the grouped halo ncclSend/ncclRecv),
                                 events on the stream it is issued on (the second stream when it overlaps the interior rows) */
    SLA_KERNEL_SUMS = 12,     /* row-sharded: per-rank partial sums made global (finalize + all-gather; in the ghost-row flows the grouped
                                 exchange that also carries a halo) */
    SLA_KERNEL_ONCHIP = 13,   /* sla_solver_step(k) as ONE persistent launch with the solver state on chip (constant-coefficient stencils that fit) */
    SLA_KERNEL_COUNT = 16
} sla_kernel_id;
/* record up to `max_launches` event pairs around launches of `kernel_id` (SLA_KERNEL_ALL: of every kernel above) from now on */
int sla_prof_start(sla_ctx_t, int kernel_id, int max_launches);
/* synchronise, return the number of recorded launches and their mean / min duration in ms (over all recorded kernels) */
int sla_prof_stop(sla_ctx_t, int *launches, double *mean_ms, double *min_ms);
/* after sla_prof_stop: the same statistics for one kernel id of the last recording */
int sla_prof_query(sla_ctx_t, int kernel_id, int *launches, double *mean_ms, double *min_ms);
/* What this GPU sustains for the access shape of the solver's vector kernels: `reads` vectors read and `writes` vectors written
 * per element (16 bytes per lane, the grid of the BLAS-1 kernels, non-temporal loads once the footprint overflows the
 * memory-side cache), `reps` launches timed one by one with HIP events.  (reads, writes) = (8, 0) pure read, (5, 3) the K4+K5
 * sweep, (2, 1) a triad.  bench.py's "measured ceiling": bytes = 8 (reads + writes) n per launch. */
int sla_stream_probe(sla_ctx_t, int reads, int writes, int64_t n, int reps, double *mean_ms, double *min_ms);
/* Rehearsal of the N > 1 point-to-point transfers on ONE rank: `pieces` grouped ncclRecv / ncclSend pairs with the context's own rank as
 * the peer move `count` doubles on the context's stream (the entry points, datatype, group calls and stream of the halo exchange and of
 * the grouped all-gather); *max_abs_err = largest difference between what was sent and what arrived (0 expected).  Needs a context
 * with an RCCL communicator (sla_ctx_create_dist; nranks may be 1); SLA_ERR_INVALID otherwise. */
int sla_dist_p2p_selftest(sla_ctx_t, int64_t count, int pieces, double *max_abs_err);
/* First contact of a multi-rank job: ONE checked collective across the context's real ranks (every rank calls it with the same
 * arguments), so that a hang or an error names its collective before anything is timed.  phase 0: ncclAllGather of `count` doubles
 * per rank; 1: the same all-gather as one group of ncclSend / ncclRecv pairs between all ranks (the pattern of the halo exchange
 * and of the overlapped all-gather); 2: the integer max all-reduce of the lowering's cross-rank decisions.  *max_abs_err: largest
 * difference between what arrived and what the peers sent (0 expected); *ms: wall-clock of the phase.  Works on loopback and
 * 1-rank contexts too.  (Serves linSolve0's sharded flow, Sparse.hs:1016-1072) */
int sla_dist_preflight(sla_ctx_t, int phase, int64_t count, double *max_abs_err, double *ms);
/* SLA_DEBUG_BINDING=1: launches, copies, collectives or device allocations issued by a thread that is not inside an entry
 * point bound to the context they belong to (HIP's current device is per thread: the bug class of multi-device fan-out,
 * invisible on a one-GPU box).  0 in a correct library; the GPU test suites assert it under the debug switch. */
long sla_debug_binding_violations(void);
/* number of HIP devices visible to this process (0 without a GPU; never fails) */
int sla_device_count(int *count);
/* ranks of the communicator behind this context: ncclCommCount for an RCCL communicator, the group size for the loopback
 * test backend, 1 for a single-GPU context */
int sla_ctx_comm_ranks(sla_ctx_t, int *nranks);
/* name of the SpMV form picked for A at lowering time and its launch geometry: "wdia" (wave-sliced (offset, value)
 * records: constant-coefficient stencils), "vdict[+xwin]" (one byte per entry), "stream+diagdict[+xwin]" (values +
 * 1-byte column codes), "stream[+xwin]" (values + i32 columns), "stream+ldspanels" (dense rows: x in LDS panels), "stream+colpanels" (irregular,
 * x > L2), "scalar"; row-sharded matrices add " x_exchange=window|allgather" */
int sla_csr_kernel_info(sla_csr_t, char *buf, int buflen);
/* Typed properties of a lowered matrix (what sla_csr_kernel_info prints, for callers that must not parse a string).
 * fold: how (#>) adds the products of a row (Common.hs:247-260 folds them left to right in ascending column order):
 *   SLA_FOLD_EXACT      that fold bit for bit on every row, reruns bit-identical (value-indexed forms, wave / exact tile forms);
 *   SLA_FOLD_REGROUPED  a FIXED regrouping for long rows (lane-group / wavefront / workgroup partial sums; on a row slab whose x arrives in
 *                       exchange groups, the tile form's fold over the column panels in the plan's visiting order): within
 *                       nnz_i * eps * sum |a_ij x_j| of the reference's value, reruns bit-identical;
 *   SLA_FOLD_RELAXED    the same set of separately rounded products added in timing order (LDS atomics of the CU-wide tile form):
 *                       same bound, NOT reproducible bit for bit from run to run.  Only with option tile_relaxed = 1 (opt-in: 17 % more
 *                       BiCGSTAB iterations per second on a 10 M-row matrix of 33 random columns per row); every default form is
 *                       SLA_FOLD_EXACT or SLA_FOLD_REGROUPED, i.e. a rerun gives the same bits.
 * x_exchange: 0 single rank, 1 all-gather of x per (#>), 2 window (halo) exchange.  struct_size as in sla_solve_info. */
typedef enum { SLA_FOLD_EXACT = 0, SLA_FOLD_REGROUPED = 1, SLA_FOLD_RELAXED = 2 } sla_fold_kind;
typedef struct {
    int32_t struct_size;  /* IN: sizeof(sla_csr_props) of the caller's header */
    int32_t fold;         /* sla_fold_kind */
    int32_t x_exchange;
    int32_t nranks;
    int64_t rows_local;   /* rows / stored entries of this rank's block */
    int64_t nnz_local;
    int32_t rowptr_bits;  /* 32 or 64 */
    int32_t reserved;
} sla_csr_props;
#define SLA_CSR_PROPS_INIT {(int32_t)sizeof(sla_csr_props), 0, 0, 0, 0, 0, 0, 0}
int sla_csr_get_props(sla_csr_t A, sla_csr_props *out);
/* Row-sharded matrices: what ONE exchange of a (#>) input moves, as planned at creation -- doubles this rank sends to /
 * receives from each peer (nranks entries each, cap >= nranks; the own slot is 0).  Window mode: the peers' parts of this rank's column
 * window; all-gather mode: whole shards.  bench.py prices the event-timed exchanges against the xGMI link peak with it.  Single-rank
 * matrices: SLA_ERR_INVALID. */
int sla_csr_exchange_plan(sla_csr_t A, int64_t *send_len, int64_t *recv_len, int cap);
/* What "lowered once" cost (fromListSM -> device, SpMatrix.hs:218-224): the wall-clock phases of this matrix's lowering as
 * "phase=milliseconds;..." (validation + narrowing + upload of the canonical arrays, one analysis per storage form, the exchange plan,
 * the tile form); bench.py reports it in its end_to_end block. */
int sla_csr_lower_info(sla_csr_t A, char *buf, int buflen);

/* ---- row-sharded exchange planning (pure host arithmetic, no GPU needed) ------------------------------ */

/* Given every rank's referenced column window windows[2*q] = cmin_q, windows[2*q+1] = cmax_q (cmax < cmin:
 * no entries) of an n-column matrix split in ceil(n/nranks)-row blocks, return for `rank` the contiguous
 * x ranges (global begin, length; per peer) it sends and receives per SpMV, and whether the window
 * exchange replaces the plain all-gather.  This is the plan sla_csr_from_* derives internally. */
int sla_plan_window_exchange(int nranks, int rank, int64_t n, const int64_t *windows, int64_t *send_begin,
                             int64_t *send_len, int64_t *recv_begin, int64_t *recv_len, int *use_window);
/* Pure host planning of the OVERLAPPED ALL-GATHER of x for all-gather-mode matrices on the tile form (BASELINE config 3a sharded;
 * DESIGN.md section 6): what rank `rank` of `nranks` does for a matrix of n columns cut into panels of 2^shift columns when the
 * gather goes out as grouped point-to-point exchanges and the tile launch runs as panel passes behind them.
 *   order 0 ("arrival", option ag_order = 0): `groups` column chunks of every shard per exchange group; panels visited own-first,
 *            then by the group that completes them -- a row is ONE left fold over the panels in that order (ascending columns
 *            inside a panel): not the reference's ascending order (Common.hs:247-260) but a fixed, documented permutation of it;
 *   order 1 ("ascending", ag_order = 1): one group per source rank in rank order, panels ascending: the reference's fold bit for bit.
 * Outputs: visit[P] (P = ceil(n / 2^shift)) the panel visiting order; pass_ptr[npass + 1] / pass_need[npass] (arrays of P + 1 / P
 * entries suffice): pass p walks visit[pass_ptr[p] .. pass_ptr[p+1]) once pass_need[p] exchange groups have completed; *ngroups. */
int sla_plan_allgather_passes(int nranks, int rank, int64_t n, int shift, int groups, int order, int32_t *visit, int32_t *pass_ptr,
                              int32_t *pass_need, int32_t *npass, int32_t *ngroups);
/* ... and the exchange groups of that plan (the same on every rank): quadruples (group, source rank, first column, end column), in
 * the order every rank posts them; group g is ONE ncclGroupStart/End launch in which each source sends its pieces to every peer.
 * pieces = cap x 4 int64; *count = quadruples written (SLA_ERR_INVALID if cap is too small). */
int sla_plan_allgather_groups(int nranks, int64_t n, int shift, int groups, int order, int64_t *pieces, int cap, int *count);

#ifdef __cplusplus
}
#endif
#endif /* SLA_HIP_H */
