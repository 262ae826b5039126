// sla_formats.cpp -- the array layouts of the reference's `vector/` package either side of the lowered matrix:
//   CSC  (vector/src/Data/Sparse/Internal/CSC.hs:17-24 cscRowIx / cscColPtr / cscVal; toCSC :51-55; fromCSC0 :61-77; transposeCSC :104-108)
//   CSB  (vector/src/Data/Sparse/Internal/CSB.hs:38-70, the layout of Buluc et al.; block order = blockIx :88-92)
//   transposeSM (src/Data/Sparse/SpMatrix.hs:717) / transposeCSR (vector/.../CSR.hs:138-141) as an owned handle.
// CSC arrays ARE the CSR arrays of the transpose: ingestion lowers them as such and transposes on the device (sla_coo_sort.hip); the
// CSC side stays attached to the result as its cached transpose, so a caller who arrives with CSC gets (<#), CGNE_ and bcgStep without
// a second sort.  CSB arrays are expanded to coordinates on the host (row-parallel over the blocks) and go through fromListSM's path.
#include <algorithm>
#include <string>
#include <vector>

#include "sla_internal.hpp"

using namespace sla;

namespace {

int single_device_only(const sla_ctx *c, const char *who) {
    if (c->nranks != 1 || c->collectives)
        return fail(SLA_ERR_INVALID, std::string(who) + ": single-device contexts only (a row-sharded matrix holds a row block: its transpose is a column block)");
    return SLA_OK;
}

}  // namespace

extern "C" {

int sla_csr_transpose(sla_csr_t A, sla_csr_t *out) {
    if (A && !A->kids.empty()) return multi_unsupported("sla_csr_transpose");
    return no_throw("sla_csr_transpose", [&]() -> int {
        if (!A || !out) return fail(SLA_ERR_INVALID, "sla_csr_transpose: bad argument");
        *out = nullptr;
        SLA_TRY(single_device_only(A->ctx, "sla_csr_transpose"));
        Bind bind(A->ctx);
        sla_csr *T = nullptr;
        SLA_TRY(csr_transposed(A, &T));
        A->transposed = nullptr;   // handed to the caller (A builds another one if (<#) asks for it later)
        *out = T;
        return SLA_OK;
    });
}

int sla_csr_from_csc(sla_ctx_t c, int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowidx, const double *val,
                     sla_csr_t *out) {
    if (c && !c->kids.empty()) return multi_unsupported("sla_csr_from_csc");
    return no_throw("sla_csr_from_csc", [&]() -> int {
        if (!c || !out || !colptr || m < 0 || n < 0) return fail(SLA_ERR_INVALID, "sla_csr_from_csc: bad argument");
        *out = nullptr;
        SLA_TRY(single_device_only(c, "sla_csr_from_csc"));
        // the n x m matrix whose rows are A's columns: validated like any canonical CSR (monotone pointers, row indices in [0, m), ascending
        // and unrepeated inside a column -- toCSC's stable sort by column gives that for triplets listed by rows, fromListSM's path for any others)
        sla_csr *T = nullptr;
        SLA_TRY(sla_csr_from_csr(c, n, m, colptr, rowidx, val, &T));
        sla_csr *A = nullptr;
        const int rc = sla_csr_transpose(T, &A);
        if (rc != SLA_OK) {
            sla_csr_destroy(T);
            return rc;
        }
        A->transposed = T;   // owned by A from here on (sla_csr_destroy)
        A->lower_log = "from_csc=1;" + A->lower_log;
        *out = A;
        return SLA_OK;
    });
}

int sla_csr_export_csc(sla_csr_t A, int64_t *colptr, int64_t *rowidx, double *val) {
    if (A && !A->kids.empty()) return multi_unsupported("sla_csr_export_csc");
    return no_throw("sla_csr_export_csc", [&]() -> int {
        if (!A) return fail(SLA_ERR_INVALID, "null matrix");
        SLA_TRY(single_device_only(A->ctx, "sla_csr_export_csc"));
        Bind bind(A->ctx);
        sla_csr *T = nullptr;
        SLA_TRY(csr_transposed(A, &T));   // stays cached on A
        return sla_csr_export(T, colptr, rowidx, val);
    });
}

int sla_csr_from_csb(sla_ctx_t c, int64_t m, int64_t n, int64_t beta, const int64_t *blkptr, const int64_t *rowix,
                     const int64_t *colix, const double *val, sla_csr_t *out) {
    return no_throw("sla_csr_from_csb", [&]() -> int {
        if (!c || !out || !blkptr || m < 0 || n < 0 || beta <= 0) return fail(SLA_ERR_INVALID, "sla_csr_from_csb: bad argument");
        *out = nullptr;
        // csbParams (CSB.hs:72-77): blocks per side; blockIx (:88-92): f(i, j) = (i div beta) + (j div beta) * nbx -- the block ROW runs fastest
        const int64_t nbx = (m + beta - 1) / beta, nby = (n + beta - 1) / beta, nblk = nbx * nby;
        if (blkptr[0] != 0) return fail(SLA_ERR_INVALID, "sla_csr_from_csb: blkptr[0] must be 0");
        for (int64_t b = 0; b < nblk; ++b)
            if (blkptr[b + 1] < blkptr[b]) return fail(SLA_ERR_INVALID, "sla_csr_from_csb: blkptr not monotone");
        const int64_t nnz = blkptr[nblk];
        if (nnz > 0 && (!rowix || !colix || !val)) return fail(SLA_ERR_INVALID, "sla_csr_from_csb: null rowix / colix / val");
        std::vector<int64_t> row((size_t)nnz), col((size_t)nnz);
        std::vector<int> bad((size_t)host_threads(), 0);
        par_rows(nblk, 1, [&](int t, int64_t lo, int64_t hi) {
            for (int64_t b = lo; b < hi; ++b) {
                const int64_t i0 = (b % nbx) * beta, j0 = (b / nbx) * beta;
                for (int64_t k = blkptr[b]; k < blkptr[b + 1]; ++k) {
                    const int64_t r = rowix[k], q = colix[k];   // relative to the block (CSB.hs:47-53): 0 .. beta - 1
                    if (r < 0 || r >= beta || q < 0 || q >= beta || i0 + r >= m || j0 + q >= n) bad[(size_t)t] = 1;
                    row[(size_t)k] = i0 + r;
                    col[(size_t)k] = j0 + q;
                }
            }
        }, 4096);
        if (std::find(bad.begin(), bad.end(), 1) != bad.end())
            return fail(SLA_ERR_OOB, "sla_csr_from_csb: block-relative index outside its block or the matrix");
        // elements of a block are unordered (consBlocks :103-107 conses them); a repeated (i, j) resolves like fromListSM's (the later one wins)
        return sla_csr_from_coo(c, m, n, nnz, row.data(), col.data(), val, SLA_DUP_LAST_WINS, out);
    });
}

}  // extern "C"
