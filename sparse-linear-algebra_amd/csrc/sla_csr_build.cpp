// sla_csr_build.cpp -- host side of "lower once": SpMatrix triples -> canonical CSR -> row blocks.
//
// Reference behaviour restated here (bit-exact on indices, checked against oracle/ in tests/):
//   fromListSM = foldl' insertSpMatrix (Data/Sparse/SpMatrix.hs:205-224, Internal/IntMap2.hs:24-28):
//     out-of-bounds index -> error ; duplicate (i,j) -> the LAST value wins;
//   traversal order of the IntMap-of-IntMap = ascending row, ascending column = the CSR layout of
//     vector/src/Data/Sparse/Internal/CSR.hs:43-50,74-78 with csPtrV row pointers
//     (vector/src/Data/Sparse/Internal/Vector/Utils.hs:12-26: rowptr[0] = 0, empty rows repeat).
#include <algorithm>
#include <numeric>

#include "sla_internal.hpp"

namespace sla {

int build_csr_from_coo(int64_t m, int64_t n, int64_t nnz, const int64_t *row, const int64_t *col,
                       const double *val, int dup_policy, HostCsr &out) {
    if (m < 0 || n < 0 || nnz < 0) return fail(SLA_ERR_INVALID, "negative dimension");
    for (int64_t k = 0; k < nnz; ++k)
        if (row[k] < 0 || row[k] >= m || col[k] < 0 || col[k] >= n)
            return fail(SLA_ERR_OOB, "insertSpMatrix : index out of bounds");
    // stable counting sort by row
    std::vector<int64_t> start((size_t)m + 1, 0);
    for (int64_t k = 0; k < nnz; ++k) start[(size_t)row[k] + 1]++;
    for (int64_t i = 0; i < m; ++i) start[(size_t)i + 1] += start[(size_t)i];
    std::vector<int64_t> perm((size_t)nnz);
    {
        std::vector<int64_t> cur(start.begin(), start.end() - 1);
        for (int64_t k = 0; k < nnz; ++k) perm[(size_t)cur[(size_t)row[k]]++] = k;
    }
    out.m = m;
    out.n = n;
    out.rowptr.assign((size_t)m + 1, 0);
    out.col.clear();
    out.val.clear();
    out.col.reserve((size_t)nnz);
    out.val.reserve((size_t)nnz);
    for (int64_t i = 0; i < m; ++i) {
        int64_t *b = perm.data() + start[(size_t)i], *e = perm.data() + start[(size_t)i + 1];
        // stable by column: members of a duplicate group stay in input order
        std::stable_sort(b, e, [&](int64_t a, int64_t c) { return col[a] < col[c]; });
        for (int64_t *q = b; q != e;) {
            int64_t *g = q;
            double v = val[*q];
            while (g + 1 != e && col[*(g + 1)] == col[*q]) {
                ++g;
                if (dup_policy == SLA_DUP_SUM) v += val[*g];
                else v = val[*g];  // last wins
            }
            out.col.push_back(col[*q]);
            out.val.push_back(v);
            q = g + 1;
        }
        out.rowptr[(size_t)i + 1] = (int64_t)out.col.size();
    }
    return SLA_OK;
}

// transposeIM2 (IntMap2.hs:88-89): rows of A^T in ascending (original row) order
void transpose_csr(const HostCsr &a, HostCsr &t) {
    t.m = a.n;
    t.n = a.m;
    const int64_t nnz = a.rowptr[(size_t)a.m];
    t.rowptr.assign((size_t)a.n + 1, 0);
    t.col.assign((size_t)nnz, 0);
    t.val.assign((size_t)nnz, 0.0);
    for (int64_t k = 0; k < nnz; ++k) t.rowptr[(size_t)a.col[(size_t)k] + 1]++;
    for (int64_t j = 0; j < a.n; ++j) t.rowptr[(size_t)j + 1] += t.rowptr[(size_t)j];
    std::vector<int64_t> cur(t.rowptr.begin(), t.rowptr.end() - 1);
    for (int64_t i = 0; i < a.m; ++i)
        for (int64_t k = a.rowptr[(size_t)i]; k < a.rowptr[(size_t)i + 1]; ++k) {
            const int64_t d = cur[(size_t)a.col[(size_t)k]]++;
            t.col[(size_t)d] = i;
            t.val[(size_t)d] = a.val[(size_t)k];
        }
}

// isDiagonalSM (SpMatrix.hs:411-415) on a row block: every local row has exactly one entry, on the diagonal
bool host_is_diagonal(int64_t rows, int64_t row_begin, const int64_t *rowptr, const int64_t *col) {
    for (int64_t i = 0; i < rows; ++i)
        if (rowptr[i + 1] - rowptr[i] != 1 || col[rowptr[i]] != row_begin + i) return false;
    return true;
}

// Greedy row blocks for the CSR-stream kernel: <= kNnzPerRowBlock entries and <= kMaxRowsPerRowBlock
// rows per block; rows longer than kNnzPerRowBlock form blocks of up to 4 such rows (<= kWaveRowMax
// entries each) or a block of their own (longer still).
void build_row_blocks(int64_t rows, const int64_t *rowptr, std::vector<int32_t> &rb, int64_t &max_row_nnz, int row_align,
                      int nnz_target) {
    if (nnz_target <= 0 || nnz_target > kNnzPerRowBlock) nnz_target = kNnzPerRowBlock;
    rb.clear();
    rb.push_back(0);
    max_row_nnz = 0;
    int64_t r0 = 0;
    while (r0 < rows) {
        int64_t r = r0, cnt = 0;
        while (r < rows && r - r0 < kMaxRowsPerRowBlock) {
            const int64_t len = rowptr[r + 1] - rowptr[r];
            if (len > max_row_nnz) max_row_nnz = len;
            if (cnt + len > (r == r0 ? kNnzPerRowBlock : nnz_target)) break;  // a single row may use the whole stage
            cnt += len;
            ++r;
        }
        // optional (SLA_ROW_ALIGN=16): end short-row blocks on a multiple of 16 rows so that the y / rowptr
        // segments of a block start and end on 128-byte lines.  Off by default: measured 1.5 % slower on the
        // 216^3 Laplacian (fewer entries per block outweigh the unsplit cache lines).
        if (row_align > 1 && r < rows && r - r0 > 2 * row_align) r -= r % row_align;
        if (r == r0) {  // the row does not fit the LDS stage: long-row block
            const int64_t len = rowptr[r + 1] - rowptr[r];
            if (len > max_row_nnz) max_row_nnz = len;
            ++r;
            if (len <= kWaveRowMax) {
                // group up to 4 consecutive long rows (one wavefront each in the kernel)
                while (r < rows && r - r0 < 4) {
                    const int64_t l2 = rowptr[r + 1] - rowptr[r];
                    if (l2 <= kNnzPerRowBlock || l2 > kWaveRowMax) break;
                    if (l2 > max_row_nnz) max_row_nnz = l2;
                    ++r;
                }
            }
        }
        rb.push_back((int32_t)r);
        r0 = r;
    }
}

}  // namespace sla
