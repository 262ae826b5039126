// sla_ilu0.cpp -- ilu0Pre (reference Sparse.hs:696-706 on top of lu, :488-527).
//
// The reference's "ILU(0)" is the COMPLETE Doolittle factorisation `lu aa` with the entries outside aa's stored positions
// filtered away afterwards.  Two modes:
//
//   exact_lu != 0   the reference's definition, entry for entry: complete LU (fill-in allowed), then the filter.  The
//                   work is that of a dense factorisation once fill sets in, so it is limited to n <= kIluExactMaxN rows
//                   -- the sizes at which the reference's own IntMap version is usable.
//   exact_lu == 0   EXTENSION (flagged as such in the header): the incomplete factorisation proper -- the same Doolittle
//                   recurrences evaluated on aa's stored positions only, any size.  Whenever `lu aa` creates no fill outside
//                   aa's pattern (tridiagonal / banded-without-gaps matrices, block-diagonal matrices with full blocks, ...)
//                   the two modes return the same factors BIT FOR BIT: every entry is the same expression over the same
//                   operands in the same order.
//
// Both run on the host, row by row: preconditioner set-up is not on the solver's hot path (like mSsorPre); what the hot path
// uses are the factors, through the level-scheduled triangular solves (sla_tri_solve).
//
// Arithmetic, as the reference's (see oracle/sla_oracle.c:orc_lu_dense for the literal loop order):
//   u_ij = a_ij - sum_k l_ik u_kj          (i <= j)    contractSub l u i j (i-1): ascending k over the STORED l_ik, k < i,
//   l_ij = (a_ij - sum_k l_ik u_kj) / u_jj (i >  j)    absent u_kj read 0 (the product is still added), acc from 0;
//   values failing isNz (|x| <= 1e-12) are not stored -- except row 0 of U and column 0 of L, which luInit copies unfiltered;
//   a pivot u_jj failing isNz while rows below remain => NeedsPivoting "solveForLij" "U(j,j)" (SLA_ERR_NEEDS_PIVOTING, bad_row = j).
#include <algorithm>
#include <cmath>
#include <vector>

#include "sla_internal.hpp"

#pragma clang fp contract(off)   // separately rounded multiply and add, like the reference (GHC never fuses)

namespace sla {

constexpr int64_t kIluExactMaxN = 4096;

namespace {
struct Ent { int64_t c; double v; };
inline bool is_nz(double v) { return !(std::fabs(v) <= 1e-12); }
inline double lookup(const std::vector<Ent> &row, int64_t c) {   // m @@! (i, c): 0 when absent
    auto it = std::lower_bound(row.begin(), row.end(), c, [](const Ent &e, int64_t cc) { return e.c < cc; });
    return (it != row.end() && it->c == c) ? it->v : 0.0;
}
}  // namespace

}  // namespace sla

using namespace sla;

extern "C" int sla_ilu0_pre(sla_csr_t A, int exact_lu, sla_csr_t *l_out, sla_csr_t *u_out, int64_t *bad_row) {
    if (A && !A->kids.empty()) return multi_unsupported("sla_ilu0_pre");
    return no_throw("sla_ilu0_pre", [&]() -> int {
        if (!A || !l_out || !u_out) return fail(SLA_ERR_INVALID, "sla_ilu0_pre: null argument");
        sla_ctx *c = A->ctx;
        if (c->collectives) return fail(SLA_ERR_INVALID, "sla_ilu0_pre: single-rank contexts only");
        Bind bind(c);
        if (A->m != A->n) return fail(SLA_ERR_DIM_MISMATCH, "ilu0Pre : the matrix must be square");
        const int64_t n = A->m;
        if (exact_lu && n > kIluExactMaxN)
            return fail(SLA_ERR_INVALID, "sla_ilu0_pre: the reference's complete-LU definition is limited to " + std::to_string(kIluExactMaxN) +
                                             " rows (dense fill-in); use exact_lu = 0, the incomplete factorisation on A's pattern");
        std::vector<int64_t> rp((size_t)n + 1), ci((size_t)std::max<int64_t>(A->nnz, 1));
        std::vector<double> va((size_t)std::max<int64_t>(A->nnz, 1));
        SLA_TRY(sla_csr_export(A, rp.data(), ci.data(), va.data()));
        auto a_at = [&](int64_t i, int64_t j, bool *present) -> double {
            const int64_t *b = ci.data() + rp[(size_t)i], *e = ci.data() + rp[(size_t)i + 1];
            const int64_t *it = std::lower_bound(b, e, j);
            const bool has = it != e && *it == j;
            if (present) *present = has;
            return has ? va[(size_t)(it - ci.data())] : 0.0;
        };
        std::vector<std::vector<Ent>> L((size_t)n), U((size_t)n);   // L: strictly lower part (the unit diagonal is implicit)
        auto contract = [&](const std::vector<Ent> &lrow, int64_t j) -> double {   // sum over the stored l_ik of l_ik * u_kj
            double acc = 0.0;
            for (const Ent &e : lrow) {
                const double prod = e.v * lookup(U[(size_t)e.c], j);
                acc = acc + prod;
            }
            return acc;
        };
        if (n > 0) {
            for (int64_t k = rp[0]; k < rp[1]; ++k) U[0].push_back({ci[(size_t)k], va[(size_t)k]});   // u0 = row 0 of aa, unfiltered
            if (!is_nz(lookup(U[0], 0))) {
                if (bad_row) *bad_row = 0;
                return fail(SLA_ERR_NEEDS_PIVOTING, "NeedsPivoting solveForLij U(0,0)");
            }
        }
        const double u00 = n > 0 ? lookup(U[0], 0) : 0.0;
        std::vector<int64_t> cand;
        for (int64_t i = 1; i < n; ++i) {
            std::vector<Ent> &lrow = L[(size_t)i];
            // ---- row i of L, columns ascending: the complete factorisation visits every j < i, the incomplete one the
            //      stored positions of row i only
            cand.clear();
            if (exact_lu) {
                for (int64_t j = 0; j < i; ++j) cand.push_back(j);
            } else {
                for (int64_t k = rp[(size_t)i]; k < rp[(size_t)i + 1] && ci[(size_t)k] < i; ++k) cand.push_back(ci[(size_t)k]);
            }
            for (int64_t j : cand) {
                bool has = false;
                const double a = a_at(i, j, &has);
                if (j == 0) {   // luInit: column 0 of L = stored a_i0 ./ u00, unfiltered
                    if (has) lrow.push_back({0, a / u00});
                    continue;
                }
                const double ujj = lookup(U[(size_t)j], j);   // (isNz: checked when row j was finished)
                const double l = (a - contract(lrow, j)) / ujj;
                if (is_nz(l)) lrow.push_back({j, l});
            }
            // ---- row i of U
            cand.clear();
            if (exact_lu) {
                for (int64_t j = i; j < n; ++j) cand.push_back(j);
            } else {
                for (int64_t k = rp[(size_t)i]; k < rp[(size_t)i + 1]; ++k)
                    if (ci[(size_t)k] >= i) cand.push_back(ci[(size_t)k]);
            }
            for (int64_t j : cand) {
                const double u = a_at(i, j, nullptr) - contract(lrow, j);
                if (is_nz(u)) U[(size_t)i].push_back({j, u});
            }
            if (i < n - 1 && !is_nz(lookup(U[(size_t)i], i))) {   // rows below still need l_ri = ... / u_ii
                if (bad_row) *bad_row = i;
                return fail(SLA_ERR_NEEDS_PIVOTING, "NeedsPivoting solveForLij U(" + std::to_string(i) + "," + std::to_string(i) + ")");
            }
        }
        // ---- sparsifyLU: keep what aa stores (the unit diagonal of L included), hand both factors back as device matrices
        auto emit = [&](const std::vector<std::vector<Ent>> &rows, bool unit_diag, sla_csr_t *out) -> int {
            std::vector<int64_t> orp((size_t)n + 1, 0), oci;
            std::vector<double> ova;
            for (int64_t i = 0; i < n; ++i) {
                bool has = false;
                for (const Ent &e : rows[(size_t)i]) {
                    (void)a_at(i, e.c, &has);
                    if (has) { oci.push_back(e.c); ova.push_back(e.v); }
                }
                if (unit_diag) {
                    (void)a_at(i, i, &has);
                    if (has) { oci.push_back(i); ova.push_back(1.0); }
                }
                orp[(size_t)i + 1] = (int64_t)oci.size();
            }
            if (oci.empty()) { oci.push_back(0); ova.push_back(0.0); }
            return sla_csr_from_csr(c, n, n, orp.data(), oci.data(), ova.data(), out);
        };
        sla_csr_t Lh = nullptr;
        SLA_TRY(emit(L, true, &Lh));
        const int rc = emit(U, false, u_out);
        if (rc != SLA_OK) {
            sla_csr_destroy(Lh);
            return rc;
        }
        *l_out = Lh;
        return SLA_OK;
    });
}
