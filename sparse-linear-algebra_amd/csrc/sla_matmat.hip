// sla_matmat.hip -- (##) and (##^): SpMatrix x SpMatrix.
//
// Reference: matMat_ / matMatUnsafeWith (src/Data/Sparse/SpMatrix.hs:768-811).  The product is STRUCTURALLY DENSE over
// (rows of m1 that hold an entry) x (columns of m2 that hold an entry): `overRows2 <$> immSM m1` maps every present row,
// `(`dott` vm1) <$> transposeIM2 (immSM m2)` every present column, and `dott x y = sum (liftI2 (*) x y)` is the ascending
// sum, from 0, of b_kj * a_ik over the indices k both hold -- an empty intersection still yields an explicit 0.0.
// Incompatible sizes: `error "matMat : incompatible matrix sizes"` (:795) -> SLA_ERR_DIM_MISMATCH.
//
// On the hot path this is small algebra (the (kn+1) x kn Hessenberg of `arnoldi`, checkArnoldi's A Q' and Q H), so one
// thread per output entry merging a CSR row of m1 with a CSR row of transpose m2 (the library's lazily built transpose) is
// plenty; separately rounded multiply and add keep the reference's summation bit for bit.
#include <hip/hip_runtime.h>

#include <limits>
#include <vector>

#include "sla_internal.hpp"

namespace sla {

template <typename RPA, typename RPB>
__global__ void __launch_bounds__(kBlock) matmat_kernel(const RPA *__restrict__ arp, const int32_t *__restrict__ acol,
                                                        const double *__restrict__ aval, const RPB *__restrict__ brp,
                                                        const int32_t *__restrict__ bcol, const double *__restrict__ bval,
                                                        const int32_t *__restrict__ rows, const int32_t *__restrict__ cols, int64_t nr,
                                                        int64_t nc, double *__restrict__ out) {
#pragma clang fp contract(off)
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= nr * nc) return;
    const int32_t i = rows[t / nc], j = cols[t % nc];
    RPA ka = arp[i];
    const RPA ea = arp[i + 1];
    RPB kb = brp[j];
    const RPB eb = brp[j + 1];
    double acc = 0.0;
    while (ka < ea && kb < eb) {   // intersectionWith (*) in ascending key order (IntM.hs:78-80)
        const int32_t ca = acol[ka], cb = bcol[kb];
        if (ca < cb) ++ka;
        else if (ca > cb) ++kb;
        else {
            const double prod = bval[kb] * aval[ka];
            acc = acc + prod;
            ++ka;
            ++kb;
        }
    }
    out[t] = acc;
}

}  // namespace sla

using namespace sla;

extern "C" int sla_csr_matmat(sla_csr_t A, sla_csr_t B, int transpose_b, sla_csr_t *out) {
    if ((A && !A->kids.empty()) || (B && !B->kids.empty())) return multi_unsupported("sla_csr_matmat");
    return no_throw("sla_csr_matmat", [&]() -> int {
        if (!A || !B || !out) return fail(SLA_ERR_INVALID, "sla_csr_matmat: null argument");
        if (A->ctx != B->ctx) return fail(SLA_ERR_INVALID, "sla_csr_matmat: operands belong to different contexts");
        sla_ctx *c = A->ctx;
        if (c->collectives) return fail(SLA_ERR_INVALID, "sla_csr_matmat: single-rank contexts only (## is not sharded, SURVEY 8(e))");
        Bind bind(c);
        const int64_t inner_b = transpose_b ? B->n : B->m, out_cols = transpose_b ? B->m : B->n;
        if (A->n != inner_b)
            return fail(SLA_ERR_DIM_MISMATCH, "matMat : incompatible matrix sizes((" + std::to_string(A->m) + "," + std::to_string(A->n) + "),(" +
                                                  std::to_string(transpose_b ? B->n : B->m) + "," + std::to_string(out_cols) + "))");
        // rows of "B by output column": A ## B walks the columns of B = rows of transpose B; A ##^ B the rows of B itself
        sla_csr *Bt = B;
        if (!transpose_b) SLA_TRY(csr_transposed(B, &Bt));
        SLA_TRY(csr_ensure_canon(A));    // (the product kernel reads the canonical arrays of both factors)
        SLA_TRY(csr_ensure_canon(Bt));
        // present rows of A / of Bt (host copies of the two row-pointer arrays: m + 1 and cols + 1 integers)
        auto present = [&](const sla_csr *M, std::vector<int32_t> &keys) -> int {
            const size_t cnt = (size_t)M->rows + 1;
            std::vector<int64_t> rp(cnt);
            if (M->rp64) {
                SLA_HIP_TRY(hipMemcpyAsync(rp.data(), M->d_rowptr, sizeof(int64_t) * cnt, hipMemcpyDeviceToHost, stream_of(c)));
                SLA_HIP_TRY(hipStreamSynchronize(stream_of(c)));
            } else {
                std::vector<int32_t> rp32(cnt);
                SLA_HIP_TRY(hipMemcpyAsync(rp32.data(), M->d_rowptr, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost, stream_of(c)));
                SLA_HIP_TRY(hipStreamSynchronize(stream_of(c)));
                for (size_t i = 0; i < cnt; ++i) rp[i] = rp32[i];
            }
            for (int64_t i = 0; i < M->rows; ++i)
                if (rp[(size_t)i + 1] > rp[(size_t)i]) keys.push_back((int32_t)i);
            return SLA_OK;
        };
        std::vector<int32_t> rows, cols;
        SLA_TRY(present(A, rows));
        SLA_TRY(present(Bt, cols));
        const int64_t nr = (int64_t)rows.size(), nc = (int64_t)cols.size();
        if (nr * nc > ((int64_t)1 << 28))
            return fail(SLA_ERR_ALLOC, "sla_csr_matmat: the structurally dense product would hold more than 2^28 entries");
        std::vector<double> val((size_t)(nr * nc));
        if (nr * nc > 0) {
            int32_t *d_rows = nullptr, *d_cols = nullptr;
            double *d_out = nullptr;
            hipError_t e = dev_malloc(c, (void **)&d_rows, sizeof(int32_t) * (size_t)nr);
            if (e == hipSuccess) e = dev_malloc(c, (void **)&d_cols, sizeof(int32_t) * (size_t)nc);
            if (e == hipSuccess) e = dev_malloc(c, (void **)&d_out, sizeof(double) * (size_t)(nr * nc));
            if (e == hipSuccess) e = hipMemcpyAsync(d_rows, rows.data(), sizeof(int32_t) * (size_t)nr, hipMemcpyHostToDevice, stream_of(c));
            if (e == hipSuccess) e = hipMemcpyAsync(d_cols, cols.data(), sizeof(int32_t) * (size_t)nc, hipMemcpyHostToDevice, stream_of(c));
            if (e == hipSuccess) {
                const dim3 grid((unsigned)((nr * nc + kBlock - 1) / kBlock)), block(kBlock);
#define SLA_MM(RA, RB)                                                                                                         \
                hipLaunchKernelGGL((matmat_kernel<RA, RB>), grid, block, 0, stream_of(c), (const RA *)A->d_rowptr, A->d_col, A->d_val,   \
                                   (const RB *)Bt->d_rowptr, Bt->d_col, Bt->d_val, d_rows, d_cols, nr, nc, d_out)
                if (A->rp64 && Bt->rp64) SLA_MM(int64_t, int64_t);
                else if (A->rp64) SLA_MM(int64_t, int32_t);
                else if (Bt->rp64) SLA_MM(int32_t, int64_t);
                else SLA_MM(int32_t, int32_t);
#undef SLA_MM
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(val.data(), d_out, sizeof(double) * (size_t)(nr * nc), hipMemcpyDeviceToHost, stream_of(c));
            if (e == hipSuccess) e = hipStreamSynchronize(stream_of(c));
            (void)hipFree(d_rows);
            (void)hipFree(d_cols);
            (void)hipFree(d_out);
            if (e != hipSuccess) return fail(SLA_ERR_HIP, std::string("sla_csr_matmat: ") + hipGetErrorString(e));
        }
        // canonical CSR of the result: a present row holds one entry per present column, ascending
        std::vector<int64_t> rp((size_t)A->m + 1, 0), ci((size_t)(nr * nc));
        {
            size_t ri = 0;
            for (int64_t i = 0; i < A->m; ++i) {
                const bool has = ri < rows.size() && rows[ri] == i;
                rp[(size_t)i + 1] = rp[(size_t)i] + (has ? nc : 0);
                if (has) {
                    for (int64_t q = 0; q < nc; ++q) ci[(size_t)(rp[(size_t)i] + q)] = cols[(size_t)q];
                    ++ri;
                }
            }
        }
        return sla_csr_from_csr(c, A->m, out_cols, rp.data(), ci.data(), val.data(), out);
    });
}
