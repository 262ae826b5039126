// sla_tri.hip -- triLowerSolve / triUpperSolve (Numeric/LinearAlgebra/Sparse.hs:750-811): one launch per dependency level of the
// level-scheduled triangle, one lane per row, ascending fold with separately rounded multiply / add, one IEEE division, then sparsifySV.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "sla_internal.hpp"
#include "sla_device.hpp"

namespace sla {

// ---------------------------------------------------------------------------------------------
// triangular solves (triLowerSolve / triUpperSolve, Sparse.hs:750-811), one dependency level per launch
// ---------------------------------------------------------------------------------------------
// Rows of one level depend only on rows of earlier levels (earlier launches), so plain loads of x are coherent.
// One lane per row: r = ascending left fold of l_ij * x_j over the triangle's side of the row (separately rounded
// multiply and add), x_i = (b_i - r) / t_ii -- the reference's arithmetic, bit for bit.  Latency-bound by nature:
// the schedule's depth times the launch latency is the floor.
__global__ void __launch_bounds__(kBlock) tri_level_kernel(const int64_t *__restrict__ tptr, const int32_t *__restrict__ tcol,
                                                             const double *__restrict__ tval, const double *__restrict__ tdiag,
                                                             const int32_t *__restrict__ order, int64_t first, int64_t count,
                                                             const double *__restrict__ b, double *x) {
    const int64_t t = first + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= first + count) return;
    const int64_t s = tptr[t], e = tptr[t + 1];
    const int i = order[t];
    const double d = tdiag[t], bi = b[i];
    double r = 0.0;
    {
#pragma clang fp contract(off)
        int64_t k = s;
        for (; k + 4 <= e; k += 4) {  // 4 gathers in flight, folded in order
            const double x0 = x[tcol[k]], x1 = x[tcol[k + 1]], x2 = x[tcol[k + 2]], x3 = x[tcol[k + 3]];
            const double p0 = tval[k] * x0, p1 = tval[k + 1] * x1, p2 = tval[k + 2] * x2, p3 = tval[k + 3] * x3;
            r = r + p0;
            r = r + p1;
            r = r + p2;
            r = r + p3;
        }
        for (; k < e; ++k) {
            const double prod = tval[k] * x[tcol[k]];
            r = r + prod;
        }
        x[i] = (bi - r) / d;
    }
}

__global__ void __launch_bounds__(kBlock) tri_sparsify_kernel(int64_t n, double *x) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        if (fabs(x[i]) <= 1e-12) x[i] = 0.0;
}

int launch_tri_level(const sla_csr *T, const sla_tri_plan *p, int64_t first, int64_t count, const double *b, double *x) {
    if (count <= 0) return SLA_OK;
    const int grid = (int)((count + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(tri_level_kernel, dim3(grid), dim3(kBlock), 0, stream_of(T->ctx), p->d_tptr, p->d_tcol, p->d_tval, p->d_tdiag,
                       p->d_order, first, count, b, x);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

int launch_tri_sparsify(sla_ctx *c, int64_t n, double *x) {
    hipLaunchKernelGGL(tri_sparsify_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, x);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

}  // namespace sla
