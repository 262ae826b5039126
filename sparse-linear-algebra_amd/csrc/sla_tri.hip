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

// ---------------------------------------------------------------------------------------------
// the same solve as ONE persistent launch (option tri_syncfree; round 5)
// ---------------------------------------------------------------------------------------------
// The level schedule costs a dependent launch per level (5.2 us x 646 levels at 216^3).  Here the workgroups of one resident grid walk
// the schedule slots in order (slot t in round t / (grid x 256): everything a slot depends on sits in an earlier slot, i.e. is held by a
// co-resident wavefront or is done), and a row waits for the rows it reads by POLLING x itself: x is pre-filled with a NaN payload no
// computation produces, a row's value is published with ONE 8-byte agent-scope store and read with agent-scope loads, so there is no
// flag beside the data and no fence.  The arithmetic is the level kernel's, entry by entry in ascending order: bit-identical results.
// A lane whose dependency does not show up within `spin_limit` polls raises *fail and leaves -- the host then runs the level schedule
// (a grid that is not co-resident, a part whose L2s do not forward agent-scope stores: never a hang).
constexpr unsigned long long kTriPending = 0x7ff8dead5a5a0001ull;

__global__ void __launch_bounds__(kBlock) tri_fill_pending_kernel(int64_t n, double *x) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        __hip_atomic_store((unsigned long long *)(x + i), kTriPending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(kBlock) tri_syncfree_kernel(const int64_t *__restrict__ tptr, const int32_t *__restrict__ tcol,
                                                               const double *__restrict__ tval, const double *__restrict__ tdiag,
                                                               const int32_t *__restrict__ order, int64_t nslots, const double *__restrict__ b,
                                                               double *x, int *fail, int spin_limit) {
#pragma clang fp contract(off)
    for (int64_t base = (int64_t)blockIdx.x * kBlock; base < nslots; base += (int64_t)gridDim.x * kBlock) {
        const int64_t t = base + threadIdx.x;
        bool done = t >= nslots;
        int64_t k = 0, e = 0;
        int i = 0;
        double d = 1.0, bi = 0.0, r = 0.0;
        if (!done) {
            k = tptr[t];
            e = tptr[t + 1];
            i = order[t];
            d = tdiag[t];
            bi = b[i];
        }
        int spins = 0;
        for (;;) {
            if (!done) {
                // up to four dependencies in flight, consumed in order as far as they have arrived
                unsigned long long v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    v[q] = k + q < e ? __hip_atomic_load((const unsigned long long *)(x + tcol[k + q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kTriPending;
                bool moved = false, chain = true;   // (in order: entry q of the batch is only taken if all before it were)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (chain && k < e && v[q] != kTriPending) {
                        const double prod = tval[k] * __longlong_as_double((long long)v[q]);
                        r = r + prod;
                        ++k;
                        moved = true;
                    } else {
                        chain = false;
                    }
                }
                if (k == e) {
                    const double xi = (bi - r) / d;
                    __hip_atomic_store((unsigned long long *)(x + i), (unsigned long long)__double_as_longlong(xi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    done = true;
                } else if (moved) {
                    spins = 0;
                } else if (++spins > spin_limit ||
                           ((spins & 255) == 0 && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {   // (a failure elsewhere: nobody waits for rows that will not come)
                    __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    done = true;
                } else {
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if (__ballot(!done) == 0) break;
        }
    }
}

int launch_tri_syncfree(const sla_csr *T, const sla_tri_plan *p, const double *b, double *x, int *d_fail) {
    sla_ctx *c = T->ctx;
    const int64_t n = T->m;
    hipStream_t st = stream_of(c);
    SLA_HIP_TRY(hipMemsetAsync(d_fail, 0, sizeof(int), st));
    hipLaunchKernelGGL(tri_fill_pending_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, st, n, x);
    // the grid must be co-resident: at most tri_grid workgroups, never more than 8 per CU
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(c->tri_grid, (int64_t)8 * c->n_cu), (n + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(tri_syncfree_kernel, dim3(grid), dim3(kBlock), 0, st, p->d_tptr, p->d_tcol, p->d_tval, p->d_tdiag, p->d_order, n, b, x, d_fail,
                       c->tri_spin);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
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
