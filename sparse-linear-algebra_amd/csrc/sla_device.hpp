// sla_device.hpp -- device-side helpers shared by the kernel translation units (sla_spmv_*.hip, sla_vec_kernels.hip, sla_arnoldi.hip, sla_spmv_tiles.hip):
// deterministic reductions, the consumer-side re-reduction of partials, the fused SpMV epilogues / prologue and the
// XCD-aware persistent walk.
#pragma once
#include <hip/hip_runtime.h>

#include "sla_internal.hpp"

namespace sla {

// Workgroups per CU the 8-per-CU kernels are compiled for: 64-bit row pointers and the heaviest epilogue (x + beta y with the norm) do not
// fit 64 registers -- at 8 they spilled up to 116 VGPRs to scratch (tools/kernel_resources.py); those instantiations get 96 registers.
template <int EPI, typename RP>
constexpr int kOcc8 = (sizeof(RP) == 8 || EPI == 7) ? 5 : 8;

// ---------------------------------------------------------------------------------------------
// reduction helpers (deterministic)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// sum over the 256 threads of the block, returned to every thread.  s4: 4 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double *s4) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((s4[0] + s4[1]) + s4[2]) + s4[3];
}

// fixed-order re-reduction of a partial array written by an EARLIER kernel
__device__ __forceinline__ double reduce_parts(const double *p, int n, int stride, double *s4) {
    double a = 0.0;
    if (n > 0 && n <= 8 * kBlock) {
        // the usual case (<= 2048 partials): all eight loads of a thread in flight at once instead of a chain of
        // dependent round trips at the head of every consumer kernel; same additions in the same order
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[(int64_t)min((int)threadIdx.x + j * kBlock, n - 1) * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if ((int)threadIdx.x + j * kBlock < n) a += v[j];
    } else {
        for (int i = threadIdx.x; i < n; i += kBlock) a += p[(int64_t)i * stride];
    }
    return block_sum(a, s4);
}

// The same re-reduction split in two so that a consumer kernel can ISSUE everything its prologue needs -- the partials of all its
// sums, the device scalars, the first vector elements of its sweep -- before it WAITS for any of it: one memory round trip at the
// head of the kernel instead of one per quantity (K2: done -> partials -> rho; K4+K5: four sums one after the other).  Same
// additions in the same order as reduce_parts (n <= 8 * kBlock partials: kMaxParts = 2048).
__device__ __forceinline__ void parts_issue(const double *p, int n, int stride, double (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[(int64_t)min((int)threadIdx.x + j * kBlock, n - 1) * stride];
}
__device__ __forceinline__ double parts_fold(const double (&v)[8], int n) {
    double a = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if ((int)threadIdx.x + j * kBlock < n) a += v[j];
    return a;
}
// block_sum of K values behind ONE pair of barriers; sK: 4 * K doubles of LDS.  Per value the same operations as block_sum.
template <int K>
__device__ __forceinline__ void block_sum_multi(double (&v)[K], double *sK) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) sK[4 * k + (threadIdx.x >> 6)] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = ((sK[4 * k] + sK[4 * k + 1]) + sK[4 * k + 2]) + sK[4 * k + 3];
}

__device__ __forceinline__ bool is_finite(double v) { return v == v && fabs(v) != INFINITY; }

// runIter's test (Sparse.hs:1047-1050) on the partials of ||A x - b||^2 left by an earlier kernel.
// Every workgroup takes the same decision; workgroup 0 publishes it.  Returns true when converged.
__device__ __forceinline__ bool residual_converged(SolverScalars *sc, const double *p, int n, int stride, double *s4) {
    const double rn = sqrt(reduce_parts(p, n, stride, s4));
    const bool conv = rn <= sc->tol;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc->resnorm = rn;
        // residual trace (sla_solve_opts.history): this is the residual of the iterate after sc->iters steps (the step that is
        // about to start has not been counted yet).  A residual may be tested twice (end of a host batch, then the next
        // step's prologue): same slot, same value.
        if (sc->hist && sc->iters >= 1 && sc->iters <= sc->hist_cap) sc->hist[sc->iters - 1] = rn;
        if (conv) { sc->done = 1; sc->flags |= SLA_FLAG_CONVERGED; }
        if (!is_finite(rn)) sc->flags |= SLA_FLAG_NONFINITE;
    }
    return conv;
}

// Arnoldi breakdown (Sparse.hs:665-667) is flagged by arn_normalize_kernel and acted upon by the kernels that FOLLOW it: the
// flag was written by an earlier launch, so every workgroup of this launch takes the same exit; workgroup 0 promotes it
// to `done` (which arn_normalize_kernel itself tests).  Solver flows never raise SLA_FLAG_BREAKDOWN.
__device__ __forceinline__ bool arn_stopped(SolverScalars *sc) {
    if (sc->done) return true;
    if (sc->flags & SLA_FLAG_BREAKDOWN) {
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) sc->done = 1;
        return true;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
// SpMV epilogues
// ---------------------------------------------------------------------------------------------
// the epilogue operands of a row (w: K1 / K3's dot operand, b of the residual forms; z: r0hat of the four-sum K3, the vector the
// CGNE forms update): loaded apart from the epilogue so that a kernel can issue them with its other loads, a whole fold early
template <int EPI, typename RP>
__device__ __forceinline__ void spmv_operands(const SpmvArgs<RP> &a, int row, double &wv, double &zv) {
    wv = 0.0;
    zv = 0.0;
    if constexpr (EPI == EPI_DOT || EPI == EPI_DOT2 || EPI == EPI_DOT4 || EPI == EPI_RES || EPI == EPI_SUB) wv = a.w[row];
    if constexpr (EPI == EPI_AXPY_DOT) {
        if (a.w) wv = a.w[row];
    }
    if constexpr (EPI == EPI_DOT4 || EPI == EPI_AXPY_DOT || EPI == EPI_XPBY_NRM) zv = a.z[row];
}
template <int EPI, typename RP>
__device__ __forceinline__ void spmv_epilogue_pre(const SpmvArgs<RP> &a, int row, double yv, double coef, double &acc1, double &acc2,
                                                  double wv, double zv) {
    if constexpr (EPI == EPI_NONE) {
        a.y[row] = yv;
    } else if constexpr (EPI == EPI_DOT) {
        a.y[row] = yv;
        acc1 += yv * wv;
    } else if constexpr (EPI == EPI_DOT2) {
        a.y[row] = yv;
        acc1 += yv * wv;
        acc2 += yv * yv;
    } else if constexpr (EPI == EPI_DOT4) {   // fused K4+K5 flow: As . r0hat and s . r0hat from the same sweep
        a.y[row] = yv;
        acc1 += yv * wv;
        acc2 += yv * yv;
        a.acc3 += yv * zv;
        a.acc4 += wv * zv;
    } else if constexpr (EPI == EPI_RES) {
        double t = yv - wv;  // (aa #> x) ^-^ b
        acc1 += t * t;
    } else if constexpr (EPI == EPI_AXPY_DOT) {
        double z = zv - coef * yv;
        a.z[row] = z;
        acc1 += z * (a.w ? wv : z);
    } else if constexpr (EPI == EPI_XPBY_NRM) {
        double z = yv + coef * zv;
        a.z[row] = z;
        acc1 += z * z;
    } else if constexpr (EPI == EPI_SUB) {
        a.y[row] = wv - yv;  // b ^-^ (aa #> x)
    }
}
template <int EPI, typename RP>
__device__ __forceinline__ void spmv_epilogue(const SpmvArgs<RP> &a, int row, double yv, double coef,
                                              double &acc1, double &acc2) {
    double wv, zv;
    spmv_operands<EPI, RP>(a, row, wv, zv);
    spmv_epilogue_pre<EPI, RP>(a, row, yv, coef, acc1, acc2, wv, zv);
}

// the two extra partial sums of EPI_DOT4 (see SpmvArgs::p3): called by every SpMV kernel after its p1 / p2 partials
template <int EPI, typename RP>
__device__ __forceinline__ void spmv_extra_partials(const SpmvArgs<RP> &a, double *s4, int tid) {
    if constexpr (EPI == EPI_DOT4) {
        const double s3 = block_sum(a.acc3, s4);
        if (tid == 0) a.p3[blockIdx.x] = s3;
        const double s4v = block_sum(a.acc4, s4);
        if (tid == 0) a.p4[blockIdx.x] = s4v;
    }
}

// Prologue shared by the SpMV kernels.  Returns false when the block must exit (solver done).
template <int EPI, typename RP>
__device__ __forceinline__ bool spmv_prologue(const SpmvArgs<RP> &a, double *s4, double &coef) {
    SolverScalars *sc = a.sc;
    coef = 0.0;
    if (sc == nullptr) return true;
    if (arn_stopped(sc)) return false;
    if (a.pres) {
        if (residual_converged(sc, a.pres, a.npres, a.pres_stride, s4)) return false;
    }
    if (a.step_begin & 1) {
        if (blockIdx.x == 0 && threadIdx.x == 0) sc->iters += 1;
    }
    const int par = (a.step_begin >> 1) & 1;
    if constexpr (EPI == EPI_AXPY_DOT) {
        if (a.pa) {  // CGNE: alpha = (r.r) / (p.p)
            coef = sc->rho2[par] / reduce_parts(a.pa, a.npa, a.pa_stride, s4);
            if (blockIdx.x == 0 && threadIdx.x == 0) sc->alpha = coef;
        } else {
            coef = sc->alpha;
        }
    } else if constexpr (EPI == EPI_XPBY_NRM) {  // CGNE: beta = (r1.r1) / (r.r)
        double rr1 = reduce_parts(a.pa, a.npa, a.pa_stride, s4);
        coef = rr1 / sc->rho2[par];
        if (blockIdx.x == 0 && threadIdx.x == 0) { sc->beta = coef; sc->rho2[par ^ 1] = rr1; }
    }
    return true;
}

// XCD-aware persistent walk over row blocks: workgroup g runs on XCD g % 8, so give XCD k the k-th
// contiguous eighth of the row blocks; its private 4 MiB L2 then holds one sliding window of x.
struct RbWalk {
    int first, step, last;
};
// G: the workgroups that take part (a multiple of 8 where the XCD shares matter); the others of the grid get an empty walk -- a
// launch whose instantiation holds fewer workgroups per CU than the grid was sized for keeps its grid (the consumers of the fused
// partial sums know it) and lets the surplus leave at once instead of running a second round
__device__ __forceinline__ RbWalk rb_walk(int nrb, int xcd_remap, int G) {
    RbWalk w;
    if ((int)blockIdx.x >= G) {
        w.first = w.last = 0;
        w.step = 1;
        return w;
    }
    if (xcd_remap && (G & 7) == 0 && nrb >= G) {
        const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3, per = (nrb + 7) >> 3;
        w.first = xcd * per + l;
        w.step = G >> 3;
        w.last = min((xcd + 1) * per, nrb);
    } else {
        w.first = blockIdx.x;
        w.step = G;
        w.last = nrb;
    }
    return w;
}
__device__ __forceinline__ RbWalk rb_walk(int nrb, int xcd_remap) { return rb_walk(nrb, xcd_remap, (int)gridDim.x); }

typedef double wd_f64x2u __attribute__((ext_vector_type(2), aligned(8)));  // a row pair of x at any 8-byte boundary
// EXEC on entry of an EXEC-masked asm fold, as an opaque value: the folds put EXEC back to it after every record.  Deliberately NOT
// __builtin_amdgcn_read_exec(): that is a plain copy of the physical register, which the compiler forwards into the "s" operand of
// the restoring s_mov_b64 -- "s_mov_b64 exec, exec", i.e. EXEC stays at the last record's odd-row mask (round 4: every row wrong).
__device__ __forceinline__ unsigned long long wd_save_exec() {
    unsigned long long e;
    asm volatile("s_mov_b64 %0, exec" : "=s"(e));
    return e;
}
typedef double wd_f64x2 __attribute__((ext_vector_type(2)));

// the fused epilogue of a row pair (row, row + 1; vb: the second row exists) of the wave-sliced kernels
// The arithmetic of the pair epilogue without its store: `out` = what would be stored; returns 0 (nothing stored: EPI_RES), 1 (into y) or
// 2 (into z).  wd_epilogue = this + the store; the wave kernel's prefetching instantiations hold the store back (sla_spmv_wave.hip, round 6).
template <int EPI>
__device__ __forceinline__ int wd_epilogue_calc(const SpmvArgs<int32_t> &a, bool vb, double ya, double yb, wd_f64x2 wv, wd_f64x2 zv, double coef,
                                                 double &acc1, double &acc2, wd_f64x2 &out) {
    out = wd_f64x2{ya, yb};
    if constexpr (EPI == EPI_NONE) {
        return 1;
    } else if constexpr (EPI == EPI_DOT) {
        acc1 += ya * wv.x;
        if (vb) acc1 += yb * wv.y;
        return 1;
    } else if constexpr (EPI == EPI_DOT2) {
        acc1 += ya * wv.x;
        acc2 += ya * ya;
        if (vb) { acc1 += yb * wv.y; acc2 += yb * yb; }
        return 1;
    } else if constexpr (EPI == EPI_DOT4) {   // (zv = the row pair of the read-only operand z)
        acc1 += ya * wv.x;
        acc2 += ya * ya;
        a.acc3 += ya * zv.x;
        a.acc4 += wv.x * zv.x;
        if (vb) { acc1 += yb * wv.y; acc2 += yb * yb; a.acc3 += yb * zv.y; a.acc4 += wv.y * zv.y; }
        return 1;
    } else if constexpr (EPI == EPI_RES) {
        const double ta = ya - wv.x, tb = yb - wv.y;  // (aa #> x) ^-^ b
        acc1 += ta * ta;
        if (vb) acc1 += tb * tb;
        return 0;
    } else if constexpr (EPI == EPI_AXPY_DOT) {
        out.x = zv.x - coef * ya;
        out.y = zv.y - coef * yb;
        acc1 += out.x * (a.w ? wv.x : out.x);
        if (vb) acc1 += out.y * (a.w ? wv.y : out.y);
        return 2;
    } else if constexpr (EPI == EPI_XPBY_NRM) {
        out.x = ya + coef * zv.x;
        out.y = yb + coef * zv.y;
        acc1 += out.x * out.x;
        if (vb) acc1 += out.y * out.y;
        return 2;
    } else {   // EPI_SUB
        out.x = wv.x - ya;  // b ^-^ (aa #> x)
        out.y = wv.y - yb;
        return 1;
    }
}
__device__ __forceinline__ void wd_store_pair(double *dst, int row, bool vb, wd_f64x2 out, bool nt_store) {
    if (vb) {
        if (nt_store) __builtin_nontemporal_store(out, (wd_f64x2 *)(dst + row));
        else *(wd_f64x2 *)(dst + row) = out;
    } else {
        dst[row] = out.x;
    }
}
template <int EPI>
__device__ __forceinline__ void wd_epilogue(const SpmvArgs<int32_t> &a, int row, bool vb, double ya, double yb, wd_f64x2 wv, wd_f64x2 zv,
                                            double coef, double &acc1, double &acc2, bool nt_store = false) {
    wd_f64x2 out;
    const int where = wd_epilogue_calc<EPI>(a, vb, ya, yb, wv, zv, coef, acc1, acc2, out);
    double *dst = where == 1 ? a.y : (where == 2 ? a.z : nullptr);
    if (dst) wd_store_pair(dst, row, vb, out, nt_store);
}

// ---------------------------------------------------------------------------------------------
// streaming vector kernels: 16-byte (double2) accesses, grid-stride loop over element pairs
// ---------------------------------------------------------------------------------------------
#define SLA_VEC_LOOP_BEGIN(n)                                                        \
    const int64_t _n2 = (n) >> 1;                                                    \
    const int64_t _gs = (int64_t)gridDim.x * kBlock;                                 \
    for (int64_t i2 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i2 < _n2; i2 += _gs) {
#define SLA_VEC_LOOP_END }
#define SLA_HAS_TAIL(n) (((n) & 1) && blockIdx.x == 0 && threadIdx.x == 0)

typedef double sla_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld2_nt(const double *p, int64_t i2) {
    const sla_d2 t = __builtin_nontemporal_load(reinterpret_cast<const sla_d2 *>(p) + i2);
    return make_double2(t.x, t.y);
}
__device__ __forceinline__ double2 ld2_t(const double *p, int64_t i2) { return reinterpret_cast<const double2 *>(p)[i2]; }
__device__ __forceinline__ void st2_nt(double *p, int64_t i2, double2 v) {
    __builtin_nontemporal_store(sla_d2{v.x, v.y}, reinterpret_cast<sla_d2 *>(p) + i2);
}
__device__ __forceinline__ void st2_t(double *p, int64_t i2, double2 v) { reinterpret_cast<double2 *>(p)[i2] = v; }
#define ld2 ld2_t
#define st2 st2_t
// Non-temporal loads in the BiCGSTAB vector kernels when the solver's vectors cannot stay in the 256 MB memory-side
// cache anyway (template NT, chosen per launch by vec_stream_nt): +12 % iterations/s at 10 M rows (7 x 80 MB), -4...-6 %
// at 1-2 M rows where the whole working set is cache-resident and the hint only loses hits.
template <bool NT>
__device__ __forceinline__ double2 ld2s(const double *p, int64_t i2) { return NT ? ld2_nt(p, i2) : ld2_t(p, i2); }


}  // namespace sla
