// NOT BUILT (round 5): the three-stage pipelined CSR-stream kernel of round 3 -- measured 7-12 % slower than spmv_stream_kernel on every workload
// (profiles/r03_pmc_l1_l2_stream_kernel.txt, LABNOTES L4), superseded by spmv_wave_kernel in round 4; removed from the library with its option
// stream_pipe, kept here for the record.

// sla_spmv_pipe.hip -- the general CSR-stream (#>) (f64 values + i32 columns) as a THREE-stage software pipeline.
// Reference semantics: Data/Sparse/Common.hs:242-260 (rows summed by one lane are the reference's ascending left fold).
//
// spmv_stream_kernel (sla_spmv_stream.hip) prefetches the col / val / rowptr streams of the next row block, but every row block still
// pays two dependent memory round trips in its own critical path that nothing of its own hides: the gather x[col] (it can only
// be issued once the columns have arrived, and the products wait for it: ~1 us from the L2) and the epilogue operands
// (w[row] / z[row] are loaded after the row's fold -- and vector loads return IN ORDER, so waiting for them also drains the
// prefetch issued before them).  Measured (tools/kbench, 216^3): the bare streams with the LDS stage, the barrier and the y
// store run at 5.6 TB/s, the full SpMV at 5.0, K1 (one operand) at 4.8 = 0.60 of the 8 TB/s peak.
//
// Here a workgroup works on three consecutive row blocks at once and issues its loads in the order they will be consumed:
//     iteration of block b:   P  products of b -> LDS          (waits for the gathers of b, issued one iteration ago)
//                             O  epilogue operands of the rows of block b+1 (w / z / running sums)
//                             G  gathers x[col] of block b+1   (its columns were streamed one iteration ago)
//                             S  col / val / rowptr streams of block b+2
//                             barrier ; R  row sums of b from LDS + fused epilogue (operands loaded one iteration ago)
// so the fold of b runs while the operands and gathers of b+1 and the streams of b+2 are in flight, and every wait is on a load
// issued a whole iteration earlier.  Values, row pointers and operands alternate between two register sets (loop unrolled by
// two) because a register copy would wait for the load that fills it.  Same row-block
// tables, same LDS layout (products double-buffered: one barrier per row block), same per-row folds and epilogue arithmetic
// as spmv_stream_kernel: identical bits.  Taken when no row is longer than a row block (the stream kernel keeps the
// long-row cases).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "sla_internal.hpp"
#include "sla_device.hpp"

namespace sla {

namespace {

struct EpiOps {   // epilogue operands of one row, loaded ahead of the fold
    double w = 0.0, z = 0.0, y0 = 0.0;
};

// (unconditional: `row` is a valid row of the slab whether or not this lane finishes one -- a lane-conditional load would put
// the loads of the pipeline into separate basic blocks, and the compiler then waits for ALL outstanding loads (vmcnt(0)) at
// every use instead of counting; the whole P / O / G / S sequence below is straight-line code for that reason)
template <int EPI, typename RP, bool YI>
__device__ __forceinline__ EpiOps epi_load(const SpmvArgs<RP> &a, int row) {
    EpiOps o;
    if constexpr (EPI == EPI_DOT || EPI == EPI_DOT2 || EPI == EPI_RES || EPI == EPI_SUB) o.w = a.w[row];
    if constexpr (EPI == EPI_DOT4) { o.w = a.w[row]; o.z = a.z[row]; }
    if constexpr (EPI == EPI_AXPY_DOT) { o.z = a.z[row]; o.w = (a.w ? a.w : a.z)[row]; }
    if constexpr (EPI == EPI_XPBY_NRM) o.z = a.z[row];
    if constexpr (YI) o.y0 = a.yinit[row];
    return o;
}

// spmv_epilogue (sla_device.hpp) on preloaded operands: the same operations in the same order
template <int EPI, typename RP>
__device__ __forceinline__ void epi_apply(const SpmvArgs<RP> &a, int row, double yv, double coef, const EpiOps &o, double &acc1, double &acc2) {
    if constexpr (EPI == EPI_NONE) {
        a.y[row] = yv;
    } else if constexpr (EPI == EPI_DOT) {
        a.y[row] = yv;
        acc1 += yv * o.w;
    } else if constexpr (EPI == EPI_DOT2) {
        a.y[row] = yv;
        acc1 += yv * o.w;
        acc2 += yv * yv;
    } else if constexpr (EPI == EPI_DOT4) {
        a.y[row] = yv;
        acc1 += yv * o.w;
        acc2 += yv * yv;
        a.acc3 += yv * o.z;
        a.acc4 += o.w * o.z;
    } else if constexpr (EPI == EPI_RES) {
        const double t = yv - o.w;
        acc1 += t * t;
    } else if constexpr (EPI == EPI_AXPY_DOT) {
        const double z = o.z - coef * yv;
        a.z[row] = z;
        acc1 += z * (a.w ? o.w : z);
    } else if constexpr (EPI == EPI_XPBY_NRM) {
        const double z = yv + coef * o.z;
        a.z[row] = z;
        acc1 += z * z;
    } else if constexpr (EPI == EPI_SUB) {
        a.y[row] = o.w - yv;
    }
}

template <int EPI, typename RP, bool YI>
__global__ void __launch_bounds__(kBlock, 8) spmv_pipe_kernel(SpmvArgs<RP> a, const RP *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                               const double *__restrict__ val, const int32_t *__restrict__ rb,
                                                               const RP *__restrict__ rbk, const double *__restrict__ xg, int xcd_remap) {
    __shared__ double s_prod[2][kNnzPerRowBlock];
    __shared__ int s_rp[2][kMaxRowsPerRowBlock + 1];
    __shared__ double s_red[4];
    const int tid = threadIdx.x;
    double coef;
    if (!spmv_prologue<EPI, RP>(a, s_red, coef)) return;
    double acc1 = 0.0, acc2 = 0.0;
    const RbWalk wk = rb_walk(a.nrb, xcd_remap);
    if (wk.first < wk.last) {
        const int nb = (wk.last - wk.first + wk.step - 1) / wk.step;   // row blocks of this workgroup
        const int blast = wk.first + (nb - 1) * wk.step;               // the last one: blocks "behind" it are clamped to it, so that
                                                                       // every G / S below is unconditional (the tail re-streams it)
        // row-block descriptors (first row, end row, first entry, end entry) of blocks b (0), b+1 (1), b+2 (2): scalar loads
        int b = wk.first;
        const int b1 = min(b + wk.step, blast), b2 = min(b + 2 * wk.step, blast);
        int r0 = rb[b], r1 = rb[b + 1];
        RP k0 = rbk[b], k1 = rbk[b + 1];
        int nr0 = rb[b1], nr1 = rb[b1 + 1], fr0 = rb[b2], fr1 = rb[b2 + 1];
        RP nk0 = rbk[b1], nk1 = rbk[b1 + 1], fk0 = rbk[b2], fk1 = rbk[b2 + 1];
        int32_t c[4];
        double vA[4], vB[4], xv[4];
        RP rpA, rpB;
        // S: the streams of one row block.  Lanes past the block's end re-read its last entry (an empty block reads the entry at
        // its position: the arrays carry zeroed slack for that), rows past its end the row pointer at its end.
        auto stream = [&](int sr0, int sr1, RP sk0, RP sk1, double (&v)[4], RP &rp) {
            const int last = max((int)(sk1 - sk0) - 1, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_nontemporal_load(col + sk0 + min(tid + j * kBlock, last));
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = __builtin_nontemporal_load(val + sk0 + min(tid + j * kBlock, last));
            rp = rowptr[sr0 + min(tid, sr1 - sr0)];
        };
        auto gather = [&]() {   // G: x[col] of the block whose columns sit in c[]
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[j] = xg[c[j]];
        };
        // which row does a lane finish in a block of nrows rows / cnt entries?  lane per row, or lane 0 of a power-of-two segment
        struct Fin { bool lane_per_row; int tpr, g, l; };
        auto fin_of = [&](int nrows, int cnt) {
            Fin f;
            f.lane_per_row = nrows > 64 || cnt <= 8 * nrows;
            int np2 = 1;
            while (np2 < nrows) np2 <<= 1;
            f.tpr = min(64, kBlock / np2);
            f.g = f.lane_per_row ? tid : tid / f.tpr;
            f.l = f.lane_per_row ? 0 : tid - f.g * f.tpr;
            return f;
        };
        auto operands = [&](int or0, int or1, RP ok0, RP ok1) {   // O: the epilogue operands of the row this lane finishes in that block
            const Fin f = fin_of(or1 - or0, (int)(ok1 - ok0));
            return epi_load<EPI, RP, YI>(a, or0 + min(f.g, max(or1 - or0 - 1, 0)));
        };
        // prime the pipeline: streams of b, operands of b, gathers of b, streams of b+1
        EpiOps opsA, opsB;
        stream(r0, r1, k0, k1, vA, rpA);
        opsA = operands(r0, r1, k0, k1);
        gather();
        stream(nr0, nr1, nk0, nk1, vB, rpB);
        int buf = 0;
        // one iteration; vcur / rpcur / opscur belong to block b (vcur / rpcur then receive the streams of block b+2), opsnext receives
        // the operands of block b+1, whose columns are in c[]
        auto iteration = [&](double (&vcur)[4], RP &rpcur, const EpiOps &opscur, EpiOps &opsnext) {
            const int nrows = r1 - r0, cnt = (int)(k1 - k0);
            double *prod = s_prod[buf];
            int *rp = s_rp[buf];
            // P
            if (tid < nrows) rp[tid] = (int)(rpcur - k0);
            if (tid == 0) rp[nrows] = cnt;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = tid + j * kBlock;
                const double p = vcur[j] * xv[j];
                if (i < cnt) prod[i] = p;
            }
            const Fin f = fin_of(nrows, cnt);
            const bool lane_per_row = f.lane_per_row;
            const int tpr = f.tpr, g = f.g, l = f.l;
            const bool fin = g < nrows && l == 0;
            const EpiOps &ops = opscur;
            // O (block b+1: consumed a whole iteration from now -- vector loads return in order, so an operand loaded for THIS
            // block's fold would stall it for a full HBM round trip), then G, S for the blocks behind
            opsnext = operands(nr0, nr1, nk0, nk1);
            gather();
            stream(fr0, fr1, fk0, fk1, vcur, rpcur);
            // descriptors three blocks ahead (scalar loads; consumed at the end of this iteration)
            const int b3 = min(b + 3 * wk.step, blast);
            const int gr0 = rb[b3], gr1 = rb[b3 + 1];
            const RP gk0 = rbk[b3], gk1 = rbk[b3 + 1];
            __syncthreads();
            // R
            if (lane_per_row) {
                if (tid < nrows) {
                    const int s = rp[tid], e = rp[tid + 1];
                    double acc = YI ? ops.y0 : 0.0;   // column-panel passes continue the running sum: still one ascending left fold
                    for (int k = s; k < e; ++k) acc += prod[k];
                    epi_apply<EPI, RP>(a, r0 + tid, acc, coef, ops, acc1, acc2);
                }
            } else {
                double acc = 0.0;
                if (g < nrows) {
                    const int e = rp[g + 1];
                    for (int k = rp[g] + l; k < e; k += tpr) acc += prod[k];
                }
                for (int off = tpr >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
                if (fin) {
                    if constexpr (YI) acc += ops.y0;
                    epi_apply<EPI, RP>(a, r0 + g, acc, coef, ops, acc1, acc2);
                }
            }
            buf ^= 1;
            b += wk.step;
            r0 = nr0; r1 = nr1; k0 = nk0; k1 = nk1;
            nr0 = fr0; nr1 = fr1; nk0 = fk0; nk1 = fk1;
            fr0 = gr0; fr1 = gr1; fk0 = gk0; fk1 = gk1;
        };
        for (int it = 0;;) {
            iteration(vA, rpA, opsA, opsB);
            if (++it >= nb) break;
            iteration(vB, rpB, opsB, opsA);
            if (++it >= nb) break;
        }
    }
    if constexpr (EPI == EPI_DOT || EPI == EPI_DOT2 || EPI == EPI_DOT4 || EPI == EPI_RES || EPI == EPI_AXPY_DOT || EPI == EPI_XPBY_NRM) {
        const double s1 = block_sum(acc1, s_red);
        if (tid == 0) a.p1[blockIdx.x] = s1;
    }
    if constexpr (EPI == EPI_DOT2 || EPI == EPI_DOT4) {
        const double s2 = block_sum(acc2, s_red);
        if (tid == 0) a.p2[blockIdx.x] = s2;
    }
    spmv_extra_partials<EPI>(a, s_red, tid);
}

template <int EPI, typename RP>
int launch_pipe_t(const sla_csr *A, const SpmvArgs<RP> &a, int grid) {
    sla_ctx *c = A->ctx;
    if (a.yinit)
        hipLaunchKernelGGL((spmv_pipe_kernel<EPI, RP, true>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, a.rowptr, a.col, a.val, a.rb, a.rbk, a.x,
                           c->xcd_remap);
    else
        hipLaunchKernelGGL((spmv_pipe_kernel<EPI, RP, false>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, a.rowptr, a.col, a.val, a.rb, a.rbk, a.x,
                           c->xcd_remap);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

template <typename RP>
int launch_pipe_rp(const sla_csr *A, int epi, const SpmvArgs<RP> &a, int grid) {
    switch (epi) {
        case EPI_NONE: return launch_pipe_t<EPI_NONE, RP>(A, a, grid);
        case EPI_DOT: return launch_pipe_t<EPI_DOT, RP>(A, a, grid);
        case EPI_DOT2: return launch_pipe_t<EPI_DOT2, RP>(A, a, grid);
        case EPI_DOT4: return launch_pipe_t<EPI_DOT4, RP>(A, a, grid);
        case EPI_RES: return launch_pipe_t<EPI_RES, RP>(A, a, grid);
        case EPI_AXPY_DOT: return launch_pipe_t<EPI_AXPY_DOT, RP>(A, a, grid);
        case EPI_XPBY_NRM: return launch_pipe_t<EPI_XPBY_NRM, RP>(A, a, grid);
        case EPI_SUB: return launch_pipe_t<EPI_SUB, RP>(A, a, grid);
    }
    return fail(SLA_ERR_INVALID, "launch_spmv_pipe: unknown epilogue");
}

}  // namespace

// is the pipelined stream kernel the one that runs a plain CSR-stream (#>) of A?  (every row fits a row block)
bool pipe_on(const sla_csr *A) { return A->ctx->stream_pipe != 0 && A->max_row_nnz <= kNnzPerRowBlock && A->nrb > 0; }

int launch_spmv_pipe(const sla_csr *A, int epi, const SpmvArgs<int32_t> &a, int grid) { return launch_pipe_rp<int32_t>(A, epi, a, grid); }
int launch_spmv_pipe(const sla_csr *A, int epi, const SpmvArgs<int64_t> &a, int grid) { return launch_pipe_rp<int64_t>(A, epi, a, grid); }

}  // namespace sla
