// sla_arnoldi_orth.hip -- one Arnoldi step's Gram-Schmidt (Numeric/LinearAlgebra/Sparse.hs:655-667) as ONE persistent launch (end of round 6).
//
// The launch flow (sla_arnoldi.hip) spends an Arnoldi step of GMRES(30) at 2 M rows like this: (#>) 20 us | dots pass 47 us | update pass 45 us |
// normalisation 6 us -- three dependent launches that stream the basis twice, write w, read it back and write it again.  Here the three become one:
//   * one 512-thread workgroup per CU owns a block of <= 8192 rows; w = aa #> q_i of its rows lives in REGISTERS from the first pass to the end;
//   * pass 1 streams the basis once (hhcoli = q_k <.> w, :655) and KEEPS the first five columns of the block on chip -- three in registers, two in LDS;
//   * a grid-wide synchronisation carries the per-workgroup partial dot products (XCD-hierarchical arrival counters, as sla_onchip.hip); every
//     workgroup folds them in the same fixed order: identical h everywhere;
//   * pass 2 (w := w ^-^ sum_k h_k q_k, :657-658, one column after the other in ascending order) reads only the columns that were not kept;
//   * a second synchronisation carries ||w||^2; every workgroup normalises its rows of w into q_{i+1} (:659-664) straight from the registers;
//     workgroup 0 writes the H column, h_{i+1,i}, the step count and the breakdown flag (:665-667) exactly as arn_normalize_kernel does.
// Bytes per step at 16 basis columns: 280 + 296 + 32 MB in three launches -> 280 + 184 MB here.  Measured (GMRES(30), 2 M-row banded matrix, same box,
// tools/arn_orth_ab.sh, profiles/r06_ab_arn_orth.txt): 7400 -> 8515 Arnoldi steps / s.  Shapes tried: four columns in registers (18 registers spilled
// around the reductions: 8160), two (8240), two register sets for the streamed columns (26 - 54 spilled: 7440 - 7880), 1024 threads x 4 row pairs (8200),
// 768 x 6 (7600).  The arithmetic is the launch flow's (same products, the
// columns subtracted in the same order); the inner products are grouped per thread / wavefront / workgroup / grid instead of per grid-stride loop, so
// H and Q agree with the launch flow to rounding (the parity tests compare both with the oracle at 1e-10 / 1e-9).
// Single-rank contexts, n <= 8192 x CUs rows; anything else -- and a launch whose workgroups cannot all be resident -- takes the launch flow.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "sla_internal.hpp"
#include "sla_device.hpp"

namespace sla {

namespace {

#ifndef SLA_AO_T
#define SLA_AO_T 512
#endif
#ifndef SLA_AO_APT
#define SLA_AO_APT 8
#endif
#ifndef SLA_AO_CR
#define SLA_AO_CR 3
#endif
constexpr int AT = SLA_AO_T;     // threads per workgroup (one workgroup per CU)
constexpr int ANW = AT / 64;
constexpr int APT = SLA_AO_APT;  // row PAIRS per thread: 2 APT AT rows per workgroup (8192)
constexpr int ACR = SLA_AO_CR;   // basis columns of the block kept in registers between the passes
constexpr int ACL = 2;           // ... and in LDS
#ifndef SLA_AO_PIPE
#define SLA_AO_PIPE 0
#endif
constexpr bool kAoPipe = SLA_AO_PIPE != 0;   // two register sets for the streamed columns (the next column's loads issued before the current one is folded): spills 26 - 54
                                             // registers at every shape tried and loses 4 - 12 % to the single set -- an A/B switch, off
constexpr int kArnOrthCols = 32; // columns one launch handles (GMRES(30): 31 basis columns at most)

typedef double ao_f64x2 __attribute__((ext_vector_type(2)));

struct ArnOrthArgs {
    int64_t n;
    const double *Q;
    int64_t ldq;
    int ncols;
    const double *w;
    double *qnext;
    double *Hcol;      // H[0 .. ncols-1, i]
    double *hsub;      // H[i + 1, i]
    SolverScalars *sc;
    double *parts;     // [kArnOrthCols + 1][G]
    unsigned *bar;     // arrival counters + the epoch word (zeroed once, at allocation)
    int first;
    int R;             // rows per workgroup (even)
    int nt;            // pass 2 reads the streamed columns non-temporally (the basis overflows the memory-side cache)
    int fault;         // test hook (option arn_orth_fault): the last workgroup leaves at once -- the others' barrier times out (a lost CU, rehearsed)
};

__device__ __forceinline__ void ao_st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ao_ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Grid-wide arrival barrier (the counters of sla_onchip.hip's oc_grid_sync; the sums travel next to it, published by the caller with agent-scope
// stores BEFORE the call).  Epochs run on across launches -- the counters are never reset: a launch reads the epoch word, uses epoch + 1 and + 2 and
// workgroup 0 stores epoch + 2 back behind the second barrier.  Returns false after ~2 s without the other workgroups (SLA_FLAG_SYNC_TIMEOUT).
__device__ __forceinline__ bool ao_grid_sync(unsigned *bar, unsigned epoch, SolverScalars *sc, int *s_ok) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned Gu = gridDim.x, g = blockIdx.x & 7, ng = Gu < 8 ? Gu : 8;
        const unsigned members = (Gu - g + 7) / 8;
        unsigned *cnt = bar + 32 * g, *top = bar + 32 * 8, *gen = bar + 32 * (9 + g);
        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == epoch * members) {
            const unsigned t = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t + 1 == epoch * ng)
                for (unsigned j = 0; j < ng; ++j) __hip_atomic_store(bar + 32 * (9 + j), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int ok = 1;
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 200000000ll) {
                __hip_atomic_fetch_or(&sc->flags, (int)SLA_FLAG_SYNC_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        *s_ok = ok;
    }
    __syncthreads();
    return *s_ok != 0;
}
// the sum of the G (<= 256) published partials of one quantity, by one wavefront, in a fixed order (lane j: parts j, j + 64, j + 128, j + 192; butterfly)
__device__ __forceinline__ double ao_wave_total(const double *parts, int G) {
    const int l = threadIdx.x & 63;
    double v[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) v[m] = l + 64 * m < G ? ao_ld_agent(parts + l + 64 * m) : 0.0;
    double s = ((v[0] + v[1]) + v[2]) + v[3];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}

__global__ void __launch_bounds__(AT) arn_orth_kernel(ArnOrthArgs a) {
    extern __shared__ __attribute__((aligned(16))) double ao_lds[];
    __shared__ int s_ok;
    __shared__ double s_red[(kArnOrthCols + 1) * ANW];
    __shared__ double s_h[kArnOrthCols];
    ao_f64x2 *ql = (ao_f64x2 *)ao_lds;                 // [ACL][APT * AT]
    SolverScalars *sc = a.sc;
    if (arn_stopped(sc)) return;                       // (a flag of an EARLIER launch: every workgroup takes the same exit)
    if (sc->flags & SLA_FLAG_SYNC_TIMEOUT) return;     // (an earlier fused step of this run lost its barrier: the host repeats the run on the launch flow)
    const int b = blockIdx.x, t = threadIdx.x, G = gridDim.x, wave = t >> 6, lane = t & 63;
    if (a.fault && b == G - 1 && G > 1) return;
    const int ncols = a.ncols;
    const unsigned e0 = __hip_atomic_load(a.bar + 32 * 18, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t r0 = (int64_t)b * a.R, r1 = min(a.n, r0 + a.R);
    const int64_t lastpair = (a.ldq >> 1) - 1;         // (columns and w hold ldq >= n doubles, ldq even: every clamped pair is readable)
    uint32_t pr[APT];                                  // this thread's row pairs (rows r0 + 2 (i AT + t), + 1) as pair indices (n < 2^32 rows here)
    uint32_t vm = 0;                                   // bits 2 i / 2 i + 1: the rows of pair i exist in this block
    ao_f64x2 wv[APT];
#pragma unroll
    for (int i = 0; i < APT; ++i) {
        const int64_t row = r0 + 2 * ((int64_t)i * AT + t);
        vm |= (row < r1 ? 1u : 0u) << (2 * i) | (row + 1 < r1 ? 1u : 0u) << (2 * i + 1);
        pr[i] = (uint32_t)min(row >> 1, lastpair);
        const ao_f64x2 x = ((const ao_f64x2 *)a.w)[pr[i]];
        wv[i].x = (vm >> (2 * i)) & 1 ? x.x : 0.0;
        wv[i].y = (vm >> (2 * i + 1)) & 1 ? x.y : 0.0;
    }
    // ---- pass 1: hhcoli = fmap (`dot` aqi) qv.  Columns 0 .. ACR-1 stay in registers, ACR .. ACR+ACL-1 in LDS ----
    ao_f64x2 qc[ACR][APT];
    auto load_col = [&](ao_f64x2 (&q)[APT], int c) {
        const ao_f64x2 *col = (const ao_f64x2 *)(a.Q + (int64_t)c * a.ldq);
#pragma unroll
        for (int i = 0; i < APT; ++i) q[i] = col[pr[i]];
    };
    auto dot_col = [&](const ao_f64x2 (&q)[APT], int c) {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            acc += q[i].x * wv[i].x;
            acc += q[i].y * wv[i].y;
        }
        acc = wave_sum(acc);
        if (lane == 0) s_red[c * ANW + wave] = acc;
    };
#pragma unroll
    for (int c = 0; c < ACR; ++c) {                    // (all of them issued back to back: they stay anyway)
        if (c < ncols) {
            load_col(qc[c], c);
        } else {
#pragma unroll
            for (int i = 0; i < APT; ++i) qc[c][i] = ao_f64x2{0.0, 0.0};
        }
    }
#pragma unroll
    for (int c = 0; c < ACR; ++c)
        if (c < ncols) dot_col(qc[c], c);
    auto keep_col = [&](const ao_f64x2 (&q)[APT], int c) {
        if (c < ACR + ACL) {
#pragma unroll
            for (int i = 0; i < APT; ++i) ql[(c - ACR) * (APT * AT) + i * AT + t] = q[i];
        }
    };
    if constexpr (kAoPipe) {
        ao_f64x2 qa[APT], qb[APT];
        if (ACR < ncols) load_col(qa, ACR);
        for (int c = ACR; c < ncols; c += 2) {
            if (c + 1 < ncols) load_col(qb, c + 1);
            keep_col(qa, c);
            dot_col(qa, c);
            if (c + 1 < ncols) {
                if (c + 2 < ncols) load_col(qa, c + 2);
                keep_col(qb, c + 1);
                dot_col(qb, c + 1);
            }
        }
    } else {
        for (int c = ACR; c < ncols; ++c) {
            ao_f64x2 q[APT];
            load_col(q, c);
            keep_col(q, c);
            dot_col(q, c);
        }
    }
    __syncthreads();
    if (t < ncols) {                                   // the wavefronts' partials of column t in a fixed order -> this workgroup's partial
        double s = s_red[t * ANW];
#pragma unroll
        for (int k = 1; k < ANW; ++k) s += s_red[t * ANW + k];
        ao_st_agent(a.parts + (size_t)t * G + b, s);
    }
    if (!ao_grid_sync(a.bar, e0 + 1, sc, &s_ok)) return;
    {   // every workgroup the same sums in the same order: identical h.  A wavefront folds columns wave, wave + ANW, ...: ALL their partials are
        // requested before the first add (one round trip to the memory side, not one per column)
        constexpr int kMine = (kArnOrthCols + ANW - 1) / ANW;
        double v[kMine][4];
#pragma unroll
        for (int k = 0; k < kMine; ++k) {
            const int c = wave + ANW * k;
#pragma unroll
            for (int m = 0; m < 4; ++m) v[k][m] = (c < ncols && lane + 64 * m < G) ? ao_ld_agent(a.parts + (size_t)c * G + lane + 64 * m) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < kMine; ++k) {
            const int c = wave + ANW * k;
            double s = ((v[k][0] + v[k][1]) + v[k][2]) + v[k][3];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            if (c < ncols && lane == 0) s_h[c] = s;
        }
    }
    __syncthreads();
    if (b == 0 && t < ncols) a.Hcol[t] = s_h[t];
    // ---- pass 2: qipnn = aqi ^-^ foldl' (^+^) (zipWith (.*) hhcoli qv), one column after the other; partial ||qipnn||^2 ----
    ao_f64x2 acc2[APT];
#pragma unroll
    for (int i = 0; i < APT; ++i) acc2[i] = ao_f64x2{0.0, 0.0};
#pragma unroll
    for (int c = 0; c < ACR; ++c) {
        if (c < ncols) {
            const double h = s_h[c];
#pragma unroll
            for (int i = 0; i < APT; ++i) {
                acc2[i].x += h * qc[c][i].x;
                acc2[i].y += h * qc[c][i].y;
            }
        }
    }
    auto axpy_col = [&](const ao_f64x2 (&q)[APT], int c) {
        const double h = s_h[c];
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            acc2[i].x += h * q[i].x;
            acc2[i].y += h * q[i].y;
        }
    };
    auto load_col2 = [&](ao_f64x2 (&q)[APT], int c) {   // the columns that were not kept: from memory again
        const ao_f64x2 *col = (const ao_f64x2 *)(a.Q + (int64_t)c * a.ldq);
        if (a.nt) {
#pragma unroll
            for (int i = 0; i < APT; ++i) q[i] = __builtin_nontemporal_load(col + pr[i]);
        } else {
#pragma unroll
            for (int i = 0; i < APT; ++i) q[i] = col[pr[i]];
        }
    };
    if constexpr (kAoPipe) {
        ao_f64x2 qa[APT], qb[APT];
        if (ACR + ACL < ncols) load_col2(qa, ACR + ACL);    // (in flight behind the LDS columns)
        for (int c = ACR; c < ncols && c < ACR + ACL; ++c) {
#pragma unroll
            for (int i = 0; i < APT; ++i) qb[i] = ql[(c - ACR) * (APT * AT) + i * AT + t];
            axpy_col(qb, c);
        }
        for (int c = ACR + ACL; c < ncols; c += 2) {
            if (c + 1 < ncols) load_col2(qb, c + 1);
            axpy_col(qa, c);
            if (c + 1 < ncols) {
                if (c + 2 < ncols) load_col2(qa, c + 2);
                axpy_col(qb, c + 1);
            }
        }
    } else {
        for (int c = ACR; c < ncols; ++c) {
            ao_f64x2 q[APT];
            if (c < ACR + ACL) {
#pragma unroll
                for (int i = 0; i < APT; ++i) q[i] = ql[(c - ACR) * (APT * AT) + i * AT + t];
            } else {
                load_col2(q, c);
            }
            axpy_col(q, c);
        }
    }
    double nrm = 0.0;
#pragma unroll
    for (int i = 0; i < APT; ++i) {
        wv[i].x = (vm >> (2 * i)) & 1 ? wv[i].x - acc2[i].x : 0.0;    // (slots past the block's last row read a neighbour's or a clamped pair: they stay zero)
        wv[i].y = (vm >> (2 * i + 1)) & 1 ? wv[i].y - acc2[i].y : 0.0;
        nrm += wv[i].x * wv[i].x;
        nrm += wv[i].y * wv[i].y;
    }
    nrm = wave_sum(nrm);
    if (lane == 0) s_red[kArnOrthCols * ANW + wave] = nrm;
    __syncthreads();
    if (t == 0) {
        double s = s_red[kArnOrthCols * ANW];
#pragma unroll
        for (int k = 1; k < ANW; ++k) s += s_red[kArnOrthCols * ANW + k];
        ao_st_agent(a.parts + (size_t)kArnOrthCols * G + b, s);
    }
    if (!ao_grid_sync(a.bar, e0 + 2, sc, &s_ok)) return;
    // ---- qip = normalize2 qipnn ; h_{i+1,i} = norm2' qipnn ; breakdown = nearZero (:659-667), as arn_normalize_kernel ----
    const double nn = sqrt(ao_wave_total(a.parts + (size_t)kArnOrthCols * G, G));
    const double inv = 1.0 / nn;
    if (b == 0 && t == 0) {
        if (a.hsub) *a.hsub = nn;
        sc->hnorm = nn;
        if (a.hsub) sc->kdone += 1;
        if (!a.first && fabs(nn) <= 1e-12) sc->flags |= SLA_FLAG_BREAKDOWN;
        __hip_atomic_store(a.bar + 32 * 18, e0 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int i = 0; i < APT; ++i) {
        const int64_t row = r0 + 2 * ((int64_t)i * AT + t);
        if ((vm >> (2 * i + 1)) & 1) {
            ((ao_f64x2 *)a.qnext)[row >> 1] = ao_f64x2{inv * wv[i].x, inv * wv[i].y};
        } else if ((vm >> (2 * i)) & 1) {
            a.qnext[row] = inv * wv[i].x;
        }
    }
}

constexpr size_t kArnOrthLds = sizeof(double) * 2 * (size_t)ACL * APT * AT;   // 128 KiB of dynamic LDS
constexpr size_t kArnOrthBarWords = 32 * 20;

}  // namespace

size_t arn_orth_bar_bytes() { return sizeof(unsigned) * kArnOrthBarWords; }

// Does the fused step apply?  A single-rank context, option arn_orth on, a block of <= 8192 rows per CU, at most 32 columns, the basis laid out for
// 16-byte loads, and the kernel resident with one workgroup per CU (asked once per context).
bool arn_orth_usable(sla_ctx *c, int64_t n, int64_t ldq, int ncols_max) {
    if (c->arn_orth == 0 || c->collectives || c->nranks != 1) return false;
    if (n < 2 * AT || (ldq & 1) || ncols_max > kArnOrthCols) return false;
    if (n > (int64_t)c->n_cu * 2 * APT * AT) return false;
    if (c->arn_orth_state == 0) {
        c->arn_orth_state = -1;
        int per_cu = 0;
        const void *kern = (const void *)arn_orth_kernel;
        if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kArnOrthLds) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, AT, kArnOrthLds) == hipSuccess && per_cu >= 1)
            c->arn_orth_state = 1;
        else
            (void)hipGetLastError();
    }
    return c->arn_orth_state > 0;
}

int launch_arn_orth(sla_ctx *c, int64_t n, const double *Q, int64_t ldq, int ncols, const double *w, double *qnext, double *Hcol, double *hsub,
                    SolverScalars *sc, double *parts, unsigned *bar, int first) {
    // rows per workgroup: the chip's CUs share the rows evenly (an even count: 16-byte row pairs), never more than the registers hold
    int64_t R = (n + c->n_cu - 1) / c->n_cu;
    R = std::max<int64_t>(R + (R & 1), 2 * AT);
    const int G = (int)((n + R - 1) / R);
    if (R > 2 * APT * AT || G > c->n_cu || ncols < 1 || ncols > kArnOrthCols) return fail(SLA_ERR_INVALID, "launch_arn_orth: outside the fused step's range");
    ArnOrthArgs a{};
    a.n = n; a.Q = Q; a.ldq = ldq; a.ncols = ncols; a.w = w; a.qnext = qnext; a.Hcol = Hcol; a.hsub = hsub; a.sc = sc; a.parts = parts; a.bar = bar;
    a.first = first;
    a.R = (int)R;
    a.fault = c->arn_orth_fault;
    a.nt = c->vec_nt < 0 ? ((int64_t)ncols * 8 * n > c->mall_bytes ? 1 : 0) : (c->vec_nt != 0 ? 1 : 0);
    hipLaunchKernelGGL(arn_orth_kernel, dim3(G), dim3(AT), kArnOrthLds, stream_of(c), a);
    SLA_HIP_TRY(hipGetLastError());
    c->arn_orth_launches += 1;
    return SLA_OK;
}

}  // namespace sla
