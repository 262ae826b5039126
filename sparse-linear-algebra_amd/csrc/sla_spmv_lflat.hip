// sla_spmv_lflat.hip -- (#>) for the MIDDLE of the matrix zoo (round 4): ~50 .. 1000 random entries per row and
// far more columns than one LDS panel holds, so that a (row, 16384-column panel) segment has only 1.5 .. 16 entries.  Such matrices
// fell to the L2-gathering forms (tile form: 0.29 .. 0.33 of the 8 TB/s peak on CSR bytes at 100 / 200 entries per row, n = 1 M:
// one 128-byte line moved per 8-byte gather), because the LDS-panel kernel spends a lane GROUP and a memory round trip per segment.
//
// Form `lflat`: the x panel in LDS like spmv_lpanel_kernel (one 1024-thread workgroup per CU; 15360 columns = 120 KiB), but ONE LANE PER SEGMENT:
//   * a panel-major second copy of the entries -- value + 16-bit column offset into the panel, 10 B per entry, the segments of a panel
//     one after the other in row order -- and the segment starts q[panel x rows + row] (4 B per segment); built on the device from
//     the canonical arrays (counts -> rocPRIM exclusive scan -> scatter), nothing crosses PCIe;
//   * a wavefront takes 64 consecutive segments (p, i0 .. i0 + 63) = one contiguous stretch of the copy, which it streams with
//     coalesced loads in chunks of 256 entries, multiplies with x from the LDS panel and stages as products in its own 2 KiB of LDS;
//   * lane l then adds the products of ITS segment (p, i0 + l) one by one in ascending column order (separately rounded multiply and
//     add) and stores the partial sum of (panel, row): 64 consecutive doubles per wavefront, no cross-lane arithmetic at all;
//   * lpanel_finish_kernel (shared with the LDS-panel form) adds a row's partials in ascending panel order and runs the fused
//     epilogue.  Summation order: left folds per (panel, row) segment, then a left fold of the segment sums -- a regrouping of the
//     reference's single left fold (Common.hs:247-260) like every GPU form for long rows: |dy_i| <= nnz_i eps sum_j |a_ij x_j|
//     (SURVEY 8(a) A1), checked per row in tests/test_gpu_lds_panels.py.
// HBM bytes per (#>): 10 B per entry + 4 B per segment + 16 B per (panel, row) partial (written, then read by the finish kernel) --
// the partials are what a long row pays for x living in LDS: taken when a segment holds >= 1.5 entries on average (the partials
// then cost less than the entries), below the LDS-panel form's 16.
#include <hip/hip_runtime.h>

#include <cstring>  // rocPRIM's texture iterator calls the host memset without including it

#include <rocprim/rocprim.hpp>

#include "sla_internal.hpp"
#include "sla_device.hpp"

namespace sla {

namespace {

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <class T> T *as() { return (T *)p; }
};

// len[p * rows + i] = entries of row i in panel p  (one thread per row; the array was zeroed)
template <typename RP>
__global__ void __launch_bounds__(256) lf_count_kernel(int64_t rows, const RP *__restrict__ rowptr, const int32_t *__restrict__ col, int W, uint32_t *len) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows; i += (int64_t)gridDim.x * 256) {
        int prev = -1;
        uint32_t cnt = 0;
        for (RP k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            const int p = col[k] / W;
            if (p != prev) {
                if (prev >= 0) len[(int64_t)prev * rows + i] = cnt;
                prev = p;
                cnt = 0;
            }
            ++cnt;
        }
        if (prev >= 0) len[(int64_t)prev * rows + i] = cnt;
    }
}

// the entries into their segments (one thread per row; q = exclusive scan of len)
template <typename RP>
__global__ void __launch_bounds__(256) lf_scatter_kernel(int64_t rows, const RP *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                          const double *__restrict__ val, int W, const uint32_t *__restrict__ q, uint16_t *col2,
                                                          double *val2) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows; i += (int64_t)gridDim.x * 256) {
        int prev = -1;
        uint32_t d = 0;
        for (RP k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            const int p = col[k] / W;
            if (p != prev) {
                prev = p;
                d = q[(int64_t)p * rows + i];
            }
            col2[d] = (uint16_t)(col[k] - p * W);
            val2[d] = val[k];
            ++d;
        }
    }
}

// weight (entries + row_cost per segment) of every (panel, row chunk) task, for the equal-weight task runs of the persistent grid
__global__ void __launch_bounds__(256) lf_task_weights_kernel(int64_t ntasks, int64_t C, int64_t rows, int64_t chunk, const uint32_t *__restrict__ q,
                                                               unsigned long long *w) {
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < ntasks; t += (int64_t)gridDim.x * 256) {
        const int64_t p = t / C, cc = t - p * C, lo = cc * chunk, hi = min(rows, lo + chunk);
        w[t] = (unsigned long long)(q[p * rows + hi] - q[p * rows + lo]);
    }
}

constexpr int kLfW = 15360;                 // columns of x per panel: 120 KiB of LDS, which leaves 32 KiB for the wavefronts' product stages
constexpr int kLfStage = 256;               // products a wavefront stages per chunk (2 KiB; 16 wavefronts)
constexpr size_t kLfLds = (size_t)kLfW * 8 + (size_t)(kLpBlock / 64) * kLfStage * 8;

// A first version let every lane LOAD its segment's entries itself (addresses ~3 entries apart across the lanes): each wave-load then
// touches ~14 cache lines instead of 4 and the L1 bounds the kernel -- 100 / 200 entries per row, n = 1 M: 0.62 / 1.21 ms, slower than
// the L2-gathering tile form (0.53 / 0.91 ms).  So the wavefront STREAMS its contiguous stretch of the copy with coalesced loads (lane l:
// entries l, l + 64, ...), multiplies with x from the LDS panel, stages the PRODUCTS in its own 2 KiB of LDS, and the lane-per-segment
// left fold reads them from there -- the structure of spmv_wave_kernel with the x gather served by LDS.
__global__ void __launch_bounds__(kLpBlock) spmv_lflat_kernel(const uint32_t *__restrict__ q, const uint16_t *__restrict__ col16,
                                                              const double *__restrict__ val, const double *__restrict__ xg, double *__restrict__ ypart,
                                                              const int32_t *__restrict__ task_begin, int rows, int n, int W, int chunk_rows, int C,
                                                              int col_lo, int col_hi, const SolverScalars *sc) {
    extern __shared__ double lf_lds[];
    double *lf_xs = lf_lds;
    if (sc && sc->done) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *stage = lf_lds + kLfW + wave * kLfStage;
    const int t0 = task_begin[blockIdx.x], t1 = task_begin[blockIdx.x + 1];
    int curp = -1;
    for (int t = t0; t < t1; ++t) {
        const int p = t / C, c = t - p * C;
        const int w0 = p * W;
        if (p != curp) {
            __syncthreads();
            const int wn = min(W, n - w0);
            // only [col_lo, col_hi] is referenced by these rows -- and, on a row slab gathering from its in-place halo window, the only
            // part of x that is backed by memory at all
            for (int j = tid; j < wn; j += kLpBlock) lf_xs[j] = (w0 + j >= col_lo && w0 + j <= col_hi) ? xg[w0 + j] : 0.0;
            __syncthreads();
            curp = p;
        }
        const int lo = c * chunk_rows, hi = min(rows, lo + chunk_rows);
        const uint32_t *qs = q + (int64_t)p * rows;
        double *yp = ypart + (int64_t)p * rows;
        for (int base = lo + wave * 64; base < hi; base += kLpBlock) {   // 64 consecutive segments per wavefront and trip
#pragma clang fp contract(off)  // a*x then +: two roundings like the reference, never an FMA
            const int i = base + lane;
            const bool has = i < hi;
            const uint32_t k = qs[has ? i : hi], e = qs[has ? i + 1 : hi];
            const uint32_t ka = (uint32_t)__builtin_amdgcn_readfirstlane((int)k), kb = (uint32_t)__builtin_amdgcn_readlane((int)e, 63);
            double acc = 0.0;
            for (uint32_t ca = ka; ca < kb; ca += kLfStage) {   // chunks of the wavefront's contiguous stretch [ka, kb)
                const uint32_t cb = min(ca + (uint32_t)kLfStage, kb);
                uint16_t cj[kLfStage / 64];
                double vj[kLfStage / 64];
#pragma unroll
                for (int u = 0; u < kLfStage / 64; ++u) {
                    const uint32_t idx = min(ca + (uint32_t)(lane + 64 * u), cb - 1);   // (clamped, unconditional: cb > ca)
                    cj[u] = __builtin_nontemporal_load(col16 + idx);
                    vj[u] = __builtin_nontemporal_load(val + idx);
                }
#pragma unroll
                for (int u = 0; u < kLfStage / 64; ++u) stage[lane + 64 * u] = vj[u] * lf_xs[cj[u]];
                // this lane's segment inside the chunk, four staged products read together, added in order
                int la = (int)(max(k, ca) - ca);
                const int ha = (int)(min(e, cb) - ca);     // (e < ca: negative, nothing to add)
                while (__builtin_amdgcn_ballot_w64(la < ha) != 0) {
                    double pj[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) pj[u] = stage[min(max(la + u, 0), kLfStage - 1)];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (la + u < ha) acc = acc + pj[u];
                    la += 4;
                }
            }
            if (has) yp[i] = acc;
        }
    }
}

}  // namespace

bool lflat_on(const sla_csr *A) { return A->use_lflat && A->ctx->lflat && A->ctx->spmv_algo == 0; }

// Lowering: after the canonical arrays are on the device.  No-op unless the structure calls for the form (see the header).
int build_lflat(sla_csr *A, int64_t n, int64_t rows, int64_t col_lo, int64_t col_hi) {
    sla_ctx *c = A->ctx;
    const int64_t nnz = A->nnz;
    if (!c->lflat || A->rp64 || rows <= 0 || nnz <= 0 || nnz >= ((int64_t)1 << 31) || !A->d_col || !A->d_val || !A->d_rowptr) return SLA_OK;
    if (A->use_wdia || A->use_vdict || A->use_diag || A->xwin_fraction >= 0.5 || (A->use_lpanel && c->lpanel)) return SLA_OK;   // stencil / banded structure, or dense rows (LDS panels)
    {   // one workgroup keeps a panel of x in 128 KiB of LDS
        int lds = 0;
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device) != hipSuccess || (size_t)lds < kLfLds) return SLA_OK;
    }
    const int64_t P = (n + kLfW - 1) / kLfW;
    const int64_t W = std::min<int64_t>(kLfW, ((n + P - 1) / P + 63) / 64 * 64);   // equal panels
    const int64_t nseg = P * rows;
    // mean segment length: from lf_min_seg10 / 10 (1.5: below that the partials cost more than the entries) up to the LDS-panel form's threshold
    if (P < 3 || P > 4096 || nseg >= ((int64_t)1 << 31) || nnz * 10 < (int64_t)c->lf_min_seg10 * nseg || (c->lflat < 2 && nnz >= (int64_t)c->lp_min_seg * nseg)) return SLA_OK;
    hipStream_t st = stream_of(c);
    DevBuf d_len, d_tmp, d_w;
    uint32_t *dq = nullptr;
    uint16_t *c2 = nullptr;
    double *v2 = nullptr, *yp = nullptr;
    int32_t *dt = nullptr;
    auto give_up = [&]() {   // (out of device memory for the copy: not an error, the other forms serve)
        (void)hipGetLastError();
        if (dq) (void)hipFree(dq);
        if (c2) (void)hipFree(c2);
        if (v2) (void)hipFree(v2);
        if (yp) (void)hipFree(yp);
        if (dt) (void)hipFree(dt);
        return SLA_OK;
    };
    hipError_t e = d_len.alloc(4 * (size_t)(nseg + 1));
    if (e == hipSuccess) e = hipMemsetAsync(d_len.p, 0, 4 * (size_t)(nseg + 1), st);
    if (e == hipSuccess) e = dev_malloc(c, (void **)&dq, 4 * (size_t)(nseg + 1) + kArraySlack);
    if (e == hipSuccess) e = dev_malloc(c, (void **)&c2, 2 * (size_t)nnz + kArraySlack);
    if (e == hipSuccess) e = dev_malloc(c, (void **)&v2, 8 * (size_t)nnz + kArraySlack);
    if (e == hipSuccess) e = dev_malloc(c, (void **)&yp, 8 * (size_t)nseg);
    if (e == hipSuccess) e = hipMemsetAsync((char *)c2 + 2 * (size_t)nnz, 0, kArraySlack, st);   // (clamped loads of an empty last segment)
    if (e == hipSuccess) e = hipMemsetAsync((char *)v2 + 8 * (size_t)nnz, 0, kArraySlack, st);
    if (e != hipSuccess) return give_up();
    const int grid = 4096;
    hipLaunchKernelGGL((lf_count_kernel<int32_t>), dim3(grid), dim3(256), 0, st, rows, (const int32_t *)A->d_rowptr, A->d_col, (int)W, d_len.as<uint32_t>());
    size_t tmp_bytes = 0;
    e = rocprim::exclusive_scan(nullptr, tmp_bytes, d_len.as<uint32_t>(), dq, 0u, (size_t)(nseg + 1), rocprim::plus<uint32_t>(), st);
    if (e == hipSuccess) e = d_tmp.alloc(tmp_bytes);
    if (e == hipSuccess) e = rocprim::exclusive_scan(d_tmp.p, tmp_bytes, d_len.as<uint32_t>(), dq, 0u, (size_t)(nseg + 1), rocprim::plus<uint32_t>(), st);
    if (e != hipSuccess) return give_up();
    hipLaunchKernelGGL((lf_scatter_kernel<int32_t>), dim3(grid), dim3(256), 0, st, rows, (const int32_t *)A->d_rowptr, A->d_col, A->d_val, (int)W, dq, c2, v2);
    if (hipGetLastError() != hipSuccess) return give_up();
    // tasks (panel, row chunk), panel-major, dealt out to one workgroup per CU in contiguous runs of equal weight (entries + a cost per
    // segment), like the LDS-panel form; the chunk of a task is a whole number of 1024-row rounds of the workgroup
    const int tasks_per_cu = std::max(1, c->lp_tasks);
    const int64_t want = std::max<int64_t>(1, (tasks_per_cu * (int64_t)c->n_cu + P - 1) / P);
    const int64_t chunk = std::max<int64_t>(kLpBlock, ((rows + want - 1) / want + kLpBlock - 1) / kLpBlock * kLpBlock);
    const int64_t C = (rows + chunk - 1) / chunk, ntasks = P * C;
    if (ntasks >= ((int64_t)1 << 31)) return give_up();
    e = d_w.alloc(8 * (size_t)ntasks);
    if (e != hipSuccess) return give_up();
    hipLaunchKernelGGL(lf_task_weights_kernel, dim3((unsigned)std::min<int64_t>(4096, (ntasks + 255) / 256)), dim3(256), 0, st, ntasks, C, rows, chunk, dq, d_w.as<unsigned long long>());
    std::vector<unsigned long long> w((size_t)ntasks);
    if (hipMemcpyAsync(w.data(), d_w.p, 8 * (size_t)ntasks, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return give_up();
    const int G = (int)std::min<int64_t>(ntasks, c->n_cu);
    std::vector<int64_t> upto((size_t)ntasks + 1, 0);
    for (int64_t t = 0; t < ntasks; ++t) {
        const int64_t cc = t % C, lo = cc * chunk, hi = std::min<int64_t>(rows, lo + chunk);
        upto[(size_t)t + 1] = upto[(size_t)t] + (int64_t)w[(size_t)t] + 2 * (hi - lo);   // (a segment: its two pointer reads and its partial)
    }
    std::vector<int32_t> tb((size_t)G + 1, 0);
    for (int g = 1; g < G; ++g) {
        const int64_t target = upto[(size_t)ntasks] / G * g;
        tb[(size_t)g] = std::max<int32_t>((int32_t)(std::lower_bound(upto.begin(), upto.end(), target) - upto.begin()), tb[(size_t)g - 1]);
    }
    tb[(size_t)G] = (int32_t)ntasks;
    e = dev_malloc(c, (void **)&dt, sizeof(int32_t) * tb.size());
    if (e == hipSuccess) e = hipMemcpy(dt, tb.data(), sizeof(int32_t) * tb.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) return give_up();
    A->lp_col_lo = (int32_t)col_lo;   // smallest / largest column these rows reference: what the panel loads may read (a sharded x is only
    A->lp_col_hi = (int32_t)col_hi;   // readable on its slab + halo)
    A->d_lfq = dq;
    A->d_lfcol = c2;
    A->d_lfval = v2;
    if (A->d_lpy) (void)hipFree(A->d_lpy);
    A->d_lpy = yp;
    if (A->d_lpt) (void)hipFree(A->d_lpt);
    A->d_lpt = dt;
    A->lp_G = G;
    A->lp_P = (int32_t)P;
    A->lp_W = (int32_t)W;
    A->lp_C = (int32_t)C;
    A->lp_chunk = (int32_t)chunk;
    A->use_lflat = true;
    return SLA_OK;
}


int launch_spmv_lflat(const sla_csr *A, int epi, const SpmvArgs<int32_t> &a, int grid) {
    sla_ctx *c = A->ctx;
    if (!(c->lp_attr & (1 << 30))) {   // per context = per device: 128 KiB of dynamic LDS
        SLA_HIP_TRY(hipFuncSetAttribute((const void *)spmv_lflat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLfLds));
        c->lp_attr |= 1 << 30;
    }
    hipLaunchKernelGGL(spmv_lflat_kernel, dim3(A->lp_G), dim3(kLpBlock), kLfLds, stream_of(c), A->d_lfq, A->d_lfcol, A->d_lfval, a.x, A->d_lpy,
                       A->d_lpt, a.rows, (int)A->n, A->lp_W, A->lp_chunk, A->lp_C, A->lp_col_lo, A->lp_col_hi, (const SolverScalars *)a.sc);
    SLA_HIP_TRY(hipGetLastError());
    return launch_lpanel_finish(A, epi, a, grid);
}

}  // namespace sla
