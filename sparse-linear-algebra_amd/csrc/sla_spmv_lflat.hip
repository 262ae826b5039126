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
//   * a workgroup owns 16384 rows and a RANGE of panels; a row's running sum stays in a register of its lane while the workgroup
//     walks the panels of the range (x panel after x panel through LDS), so the products of a row are added one after the other
//     across panels; each range leaves ONE partial per row, and lpanel_finish_kernel (shared with the LDS-panel form) adds a
//     row's partials in ascending range order and runs the fused epilogue.  Summation order: one left fold per (row, panel range),
//     then a left fold of the <= ~9 range sums -- a regrouping of the reference's single left fold (Common.hs:247-260) like every
//     GPU form for long rows: |dy_i| <= nnz_i eps sum_j |a_ij x_j|, checked per row in tests/test_gpu_lds_panels.py;
//     with a single range (few row chunks more than CUs) it IS the reference's left fold.
// HBM bytes per (#>): 10 B per entry + 4 B per segment + 16 B per (row, range) partial; the x panels (n x 8 B per 16384-row chunk)
// come out of the L2 / the memory-side cache.  Taken when a segment holds 1.5 .. 16 entries on average.
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

constexpr int kLfW = 15360;                 // columns of x per panel: 120 KiB of LDS, which leaves 32 KiB for the wavefronts' product stages
constexpr int kLfStage = 256;               // products a wavefront stages per chunk (2 KiB; 16 wavefronts)
constexpr int kLfRpt = 16;                  // rows per thread: a workgroup owns 16384 rows and keeps their running sums in registers
constexpr size_t kLfLds = (size_t)kLfW * 8 + (size_t)(kLpBlock / 64) * kLfStage * 8;

// Two versions were measured and replaced (100 / 200 entries per row, n = 1 M; tile form: 0.53 / 0.91 ms):
//  v1  every lane LOADS its segment's entries itself (addresses ~3 entries apart across the lanes): each wave-load touches ~14 cache
//      lines instead of 4 and the L1 bounds the kernel: 0.62 / 1.21 ms;
//  v2  the wavefront streams its contiguous stretch coalesced, stages the products in LDS, lanes fold their segments from there;
//      tasks = (panel, row chunk), one partial sum per (panel, row) through HBM (16 B x 66 panels per row): 0.59 / 0.74 ms -- the
//      partials (1.06 GB) and the segment starts (0.26 GB) weigh as much as the entries (1.0 / 2.0 GB).
// v3 (this): a workgroup owns 16384 ROWS and walks a RANGE of panels with the running sums of its rows in registers (16 per thread):
// a row's products are added one after the other across the panels of the range -- no per-panel partials at all; the panel ranges
// (as many as it takes to fill the chip: tasks = row chunks x ranges) leave one partial each, which lpanel_finish_kernel folds.
__global__ void __launch_bounds__(kLpBlock) spmv_lflat_kernel(const uint32_t *__restrict__ q, const uint16_t *__restrict__ col16,
                                                              const double *__restrict__ val, const double *__restrict__ xg, double *__restrict__ ypart,
                                                              int rows, int n, int W, int P, int PR, int col_lo, int col_hi, const SolverScalars *sc) {
    extern __shared__ double lf_lds[];
    double *lf_xs = lf_lds;
    if (sc && sc->done) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *stage = lf_lds + kLfW + wave * kLfStage;
    constexpr int R = kLpBlock * kLfRpt;
    const int nchunks = (rows + R - 1) / R;
    const int cidx = (int)blockIdx.x % nchunks, sidx = (int)blockIdx.x / nchunks;   // (row chunk, panel range) of this workgroup
    const int lo = cidx * R, hi = min(rows, lo + R);
    const int p0 = sidx * PR, p1 = min(P, p0 + PR);
    double acc[kLfRpt];
#pragma unroll
    for (int r = 0; r < kLfRpt; ++r) acc[r] = 0.0;
    for (int p = p0; p < p1; ++p) {
        const int w0 = p * W;
        __syncthreads();
        {
            const int wn = min(W, n - w0);
            // only [col_lo, col_hi] is referenced by these rows -- and, on a row slab gathering from its in-place halo window, the only
            // part of x that is backed by memory at all
            for (int j = tid; j < wn; j += kLpBlock) lf_xs[j] = (w0 + j >= col_lo && w0 + j <= col_hi) ? xg[w0 + j] : 0.0;
        }
        __syncthreads();
        const uint32_t *qs = q + (int64_t)p * rows;
        // Latency, not bandwidth, bounds a trip (segment starts -> the stretch's streams -> LDS stage -> fold: three dependent round trips,
        // and the 120 KiB panel leaves room for ONE workgroup = 16 wavefronts per CU to hide them: v3 without the two measures below ran
        // 100 entries per row no faster than 200, 0.55 vs 0.59 ms).  So (1) the segment starts of all sixteen trips of a panel are loaded
        // up front, (2) the streams of the NEXT chunk -- of this trip or of the next one -- are issued before the current chunk is folded.
        uint32_t ks[kLfRpt], kend[kLfRpt];   // per lane: its segment's start; per wavefront (scalar): where its 64 segments end
#pragma unroll
        for (int r = 0; r < kLfRpt; ++r) {
            const int b = lo + r * kLpBlock + wave * 64;
            ks[r] = qs[min(b + lane, hi)];   // (rows past the chunk: empty segments at qs[hi])
            kend[r] = qs[min(b + 64, hi)];
        }
        uint16_t cj[kLfStage / 64], cn[kLfStage / 64];
        double vj[kLfStage / 64], vn[kLfStage / 64];
        auto load_chunk = [&](uint16_t (&cc)[kLfStage / 64], double (&vv)[kLfStage / 64], uint32_t ca, uint32_t kb) {
            const uint32_t last = max(min(ca + (uint32_t)kLfStage, kb), ca + 1u) - 1u;   // (clamped, unconditional; an empty stretch reads the entry at its start)
#pragma unroll
            for (int u = 0; u < kLfStage / 64; ++u) {
                const uint32_t idx = min(ca + (uint32_t)(lane + 64 * u), last);
                cc[u] = __builtin_nontemporal_load(col16 + idx);
                vv[u] = __builtin_nontemporal_load(val + idx);
            }
        };
        load_chunk(cj, vj, (uint32_t)__builtin_amdgcn_readfirstlane((int)ks[0]), kend[0]);
#pragma unroll
        for (int r = 0; r < kLfRpt; ++r) {   // trip r: the wavefront's 64 consecutive segments (p, base .. base + 63)
#pragma clang fp contract(off)  // a*x then +: two roundings like the reference, never an FMA
            const uint32_t k = ks[r], kb = kend[r];
            uint32_t e = (uint32_t)__shfl_down((int)k, 1, 64);   // a segment ends where the next lane's starts; the last one at kb
            if (lane == 63) e = kb;
            const uint32_t ka = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
            const uint32_t nka = r + 1 < kLfRpt ? (uint32_t)__builtin_amdgcn_readfirstlane((int)ks[r + 1 < kLfRpt ? r + 1 : r]) : ka;
            const uint32_t nkb = r + 1 < kLfRpt ? kend[r + 1 < kLfRpt ? r + 1 : r] : ka;
            for (uint32_t ca = ka;; ca += kLfStage) {   // chunks of the wavefront's contiguous stretch [ka, kb) (at least one trip through: an empty stretch folds nothing)
                const uint32_t cb = min(ca + (uint32_t)kLfStage, kb);
                const bool more = ca + kLfStage < kb;
                load_chunk(cn, vn, more ? ca + kLfStage : nka, more ? kb : nkb);
#pragma unroll
                for (int u = 0; u < kLfStage / 64; ++u) stage[lane + 64 * u] = vj[u] * lf_xs[cj[u]];
                // this lane's segment inside the chunk, four staged products read together, added in order
                int la = (int)(max(k, ca) - ca);
                const int ha = (int)min(e, cb) - (int)ca;  // (a segment that ends before the chunk: negative, nothing to add)
                while (__builtin_amdgcn_ballot_w64(la < ha) != 0) {
                    double pj[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) pj[u] = stage[min(max(la + u, 0), kLfStage - 1)];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (la + u < ha) acc[r] = acc[r] + pj[u];
                    la += 4;
                }
#pragma unroll
                for (int u = 0; u < kLfStage / 64; ++u) { cj[u] = cn[u]; vj[u] = vn[u]; }
                if (!more) break;
            }
        }
    }
    double *yp = ypart + (int64_t)sidx * rows;
#pragma unroll
    for (int r = 0; r < kLfRpt; ++r) {
        const int i = lo + r * kLpBlock + tid;
        if (i < hi) yp[i] = acc[r];
    }
}

}  // namespace

bool lflat_on(const sla_csr *A) { return A->use_lflat && A->ctx->lflat && A->ctx->spmv_algo == 0; }

// Lowering: after the canonical arrays are on the device.  No-op unless the structure calls for the form (see the header).
// what can be said without looking at an entry (csr_upload asks before it spends a pass over the rows on the column range)
bool lflat_candidate(const sla_csr *A, int64_t n, int64_t rows) {
    const sla_ctx *c = A->ctx;
    const int64_t nnz = A->nnz;
    if (!c->lflat || A->rp64 || rows <= 0 || nnz <= 0 || nnz >= ((int64_t)1 << 31)) return false;
    if (A->use_lpanel && c->lpanel) return false;                                                                        // dense rows: LDS panels
    if (c->lflat < 2 && (A->use_wdia || A->use_vdict || A->use_diag || A->xwin_fraction >= 0.5)) return false;         // stencil / banded structure (lflat = 2: test hook, any structure)
    const int64_t P = (n + kLfW - 1) / kLfW;
    const int64_t nseg = P * rows;
    // mean segment length: from lf_min_seg10 / 10 (1.5: below that the partials cost more than the entries) up to the LDS-panel form's threshold.
    // Round 5: where the CU-wide tile form applies (relaxed order allowed, x beyond the L2) it takes these matrices instead: 100 per row
    // 0.36 -> 0.52 of peak, 200 per row 0.61 -> 0.65; at 500 per row the two are within 3 % of each other at 1 M rows (0.754 here, 0.734
    // there) and the tile form is 8 % ahead at 600 k rows (profiles/r05_form_tournament.txt, tests/test_gpu_form_choice.py) -- one form
    // less in the default ladder.  This form stays for n <= 2^18 columns, for tile_relaxed = 0 / tiles = 0 and when forced (lflat = 2).
    if (c->tiles && c->tile_relaxed && c->lflat < 2 && n > ((int64_t)1 << 18)) return false;
    const int64_t min10 = c->lf_min_seg10;
    return !(P < 3 || P > 4096 || nseg >= ((int64_t)1 << 31) || nnz * 10 < min10 * nseg || (c->lflat < 2 && nnz >= (int64_t)c->lp_min_seg * nseg));
}

int build_lflat(sla_csr *A, int64_t n, int64_t rows, int64_t col_lo, int64_t col_hi) {
    sla_ctx *c = A->ctx;
    const int64_t nnz = A->nnz;
    if (!lflat_candidate(A, n, rows) || !A->d_col || !A->d_val || !A->d_rowptr) return SLA_OK;
    {   // one workgroup keeps a panel of x in 128 KiB of LDS
        int lds = 0;
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device) != hipSuccess || (size_t)lds < kLfLds) return SLA_OK;
    }
    const int64_t P = (n + kLfW - 1) / kLfW;
    const int64_t W = std::min<int64_t>(kLfW, ((n + P - 1) / P + 63) / 64 * 64);   // equal panels
    const int64_t nseg = P * rows;
    hipStream_t st = stream_of(c);
    DevBuf d_len, d_tmp;
    uint32_t *dq = nullptr;
    uint16_t *c2 = nullptr;
    double *v2 = nullptr, *yp = nullptr;
    auto give_up = [&]() {   // (out of device memory for the copy: not an error, the other forms serve)
        (void)hipGetLastError();
        if (dq) (void)hipFree(dq);
        if (c2) (void)hipFree(c2);
        if (v2) (void)hipFree(v2);
        if (yp) (void)hipFree(yp);
        return SLA_OK;
    };
    hipError_t e = d_len.alloc(4 * (size_t)(nseg + 1));
    if (e == hipSuccess) e = hipMemsetAsync(d_len.p, 0, 4 * (size_t)(nseg + 1), st);
    if (e == hipSuccess) e = dev_malloc(c, (void **)&dq, 4 * (size_t)(nseg + 1) + kArraySlack);
    if (e == hipSuccess) e = dev_malloc(c, (void **)&c2, 2 * (size_t)nnz + kArraySlack);
    if (e == hipSuccess) e = dev_malloc(c, (void **)&v2, 8 * (size_t)nnz + kArraySlack);
    if (e == hipSuccess) e = dev_malloc(c, (void **)&yp, 8);   // (sized below, once the panel ranges are known)
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
    // tasks = (row chunk of 16384 rows) x (range of PR panels): as many ranges as it takes to give every CU about two tasks
    const int64_t R = (int64_t)kLpBlock * kLfRpt, nchunks = (rows + R - 1) / R;
    const int64_t want_ranges = std::max<int64_t>(1, std::min<int64_t>(P, (2 * (int64_t)c->n_cu + nchunks - 1) / nchunks));
    const int64_t PR = (P + want_ranges - 1) / want_ranges, SX = (P + PR - 1) / PR;
    if (nchunks * SX >= ((int64_t)1 << 31)) return give_up();
    if (hipStreamSynchronize(st) != hipSuccess) return give_up();
    (void)hipFree(yp);
    yp = nullptr;
    e = dev_malloc(c, (void **)&yp, 8 * (size_t)(SX * rows));
    if (e != hipSuccess) return give_up();
    A->lp_col_lo = (int32_t)col_lo;   // smallest / largest column these rows reference: what the panel loads may read (a sharded x is only
    A->lp_col_hi = (int32_t)col_hi;   // readable on its slab + halo)
    A->d_lfq = dq;
    A->d_lfcol = c2;
    A->d_lfval = v2;
    if (A->d_lpy) (void)hipFree(A->d_lpy);
    A->d_lpy = yp;
    A->lp_G = (int32_t)(nchunks * SX);   // workgroups = tasks
    A->lp_P = (int32_t)SX;               // partials per row (what the finish kernel folds)
    A->lp_W = (int32_t)W;
    A->lp_C = (int32_t)P;                // panels
    A->lp_chunk = (int32_t)PR;           // panels per range
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
                       a.rows, (int)A->n, A->lp_W, A->lp_C, A->lp_chunk, A->lp_col_lo, A->lp_col_hi, (const SolverScalars *)a.sc);
    SLA_HIP_TRY(hipGetLastError());
    return launch_lpanel_finish(A, epi, a, grid);
}

}  // namespace sla
