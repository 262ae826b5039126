// sla_lower.cpp -- "lower once": what fromListSM / fromCSR leave on the device for (#>) (Data/Sparse/SpMatrix.hs:180-260, Common.hs:242-260).
// csr_upload validates a rank's row block, uploads the canonical CSR arrays and runs one lowering analysis per storage form
// (diagonal dictionary, value-indexed pairs / wave-sliced records / LDS windows / plane march, variable-coefficient slices, LDS
// panels, column panels; the tile form lives in sla_lower_tiles.cpp); on row-sharded contexts it also derives the halo exchange
// plan and the interior / boundary step lists of the overlapped (#>).  Host code: set-up is off the solver path.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <sched.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <future>
#include <memory>
#include <cstring>
#include <limits>
#include <thread>

#include "sla_internal.hpp"

namespace sla {

// sharded only: all-gather every rank's referenced column window and derive the exchange plan
static int build_xplan(sla_csr *A, int64_t rows, const int64_t *rowptr, const int64_t *col) {
    sla_ctx *c = A->ctx;
    if (!c->collectives) return SLA_OK;
    int64_t w[2] = {1, 0};  // empty
    const int64_t nnz = rowptr[rows];
    if (nnz > 0) {
        w[0] = col[0];
        w[1] = col[0];
        for (int64_t i = 0; i < rows; ++i)
            if (rowptr[i + 1] > rowptr[i]) {  // canonical CSR: first / last entry of a row are its min / max
                w[0] = std::min(w[0], col[rowptr[i]]);
                w[1] = std::max(w[1], col[rowptr[i + 1] - 1]);
            }
    }
    // ship the two int64 as raw 8-byte words through the f64 all-gather (no arithmetic touches them)
    double *d = c->d_result + 64;
    SLA_HIP_TRY(hipMemcpyAsync(d, w, sizeof(w), hipMemcpyHostToDevice, stream_of(c)));
    SLA_TRY(dist_allgather_f64(c, d, d + 8, 2));
    std::vector<int64_t> all((size_t)2 * c->nranks);
    SLA_HIP_TRY(hipMemcpyAsync(all.data(), d + 8, sizeof(int64_t) * all.size(), hipMemcpyDeviceToHost, stream_of(c)));
    SLA_HIP_TRY(hipStreamSynchronize(stream_of(c)));
    A->xplan = new XPlan();
    plan_window_exchange(c->nranks, c->rank, A->n, all.data(), *A->xplan);
    return SLA_OK;
}

// Row-sharded comm / compute overlap.  For the wave-sliced forms the 512-row steps are split into the
// INTERIOR ones -- every row references columns of this rank's own block only: they can run while the halo exchange is in
// flight -- and the BOUNDARY ones, each list in the visiting order of the full walk (the plane-tiled `sched` when there is
// one).  Taken when the window exchange is in use and most steps are interior.
static int build_overlap_lists(sla_csr *A, int64_t m, int64_t n, int64_t row_begin, int64_t rows, const int64_t *rowptr, const int64_t *col) {
    sla_ctx *c = A->ctx;
    if (!c->collectives || c->overlap < 0 || !A->use_wdia || !A->xplan || m != n || rows == 0) return SLA_OK;
    if (!(A->xplan->use_window || c->x_exchange == 2) || c->x_exchange == 1) return SLA_OK;
    const int64_t lo = row_begin, hi = row_begin + rows;
    std::vector<char> bnd((size_t)A->nblk_wd, 0);
    for (int64_t i = 0; i < rows; ++i)
        if (rowptr[i + 1] > rowptr[i] && (col[rowptr[i]] < lo || col[rowptr[i + 1] - 1] >= hi)) bnd[(size_t)(i / 512)] = 1;   // canonical CSR: min / max column
    std::vector<int32_t> li, lb;
    for (int32_t t = 0; t < A->nblk_wd; ++t) {
        const int32_t s = A->h_wsched.empty() ? t : A->h_wsched[(size_t)t];
        (bnd[(size_t)s] ? lb : li).push_back(s);
    }
    if (lb.empty() || li.size() < lb.size()) return SLA_OK;   // nothing to exchange for / too little to hide it behind
    SLA_HIP_TRY(dev_malloc(c, (void **)&A->d_ov_int, sizeof(int32_t) * li.size()));
    SLA_HIP_TRY(dev_malloc(c, (void **)&A->d_ov_bnd, sizeof(int32_t) * lb.size()));
    SLA_HIP_TRY(hipMemcpy(A->d_ov_int, li.data(), sizeof(int32_t) * li.size(), hipMemcpyHostToDevice));
    SLA_HIP_TRY(hipMemcpy(A->d_ov_bnd, lb.data(), sizeof(int32_t) * lb.size(), hipMemcpyHostToDevice));
    A->ov_nint = (int32_t)li.size();
    A->ov_nbnd = (int32_t)lb.size();
    return SLA_OK;
}


// A rank whose input fails validation still owes its peers the agreement collective of csr_upload (they would block in it
// forever): contribute "failed", then report the local error.
int csr_reject(sla_ctx *c, int rc) {
    if (c && c->collectives) {
        const std::string msg = sla_last_error();
        int agree = 2;
        (void)dist_allreduce_max_i32(c, &agree);
        set_error(msg);
    }
    return rc;
}

// Column panels for irregular matrices (see launch_spmv_panels): panel p = the entries with column in
// [p W, (p+1) W), as a CSR view over the same rows.  Worth it when x does not fit the XCD-private L2 and
// every row still has about one entry per panel.
static int build_panels(sla_csr *A, int64_t m, int64_t n, int64_t row_begin, int64_t rows, const int64_t *rowptr,
                        const int64_t *col, const double *val) {
    sla_ctx *c = A->ctx;
    const int64_t nnz = rowptr[rows];
    if (!c->panels || A->use_diag || A->use_wdia || A->use_vdict || A->xwin_fraction >= 0.5 || rows == 0) return SLA_OK;   // (stencil / banded structure: the value-indexed forms skip the offset dictionary and the window statistics, round 4)
    const int64_t W = std::max<int64_t>(c->panel_cols, 1);
    if (n <= 2 * W) return SLA_OK;                                   // x (nearly) fits the L2 already
    int64_t P = std::min<int64_t>((n + W - 1) / W, nnz / rows);      // >= ~1 entry per row per panel
    if (P < 2) return SLA_OK;
    const int64_t Wp = (n + P - 1) / P;
    std::vector<int64_t> cur(rowptr, rowptr + rows), prp((size_t)rows + 1), pcol;
    std::vector<double> pval;
    for (int64_t p = 0; p < P; ++p) {
        const int64_t chi = std::min<int64_t>(n, (p + 1) * Wp);
        pcol.clear();
        pval.clear();
        prp[0] = 0;
        for (int64_t i = 0; i < rows; ++i) {
            int64_t k = cur[(size_t)i];
            const int64_t e = rowptr[i + 1];
            while (k < e && col[k] < chi) {
                pcol.push_back(col[k]);
                pval.push_back(val[k]);
                ++k;
            }
            cur[(size_t)i] = k;
            prp[(size_t)i + 1] = (int64_t)pcol.size();
        }
        sla_csr *V = nullptr;
        SLA_TRY(csr_upload(c, m, n, row_begin, rows, prp.data(), pcol.data(), pval.data(), &V, true));
        V->is_panel_view = true;
        A->panels.push_back(V);
    }
    SLA_HIP_TRY(dev_malloc(c, (void **)&A->d_panel_y, sizeof(double) * (size_t)std::max<int64_t>(rows, 1)));
    return SLA_OK;
}

// cores this process may use: the affinity mask capped by a container CPU quota (cgroup v2 cpu.max, cgroup v1 cfs_quota_us)
static int host_cores_available() {
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::max(1, std::min(n, CPU_COUNT(&set)));
    long long quota = -1, period = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (fscanf(g, "%lld", &quota) != 1) quota = -1;
        fclose(g);
        if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(h, "%lld", &period) != 1) period = 0;
            fclose(h);
        }
    }
    if (quota > 0 && period > 0) n = std::max(1, std::min<int>(n, (int)(quota / period)));
    return n;
}

int host_threads() {
    static const int t = [] {
        const char *s = getenv("SLA_HOST_THREADS");
        // (default: 16 threads, 32 on hosts with >= 64 hardware threads -- the MI355X boxes have 256; the pair-coding pass at 70 M entries
        // measured 111 / 56 / 42 ms at 8 / 16 / 32 threads and nothing more at 64)
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        // Round 5: capped by what the process may actually USE.  The GPU boxes run this in a container with a CFS quota of 16 cores: 32
        // analysis threads next to the 8 staging threads of the upload overran it, and a throttled group stalls ALL its threads for the
        // rest of the period -- the x-window statistics of config 3a took 72 ms inside bench.py against 6 ms stand-alone, the 4 GB upload
        // 240 ms against 90.
        int v = s ? atoi(s) : std::min((int)(hw >= 64 ? 32u : std::min(16u, hw)), host_cores_available());
        // one process per GPU (torchrun: LOCAL_WORLD_SIZE ranks in this container share the cores and the quota): an equal share, at least 2
        if (const char *lw = s ? nullptr : getenv("LOCAL_WORLD_SIZE")) {
            const int ranks = atoi(lw);
            if (ranks > 1) v = std::max(2, std::min(v, host_cores_available() / ranks));
        }
        return std::max(1, std::min(v, 64));
    }();
    return t;
}

// ---------------------------------------------------------------------------------------------------------------
// csr_upload = one lowering analysis per function.  What they share travels in `Low`; SLA_LOW_LOCALS re-opens it under the
// names the analyses use.
// ---------------------------------------------------------------------------------------------------------------
// The canonical entry arrays go up in the background and the upload can be called off (`stop`): a matrix that turns out to be
// value-indexed gets its col / val written by a device kernel from its 1-byte codes (vd_expand_kernel below) instead of over PCIe.  The
// entry copies wait for the analysis's decision (`decided`: it falls right after the pair-coding pass, or after the first 257 distinct
// pairs of a matrix that is not value-indexed -- microseconds) instead of racing it for the host's memory bandwidth.
struct CanonUpload {
    std::atomic<int> stop{0};
    std::atomic<int> decided{0};
    // the canonical-CSR check of the caller's columns rides on the narrowing of the upload (set before `decided`): 0 fine, 1 out of
    // bounds, 2 not ascending inside a row -- validate_columns' codes, the lowest kind wins
    bool validate = false;
    std::atomic<int> col_bad{0};
    int64_t done_col = 0, done_val = 0;   // entries [0, done) are on the device
};
struct Low {
    sla_ctx *c;
    sla_csr *A;
    int64_t m, n, row_begin, rows, nnz;
    const int64_t *rowptr, *col;
    const double *val;
    bool panel_view, dbg_lower;
    hipError_t err = hipSuccess;
    CanonUpload *cu = nullptr;          // (csr_upload's; low_value_indexed reports its decision there)
    bool validate = false;              // the caller's columns still have to be checked (bounds, ascending inside a row)
    int col_verdict = -1;               // -1: not checked yet; 0 fine, 1 out of bounds, 2 not ascending (validate_columns' codes)
    std::chrono::steady_clock::time_point t_sub = std::chrono::steady_clock::now();
    void sub(const char *what) {        // SLA_DEBUG_LOWER: times inside one analysis
        if (!dbg_lower || panel_view) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[sla] lowering:     . %-34s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(t - t_sub).count());
        t_sub = t;
    }
    std::vector<int32_t> rb;            // row-block starts
    std::vector<int64_t> offs;          // sorted distinct diagonal offsets (<= 256) when the matrix has that structure
    std::vector<uint8_t> dcodes;        // per entry: index into offs
    // (every lowered array carries kArraySlack zeroed bytes behind its end: the pipelined stream kernel reads whole row blocks
    // with clamped, unconditional loads, and an empty row block at the very end of the matrix reads "its" first entry there)
    void upload(void **dst, const void *src, size_t bytes) {
        if (err != hipSuccess) return;
        err = dev_malloc(c, dst, bytes + kArraySlack);
        if (err == hipSuccess && bytes) err = xfer_copy(c, *dst, src, bytes, hipMemcpyHostToDevice);
        if (err == hipSuccess) err = hipMemset((char *)*dst + bytes, 0, kArraySlack);
    }
};
#define SLA_LOW_LOCALS(L)                                                                                                  \
    [[maybe_unused]] sla_ctx *c = (L).c;                                                                                   \
    [[maybe_unused]] sla_csr *A = (L).A;                                                                                   \
    [[maybe_unused]] const int64_t m = (L).m, n = (L).n, row_begin = (L).row_begin, rows = (L).rows, nnz = (L).nnz;        \
    [[maybe_unused]] const int64_t *rowptr = (L).rowptr, *col = (L).col;                                                   \
    [[maybe_unused]] const double *val = (L).val;                                                                          \
    [[maybe_unused]] const bool panel_view = (L).panel_view, dbg_lower = (L).dbg_lower;                                    \
    [[maybe_unused]] hipError_t &err = (L).err;                                                                            \
    [[maybe_unused]] std::vector<int32_t> &rb = (L).rb;                                                                    \
    [[maybe_unused]] std::vector<int64_t> &offs = (L).offs;                                                                \
    [[maybe_unused]] std::vector<uint8_t> &dcodes = (L).dcodes;                                                            \
    [[maybe_unused]] auto upload = [&](void **dst_, const void *src_, size_t bytes_) { (L).upload(dst_, src_, bytes_); }

// canonical CSR arrays (i32 columns, i32 / i64 row pointers) + the row-block tables of the general kernels.
// The entry arrays go up in chunks and the upload can be called off (`stop`): a matrix that turns out to be value-indexed gets the rest
// of its col / val written by a device kernel from its 1-byte codes (vd_expand_kernel below) instead of over PCIe.
static void low_csr_arrays(Low &L, CanonUpload *cu) {
    SLA_LOW_LOCALS(L);
    auto stopped = [&] { return cu && cu->stop.load(std::memory_order_relaxed) != 0; };
    auto await_decision = [&] {
        while (cu && !cu->decided.load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(50));   // (not a spin: the analysis wants the cores)
    };
    // the values go up on a second host thread while this one narrows and uploads the indices (pageable copies are
    // bound by the staging memcpy of the calling thread, not by the link)
    hipError_t err_val = hipSuccess;
    struct Joiner {   // (a host allocation failing below must not unwind past a joinable thread)
        std::thread &t;
        ~Joiner() { if (t.joinable()) t.join(); }
    };
    std::thread val_up([&] {
        Bind bind(c);   // (a new thread starts on device 0)
        await_decision();   // (a value-indexed matrix does not even get the arrays allocated here: csr_ensure_canon, if ever)
        if (stopped()) return;
        if (err_val == hipSuccess) err_val = dev_malloc(c, (void **)&A->d_val, sizeof(double) * (size_t)nnz + kArraySlack);
        if (err_val == hipSuccess) err_val = hipMemset((char *)A->d_val + sizeof(double) * (size_t)nnz, 0, kArraySlack);
        size_t done_b = 0;
        if (err_val == hipSuccess && nnz && !stopped())
            err_val = xfer_copy(c, A->d_val, val, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, cu ? &cu->stop : nullptr, &done_b);
        if (cu) cu->done_val = (int64_t)(done_b / sizeof(double));
    });
    Joiner val_up_joiner{val_up};
    // ... and the indices on a third (round 5: they used to follow the row pointers and row-block tables on THIS thread -- 40-100 ms of
    // allocations, narrowing and small copies at 10 M rows during which 1.3 GB of columns waited)
    hipError_t err_col = hipSuccess;
    std::thread col_up([&] {
        Bind bind(c);
        await_decision();
        if (!stopped()) {   // the caller's int64 column indices are narrowed on their way into the pinned slots (no 4 B-per-entry host copy, no pass of its own)
            if (err_col == hipSuccess) err_col = dev_malloc(c, (void **)&A->d_col, sizeof(int32_t) * (size_t)nnz + kArraySlack);
            if (err_col == hipSuccess) {   // the slack repeats the last valid column (zero = x[0] can lie outside a slab's vector: see launch_col_slack_fill)
                const std::vector<int32_t> tail(kArraySlack / sizeof(int32_t), nnz ? (int32_t)col[nnz - 1] : 0);
                err_col = hipMemcpy((char *)A->d_col + sizeof(int32_t) * (size_t)nnz, tail.data(), kArraySlack, hipMemcpyHostToDevice);
            }
            size_t done_b = 0;
            struct NarrowCtx { const int64_t *col, *rowptr; int64_t rows, n; bool validate; std::atomic<int> *bad; } nctx{col, rowptr, rows, n, cu && cu->validate, cu ? &cu->col_bad : nullptr};
            if (err_col == hipSuccess && nnz && !stopped())
                err_col = xfer_copy(c, A->d_col, nullptr, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, cu ? &cu->stop : nullptr, &done_b,
                                [](void *slot, size_t off, size_t len, const void *ctx) {
                                    const NarrowCtx &x = *(const NarrowCtx *)ctx;
                                    const int64_t k0 = (int64_t)(off / sizeof(int32_t)), cnt = (int64_t)(len / sizeof(int32_t));
                                    const int64_t *col64 = x.col + k0;
                                    int32_t *o = (int32_t *)slot;
                                    if (!x.validate) {
                                        for (int64_t q = 0; q < cnt; ++q) o[q] = (int32_t)col64[q];
                                        return;
                                    }
                                    // (round 5: the check used to be a pass of its own over the 2.6 GB of config 3a's columns -- 167 ms on the main
                                    // thread with the upload waiting for its verdict.  Here only the bits the narrowing drops are kept: a column
                                    // with a bit from 31 up is negative or >= 2^31 = out of bounds; the rest of the check -- < n, ascending
                                    // inside a row -- runs on the device over the narrowed copy: validate_columns_device)
                                    uint64_t acc = 0;
                                    for (int64_t q = 0; q < cnt; ++q) {
                                        acc |= (uint64_t)col64[q];
                                        o[q] = (int32_t)col64[q];
                                    }
                                    if (acc >> 31) x.bad->store(1, std::memory_order_relaxed);
                                }, &nctx);
            if (cu) cu->done_col = (int64_t)(done_b / sizeof(int32_t));
        }
    });
    Joiner col_up_joiner{col_up};
    // row pointers and row-block tables first: every form needs them
    if (A->rp64) {
        upload(&A->d_rowptr, rowptr, sizeof(int64_t) * (size_t)(rows + 1));
        std::vector<int64_t> rbk(rb.size());
        for (size_t b = 0; b < rb.size(); ++b) rbk[b] = rowptr[rb[b]];
        upload(&A->d_rbk, rbk.data(), sizeof(int64_t) * rbk.size());
    } else {
        std::vector<int32_t> rbk(rb.size());
        for (size_t b = 0; b < rb.size(); ++b) rbk[b] = (int32_t)rowptr[rb[b]];
        upload(&A->d_rbk, rbk.data(), sizeof(int32_t) * rbk.size());
        // (kRowptrPad entries = nnz behind the last one: spmv_wave_kernel reads the row pointers of whole 128-row blocks unclamped,
        // rows past the end of the matrix are empty)
        std::vector<int32_t> rp32((size_t)rows + 1 + kRowptrPad, (int32_t)nnz);
        par_rows(rows + 1, 1, [&](int, int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i) rp32[(size_t)i] = (int32_t)rowptr[i];
        });
        upload(&A->d_rowptr, rp32.data(), sizeof(int32_t) * rp32.size());
    }
    upload((void **)&A->d_rb, rb.data(), sizeof(int32_t) * rb.size());
    L.sub("[upload thread] row pointers + row blocks up");
    col_up.join();
    val_up.join();
    if (err == hipSuccess) err = err_col;
    if (err == hipSuccess) err = err_val;
}

// Canonical col / val of a value-indexed matrix from its 1-byte codes: col[k] = row + offset[code[k]], val[k] = value[code[k]] -- the
// pair table holds the values' bit patterns, so the arrays are the caller's bit for bit.  At 70 M entries: 0.3 ms on the device against
// 0.11 s for the 0.84 GB over PCIe from pageable memory, which was the critical path of sla_csr_from_csr (round 4).
__global__ void __launch_bounds__(256) vd_expand_kernel(int64_t rows, int64_t row_begin, const int32_t *__restrict__ rowptr,
                                                        const uint8_t *__restrict__ code, const int32_t *__restrict__ doff,
                                                        const double *__restrict__ dval, int32_t *__restrict__ col, double *__restrict__ val,
                                                        int64_t from_col, int64_t from_val) {
    // A wavefront takes 64 consecutive rows: their entries are one contiguous range, walked lane by lane (coalesced loads of the codes,
    // coalesced stores of col / val); the row of an entry is a binary search over the 65 row pointers in LDS.  (The first version gave a
    // thread a row: 28- and 56-byte pieces at row-strided addresses -- 12-23 ms for 70 M entries; this one: < 1 ms.)
    __shared__ int32_t s_off[256];
    __shared__ double s_val[256];
    __shared__ int32_t s_rp[4][65];
    s_off[threadIdx.x] = doff[threadIdx.x];
    s_val[threadIdx.x] = dval[threadIdx.x];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t nblk = (rows + 63) / 64;
    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < nblk; blk += (int64_t)gridDim.x * 4) {   // (wavefront-uniform trip count per wavefront: no barrier inside)
        const int64_t r0 = blk * 64;
        s_rp[wave][lane] = rowptr[std::min<int64_t>(r0 + lane, rows)];
        if (lane == 0) s_rp[wave][64] = rowptr[std::min<int64_t>(r0 + 64, rows)];
        __builtin_amdgcn_wave_barrier();
        const int32_t k0 = s_rp[wave][0], k1 = s_rp[wave][64];
        for (int32_t k = k0 + lane; k < k1; k += 64) {
            int lo = 0, hi = 64;
#pragma unroll
            for (int step = 0; step < 6; ++step) {
                const int mid = (lo + hi) >> 1;
                if (s_rp[wave][mid] <= k) lo = mid;
                else hi = mid;
            }
            const int cd = code[k];
            if (k >= from_col) col[k] = (int32_t)(row_begin + r0 + lo + s_off[cd]);
            if (k >= from_val) val[k] = s_val[cd];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// LDS x window of every row block (spmv_xwin_kernel) and the share of the entries that fall inside
static void low_xwin_statistics(Low &L) {
    SLA_LOW_LOCALS(L);
    if (!panel_view) {
        // LDS x window of each row block: kXWin columns starting kXWinHalo left of its first diagonal column
        std::vector<int32_t> rbw(rb.size(), 0);
        int64_t inside = 0, total = 0;
        const int64_t wmax = std::max<int64_t>(0, n - kXWin);
        std::vector<int64_t> part_in((size_t)host_threads(), 0), part_tot((size_t)host_threads(), 0);
        // (the share is a go / no-go statistic with its threshold at one half: from 2^16 row blocks on every (blocks / 2^15)-th block is
        // counted -- the full count read all of col again, 0.16 s at 330 M entries; since round 5 from 2^14 blocks on, every (blocks / 2^12)-th)
        const int64_t nblk = (int64_t)rb.size() - 1, samp = nblk >= ((int64_t)1 << 14) ? nblk >> 12 : 1;   // (round 5: 4096 sampled blocks = 4 M entries decide a threshold at one half as well as 32768 did)
        par_rows(nblk, 1, [&](int t, int64_t blo, int64_t bhi) {
            int64_t in = 0, tot = 0;
            for (int64_t b = blo; b < bhi; ++b) {
                const int64_t w = std::min<int64_t>(wmax, std::max<int64_t>(0, row_begin + rb[(size_t)b] - kXWinHalo));
                rbw[(size_t)b] = (int32_t)w;
                if (b % samp != 0) continue;
                const int64_t k0 = rowptr[rb[(size_t)b]], k1 = rowptr[rb[(size_t)b + 1]];
                if (k1 - k0 > kNnzPerRowBlock) continue;  // long-row blocks gather from global memory
                tot += k1 - k0;
                for (int64_t k = k0; k < k1; ++k) in += (col[k] >= w && col[k] < w + kXWin) ? 1 : 0;
            }
            part_in[(size_t)t] = in;
            part_tot[(size_t)t] = tot;
        }, 4096);
        for (size_t t = 0; t < part_in.size(); ++t) { inside += part_in[t]; total += part_tot[t]; }
        A->xwin_fraction = total ? (double)inside / (double)total : 0.0;
        A->use_xwin = A->xwin_fraction >= 0.5;
        upload((void **)&A->d_rbw, rbw.data(), sizeof(int32_t) * rbw.size());
    }
}

// dictionary of diagonal offsets + 1-byte column codes (spmv_diag_kernel): <= 256 distinct col - row values
static void low_diagonal_dictionary(Low &L) {
    SLA_LOW_LOCALS(L);
    if (!panel_view) {
        // dictionary of diagonal offsets: worthwhile (and representable in a byte) when col - row takes at
        // most 256 distinct values, i.e. for stencil / banded structure
        bool ok = nnz > 0;
        {   // distinct offsets: per-thread sets, merged
            std::vector<std::vector<int64_t>> loc((size_t)host_threads());
            std::vector<char> bad((size_t)host_threads(), 0);
            par_rows(rows, 1, [&](int t, int64_t lo, int64_t hi) {
                std::vector<int64_t> &mine = loc[(size_t)t];
                for (int64_t i = lo; i < hi && !bad[(size_t)t]; ++i) {
                    const int64_t gr = row_begin + i;
                    for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
                        const int64_t d = col[k] - gr;
                        bool found = false;
                        for (int64_t o : mine) if (o == d) { found = true; break; }   // <= 256 entries: a linear scan is fine
                        if (!found) {
                            if (mine.size() == 256) { bad[(size_t)t] = 1; break; }
                            mine.push_back(d);
                        }
                    }
                }
            });
            for (size_t t = 0; t < loc.size() && ok; ++t) {
                if (bad[t]) ok = false;
                for (int64_t d : loc[t]) {
                    if (std::find(offs.begin(), offs.end(), d) != offs.end()) continue;
                    if (offs.size() == 256) { ok = false; break; }
                    offs.push_back(d);
                }
            }
            if (!ok) offs.clear();
        }
        if (ok) {
            std::sort(offs.begin(), offs.end());
            std::vector<int32_t> dict(256, (int32_t)offs.back());
            for (size_t t = 0; t < offs.size(); ++t) dict[t] = (int32_t)offs[t];
            std::vector<uint8_t> &codes = dcodes;
            codes.resize((size_t)nnz);
            par_rows(rows, 1, [&](int, int64_t lo, int64_t hi) {
                for (int64_t i = lo; i < hi; ++i) {
                    const int64_t gr = row_begin + i;
                    for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
                        codes[(size_t)k] = (uint8_t)(std::lower_bound(offs.begin(), offs.end(), col[k] - gr) - offs.begin());
                }
            });
            upload((void **)&A->d_code, codes.data(), codes.size());
            upload((void **)&A->d_dict, dict.data(), sizeof(int32_t) * dict.size());
            A->use_diag = true;
            A->ndiag = (int)offs.size();
        }
    }
}

// value-indexed forms for constant-coefficient stencils: (offset, value) pair dictionary + byte codes (spmv_vdict_kernel), the
// wave-sliced records and the plane-tiled visiting order (spmv_wdia_kernel)
static void low_value_indexed(Low &L) {
    SLA_LOW_LOCALS(L);
    if (!panel_view && !A->rp64 && nnz > 0 && A->max_row_nnz <= kVdMaxRowNnz) {
        // value-indexed form: dictionary of (col - row, value bit pattern) pairs, one byte per entry
        struct Pair { int64_t off; uint64_t bits; };
        auto bits_of = [](double v) { uint64_t u; memcpy(&u, &v, 8); return u; };
        constexpr int kSlots = 1024;                       // open addressing, <= 256 live keys
        struct PairTable {
            std::vector<int> slot = std::vector<int>(kSlots, -1);
            std::vector<Pair> pairs;
            int find(int64_t off, uint64_t bits, bool insert) {
                uint64_t h = ((uint64_t)off * 0x9E3779B97F4A7C15ull) ^ (bits * 0xC2B2AE3D27D4EB4Full);
                h ^= h >> 29;
                for (int i = (int)(h & (kSlots - 1));; i = (i + 1) & (kSlots - 1)) {
                    const int id = slot[(size_t)i];
                    if (id < 0) {
                        if (!insert || pairs.size() == 256) return -1;
                        slot[(size_t)i] = (int)pairs.size();
                        pairs.push_back({off, bits});
                        return (int)pairs.size() - 1;
                    }
                    if (pairs[(size_t)id].off == off && pairs[(size_t)id].bits == bits) return id;
                }
            }
        };
        PairTable tab;                                     // the matrix's table: per-thread tables, merged
        std::vector<Pair> &pairs = tab.pairs;
        auto find = [&](int64_t off, uint64_t bits, bool insert) -> int { return tab.find(off, bits, insert); };
        bool ok = true;
        // ONE pass over the entries (round 4: the table pass and the code pass each read col + val, 1.1 GB at 70 M entries): every
        // thread codes its rows against its OWN table while it builds it; once the tables are merged and ranked the thread-local
        // codes are translated in place (1 B per entry)
        L.sub("(start)");
        // (not value-initialised: a zero fill of 70 MB on this thread cost 12 ms at 216^3; every entry's byte is written by the pass below,
        // whose threads also take the first-touch page faults)
        const size_t codes_n = ((size_t)nnz + 3) / 4 * 4 + 16;
        std::unique_ptr<uint8_t[]> codes_buf(new uint8_t[codes_n]);
        uint8_t *codes = codes_buf.get();
        memset(codes + nnz, 0, codes_n - (size_t)nnz);
        L.sub("codes allocation");
        hipError_t err_code = hipSuccess;
        std::thread code_up;
        struct Joiner {   // (the code upload reads codes[]: joined before the buffer goes, whichever way this block is left)
            std::thread &t;
            ~Joiner() { if (t.joinable()) t.join(); }
        } code_up_joiner{code_up};
        std::vector<PairTable> loc((size_t)host_threads());
        std::vector<int> vcol((size_t)host_threads(), 0);
        std::vector<int64_t> cr_lo((size_t)host_threads(), n), cr_hi((size_t)host_threads(), -1);   // smallest / largest column (taken along in the pass below)
        {
            std::vector<char> bad((size_t)host_threads(), 0);
            par_rows(rows, 1, [&](int t, int64_t lo, int64_t hi) {
                PairTable &mine = loc[(size_t)t];
                int64_t cmin = n, cmax = -1;
                int vbad = 0;   // (the canonical-CSR check of the caller's columns rides along: the pass reads every column anyway)
                // (a stencil row repeats its predecessor's pairs position by position: the id found at position j of the previous row is
                // tried first -- two compares instead of a hash and a probe for all but the boundary rows)
                int recent[kVdMaxRowNnz + 1];
                for (int &r_ : recent) r_ = -1;
                for (int64_t i = lo; i < hi && !bad[(size_t)t]; ++i) {
                    const int64_t gr = row_begin + i, k0 = rowptr[i], k1 = rowptr[i + 1];
                    if (k1 > k0) {   // (columns ascend inside a row)
                        cmin = std::min<int64_t>(cmin, col[k0]);
                        cmax = std::max<int64_t>(cmax, col[k1 - 1]);
                    }
                    for (int64_t k = k0; k < k1; ++k) {
                        if (col[k] < 0 || col[k] >= n) vbad = 1;
                        else if (k > k0 && col[k] <= col[k - 1] && vbad == 0) vbad = 2;
                        const int64_t off = col[k] - gr;
                        const uint64_t bits = bits_of(val[k]);
                        int id = recent[k - k0];
                        if (id < 0 || mine.pairs[(size_t)id].off != off || mine.pairs[(size_t)id].bits != bits) {
                            id = mine.find(off, bits, true);
                            if (id < 0) { bad[(size_t)t] = 1; break; }
                            recent[k - k0] = id;
                        }
                        codes[(size_t)k] = (uint8_t)id;
                    }
                }
                cr_lo[(size_t)t] = cmin;
                cr_hi[(size_t)t] = cmax;
                vcol[(size_t)t] = vbad;
            });
            if (L.validate && std::find(bad.begin(), bad.end(), 1) == bad.end()) {   // (every thread went through all of its rows)
                L.col_verdict = std::find(vcol.begin(), vcol.end(), 1) != vcol.end() ? 1 : std::find(vcol.begin(), vcol.end(), 2) != vcol.end() ? 2 : 0;
                if (L.col_verdict != 0) ok = false;
            }
            for (size_t t = 0; t < loc.size() && ok; ++t) {
                if (bad[t]) ok = false;
                for (const Pair &pr : loc[t].pairs)
                    if (ok && find(pr.off, pr.bits, true) < 0) ok = false;
            }
        }
        L.sub("pair coding pass");
        if (L.cu) {   // the decision the background upload of col / val waits for
            if (ok && c->canon_device) L.cu->stop.store(1, std::memory_order_relaxed);
            L.cu->decided.store(1, std::memory_order_release);
        }
        if (ok) {
            // canonical table order: by offset, then by value bits (independent of the input order)
            std::vector<int> order(pairs.size()), rank(pairs.size());
            for (size_t t = 0; t < order.size(); ++t) order[t] = (int)t;
            std::sort(order.begin(), order.end(), [&](int x, int y) {
                return pairs[(size_t)x].off != pairs[(size_t)y].off ? pairs[(size_t)x].off < pairs[(size_t)y].off
                                                                      : pairs[(size_t)x].bits < pairs[(size_t)y].bits;
            });
            std::vector<int32_t> doff(256, 0);
            std::vector<double> dval(256, 0.0);
            for (size_t t = 0; t < order.size(); ++t) {
                rank[(size_t)order[t]] = (int)t;
                doff[t] = (int32_t)pairs[(size_t)order[t]].off;
                memcpy(&dval[t], &pairs[(size_t)order[t]].bits, 8);
            }
            par_rows(rows, 1, [&](int t, int64_t lo, int64_t hi) {   // (the same row partition as above: thread t translates its own codes)
                uint8_t lut[256];
                const PairTable &mine = loc[(size_t)t];
                for (size_t q = 0; q < mine.pairs.size(); ++q) lut[q] = (uint8_t)rank[(size_t)find(mine.pairs[q].off, mine.pairs[q].bits, false)];
                for (int64_t k = rowptr[lo]; k < rowptr[hi]; ++k) codes[(size_t)k] = lut[codes[(size_t)k]];
            });
            L.sub("code translation");
            // (the 1 B per entry goes up on its own thread while this one builds the wave slices: 3-15 ms at 70 M entries)
            code_up = std::thread([&L, A, codes, codes_n, &err_code] {
                Bind bind(L.c);
                err_code = dev_malloc(L.c, (void **)&A->d_vcode, codes_n + kArraySlack);
                if (err_code == hipSuccess) err_code = hipMemcpy(A->d_vcode, codes, codes_n, hipMemcpyHostToDevice);
                if (err_code == hipSuccess) err_code = hipMemset((char *)A->d_vcode + codes_n, 0, kArraySlack);
            });
            upload((void **)&A->d_vdoff, doff.data(), sizeof(int32_t) * doff.size());
            upload((void **)&A->d_vdval, dval.data(), sizeof(double) * dval.size());
            A->use_vdict = true;
            A->npairs = (int)pairs.size();
            A->nblk_vd = (int32_t)((rows + kVdRows - 1) / kVdRows);
            // wave-sliced form: per 128 rows (two per lane) the sorted union of the pair codes with an even-row and
            // an odd-row lane mask each.  Taken when the slices are reasonably full (>= 1/4 of the row slots busy
            // on average) and x is addressable with a 32-bit byte offset.
            const int64_t nsl = (rows + 127) / 128;
            std::vector<int32_t> wptr((size_t)nsl + 1, 0);
            std::vector<uint64_t> wme, wmo;
            std::vector<double> wval;
            std::vector<int32_t> woff;
            std::vector<uint8_t> wcode;
            bool wok = n < ((int64_t)1 << 28);
            if (wok) {   // slices are independent: each thread builds the records of a contiguous range of slices
                struct Part { std::vector<uint64_t> me, mo; std::vector<double> v; std::vector<int32_t> o, cnt; std::vector<uint8_t> cd; };
                std::vector<Part> part((size_t)host_threads());
                const int T = par_rows(rows, 128, [&](int t, int64_t lo, int64_t hi) {
                    Part &P = part[(size_t)t];
                    uint64_t lane_mask[2][256];
                    for (int64_t rlo = lo; rlo < hi; rlo += 128) {
                        uint64_t present[4] = {0, 0, 0, 0};
                        const int64_t rhi = std::min<int64_t>(hi, rlo + 128);
                        for (int64_t i = rlo; i < rhi; ++i)
                            for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
                                const int cd = codes[(size_t)k];
                                if (!((present[cd >> 6] >> (cd & 63)) & 1)) {
                                    present[cd >> 6] |= 1ull << (cd & 63);
                                    lane_mask[0][cd] = lane_mask[1][cd] = 0;
                                }
                                lane_mask[(i - rlo) & 1][cd] |= 1ull << ((i - rlo) >> 1);
                            }
                        const size_t first = P.me.size();
                        for (int cd = 0; cd < 256; ++cd)   // ascending code = ascending (offset, value bits)
                            if ((present[cd >> 6] >> (cd & 63)) & 1) {
                                P.me.push_back(lane_mask[0][cd]);
                                P.mo.push_back(lane_mask[1][cd]);
                                P.v.push_back(dval[(size_t)cd]);
                                P.o.push_back(doff[(size_t)cd]);
                                P.cd.push_back((uint8_t)cd);
                            }
                        P.cnt.push_back((int32_t)(P.me.size() - first));
                    }
                });
                L.sub("wave slices");
                int64_t sl = 0;
                for (int t = 0; t < T && wok; ++t) {
                    const Part &P = part[(size_t)t];
                    for (int32_t cnt : P.cnt) {
                        if (cnt > kWdMaxSliceRecords) wok = false;   // the records past the 8 pipelined ones go chunk by chunk
                        wptr[(size_t)sl + 1] = wptr[(size_t)sl] + cnt;
                        ++sl;
                    }
                    wme.insert(wme.end(), P.me.begin(), P.me.end());
                    wmo.insert(wmo.end(), P.mo.begin(), P.mo.end());
                    wval.insert(wval.end(), P.v.begin(), P.v.end());
                    woff.insert(woff.end(), P.o.begin(), P.o.end());
                    wcode.insert(wcode.end(), P.cd.begin(), P.cd.end());
                }
                if ((int64_t)wme.size() * 32 > nnz + 2048) wok = false;   // < 1/4 full: the byte-code kernel is the better form
            }
            L.sub("wave slice merge");
            if (wok) {
                A->nwent = (int64_t)wme.size();
                for (int t = 0; t < 8; ++t) { wme.push_back(0); wmo.push_back(0); wval.push_back(0.0); woff.push_back(0); }
                upload((void **)&A->d_wptr, wptr.data(), sizeof(int32_t) * wptr.size());
                upload((void **)&A->d_wme, wme.data(), sizeof(uint64_t) * wme.size());
                upload((void **)&A->d_wmo, wmo.data(), sizeof(uint64_t) * wmo.size());
                upload((void **)&A->d_wval, wval.data(), sizeof(double) * wval.size());
                upload((void **)&A->d_woff, woff.data(), sizeof(int32_t) * woff.size());
                A->use_wdia = true;
                A->nslices = (int32_t)nsl;
                A->nblk_wd = (int32_t)((nsl + 3) / 4);
                {   // LDS windows: cluster the offsets (doff[] is ascending); a record then names an element of the staged buffer
                    WdWin W;
                    bool lok = true;
                    int32_t wmax[kWdWinMax] = {};
                    for (int t = 0; t < A->npairs && lok; ++t) {
                        const int32_t o = doff[(size_t)t];
                        if (W.n > 0 && o - wmax[W.n - 1] < kWdWinMerge) { wmax[W.n - 1] = o; continue; }
                        if (W.n == kWdWinMax) { lok = false; break; }
                        W.omin[W.n] = o & ~1;        // (floor to even: staged as aligned 16-byte pairs)
                        wmax[W.n] = o;
                        ++W.n;
                    }
                    for (int k = 0; k < W.n && lok; ++k) {
                        const int32_t elems = wmax[k] - W.omin[k] + 512 + 2;   // + 1: odd first row of a slab, + 1: second row of the last pair
                        W.pb[k + 1] = W.pb[k] + (elems + 1) / 2;
                    }
                    W.pairs = W.pb[W.n];
                    if (lok && W.n > 0 && W.pairs <= kWdWinMaxPairs && A->npairs <= 8) {
                        // uniform records: every slice carries all pairs in table order; a pair it does not use has empty masks
                        WdUni U;
                        U.n = A->npairs;
                        for (int t = 0; t < A->npairs; ++t) {
                            int k = 0;
                            while (k + 1 < W.n && doff[(size_t)t] >= W.omin[k + 1]) ++k;
                            U.lpos[t] = 2 * W.pb[k] + (doff[(size_t)t] - W.omin[k]);
                            U.val[t] = dval[(size_t)t];
                        }

                        std::vector<uint64_t> wum((size_t)nsl * 16, 0);
                        par_rows(nsl, 1, [&](int, int64_t slo, int64_t shi) {
                            for (int64_t sl2 = slo; sl2 < shi; ++sl2)
                                for (int32_t e = wptr[(size_t)sl2]; e < wptr[(size_t)sl2 + 1]; ++e) {
                                    wum[(size_t)sl2 * 16 + wcode[(size_t)e]] = wme[(size_t)e];
                                    wum[(size_t)sl2 * 16 + 8 + wcode[(size_t)e]] = wmo[(size_t)e];
                                }
                        }, 16384);
                        upload((void **)&A->d_wum, wum.data(), sizeof(uint64_t) * wum.size());
                        L.sub("slice uploads + uniform masks");
                        int64_t clo = n, chi = -1;                // (from the pair-coding pass: no pass of its own)
                        for (size_t t = 0; t < cr_lo.size(); ++t) { clo = std::min(clo, cr_lo[t]); chi = std::max(chi, cr_hi[t]); }
                        L.sub("column range");
                        A->wd_col_lo = (int32_t)clo;
                        A->wd_col_hi = (int32_t)chi;
                        // x[own row] from the staged buffer (an epilogue operand that is the gathered vector): offset 0 inside a
                        // window, and every own row inside the column range the windows are filled for
                        if (clo <= row_begin && row_begin + rows - 1 <= chi)
                            for (int k = 0; k < W.n; ++k)
                                if (W.omin[k] <= 0 && 0 <= wmax[k]) U.lpos0 = 2 * W.pb[k] - W.omin[k];
                        A->wd_win = W;
                        A->wd_uni = U;
                        A->wd_lds = true;
                        // Plane march (spmv_wdia_march_kernel): three windows {-D}, {in-plane}, {+D} with ONE pair in each far
                        // window, D even, the in-plane window <= 512 pairs and around offset 0; a row slab of a sharded matrix is walked like a
                        // matrix of its own (planes counted from its first row, which must be even: aligned 16-byte pairs of x).  The masks
                        // are laid out per (tile, plane, wavefront) for rows plane * D + tile * 512 + wavefront * 128 + [0, 128)
                        // -- a plane is not a whole number of 128-row slices (216^2 = 364.5 of them).
                        const int np = A->npairs;
                        const int64_t D = np >= 3 ? (int64_t)doff[(size_t)np - 1] : 0;
                        if (W.n == 3 && (np == 5 || np == 7) && (row_begin & 1) == 0 && m == n && D >= 1024 && (D & 1) == 0 &&
                            doff[0] == -D && doff[1] >= W.omin[1] && doff[(size_t)np - 2] <= wmax[1] && W.omin[1] <= 0 && wmax[1] >= 0 &&
                            W.pb[2] - W.pb[1] <= 512 && rows >= 2 * D) {
                            WdMarch G;
                            G.D = (int32_t)D;
                            G.T = (int32_t)((D + 511) / 512);
                            G.planes = (int32_t)((rows + D - 1) / D);
                            G.omin = W.omin[1];
                            G.pairs = W.pb[2] - W.pb[1];
                            const int slots = std::max(1, c->wd_march_occ) * c->n_cu;
                            G.S = std::max(1, std::min(G.planes, slots / G.T));
                            G.PS = (G.planes + G.S - 1) / G.S;
                            G.S = (G.planes + G.PS - 1) / G.PS;
                            G.ntasks = G.T * G.S;
                            WdUni M;
                            M.n = np;
                            for (int t = 0; t < np; ++t) {
                                const int64_t o = t == 0 ? 0 : t == np - 1 ? 0 : (int64_t)doff[(size_t)t];   // (relative to the pair's own plane)
                                M.lpos[t] = (int32_t)(o - G.omin);
                                M.val[t] = dval[(size_t)t];
                            }
                            M.lpos0 = -G.omin;
                            std::vector<uint64_t> wm((size_t)G.T * (size_t)G.planes * 4 * 16, 0);
                            par_rows(G.planes, 1, [&](int, int64_t plo, int64_t phi) {   // (the slices of a plane belong to one thread)
                                for (int64_t kk = plo; kk < phi; ++kk) {
                                    const int64_t r0 = kk * D, r1 = std::min<int64_t>(rows, r0 + D);
                                    for (int64_t i = r0; i < r1; ++i) {
                                        const int64_t pos = i - r0, tile = pos >> 9;
                                        const size_t sl2 = ((size_t)(tile * G.planes + kk) * 4 + (size_t)((pos >> 7) & 3)) * 16;
                                        const uint64_t bit = 1ull << ((pos & 127) >> 1);
                                        for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) wm[sl2 + (size_t)(pos & 1) * 8 + codes[(size_t)k]] |= bit;
                                    }
                                }
                            }, 4);
                            L.sub("march masks");
                            upload((void **)&A->d_wum_m, wm.data(), sizeof(uint64_t) * wm.size());
                            A->wd_mg = G;
                            A->wd_muni = M;
                            A->wd_march = true;
                            wd_march_prepare();
                            L.sub("march upload + prepare");
                        }
                    }
                }
                // Visiting order of the 512-row steps.  A 3-D stencil row touches x one PLANE (the far diagonal, D rows)
                // behind and ahead; swept in row order, a line of x is needed again 2 D rows later, by which time
                // the vectors streaming through the 4 MiB L2 have evicted it (216^3: D = 46656, 1.35 extra reads
                // of x measured).  So the sweep is tiled: the steps are grouped by their position inside the plane
                // (tiles of `tile` steps) and each tile is walked plane after plane, which makes the three touches
                // of a line neighbours in time.  Only the order changes; every step is still done exactly once.
                L.sub("(before the visiting order)");
                int64_t far = 0;
                for (int t = 0; t < A->npairs; ++t) far = std::max<int64_t>(far, std::llabs((long long)doff[(size_t)t]));
                const double bpp = (double)far / 512.0;   // steps per plane
                int tile = c->wd_tile;
                if (tile < 0) tile = (far * 8 >= (256 << 10) && far * 4 <= rows) ? (int)std::max(8.0, bpp / 6.0 + 0.5) : 0;
                if (tile > 0 && bpp > 2.0 * tile) {
                    std::vector<int32_t> sched((size_t)A->nblk_wd);
                    std::vector<int32_t> key((size_t)A->nblk_wd);
                    for (int32_t bb = 0; bb < A->nblk_wd; ++bb) {
                        sched[(size_t)bb] = bb;
                        const double pos = fmod((double)bb, bpp);   // position of the step inside its plane, in steps
                        key[(size_t)bb] = (int32_t)(pos / tile);
                    }
                    std::stable_sort(sched.begin(), sched.end(), [&](int32_t x, int32_t y) { return key[(size_t)x] < key[(size_t)y]; });
                    upload((void **)&A->d_wsched, sched.data(), sizeof(int32_t) * sched.size());
                    A->h_wsched = sched;
                }
                L.sub("visiting order");
            }
        }
        // (returning 70 MB to the system took 5-10 ms on this thread at 216^3: the codes are released behind the call's back)
        if (code_up.joinable()) code_up.join();
        if (err == hipSuccess) err = err_code;
        L.sub("code upload (rest)");
        defer_release(c, [p = codes_buf.release()] { delete[] p; });
    }
}

// wave-sliced form for variable coefficients (wdia-vv): diagonal records with per-row value blocks
static void low_wave_sliced_variable(Low &L) {
    SLA_LOW_LOCALS(L);
    if (!panel_view && A->use_diag && !A->use_wdia && !A->rp64 && c->wdia_vv && n < ((int64_t)1 << 28) && nnz > 0) {
        // Wave-sliced form for VARIABLE coefficients (banded / stencil structure, arbitrary values): per 128-row slice the
        // sorted union of its diagonal offsets with the two row masks each, and per record a block of 128 values laid out
        // like the rows (lane l holds rows 2l, 2l+1: one 16-byte load).  8 B per stored slot instead of 8 + 1 B per entry
        // plus rowptr, no codes, no LDS.  Taken when at least half of the slots hold an entry.
        const int64_t nsl = (rows + 127) / 128;
        std::vector<int32_t> wptr((size_t)nsl + 1, 0);
        std::vector<uint64_t> wme, wmo;
        std::vector<int32_t> woff;
        std::vector<double> wvb;
        uint64_t lane_mask[2][256];
        int slot_of[256];
        bool wok = true;
        for (int64_t sl = 0; sl < nsl && wok; ++sl) {
            uint64_t present[4] = {0, 0, 0, 0};
            const int64_t rlo = sl * 128, rhi = std::min<int64_t>(rows, rlo + 128);
            for (int64_t i = rlo; i < rhi; ++i)
                for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
                    const int cd = dcodes[(size_t)k];
                    if (!((present[cd >> 6] >> (cd & 63)) & 1)) {
                        present[cd >> 6] |= 1ull << (cd & 63);
                        lane_mask[0][cd] = lane_mask[1][cd] = 0;
                    }
                    lane_mask[(i - rlo) & 1][cd] |= 1ull << ((i - rlo) >> 1);
                }
            const size_t first = wme.size();
            for (int cd = 0; cd < 256; ++cd)   // ascending code = ascending offset
                if ((present[cd >> 6] >> (cd & 63)) & 1) {
                    slot_of[cd] = (int)(wme.size() - first);
                    wme.push_back(lane_mask[0][cd]);
                    wmo.push_back(lane_mask[1][cd]);
                    woff.push_back((int32_t)offs[(size_t)cd]);
                }
            wvb.resize(wme.size() * 128, 0.0);
            for (int64_t i = rlo; i < rhi; ++i)
                for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
                    wvb[(first + (size_t)slot_of[dcodes[(size_t)k]]) * 128 + (size_t)(i - rlo)] = val[k];
            wptr[(size_t)sl + 1] = (int32_t)wme.size();
            if ((int64_t)wme.size() * 64 > nnz + 4096) wok = false;   // less than half of the slots used
            if (wptr[(size_t)sl + 1] - wptr[(size_t)sl] > kWdMaxSliceRecords) wok = false;
        }
        if (wok) {
            A->nwent = (int64_t)wme.size();
            for (int t = 0; t < 8; ++t) { wme.push_back(0); wmo.push_back(0); woff.push_back(0); }
            wvb.resize(wme.size() * 128, 0.0);
            upload((void **)&A->d_wptr, wptr.data(), sizeof(int32_t) * wptr.size());
            upload((void **)&A->d_wme, wme.data(), sizeof(uint64_t) * wme.size());
            upload((void **)&A->d_wmo, wmo.data(), sizeof(uint64_t) * wmo.size());
            upload((void **)&A->d_woff, woff.data(), sizeof(int32_t) * woff.size());
            upload((void **)&A->d_wvblk, wvb.data(), sizeof(double) * wvb.size());
            A->use_wdia = true;
            A->wd_vv = true;
            A->nslices = (int32_t)nsl;
            A->nblk_wd = (int32_t)((nsl + 3) / 4);
        }
    }
}

// LDS-panel form for dense rows (spmv_lpanel_kernel): per (panel, row) entry ranges + the task runs of the persistent grid
static void low_lds_panels(Low &L) {
    SLA_LOW_LOCALS(L);
    if (!panel_view && !A->use_wdia && !A->use_vdict && rows > 0 && err == hipSuccess) {
        // LDS-panel form (spmv_lpanel_kernel): worthwhile when a row has enough entries per kLpW-column panel to
        // keep a wavefront's lanes busy, affordable when the (panel, row) pointer table stays a fraction of the matrix
        const int64_t P = (n + kLpW - 1) / kLpW;
        const int64_t W = ((n + P - 1) / P + 63) / 64 * 64;   // equal panels (a narrow last panel would be all short segments)
        const size_t rpsz = A->rp64 ? sizeof(int64_t) : sizeof(int32_t);
        const int64_t min_seg = c->lp_min_seg;
        if (P <= 4096 && nnz >= min_seg * rows * P && (P + 1) * rows * (int64_t)rpsz <= nnz * 12 / 4) {
            std::vector<int64_t> pp((size_t)((P + 1) * rows));
            par_rows(rows, 1, [&](int, int64_t lo, int64_t hi) {
                for (int64_t i = lo; i < hi; ++i) {
                    const int64_t *cb = col + rowptr[i], *ce = col + rowptr[i + 1];
                    const int64_t *cur = cb;
                    for (int64_t p = 0; p <= P; ++p) {
                        cur = std::lower_bound(cur, ce, p * W);
                        pp[(size_t)(p * rows + i)] = rowptr[i] + (cur - cb);
                    }
                }
            }, 4096);
            if (A->rp64) {
                upload(&A->d_lpp, pp.data(), sizeof(int64_t) * pp.size());
            } else {
                std::vector<int32_t> pp32(pp.begin(), pp.end());
                upload(&A->d_lpp, pp32.data(), sizeof(int32_t) * pp32.size());
            }
            if (err == hipSuccess) err = dev_malloc(c, (void **)&A->d_lpy, sizeof(double) * (size_t)(P * rows));
            // panel-major second copy of the entries (sla_spmv_lpanel.hip: lp_reorder_kernel), built on the device from the arrays just
            // uploaded; when it exists d_lpp is replaced by the P x rows + 1 segment starts into it.  12 B per entry: taken while it
            // (10 B: 16-bit panel offsets) stays below 48 GB and the allocation succeeds (a failure here is not an error: the row-major arrays serve)
            if (err == hipSuccess && c->lp_copy && nnz * 12 <= ((int64_t)48 << 30) && A->d_col && A->d_val) {
                std::vector<int64_t> q((size_t)(P * rows) + 1, 0);
                for (int64_t p = 0; p < P; ++p)
                    for (int64_t i = 0; i < rows; ++i)
                        q[(size_t)(p * rows + i) + 1] = q[(size_t)(p * rows + i)] + ((pp[(size_t)((p + 1) * rows + i)] - pp[(size_t)(p * rows + i)] + 1) & ~(int64_t)1);   // (whole pairs)
                void *dq = nullptr;
                uint16_t *c2 = nullptr;
                double *v2 = nullptr;
                const size_t nnz2 = (size_t)q.back();          // entries + pads
                hipError_t e2 = W < 65535 && (A->rp64 || nnz2 < ((size_t)1 << 31)) ? dev_malloc(c, (void **)&c2, sizeof(uint16_t) * nnz2 + kArraySlack) : hipErrorInvalidValue;
                if (e2 == hipSuccess) e2 = dev_malloc(c, (void **)&v2, sizeof(double) * nnz2 + kArraySlack);
                if (e2 == hipSuccess) e2 = dev_malloc(c, &dq, rpsz * q.size() + kArraySlack);
                if (e2 == hipSuccess) {
                    if (A->rp64) {
                        e2 = hipMemcpy(dq, q.data(), sizeof(int64_t) * q.size(), hipMemcpyHostToDevice);
                    } else {
                        std::vector<int32_t> q32(q.begin(), q.end());
                        e2 = hipMemcpy(dq, q32.data(), sizeof(int32_t) * q32.size(), hipMemcpyHostToDevice);
                    }
                }
                if (e2 == hipSuccess && launch_lp_reorder(c, A->rp64, A->d_lpp, dq, A->d_col, A->d_val, c2, v2, rows, P, (int32_t)W) != SLA_OK) e2 = hipErrorUnknown;
                if (e2 == hipSuccess) {
                    (void)hipFree(A->d_lpp);
                    A->d_lpp = dq;
                    A->d_lpcol = c2;
                    A->d_lpval = v2;
                } else {
                    (void)hipGetLastError();
                    if (c2) (void)hipFree(c2);
                    if (v2) (void)hipFree(v2);
                    if (dq) (void)hipFree(dq);
                }
            }
            // row chunks: ~32 tasks per workgroup of the persistent grid (measured: 8 -> 0.936 ms, 32 -> 0.900 ms, 64 ->
            // 0.902 ms on the 200k-row 1 % matrix), at least 64 rows (4 per wavefront) each
            const int tasks_per_cu = std::max(1, c->lp_tasks);
            const int64_t want = std::max<int64_t>(1, (tasks_per_cu * (int64_t)c->n_cu + P - 1) / P);
            const int64_t chunk = std::max<int64_t>(64, (rows + want - 1) / want);
            const int64_t C = (rows + chunk - 1) / chunk;
            A->lp_P = (int32_t)P;
            A->lp_W = (int32_t)W;
            {   // lanes per segment ~ half the mean segment length (so that two strided loads cover a typical segment)
                const int64_t seg = nnz / (rows * P);
                // measured, 200 k rows x 13 panels, ms per (#>) [64 / 32 / 16 / 8 lanes]: segment 153: 0.90 / 1.15 / 1.33 / 1.80;
                // 92: 0.68 / 0.73 / 0.87 / 1.15; 61: 0.65 / 0.52 / 0.59 / 0.77; 30: 0.61 / 0.36 / 0.36 / 0.43; 15: 0.58 / 0.32 /
                // 0.26 / 0.27 (stream kernel: 2.21 / 1.39 / 1.07 / 0.56 / 0.28)
                A->lp_cfg = seg >= 80 ? 0 : seg >= 40 ? 1 : 2;
                if (c->lp_cfg >= 0) A->lp_cfg = std::min(3, c->lp_cfg);
            }
            int64_t clo = n, chi = -1;
            for (int64_t i = 0; i < rows; ++i)
                if (rowptr[i + 1] > rowptr[i]) {   // canonical CSR: first / last entry of a row are its min / max column
                    clo = std::min(clo, col[rowptr[i]]);
                    chi = std::max(chi, col[rowptr[i + 1] - 1]);
                }
            A->lp_col_lo = (int32_t)clo;
            A->lp_col_hi = (int32_t)chi;
            A->lp_chunk = (int32_t)chunk;
            A->lp_C = (int32_t)C;
            // tasks (panel-major) are dealt out in contiguous runs of equal ENTRY counts, one run per workgroup
            const int64_t ntasks = P * C;
            const int G = (int)std::min<int64_t>(ntasks, c->n_cu);
            // (a segment costs a memory round trip however short it is -- with entries alone balanced, workgroups holding
            // 34-entry segments took 2.4x as long as those with 154-entry ones: weigh a row like row_cost entries)
            const int64_t row_cost = c->lp_rowcost;
            std::vector<int64_t> upto((size_t)ntasks + 1, 0);   // weight before task t
            for (int64_t t = 0; t < ntasks; ++t) {   // (panel-major; row-chunk-major was tried: 0.906 -> 0.971 ms, x reloaded per task)
                const int64_t p = t / C, cc = t % C;
                const int64_t lo = cc * chunk, hi = std::min<int64_t>(rows, lo + chunk);
                int64_t w = 0;
                for (int64_t i = lo; i < hi; ++i) w += pp[(size_t)((p + 1) * rows + i)] - pp[(size_t)(p * rows + i)];
                upto[(size_t)t + 1] = upto[(size_t)t] + w + row_cost * (hi - lo);
            }
            std::vector<int32_t> tb((size_t)G + 1, 0);
            for (int g = 1; g < G; ++g) {
                const int64_t target = upto[(size_t)ntasks] / G * g;
                tb[(size_t)g] = (int32_t)(std::lower_bound(upto.begin(), upto.end(), target) - upto.begin());
                tb[(size_t)g] = std::max(tb[(size_t)g], tb[(size_t)g - 1]);
            }
            tb[(size_t)G] = (int32_t)ntasks;
            if (dbg_lower) {
                fprintf(stderr, "[sla] lpanel: P=%lld C=%lld chunk=%lld G=%d total=%lld tb:", (long long)P, (long long)C, (long long)chunk, G, (long long)upto[(size_t)ntasks]);
                for (int g = 0; g <= G; g += std::max(1, G / 16)) fprintf(stderr, " %d", tb[(size_t)g]);
                fprintf(stderr, "\n");
            }
            A->lp_G = G;
            upload((void **)&A->d_lpt, tb.data(), sizeof(int32_t) * tb.size());
            A->use_lpanel = err == hipSuccess;
        }
    }
}

// Canonical col / val of a value-indexed matrix, written on the device from its 1-byte codes the first time something needs them
// (sla_csr_export and through it transpose / the preconditioner builders, (##), a CSR kernel once the value-indexed form is peeled off by
// an option).  Never called inside a stream capture: it allocates.
int csr_ensure_canon(sla_csr *A) {
    if (!A || !A->canon_lazy) return SLA_OK;
    sla_ctx *c = A->ctx;
    Bind bind(c);
    const int64_t nnz = A->nnz;
    if (!A->d_vcode || !A->d_vdoff || !A->d_vdval || A->rp64) return fail(SLA_ERR_INVALID, "csr_ensure_canon: the matrix has no value-indexed codes");
    hipError_t e = hipSuccess;
    if (!A->d_col) {
        e = dev_malloc(c, (void **)&A->d_col, sizeof(int32_t) * (size_t)nnz + kArraySlack);
        if (e == hipSuccess) e = hipMemsetAsync((char *)A->d_col + sizeof(int32_t) * (size_t)nnz, 0, kArraySlack, stream_of(c));
    }
    if (e == hipSuccess && !A->d_val) {
        e = dev_malloc(c, (void **)&A->d_val, sizeof(double) * (size_t)nnz + kArraySlack);
        if (e == hipSuccess) e = hipMemsetAsync((char *)A->d_val + sizeof(double) * (size_t)nnz, 0, kArraySlack, stream_of(c));
    }
    if (e != hipSuccess) return fail(SLA_ERR_ALLOC, std::string("canonical arrays of a value-indexed matrix: ") + hipGetErrorString(e));
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((A->rows + 255) / 256, (int64_t)c->n_cu * 16));
    hipLaunchKernelGGL(vd_expand_kernel, dim3(grid), dim3(256), 0, stream_of(c), A->rows, A->row_begin, (const int32_t *)A->d_rowptr, A->d_vcode,
                       A->d_vdoff, A->d_vdval, A->d_col, A->d_val, (int64_t)0, (int64_t)0);
    SLA_HIP_TRY(hipGetLastError());
    SLA_TRY(launch_col_slack_fill(c, A->d_col, nnz));
    SLA_HIP_TRY(hipStreamSynchronize(stream_of(c)));
    A->canon_lazy = false;
    return SLA_OK;
}

// canonical-CSR check of the caller's columns (sla_csr_from_csr_rows): 0 fine, 1 some index out of bounds, 2 not strictly ascending
// inside a row; the lowest-numbered kind of violation wins so that the verdict does not depend on the thread count
static int validate_columns(int64_t rows, int64_t n, const int64_t *rowptr, const int64_t *col) {
    std::vector<int> bad((size_t)host_threads(), 0);
    par_rows(rows, 1, [&](int t, int64_t lo, int64_t hi) {
        int b_ = 0;
        for (int64_t i = lo; i < hi && b_ != 1; ++i)
            for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
                if (col[k] < 0 || col[k] >= n) { b_ = 1; break; }
                if (k > rowptr[i] && col[k] <= col[k - 1]) b_ = 2;
            }
        bad[(size_t)t] = b_;
    });
    if (std::find(bad.begin(), bad.end(), 1) != bad.end()) return 1;
    return std::find(bad.begin(), bad.end(), 2) != bad.end() ? 2 : 0;
}

int csr_upload(sla_ctx *c, int64_t m, int64_t n, int64_t row_begin, int64_t rows, const int64_t *rowptr,
               const int64_t *col, const double *val, sla_csr **out, bool panel_view, bool validate_cols) {
    const int64_t nnz = rowptr[rows];
    if (n > (int64_t)std::numeric_limits<int32_t>::max() || rows >= (int64_t)std::numeric_limits<int32_t>::max()) {
        // (rows differs per rank: a rank that fails here alone still owes its peers the agreement collective below)
        const int rc = fail(SLA_ERR_INVALID, "matrix dimension exceeds the 32-bit device index width");
        return panel_view ? rc : csr_reject(c, rc);
    }
    static const bool dbg_lower = getenv("SLA_DEBUG_LOWER") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    sla_csr *A = new sla_csr();
    auto lap = [&](const char *what) {   // phase times: kept with the matrix (sla_csr_lower_info), printed under SLA_DEBUG_LOWER
        if (panel_view) return;
        const auto t = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(t - t_last).count();
        char buf[96];
        snprintf(buf, sizeof(buf), "%s=%.2f;", what, ms);
        A->lower_log += buf;
        if (dbg_lower) fprintf(stderr, "[sla] lowering: %-28s %7.1f ms\n", what, ms);
        t_last = t;
    };
    A->ctx = c;
    A->m = m;
    A->n = n;
    A->row_begin = row_begin;
    A->rows = rows;
    A->nnz = nnz;
    // (SLA_FORCE_RP64=1: test hook -- run the 64-bit row-pointer instantiations of the kernels on small matrices)
    // (SLA_FORCE_RP64=2: the parent only -- its column-panel views keep their natural 32-bit width, the mixed case of a > 2^31-entry matrix)
    A->rp64 = nnz > (int64_t)std::numeric_limits<int32_t>::max() || c->force_rp64 == 1 || (c->force_rp64 == 2 && !panel_view);
    Low L{c, A, m, n, row_begin, rows, nnz, rowptr, col, val, panel_view, dbg_lower};
    hipError_t &err = L.err;
    int diag_not = 1;
    // The caller's columns are checked NEXT TO the first analyses (which compare and subtract column values, never index with them); the
    // verdict is taken before anything uses a column as an index (the LDS-panel / flat / tile builders, the exchange plan).
    int bad_cols = 0;
    L.validate = validate_cols && nnz > 0;
    try {   // (a host allocation failing inside an analysis leaves through the agreement collective below like a device one)
        // The row blocks (a sequential greedy pass over the rows, 5 ms at 10 M) and isDiagonalSM are built at the head of the background
        // thread below; here only the longest row, which the first analysis asks for (parallel).
        {
            std::vector<int64_t> mx((size_t)host_threads(), 0);
            par_rows(rows, 1, [&](int t, int64_t lo, int64_t hi) {
                int64_t v = 0;
                for (int64_t i = lo; i < hi; ++i) v = std::max(v, rowptr[i + 1] - rowptr[i]);
                mx[(size_t)t] = v;
            });
            A->max_row_nnz = *std::max_element(mx.begin(), mx.end());
        }
        std::atomic<int> rb_ready{0};
        auto await_rb = [&] {              // (the analyses that walk row blocks: not the value-indexed one)
            while (!rb_ready.load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(50));
        };
        lap("row blocks");
        // The canonical arrays (narrowed columns, values, row pointers, row-block tables: 12 B per entry over PCIe from pageable memory)
        // go up on a background thread WHILE the host analyses of the storage forms run (round 4: the two were 106 ms + 106 ms in a row
        // at 70 M entries; the upload is bound by the staging copies of one or two threads, the analyses use the other cores).
        CanonUpload cu;
        Low Lup = L;                       // (own error slot)
        L.cu = &cu;
        std::thread up([&] {
            Bind bind(c);                  // (a new thread starts on device 0)
            struct Ready { std::atomic<int> &f; ~Ready() { f.store(1, std::memory_order_release); } };
            try {
                {
                    Ready ready{rb_ready};   // (set however the block is left: the main thread waits for it)
                    int64_t mx_unused = 0;
                    build_row_blocks(rows, rowptr, L.rb, mx_unused, c->row_align, c->rb_nnz);
                    A->nrb = (int32_t)L.rb.size() - 1;
                    diag_not = host_is_diagonal(rows, row_begin, rowptr, col) ? 0 : 1;
                    Lup.rb = L.rb;
                }
                low_csr_arrays(Lup, &cu);
            } catch (const std::bad_alloc &) {
                if (Lup.err == hipSuccess) Lup.err = hipErrorOutOfMemory;
            }
        });
        struct Joiner {                    // (an exception below must not unwind past a joinable thread -- nor leave it waiting for the decision)
            std::thread &t;
            CanonUpload &cu;
            ~Joiner() { cu.decided.store(1, std::memory_order_release); if (t.joinable()) t.join(); }
        } up_joiner{up, cu};
        // (an analysis whose forms the context's options peel off is not run: `wdia = 0 vdict = 0 diag = 0` -- bench.py's plain-CSR block -- paid
        // 48 + 15 ms for dictionaries nothing would read; options steer the lowering of matrices created AFTER they are set)
        if (c->wdia || c->vdict) low_value_indexed(L);
        L.sub("visiting order + return");
        bool verdict_with_upload = false;
        if (L.validate) {   // the pair-coding pass checked the columns on its way unless it left early (not a value-indexed matrix): then the
            // upload thread does, while it narrows them (nothing between here and its join uses a column as an INDEX: the window / offset
            // analyses compare and subtract column values)
            if (L.col_verdict >= 0) {
                bad_cols = L.col_verdict;
                if (bad_cols && err == hipSuccess) err = hipErrorInvalidValue;   // (nothing below runs; reported after the agreement)
            } else if (cu.stop.load() == 0 && nnz > 0) {
                cu.validate = true;
                verdict_with_upload = true;
            } else {
                bad_cols = validate_columns(rows, n, rowptr, col);
                if (bad_cols && err == hipSuccess) err = hipErrorInvalidValue;
            }
        }
        // value-indexed after all: the rest of the canonical entry arrays is written on the device from the codes (option canon_device)
        cu.decided.store(1, std::memory_order_release);   // (matrices the analysis did not look at: rp64, rows too long, no entries)
        lap("pair dictionary + wave slices");
        // (the LDS x windows of the row blocks serve the CSR-stream / dictionary-code kernels only: skipped with them, see below)
        await_rb();
        if (L.rb.empty()) throw std::bad_alloc();   // (the background thread could not build them)
        if (!(A->use_wdia && c->wdia && c->diag_lazy)) low_xwin_statistics(L);
        lap("x-window statistics");
        // (the 1-byte column codes serve the dictionary-code kernel and the variable-coefficient slices: a matrix that just took the
        // constant-coefficient wave-sliced form needs neither -- unless the context's knobs peel that form off again (A/B runs), which
        // is decided when the matrix is created: two passes over the entries and 1 B per entry of upload saved, round 4)
        if (!(A->use_wdia && c->wdia && c->diag_lazy) && c->diag) low_diagonal_dictionary(L);
        lap("offset dictionary");
        low_wave_sliced_variable(L);
        lap("variable-coefficient slices");
        L.sub("(before the join)");
        up.join();
        L.sub("join of the upload thread");
        if (verdict_with_upload) {
            // (an upload that failed or was skipped checked nothing: the separate pass then)
            if (Lup.err == hipSuccess && cu.done_col == nnz && A->d_col && A->d_rowptr) {
                bad_cols = cu.col_bad.load();
                if (!bad_cols && validate_columns_device(A, n, &bad_cols) != SLA_OK) bad_cols = validate_columns(rows, n, rowptr, col);
            } else {
                bad_cols = validate_columns(rows, n, rowptr, col);
            }
            if (bad_cols && err == hipSuccess) err = hipErrorInvalidValue;
        }
        if (err == hipSuccess) err = Lup.err;
        if (err == hipSuccess && cu.stop.load()) {
            // value-indexed: nothing of col / val crossed PCIe.  The arrays are written on the device from the codes -- now, or (option
            // canon_lazy, default) when something first asks for them: export, transpose, a CSR kernel after the form was peeled off
            A->canon_lazy = true;
            A->lower_log += "canonical entries over PCIe (fraction)=0.000;";
            if (!(c->canon_lazy && spmv_value_indexed(A, false))) {
                const int rc_c = csr_ensure_canon(A);
                if (rc_c != SLA_OK) err = hipErrorOutOfMemory;
                L.sub("col / val written on the device");
            } else {
                A->lower_log += "canonical arrays lazy=1;";
            }
        }
        lap("canonical CSR upload (rest)");
        low_lds_panels(L);                 // (its panel-major copy is written by a device kernel from the canonical arrays)
        lap("LDS panel table");
        if (!panel_view && err == hipSuccess && !(A->use_lpanel && c->lpanel) && c->lflat && nnz > 0 &&
            (c->lflat == 2 || !(A->use_wdia || A->use_vdict || A->use_diag)) && lflat_candidate(A, n, rows)) {   // medium rows without band structure: the flat LDS-panel form (device-built)
            std::vector<int64_t> plo((size_t)host_threads(), n), phi((size_t)host_threads(), -1);
            par_rows(rows, 1, [&](int t, int64_t lo, int64_t hi) {   // (canonical CSR: first / last entry of a row are its min / max column)
                int64_t a = n, b = -1;
                for (int64_t i = lo; i < hi; ++i)
                    if (rowptr[i + 1] > rowptr[i]) { a = std::min(a, col[rowptr[i]]); b = std::max(b, col[rowptr[i + 1] - 1]); }
                plo[(size_t)t] = a;
                phi[(size_t)t] = b;
            });
            const int rc_lf = build_lflat(A, n, rows, *std::min_element(plo.begin(), plo.end()), *std::max_element(phi.begin(), phi.end()));
            if (rc_lf != SLA_OK && err == hipSuccess) err = hipErrorUnknown;
            lap("flat LDS-panel form");
        }
    } catch (const std::bad_alloc &) {
        if (err == hipSuccess) err = hipErrorOutOfMemory;
    }
    if (panel_view) {
        if (err != hipSuccess) {
            sla_csr_destroy(A);
            return fail(SLA_ERR_ALLOC, std::string("CSR upload: ") + hipGetErrorString(err));
        }
        *out = A;
        return SLA_OK;
    }
    // Every rank enters the agreement collective, failed or not (a rank returning early would leave its peers blocked in
    // it): the all-reduced maximum carries isDiagonalSM's verdict in bit 0 and "some rank failed" as a value >= 2.
    int agree = err != hipSuccess ? 2 : diag_not;
    int rc = dist_allreduce_max_i32(c, &agree);
    if (bad_cols) {   // (every rank has been through the agreement: its peers learn "failed on another rank")
        sla_csr_destroy(A);
        return bad_cols == 1 ? fail(SLA_ERR_OOB, "insertSpMatrix : index out of bounds")
                             : fail(SLA_ERR_INVALID, "columns must be strictly ascending inside a row (canonical CSR)");
    }
    if (err != hipSuccess) {
        sla_csr_destroy(A);
        return fail(SLA_ERR_ALLOC, std::string("CSR upload: ") + hipGetErrorString(err));
    }
    if (rc == SLA_OK && agree >= 2) rc = fail(SLA_ERR_INVALID, "matrix creation failed on another rank of the row-sharded job");
    if (rc != SLA_OK) {
        sla_csr_destroy(A);
        return rc;
    }
    A->is_diagonal = agree == 0;
    lap("agreement");
    rc = build_xplan(A, rows, rowptr, col);
    if (rc == SLA_OK) rc = build_overlap_lists(A, m, n, row_begin, rows, rowptr, col);
    lap("exchange plan");
    if (rc == SLA_OK) rc = build_tiles(A, n, rows, rowptr, col, val);
    lap("tile form");
    {   // (contains the ranks' agreement on the exchange pattern: every rank of a sharded context gets here, failed or not)
        const int rc_ag = build_ag_plan(A, rc != SLA_OK);
        if (rc == SLA_OK) rc = rc_ag;
    }
    if (rc == SLA_OK && !(A->use_lpanel && c->lpanel) && !(A->use_lflat && c->lflat) && !A->use_tiles) rc = build_panels(A, m, n, row_begin, rows, rowptr, col, val);
    lap("column panels");
    if (rc != SLA_OK) {
        sla_csr_destroy(A);
        return rc;
    }
    *out = A;
    return SLA_OK;
}

}  // namespace sla
