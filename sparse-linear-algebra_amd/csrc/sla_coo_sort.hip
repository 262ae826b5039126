// sla_coo_sort.hip -- device side of "lower once" for large triple lists.
//
// fromListSM (Data/Sparse/SpMatrix.hs:205-224) = foldl' of IntMap inserts: ascending (row, col) order,
// the LAST duplicate wins.  On the device: a STABLE radix sort (rocPRIM) of the 64-bit key (row << 32 | col)
// carrying the input position keeps duplicates in input order, the last member of each key group is kept
// (or the group is summed left-to-right in input order for SLA_DUP_SUM), an exclusive scan gives the output
// slots and a binary search per row gives csPtrV's row pointers (vector/.../Vector/Utils.hs:12-26).
// The result is bit-identical to the host builder in sla_csr_build.cpp (tests/test_gpu_edge_cases.py).
#include <hip/hip_runtime.h>

#include <cstring>  // rocPRIM's texture iterator calls the host memset without including it

#include <rocprim/rocprim.hpp>

#include "sla_internal.hpp"

namespace sla {

namespace {

__global__ void __launch_bounds__(256) coo_keys_kernel(int64_t nnz, const int64_t *row, const int64_t *col, uint64_t *key,
                                                        uint32_t *idx) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * 256) {
        key[i] = ((uint64_t)row[i] << 32) | (uint64_t)(uint32_t)col[i];
        idx[i] = (uint32_t)i;
    }
}

// flag[i] = 1 when sorted entry i is the LAST of its (row, col) group
__global__ void __launch_bounds__(256) coo_flag_kernel(int64_t nnz, const uint64_t *key, uint32_t *flag) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * 256)
        flag[i] = (i + 1 == nnz || key[i + 1] != key[i]) ? 1u : 0u;
}

__global__ void __launch_bounds__(256) coo_emit_kernel(int64_t nnz, const uint64_t *key, const uint32_t *idx, const uint32_t *flag,
                                                        const uint32_t *pos, const double *val, int sum_dups, int64_t *col_out,
                                                        double *val_out, int64_t *row_out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * 256) {
        if (!flag[i]) continue;
        const uint64_t k = key[i];
        double v = val[idx[i]];  // last member = last in input order (stable sort)
        if (sum_dups) {          // left-to-right in input order, like the host builder
            int64_t j = i;
            while (j > 0 && key[j - 1] == k) --j;
            v = val[idx[j]];
            for (int64_t t = j + 1; t <= i; ++t) v += val[idx[t]];
        }
        const uint32_t p = pos[i];
        col_out[p] = (int64_t)(uint32_t)k;
        row_out[p] = (int64_t)(k >> 32);
        val_out[p] = v;
    }
}

// rowptr[r] = first output slot whose row is >= r  (csPtrV: empty rows repeat the previous value)
__global__ void __launch_bounds__(256) coo_rowptr_kernel(int64_t m, int64_t nout, const int64_t *row_out, int64_t *rowptr) {
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r <= m; r += (int64_t)gridDim.x * 256) {
        int64_t lo = 0, hi = nout;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (row_out[mid] < r) lo = mid + 1;
            else hi = mid;
        }
        rowptr[r] = lo;
    }
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <class T> T *as() { return (T *)p; }
};

}  // namespace

bool device_coo_supported(int64_t m, int64_t n, int64_t nnz) {
    return m < ((int64_t)1 << 31) && n < ((int64_t)1 << 32) && nnz > 0 && nnz < ((int64_t)1 << 32) - 1;
}

// Indices must already be bounds-checked by the caller.  Synchronises the context stream.
int device_coo_to_csr(sla_ctx *c, int64_t m, int64_t n, int64_t nnz, const int64_t *row, const int64_t *col,
                      const double *val, int dup_policy, HostCsr &out) {
    hipStream_t st = stream_of(c);
    DevBuf d_row, d_col, d_val, d_key, d_key2, d_idx, d_idx2, d_flag, d_pos, d_tmp, d_colo, d_valo, d_rowo, d_rp;
    const size_t N = (size_t)nnz;
    SLA_HIP_TRY(d_row.alloc(8 * N));
    SLA_HIP_TRY(d_col.alloc(8 * N));
    SLA_HIP_TRY(d_val.alloc(8 * N));
    SLA_HIP_TRY(d_key.alloc(8 * N));
    SLA_HIP_TRY(d_key2.alloc(8 * N));
    SLA_HIP_TRY(d_idx.alloc(4 * N));
    SLA_HIP_TRY(d_idx2.alloc(4 * N));
    {   // the three triple arrays go up side by side (sla_xfer.cpp: own pinned lanes; round 4 -- one pageable hipMemcpyAsync after the other
        // took 0.25 s for the 1.7 GB of 70 M triples)
        hipError_t e3[3] = {hipSuccess, hipSuccess, hipSuccess};
        void *dst3[3] = {d_row.p, d_col.p, d_val.p};
        const void *src3[3] = {row, col, val};
        std::thread th[2];
        for (int t = 0; t < 2; ++t)
            th[t] = std::thread([&, t] {
                Bind bind(c);   // (a new thread starts on device 0)
                e3[t + 1] = xfer_copy(c, dst3[t + 1], src3[t + 1], 8 * N, hipMemcpyHostToDevice);
            });
        e3[0] = xfer_copy(c, dst3[0], src3[0], 8 * N, hipMemcpyHostToDevice);
        for (auto &t : th) t.join();
        for (hipError_t e : e3) SLA_HIP_TRY(e);
    }
    const int grid = (int)std::min<int64_t>((nnz + 255) / 256, 4096);
    hipLaunchKernelGGL(coo_keys_kernel, dim3(grid), dim3(256), 0, st, nnz, d_row.as<int64_t>(), d_col.as<int64_t>(),
                       d_key.as<uint64_t>(), d_idx.as<uint32_t>());
    // stable LSD radix sort on the significant key bits only
    int row_bits = 1, col_bits = 1;
    while (((int64_t)1 << row_bits) < m) ++row_bits;
    while (((int64_t)1 << col_bits) < n) ++col_bits;
    size_t tmp_bytes = 0;
    SLA_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_key.as<uint64_t>(), d_key2.as<uint64_t>(), d_idx.as<uint32_t>(),
                                          d_idx2.as<uint32_t>(), N, 0, 32 + row_bits, st));
    SLA_HIP_TRY(d_tmp.alloc(tmp_bytes));
    SLA_HIP_TRY(rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, d_key.as<uint64_t>(), d_key2.as<uint64_t>(), d_idx.as<uint32_t>(),
                                          d_idx2.as<uint32_t>(), N, 0, 32 + row_bits, st));
    (void)col_bits;
    // rows / cols are no longer needed on the device: reuse their storage for flags / positions
    uint32_t *flag = d_row.as<uint32_t>(), *pos = d_col.as<uint32_t>();
    hipLaunchKernelGGL(coo_flag_kernel, dim3(grid), dim3(256), 0, st, nnz, d_key2.as<uint64_t>(), flag);
    size_t scan_bytes = 0;
    SLA_HIP_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, flag, pos, 0u, N, rocprim::plus<uint32_t>(), st));
    DevBuf d_scan;
    SLA_HIP_TRY(d_scan.alloc(scan_bytes));
    SLA_HIP_TRY(rocprim::exclusive_scan(d_scan.p, scan_bytes, flag, pos, 0u, N, rocprim::plus<uint32_t>(), st));
    uint32_t last_pos = 0, last_flag = 0;
    SLA_HIP_TRY(hipMemcpyAsync(&last_pos, pos + (N - 1), 4, hipMemcpyDeviceToHost, st));
    SLA_HIP_TRY(hipMemcpyAsync(&last_flag, flag + (N - 1), 4, hipMemcpyDeviceToHost, st));
    SLA_HIP_TRY(hipStreamSynchronize(st));
    const int64_t nout = (int64_t)last_pos + (int64_t)last_flag;
    SLA_HIP_TRY(d_colo.alloc(8 * (size_t)nout));
    SLA_HIP_TRY(d_valo.alloc(8 * (size_t)nout));
    SLA_HIP_TRY(d_rowo.alloc(8 * (size_t)nout));
    SLA_HIP_TRY(d_rp.alloc(8 * (size_t)(m + 1)));
    hipLaunchKernelGGL(coo_emit_kernel, dim3(grid), dim3(256), 0, st, nnz, d_key2.as<uint64_t>(), d_idx2.as<uint32_t>(), flag, pos,
                       d_val.as<double>(), dup_policy == SLA_DUP_SUM ? 1 : 0, d_colo.as<int64_t>(), d_valo.as<double>(),
                       d_rowo.as<int64_t>());
    const int grid_r = (int)std::min<int64_t>((m + 1 + 255) / 256, 4096);
    hipLaunchKernelGGL(coo_rowptr_kernel, dim3(grid_r), dim3(256), 0, st, m, nout, d_rowo.as<int64_t>(), d_rp.as<int64_t>());
    SLA_HIP_TRY(hipGetLastError());
    out.m = m;
    out.n = n;
    out.rowptr.resize((size_t)m + 1);
    out.col.resize((size_t)nout);
    out.val.resize((size_t)nout);
    SLA_HIP_TRY(hipStreamSynchronize(st));
    {
        hipError_t e2 = hipSuccess;
        std::thread th;
        if (nout)
            th = std::thread([&] {
                Bind bind(c);
                e2 = xfer_copy(c, out.val.data(), d_valo.p, 8 * (size_t)nout, hipMemcpyDeviceToHost);
            });
        hipError_t e1 = xfer_copy(c, out.rowptr.data(), d_rp.p, 8 * (size_t)(m + 1), hipMemcpyDeviceToHost);
        if (e1 == hipSuccess && nout) e1 = xfer_copy(c, out.col.data(), d_colo.p, 8 * (size_t)nout, hipMemcpyDeviceToHost);
        if (th.joinable()) th.join();
        SLA_HIP_TRY(e1);
        SLA_HIP_TRY(e2);
    }
    return SLA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// transposeSM (SpMatrix.hs:717) of a lowered matrix, on the device (round 4).  The first (<#) / cgneStep of a matrix used to export its
// 12 B per entry to the host, transpose there on one thread and lower the result: 0.99 s at 216^3, ten times sla_csr_from_csr.  Here the
// entries are sorted by (column, row) where they are (stable rocPRIM radix sort of 64-bit keys carrying the entry's position), the
// transposed CSR arrays are emitted on the device and only then copied down for the lowering analyses of the transpose.
namespace {

template <typename RP>
__global__ void __launch_bounds__(256) tr_keys_kernel(int64_t rows, int64_t nnz, const RP *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                       uint64_t *key, uint32_t *idx) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * 256) {
        int64_t lo = 0, hi = rows;                    // last row whose first entry is <= k (empty rows: the last of equal starts)
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)rowptr[mid] <= k) lo = mid;
            else hi = mid;
        }
        key[k] = ((uint64_t)(uint32_t)col[k] << 32) | (uint64_t)(uint32_t)lo;
        idx[k] = (uint32_t)k;
    }
}

__global__ void __launch_bounds__(256) tr_emit_kernel(int64_t nnz, const uint64_t *__restrict__ key, const uint32_t *__restrict__ idx,
                                                       const double *__restrict__ val, int64_t *trow, int64_t *tcol, double *tval) {
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < nnz; o += (int64_t)gridDim.x * 256) {
        const uint64_t k = key[o];
        trow[o] = (int64_t)(k >> 32);
        tcol[o] = (int64_t)(uint32_t)k;
        tval[o] = val[idx[o]];
    }
}

}  // namespace

// t = transpose of A's row block: t.m = A->n rows, columns = LOCAL row ids 0 .. A->rows - 1 (like transpose_csr of the exported block).
// *done = false: this path could not take the matrix (sizes, device memory) -- the host path does then.
int device_transpose_to_host(sla_csr *A, HostCsr &t, bool *done) {
    *done = false;
    sla_ctx *c = A->ctx;
    const int64_t nnz = A->nnz, rows = A->rows;
    if (nnz <= 0 || nnz >= ((int64_t)1 << 32) - 1 || rows >= ((int64_t)1 << 32) || A->n >= ((int64_t)1 << 32)) return SLA_OK;
    SLA_TRY(csr_ensure_canon(A));
    hipStream_t st = stream_of(c);
    DevBuf d_key, d_key2, d_idx, d_idx2, d_trow, d_tcol, d_tval, d_rp, d_tmp;
    const size_t N = (size_t)nnz;
    hipError_t e = d_key.alloc(8 * N);
    if (e == hipSuccess) e = d_key2.alloc(8 * N);
    if (e == hipSuccess) e = d_idx.alloc(4 * N);
    if (e == hipSuccess) e = d_idx2.alloc(4 * N);
    if (e == hipSuccess) e = d_trow.alloc(8 * N);
    if (e == hipSuccess) e = d_tcol.alloc(8 * N);
    if (e == hipSuccess) e = d_tval.alloc(8 * N);
    if (e == hipSuccess) e = d_rp.alloc(8 * (size_t)(A->n + 1));
    int col_bits = 1;
    while (((int64_t)1 << col_bits) < A->n) ++col_bits;
    size_t tmp_bytes = 0;
    if (e == hipSuccess)
        e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_key.as<uint64_t>(), d_key2.as<uint64_t>(), d_idx.as<uint32_t>(), d_idx2.as<uint32_t>(), N, 0,
                                      (unsigned)(32 + col_bits), st);
    if (e == hipSuccess) e = d_tmp.alloc(tmp_bytes);
    if (e != hipSuccess) {   // (no room for the scratch: not an error)
        (void)hipGetLastError();
        return SLA_OK;
    }
    const int grid = (int)std::min<int64_t>((nnz + 255) / 256, 8192);
    if (A->rp64) hipLaunchKernelGGL((tr_keys_kernel<int64_t>), dim3(grid), dim3(256), 0, st, rows, nnz, (const int64_t *)A->d_rowptr, A->d_col, d_key.as<uint64_t>(), d_idx.as<uint32_t>());
    else hipLaunchKernelGGL((tr_keys_kernel<int32_t>), dim3(grid), dim3(256), 0, st, rows, nnz, (const int32_t *)A->d_rowptr, A->d_col, d_key.as<uint64_t>(), d_idx.as<uint32_t>());
    SLA_HIP_TRY(hipGetLastError());
    SLA_HIP_TRY(rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, d_key.as<uint64_t>(), d_key2.as<uint64_t>(), d_idx.as<uint32_t>(), d_idx2.as<uint32_t>(), N, 0,
                                          (unsigned)(32 + col_bits), st));
    hipLaunchKernelGGL(tr_emit_kernel, dim3(grid), dim3(256), 0, st, nnz, d_key2.as<uint64_t>(), d_idx2.as<uint32_t>(), A->d_val, d_trow.as<int64_t>(),
                       d_tcol.as<int64_t>(), d_tval.as<double>());
    const int grid_r = (int)std::min<int64_t>((A->n + 1 + 255) / 256, 4096);
    hipLaunchKernelGGL(coo_rowptr_kernel, dim3(grid_r), dim3(256), 0, st, A->n, nnz, d_trow.as<int64_t>(), d_rp.as<int64_t>());
    SLA_HIP_TRY(hipGetLastError());
    SLA_HIP_TRY(hipStreamSynchronize(st));
    t.m = A->n;
    t.n = A->rows;
    t.rowptr.resize((size_t)A->n + 1);
    t.col.resize(N);
    t.val.resize(N);
    {
        hipError_t e2 = hipSuccess;
        std::thread th([&] {
            Bind bind(c);
            e2 = xfer_copy(c, t.val.data(), d_tval.p, 8 * N, hipMemcpyDeviceToHost);
        });
        hipError_t e1 = xfer_copy(c, t.rowptr.data(), d_rp.p, 8 * (size_t)(A->n + 1), hipMemcpyDeviceToHost);
        if (e1 == hipSuccess) e1 = xfer_copy(c, t.col.data(), d_tcol.p, 8 * N, hipMemcpyDeviceToHost);
        th.join();
        SLA_HIP_TRY(e1);
        SLA_HIP_TRY(e2);
    }
    *done = true;
    return SLA_OK;
}

// Canonical-CSR check of the caller's columns ON THE DEVICE (sla_csr_from_csr_rows, round 5): the narrowed int32 columns and the row
// pointers are in HBM already, and a pass over them costs ~1 ms at 330 M entries where the host pass cost 167 ms (its own) or slowed
// the upload's narrowing threads by as much (fused).  The host only ORs the int64 columns together while it narrows them: any bit
// from 31 up = a negative or >= 2^31 column = out of bounds, which the narrowed copy could no longer show.
// verdict: 0 fine, 1 some column >= n, 2 not strictly ascending inside a row (validate_columns' codes; the lowest kind wins).
template <typename RP>
__global__ void __launch_bounds__(256) validate_cols_kernel(int64_t rows, int64_t n, const RP *__restrict__ rowptr, const int32_t *__restrict__ col, int *flags) {
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows; i += (int64_t)gridDim.x * 256) {
        int64_t prev = -1;
        for (RP k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            const int64_t cv = (int64_t)(uint32_t)col[k];     // (negative after narrowing = bit 31 set: caught on the host, >= n here too)
            if (cv >= n) bad |= 1;
            else if (cv <= prev) bad |= 2;
            prev = cv;
        }
    }
    if (bad & 1) flags[0] = 1;
    if (bad & 2) flags[1] = 1;
}
int validate_columns_device(sla_csr *A, int64_t n, int *verdict) {
    sla_ctx *c = A->ctx;
    *verdict = 0;
    if (A->nnz <= 0 || A->rows <= 0) return SLA_OK;
    if (!A->d_col || !A->d_rowptr) return fail(SLA_ERR_INVALID, "validate_columns_device: the canonical arrays are not on the device");
    int *d = (int *)(c->d_result + 1536);   // two ints of the context's scratch
    hipStream_t st = stream_of(c);
    SLA_HIP_TRY(hipMemsetAsync(d, 0, 2 * sizeof(int), st));
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((A->rows + 255) / 256, (int64_t)c->n_cu * 32));
    if (A->rp64) hipLaunchKernelGGL((validate_cols_kernel<int64_t>), dim3(grid), dim3(256), 0, st, A->rows, n, (const int64_t *)A->d_rowptr, A->d_col, d);
    else hipLaunchKernelGGL((validate_cols_kernel<int32_t>), dim3(grid), dim3(256), 0, st, A->rows, n, (const int32_t *)A->d_rowptr, A->d_col, d);
    SLA_HIP_TRY(hipGetLastError());
    int h[2] = {0, 0};
    SLA_HIP_TRY(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st));
    SLA_HIP_TRY(hipStreamSynchronize(st));
    *verdict = h[0] ? 1 : h[1] ? 2 : 0;
    return SLA_OK;
}

}  // namespace sla
