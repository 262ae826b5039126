// sla_tiles_build.hip -- the tile form of sla_lower_tiles.cpp built ON THE DEVICE (round 4: "lowered once" for BASELINE config 3a cost
// 1.2 s of host re-ordering and 4 GB of PCIe for the second, tile-major copy of the entries -- 390 BiCGSTAB steps at the
// steady-state rate, while linSolve0 runs <= 200).  The canonical CSR arrays are on the device already; the tile order is a sort:
//
//   entry k of row i (slice s, panel j = col >> shift, layer l = its rank inside the (row, panel) segment, c = col inside the panel)
//   ->  key (s, j, l, c);  STABLE radix sort of (key, k) (rocPRIM): the entries of one (slice, panel, layer) in ascending COLUMN order
//   (equal columns: input order = ascending rows);  tlidx / tlval gathered through the sorted positions, bit 31 of tlidx marks the
//   first entry of every layer >= 1 of a tile;  tloff by binary search for every (slice, panel) boundary.
//
// Bit-identical to the host builder (tests/test_gpu_tiles.py runs both: option tiles_device).  Reference semantics of the form: the
// row's left fold over ascending columns, Data/Sparse/Common.hs:247-260 (kernel: sla_spmv_tiles.hip).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>  // rocPRIM's texture iterator calls the host memset without including it

#include <rocprim/rocprim.hpp>

#include "sla_internal.hpp"

namespace sla {

namespace {

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <class T> T *as() { return (T *)p; }
};

// longest (row, panel) segment: one thread per row walks its entries (rows are short in the matrices that take this form)
template <typename RP>
__global__ void __launch_bounds__(256) tile_maxseg_kernel(int64_t rows, const RP *__restrict__ rowptr, const int32_t *__restrict__ col, int shift,
                                                           unsigned *maxseg) {
    unsigned mx = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows; i += (int64_t)gridDim.x * 256) {
        int prev = -1;
        unsigned len = 0;
        for (RP k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            const int j = col[k] >> shift;
            len = j == prev ? len + 1 : 1;
            prev = j;
            mx = max(mx, len);
        }
    }
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off, 64));
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(maxseg, mx);
}

// key = ((slice << pbits | panel) << lbits | layer) << shift | column inside the panel ; value = the entry's position in the canonical arrays
template <typename RP>
__global__ void __launch_bounds__(256) tile_keys_kernel(int S, const int32_t *__restrict__ srow, const RP *__restrict__ rowptr,
                                                         const int32_t *__restrict__ col, int shift, int lbits, int pbits, uint64_t *key, uint32_t *idx) {
    for (int s = blockIdx.x; s < S; s += gridDim.x) {
        const uint64_t hi = (uint64_t)s << (pbits + lbits + shift);
        const uint32_t cmask = (1u << shift) - 1u;
        for (int i = srow[s] + (int)threadIdx.x; i < srow[s + 1]; i += 256) {
            int prev = -1;
            unsigned layer = 0;
            for (RP k = rowptr[i]; k < rowptr[i + 1]; ++k) {
                const int j = col[k] >> shift;
                layer = j == prev ? layer + 1 : 0;
                prev = j;
                key[k] = hi | ((uint64_t)j << (lbits + shift)) | ((uint64_t)layer << shift) | ((uint32_t)col[k] & cmask);
                idx[k] = (uint32_t)k;
            }
        }
    }
}

// the entries in tile order; layer boundaries INSIDE a tile are counted (each costs the kernel one more LDS pass)
template <typename RP>
__global__ void __launch_bounds__(256) tile_emit_kernel(int64_t nnz, const uint64_t *__restrict__ key, const uint32_t *__restrict__ idx,
                                                         const int32_t *__restrict__ srow, const RP *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                         const double *__restrict__ val, const int32_t *__restrict__ row_of_entry, int shift, int lbits,
                                                         int pbits, uint32_t *tidx, double *tval, unsigned long long *nbreaks) {
    const uint32_t cmask = (1u << shift) - 1u;
    unsigned brk = 0;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < nnz; o += (int64_t)gridDim.x * 256) {
        const uint64_t kk = key[o];
        const uint32_t k = idx[o];
        const int s = (int)(kk >> (pbits + lbits + shift));
        const int r = row_of_entry[k] - srow[s];
        uint32_t first = 0;                                          // first entry of a layer >= 1: same (slice, panel) as its predecessor, another layer
        if (o > 0) {
            const uint64_t kp = key[o - 1];
            first = (kp >> (lbits + shift)) == (kk >> (lbits + shift)) && (kp >> shift) != (kk >> shift);
        }
        tidx[o] = (first << 31) | ((uint32_t)r << shift) | ((uint32_t)col[k] & cmask);
        tval[o] = val[k];
        brk += first;
    }
    for (int off = 32; off > 0; off >>= 1) brk += (unsigned)__shfl_xor((int)brk, off, 64);
    if ((threadIdx.x & 63) == 0 && brk) atomicAdd(nbreaks, (unsigned long long)brk);
}

// row of every entry (the emit kernel needs it; one thread per row)
template <typename RP>
__global__ void __launch_bounds__(256) tile_rows_kernel(int64_t rows, const RP *__restrict__ rowptr, int32_t *row_of_entry) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows; i += (int64_t)gridDim.x * 256)
        for (RP k = rowptr[i]; k < rowptr[i + 1]; ++k) row_of_entry[k] = (int32_t)i;
}

// tloff[s * (P + 1) + j] = first sorted position with (slice, panel) >= (s, j), relative to the slice's first entry
template <typename RP>
__global__ void __launch_bounds__(256) tile_offsets_kernel(int S, int P, int64_t nnz, const uint64_t *__restrict__ key, const int32_t *__restrict__ srow,
                                                            const RP *__restrict__ rowptr, int shift, int lbits, int pbits, uint32_t *toff) {
    const int64_t total = (int64_t)S * (P + 1);
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int s = (int)(t / (P + 1)), j = (int)(t - (int64_t)s * (P + 1));
        const uint64_t want = ((uint64_t)s << (pbits + lbits + shift)) | ((uint64_t)j << (lbits + shift));
        int64_t lo = (int64_t)rowptr[srow[s]], hi = (int64_t)rowptr[srow[s + 1]];   // the slice owns the same entry range in both orders
        const int64_t base = lo;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (key[mid] < want) lo = mid + 1;
            else hi = mid;
        }
        toff[t] = (uint32_t)(lo - base);
    }
}

// ---- CU-wide slices (sla_spmv_ctiles.hip, round 5) ----------------------------------------------------------------------------
// key = (slice << pbits | panel) << shift | column inside the panel; STABLE sort: equal columns stay in input order = ascending rows.
// Inside a tile the sorted entries are dealt to the workgroup's four wavefronts in 64-entry groups round-robin; the layout is
// [slice][wavefront][panel], toff[(s * 4 + w) * (P + 1) + j] relative to the slice's first entry.
// The build runs over CHUNKS of whole slices [s_lo, s_hi) = entries [k_lo, k_hi): the scratch (sort keys and positions twice, the row of
// every entry, the tile starts) is sized for one chunk -- 9.2 GB for config 3a in one piece made every third lowering of a process wait
// 1.5 s in hipMalloc (LABNOTES R5).  key / idx / row_of_entry / bs are chunk-local arrays.
// Row-owning variant (round 6, `wb` = 2): the wavefront that owns the entry's row -- (row - slice's first row) & 3 -- sits between the panel and the
// column in the key, so that a tile's run splits into four column-sorted sub-runs, one per wavefront (row_of_entry is filled first then).
template <typename RP>
__global__ void __launch_bounds__(256) ctile_keys_kernel(int s_lo, int s_hi, const int32_t *__restrict__ srow, const RP *__restrict__ rowptr,
                                                          const int32_t *__restrict__ col, int shift, int pbits, int64_t k_lo, uint64_t *key, uint32_t *idx,
                                                          int wb, const int32_t *__restrict__ row_of_entry) {
    for (int s = s_lo + (int)blockIdx.x; s < s_hi; s += gridDim.x) {
        const uint64_t hi = (uint64_t)s << (pbits + wb + shift);
        const uint32_t cmask = (1u << shift) - 1u;
        const int32_t r0 = srow[s];
        const RP k0 = rowptr[r0], k1 = rowptr[srow[s + 1]];
        for (RP k = k0 + (RP)threadIdx.x; k < k1; k += 256) {
            const uint64_t pw = wb ? (((uint64_t)(col[k] >> shift) << wb) | (uint64_t)((row_of_entry[(int64_t)k - k_lo] - r0) & 3)) : (uint64_t)(col[k] >> shift);
            key[(int64_t)k - k_lo] = hi | (pw << shift) | ((uint32_t)col[k] & cmask);
            idx[(int64_t)k - k_lo] = (uint32_t)((int64_t)k - k_lo);   // chunk-local (round 6: the matrix may hold more than 2^32 entries, a chunk never does)
        }
    }
}
template <typename RP>
__global__ void __launch_bounds__(256) ctile_rows_kernel(int64_t r_lo, int64_t r_hi, const RP *__restrict__ rowptr, int64_t k_lo, int32_t *row_of_entry) {
    for (int64_t i = r_lo + (int64_t)blockIdx.x * 256 + threadIdx.x; i < r_hi; i += (int64_t)gridDim.x * 256)
        for (RP k = rowptr[i]; k < rowptr[i + 1]; ++k) row_of_entry[(int64_t)k - k_lo] = (int32_t)i;
}
// bs[(s - s_lo) * (P + 1) + j] = first sorted position with (slice, panel) >= (s, j), relative to the slice's first entry
template <typename RP>
__global__ void __launch_bounds__(256) ctile_bounds_kernel(int s_lo, int s_hi, int P, const uint64_t *__restrict__ key, const int32_t *__restrict__ srow,
                                                            const RP *__restrict__ rowptr, int shift, int pbits, int64_t k_lo, uint32_t *bs) {
    const int64_t total = (int64_t)(s_hi - s_lo) * (P + 1);
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int sl = (int)(t / (P + 1)), j = (int)(t - (int64_t)sl * (P + 1)), s = s_lo + sl;
        const uint64_t want = ((uint64_t)s << (pbits + shift)) | ((uint64_t)j << shift);
        int64_t lo = (int64_t)rowptr[srow[s]], hi = (int64_t)rowptr[srow[s + 1]];   // the slice owns the same entry range in both orders
        const int64_t base = lo;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (key[mid - k_lo] < want) lo = mid + 1;
            else hi = mid;
        }
        bs[t] = (uint32_t)(lo - base);
    }
}

// entries of wavefront w's share of tile (s, j) -> toff[(s * 4 + w) * (P + 1) + j] (counts; scanned below)
__global__ void __launch_bounds__(256) ctile_counts_kernel(int s_lo, int s_hi, int P, const uint32_t *__restrict__ bs, uint32_t *toff) {
    const int64_t total = (int64_t)(s_hi - s_lo) * P;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int sl = (int)(t / P), j = (int)(t - (int64_t)sl * P), s = s_lo + sl;
        const uint32_t *b = bs + (size_t)sl * (P + 1) + j;
        const uint32_t n0 = b[1] - b[0], G = (n0 + 63) >> 6, tail = n0 & 63;
        for (uint32_t w = 0; w < 4; ++w) {
            uint32_t ng = G > w ? (G - w + 3) >> 2 : 0, na = ng * 64;
            if (tail && G > 0 && ((G - 1) & 3) == w) na -= 64 - tail;
            toff[((size_t)s * 4 + w) * (size_t)(P + 1) + (size_t)j] = na;
        }
    }
}
// exclusive scan of every slice's 4 x (P + 1) counts in [wavefront][panel] order (the last slot of a wavefront's row = the next one's start)
__global__ void __launch_bounds__(64) ctile_scan_kernel(int s_lo, int s_hi, int P, uint32_t *toff) {
    const int s = s_lo + blockIdx.x * 64 + threadIdx.x;
    if (s >= s_hi) return;
    uint32_t *o = toff + (size_t)s * 4 * (size_t)(P + 1);
    uint32_t run = 0;
    for (int w = 0; w < 4; ++w) {
        for (int j = 0; j < P; ++j) { const uint32_t c = o[(size_t)w * (P + 1) + j]; o[(size_t)w * (P + 1) + j] = run; run += c; }
        o[(size_t)w * (P + 1) + P] = run;
    }
}

template <typename RP>
__global__ void __launch_bounds__(256) ctile_emit_kernel(int64_t k_lo, int64_t n, int s_lo, int P, const uint64_t *__restrict__ key, const uint32_t *__restrict__ idx,
                                                          const int32_t *__restrict__ srow, const RP *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                          const double *__restrict__ val, const int32_t *__restrict__ row_of_entry,
                                                          const uint32_t *__restrict__ bs, const uint32_t *__restrict__ toff, int shift, int pbits,
                                                          uint32_t *tidx, double *tval) {
    const uint32_t cmask = (1u << shift) - 1u;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (int64_t)gridDim.x * 256) {
        const uint64_t kk = key[o];
        const int64_t k = k_lo + (int64_t)idx[o];
        const int s = (int)(kk >> (pbits + shift));
        const int j = (int)((kk >> shift) & (((uint64_t)1 << pbits) - 1));
        const int64_t base = (int64_t)rowptr[srow[s]];
        const uint32_t t = (uint32_t)(o + k_lo - base) - bs[(size_t)(s - s_lo) * (P + 1) + j], g = t >> 6, w = g & 3;
        const int64_t dst = base + toff[((size_t)s * 4 + w) * (size_t)(P + 1) + (size_t)j] + ((g >> 2) << 6) + (t & 63);
        tidx[dst] = ((uint32_t)(row_of_entry[k - k_lo] - srow[s]) << shift) | ((uint32_t)col[k] & cmask);
        tval[dst] = val[k];
    }
}

// ---- row-owning layout: bounds per (slice, panel, wavefront), counts, emit --------------------------------------------------------------
// bs4[((s - s_lo) * P + j) * 4 + w] = first sorted position with (slice, panel, wavefront) >= (s, j, w), relative to the slice's first entry
// (one more slot per slice: its entry count)
template <typename RP>
__global__ void __launch_bounds__(256) ctile_bounds4_kernel(int s_lo, int s_hi, int P, const uint64_t *__restrict__ key, const int32_t *__restrict__ srow,
                                                             const RP *__restrict__ rowptr, int shift, int pbits, int64_t k_lo, uint32_t *bs4) {
    const int64_t per = (int64_t)P * 4 + 1, total = (int64_t)(s_hi - s_lo) * per;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int sl = (int)(t / per), q = (int)(t - (int64_t)sl * per), s = s_lo + sl;   // q = panel * 4 + wavefront (P * 4: the end)
        const uint64_t want = ((uint64_t)s << (pbits + 2 + shift)) | ((uint64_t)q << shift);
        int64_t lo = (int64_t)rowptr[srow[s]], hi = (int64_t)rowptr[srow[s + 1]];
        const int64_t base = lo;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (key[mid - k_lo] < want) lo = mid + 1;
            else hi = mid;
        }
        bs4[t] = (uint32_t)(lo - base);
    }
}
__global__ void __launch_bounds__(256) ctile_counts4_kernel(int s_lo, int s_hi, int P, const uint32_t *__restrict__ bs4, uint32_t *toff) {
    const int64_t total = (int64_t)(s_hi - s_lo) * P * 4;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int sl = (int)(t / ((int64_t)P * 4)), q = (int)(t - (int64_t)sl * P * 4), j = q >> 2, w = q & 3, s = s_lo + sl;
        const uint32_t *b = bs4 + (size_t)sl * ((size_t)P * 4 + 1) + q;
        toff[((size_t)s * 4 + w) * (size_t)(P + 1) + (size_t)j] = b[1] - b[0];
    }
}
template <typename RP>
__global__ void __launch_bounds__(256) ctile_emit4_kernel(int64_t k_lo, int64_t n, int s_lo, int P, const uint64_t *__restrict__ key, const uint32_t *__restrict__ idx,
                                                           const int32_t *__restrict__ srow, const RP *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                           const double *__restrict__ val, const int32_t *__restrict__ row_of_entry,
                                                           const uint32_t *__restrict__ bs4, const uint32_t *__restrict__ toff, int shift, int pbits,
                                                           uint32_t *tidx, double *tval) {
    const uint32_t cmask = (1u << shift) - 1u;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (int64_t)gridDim.x * 256) {
        const uint64_t kk = key[o];
        const int64_t k = k_lo + (int64_t)idx[o];
        const int s = (int)(kk >> (pbits + 2 + shift));
        const int q = (int)((kk >> shift) & (((uint64_t)1 << (pbits + 2)) - 1)), j = q >> 2, w = q & 3;
        const int64_t base = (int64_t)rowptr[srow[s]];
        const uint32_t t = (uint32_t)(o + k_lo - base) - bs4[(size_t)(s - s_lo) * ((size_t)P * 4 + 1) + (size_t)q];
        const int64_t dst = base + toff[((size_t)s * 4 + w) * (size_t)(P + 1) + (size_t)j] + t;
        tidx[dst] = ((uint32_t)(row_of_entry[k - k_lo] - srow[s]) << shift) | ((uint32_t)col[k] & cmask);
        tval[dst] = val[k];
    }
}

int bits_for(uint64_t v) {
    int b = 1;
    while (b < 63 && (v >> b)) ++b;
    return b;
}

}  // namespace

// Builds d_tlidx / d_tlval / d_tloff of A from its canonical device arrays.  srow: the slice row starts (host, S + 1).  Outputs the
// longest segment and the number of layer boundaries.  Returns SLA_OK with *done = false when this path cannot take the matrix
// (the host builder does then).
int build_tiles_device(sla_csr *A, const std::vector<int32_t> &srow, int shift, int64_t P, int64_t *maxseg_out, int64_t *nbreaks_out, bool *done) {
    *done = false;
    sla_ctx *c = A->ctx;
    const int64_t nnz = A->nnz, rows = A->rows, S = (int64_t)srow.size() - 1;
    if (!A->d_col || !A->d_val || !A->d_rowptr || nnz <= 0 || nnz >= ((int64_t)1 << 31) || S <= 0) return SLA_OK;
    hipStream_t st = stream_of(c);
    DevBuf d_srow, d_stat, d_key, d_key2, d_idx, d_idx2, d_rows, d_tmp;
    auto launch_ok = [&]() { return hipGetLastError() == hipSuccess; };
    hipError_t e = d_srow.alloc(sizeof(int32_t) * srow.size());
    if (e == hipSuccess) e = hipMemcpyAsync(d_srow.p, srow.data(), sizeof(int32_t) * srow.size(), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = d_stat.alloc(16);
    if (e == hipSuccess) e = hipMemsetAsync(d_stat.p, 0, 16, st);
    if (e != hipSuccess) return SLA_OK;   // (no device memory for the scratch: the host path may still work)
    const int grid = 4096;
    unsigned long long h_stat[2] = {0, 0};
    if (A->rp64) hipLaunchKernelGGL((tile_maxseg_kernel<int64_t>), dim3(grid), dim3(256), 0, st, rows, (const int64_t *)A->d_rowptr, A->d_col, shift, d_stat.as<unsigned>());
    else hipLaunchKernelGGL((tile_maxseg_kernel<int32_t>), dim3(grid), dim3(256), 0, st, rows, (const int32_t *)A->d_rowptr, A->d_col, shift, d_stat.as<unsigned>());
    if (!launch_ok()) return SLA_OK;
    SLA_HIP_TRY(hipMemcpyAsync(h_stat, d_stat.p, 8, hipMemcpyDeviceToHost, st));
    SLA_HIP_TRY(hipStreamSynchronize(st));
    const int64_t maxseg = (int64_t)(unsigned)h_stat[0];
    const int lbits = bits_for((uint64_t)std::max<int64_t>(maxseg, 1)), pbits = bits_for((uint64_t)P), sbits = bits_for((uint64_t)S);
    if (lbits + pbits + sbits + shift > 62) return SLA_OK;
    // scratch: keys / positions twice (radix sort ping-pong), the row of every entry, rocPRIM's own
    e = d_key.alloc(8 * (size_t)nnz);
    if (e == hipSuccess) e = d_key2.alloc(8 * (size_t)nnz);
    if (e == hipSuccess) e = d_idx.alloc(4 * (size_t)nnz);
    if (e == hipSuccess) e = d_idx2.alloc(4 * (size_t)nnz);
    if (e == hipSuccess) e = d_rows.alloc(4 * (size_t)nnz);
    size_t tmp_bytes = 0;
    if (e == hipSuccess)
        e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_key.as<uint64_t>(), d_key2.as<uint64_t>(), d_idx.as<uint32_t>(), d_idx2.as<uint32_t>(),
                                      (size_t)nnz, 0, (unsigned)(lbits + pbits + sbits + shift), st);
    if (e == hipSuccess) e = d_tmp.alloc(tmp_bytes);
    if (e == hipSuccess && !A->d_tlidx) e = dev_malloc(c, (void **)&A->d_tlidx, sizeof(uint32_t) * (size_t)nnz + 64);
    if (e == hipSuccess && !A->d_tlval) e = dev_malloc(c, (void **)&A->d_tlval, sizeof(double) * (size_t)nnz + 64);
    if (e == hipSuccess && !A->d_tloff) e = dev_malloc(c, (void **)&A->d_tloff, sizeof(uint32_t) * (size_t)(S * (P + 1)) + 64);
    auto give_up = [&]() {   // out of device memory for the scratch or the copy: release what was taken, let the host path decide
        (void)hipGetLastError();
        if (A->d_tlidx) { (void)hipFree(A->d_tlidx); A->d_tlidx = nullptr; }
        if (A->d_tlval) { (void)hipFree(A->d_tlval); A->d_tlval = nullptr; }
        if (A->d_tloff) { (void)hipFree(A->d_tloff); A->d_tloff = nullptr; }
        return SLA_OK;
    };
    if (e != hipSuccess) return give_up();
#define SLA_RP_LAUNCH(KERNEL, GRID, ...)                                                                                                   \
    do {                                                                                                                                   \
        if (A->rp64) hipLaunchKernelGGL((KERNEL<int64_t>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__);                                      \
        else hipLaunchKernelGGL((KERNEL<int32_t>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__);                                              \
    } while (0)
    if (A->rp64) {
        hipLaunchKernelGGL((tile_keys_kernel<int64_t>), dim3((unsigned)std::min<int64_t>(S, 65535)), dim3(256), 0, st, (int)S, d_srow.as<int32_t>(), (const int64_t *)A->d_rowptr, A->d_col, shift, lbits, pbits, d_key.as<uint64_t>(), d_idx.as<uint32_t>());
        hipLaunchKernelGGL((tile_rows_kernel<int64_t>), dim3(grid), dim3(256), 0, st, rows, (const int64_t *)A->d_rowptr, d_rows.as<int32_t>());
    } else {
        hipLaunchKernelGGL((tile_keys_kernel<int32_t>), dim3((unsigned)std::min<int64_t>(S, 65535)), dim3(256), 0, st, (int)S, d_srow.as<int32_t>(), (const int32_t *)A->d_rowptr, A->d_col, shift, lbits, pbits, d_key.as<uint64_t>(), d_idx.as<uint32_t>());
        hipLaunchKernelGGL((tile_rows_kernel<int32_t>), dim3(grid), dim3(256), 0, st, rows, (const int32_t *)A->d_rowptr, d_rows.as<int32_t>());
    }
    if (!launch_ok()) return give_up();
    e = rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, d_key.as<uint64_t>(), d_key2.as<uint64_t>(), d_idx.as<uint32_t>(), d_idx2.as<uint32_t>(), (size_t)nnz, 0,
                                  (unsigned)(lbits + pbits + sbits + shift), st);
    if (e != hipSuccess) return give_up();
    if (A->rp64) {
        hipLaunchKernelGGL((tile_emit_kernel<int64_t>), dim3(grid), dim3(256), 0, st, nnz, d_key2.as<uint64_t>(), d_idx2.as<uint32_t>(), d_srow.as<int32_t>(), (const int64_t *)A->d_rowptr, A->d_col, A->d_val, d_rows.as<int32_t>(), shift, lbits, pbits, A->d_tlidx, A->d_tlval, d_stat.as<unsigned long long>() + 1);
        hipLaunchKernelGGL((tile_offsets_kernel<int64_t>), dim3(grid), dim3(256), 0, st, (int)S, (int)P, nnz, d_key2.as<uint64_t>(), d_srow.as<int32_t>(), (const int64_t *)A->d_rowptr, shift, lbits, pbits, A->d_tloff);
    } else {
        hipLaunchKernelGGL((tile_emit_kernel<int32_t>), dim3(grid), dim3(256), 0, st, nnz, d_key2.as<uint64_t>(), d_idx2.as<uint32_t>(), d_srow.as<int32_t>(), (const int32_t *)A->d_rowptr, A->d_col, A->d_val, d_rows.as<int32_t>(), shift, lbits, pbits, A->d_tlidx, A->d_tlval, d_stat.as<unsigned long long>() + 1);
        hipLaunchKernelGGL((tile_offsets_kernel<int32_t>), dim3(grid), dim3(256), 0, st, (int)S, (int)P, nnz, d_key2.as<uint64_t>(), d_srow.as<int32_t>(), (const int32_t *)A->d_rowptr, shift, lbits, pbits, A->d_tloff);
    }
#undef SLA_RP_LAUNCH
    if (!launch_ok()) return give_up();
    SLA_HIP_TRY(hipMemcpyAsync(h_stat, d_stat.p, 16, hipMemcpyDeviceToHost, st));
    SLA_HIP_TRY(hipStreamSynchronize(st));
    *maxseg_out = maxseg;
    *nbreaks_out = (int64_t)h_stat[1];
    *done = true;
    return SLA_OK;
}

// The CU-wide layout of sla_spmv_ctiles.hip from A's canonical device arrays (same contract as build_tiles_device).
int build_ctiles_device(sla_csr *A, const std::vector<int32_t> &srow, int shift, int64_t P, const int64_t *rowptr_host, bool *done) {
    *done = false;
    sla_ctx *c = A->ctx;
    const int64_t nnz = A->nnz, S = (int64_t)srow.size() - 1;
    // (no bound on nnz: the build runs over chunks of <= 2^26 entries with chunk-local positions, every global offset is 64-bit, and a slice's
    // own offsets fit 32 bits by the caller's check; round 5 refused >= 2^31 entries here and the 2.2e9-entry run of tools/big_nnz.py spent 17.4 of
    // its 18.5 s of lowering in the host builder)
    if (!A->d_col || !A->d_val || !A->d_rowptr || nnz <= 0 || S <= 0) return SLA_OK;
    hipStream_t st = stream_of(c);
    static const bool dbg = getenv("SLA_DEBUG_LOWER") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {   // SLA_DEBUG_LOWER: where the build's time goes (device work included: synchronises)
        if (!dbg) return;
        (void)hipStreamSynchronize(st);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[sla] lowering:     . [cu tiles] %-28s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    // chunks of whole slices holding <= ~2^26 entries each (one slice at least)
    int64_t chunk_target = (int64_t)1 << 26;
    if (const char *ev = getenv("SLA_TILE_BUILD_CHUNK")) chunk_target = std::max<int64_t>(1, atoll(ev));   // (test hook: many chunks on small matrices)
    std::vector<int32_t> cs{0};                       // chunk c = slices [cs[c], cs[c + 1])
    int64_t nmax = 0, smax = 0;
    for (int64_t s = 0; s < S;) {
        const int64_t k_lo = rowptr_host[srow[(size_t)s]];
        int64_t e = s + 1;
        while (e < S && rowptr_host[srow[(size_t)e + 1]] - k_lo <= chunk_target) ++e;
        nmax = std::max(nmax, rowptr_host[srow[(size_t)e]] - k_lo);
        smax = std::max(smax, e - s);
        cs.push_back((int32_t)e);
        s = e;
    }
    DevBuf d_srow, d_key, d_key2, d_idx, d_idx2, d_rows, d_tmp, d_bs;
    auto launch_ok = [&]() { return hipGetLastError() == hipSuccess; };
    const bool rowown = A->tl_rowown;   // (round 6: every row of a slice owned by one wavefront -- reproducible, ascending row sums)
    const int wb = rowown ? 2 : 0;
    const int pbits = bits_for((uint64_t)P), sbits = bits_for((uint64_t)S);
    const int keybits = sbits + pbits + wb + shift;
    if (keybits > 62) return SLA_OK;
    const size_t ntoff = (size_t)S * 4 * (size_t)(P + 1);
    hipError_t e = d_srow.alloc(sizeof(int32_t) * srow.size());
    if (e == hipSuccess) e = hipMemcpyAsync(d_srow.p, srow.data(), sizeof(int32_t) * srow.size(), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = d_key.alloc(8 * (size_t)nmax);
    if (e == hipSuccess) e = d_key2.alloc(8 * (size_t)nmax);
    if (e == hipSuccess) e = d_idx.alloc(4 * (size_t)nmax);
    if (e == hipSuccess) e = d_idx2.alloc(4 * (size_t)nmax);
    if (e == hipSuccess) e = d_rows.alloc(4 * (size_t)nmax);
    if (e == hipSuccess) e = d_bs.alloc(4 * ((size_t)smax * (rowown ? (size_t)P * 4 + 1 : (size_t)(P + 1)) + 8));
    size_t tmp_bytes = 0;
    if (e == hipSuccess)
        e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_key.as<uint64_t>(), d_key2.as<uint64_t>(), d_idx.as<uint32_t>(), d_idx2.as<uint32_t>(),
                                      (size_t)nmax, 0, (unsigned)keybits, st);
    if (e == hipSuccess) e = d_tmp.alloc(tmp_bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); return SLA_OK; }   // (no device memory for the scratch: the host path may still work)
    lap("scratch allocations");
    if (!A->d_tlidx) e = dev_malloc(c, (void **)&A->d_tlidx, sizeof(uint32_t) * (size_t)nnz + 64);
    if (e == hipSuccess && !A->d_tlval) e = dev_malloc(c, (void **)&A->d_tlval, sizeof(double) * (size_t)nnz + 64);
    if (e == hipSuccess && !A->d_tloff) e = dev_malloc(c, (void **)&A->d_tloff, sizeof(uint32_t) * ntoff + 64);
    auto give_up = [&]() {
        (void)hipGetLastError();
        if (A->d_tlidx) { (void)hipFree(A->d_tlidx); A->d_tlidx = nullptr; }
        if (A->d_tlval) { (void)hipFree(A->d_tlval); A->d_tlval = nullptr; }
        if (A->d_tloff) { (void)hipFree(A->d_tloff); A->d_tloff = nullptr; }
        return SLA_OK;
    };
    if (e != hipSuccess) return give_up();
    lap("tile array allocations");
    const int grid = 4096;
    for (size_t ch = 0; ch + 1 < cs.size(); ++ch) {
        const int s_lo = cs[ch], s_hi = cs[ch + 1];
        const int64_t r_lo = srow[(size_t)s_lo], r_hi = srow[(size_t)s_hi], k_lo = rowptr_host[r_lo], n = rowptr_host[r_hi] - k_lo;
        if (n <= 0) {   // slices without entries: their offset rows are all zero
            SLA_HIP_TRY(hipMemsetAsync(A->d_tloff + (size_t)s_lo * 4 * (size_t)(P + 1), 0, sizeof(uint32_t) * (size_t)(s_hi - s_lo) * 4 * (size_t)(P + 1), st));
            continue;
        }
        const unsigned gs = (unsigned)std::min<int64_t>(s_hi - s_lo, 65535);
        const unsigned gt = (unsigned)std::min<int64_t>(((int64_t)(s_hi - s_lo) * (P + 1) + 255) / 256, 65535);
        if (A->rp64) {   // (the rows first: the row-owning keys read them)
            hipLaunchKernelGGL((ctile_rows_kernel<int64_t>), dim3(grid), dim3(256), 0, st, r_lo, r_hi, (const int64_t *)A->d_rowptr, k_lo, d_rows.as<int32_t>());
            hipLaunchKernelGGL((ctile_keys_kernel<int64_t>), dim3(gs), dim3(256), 0, st, s_lo, s_hi, d_srow.as<int32_t>(), (const int64_t *)A->d_rowptr, A->d_col, shift, pbits, k_lo, d_key.as<uint64_t>(), d_idx.as<uint32_t>(), wb, d_rows.as<int32_t>());
        } else {
            hipLaunchKernelGGL((ctile_rows_kernel<int32_t>), dim3(grid), dim3(256), 0, st, r_lo, r_hi, (const int32_t *)A->d_rowptr, k_lo, d_rows.as<int32_t>());
            hipLaunchKernelGGL((ctile_keys_kernel<int32_t>), dim3(gs), dim3(256), 0, st, s_lo, s_hi, d_srow.as<int32_t>(), (const int32_t *)A->d_rowptr, A->d_col, shift, pbits, k_lo, d_key.as<uint64_t>(), d_idx.as<uint32_t>(), wb, d_rows.as<int32_t>());
        }
        if (!launch_ok()) return give_up();
        e = rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, d_key.as<uint64_t>(), d_key2.as<uint64_t>(), d_idx.as<uint32_t>(), d_idx2.as<uint32_t>(), (size_t)n, 0,
                                      (unsigned)keybits, st);
        if (e != hipSuccess) return give_up();
        if (rowown) {
            const unsigned gt4 = (unsigned)std::min<int64_t>(((int64_t)(s_hi - s_lo) * (P * 4 + 1) + 255) / 256, 65535);
            if (A->rp64) hipLaunchKernelGGL((ctile_bounds4_kernel<int64_t>), dim3(gt4), dim3(256), 0, st, s_lo, s_hi, (int)P, d_key2.as<uint64_t>(), d_srow.as<int32_t>(), (const int64_t *)A->d_rowptr, shift, pbits, k_lo, d_bs.as<uint32_t>());
            else hipLaunchKernelGGL((ctile_bounds4_kernel<int32_t>), dim3(gt4), dim3(256), 0, st, s_lo, s_hi, (int)P, d_key2.as<uint64_t>(), d_srow.as<int32_t>(), (const int32_t *)A->d_rowptr, shift, pbits, k_lo, d_bs.as<uint32_t>());
            hipLaunchKernelGGL(ctile_counts4_kernel, dim3(gt4), dim3(256), 0, st, s_lo, s_hi, (int)P, d_bs.as<uint32_t>(), A->d_tloff);
            hipLaunchKernelGGL(ctile_scan_kernel, dim3((unsigned)((s_hi - s_lo + 63) / 64)), dim3(64), 0, st, s_lo, s_hi, (int)P, A->d_tloff);
            if (A->rp64)
                hipLaunchKernelGGL((ctile_emit4_kernel<int64_t>), dim3(grid), dim3(256), 0, st, k_lo, n, s_lo, (int)P, d_key2.as<uint64_t>(), d_idx2.as<uint32_t>(), d_srow.as<int32_t>(), (const int64_t *)A->d_rowptr, A->d_col, A->d_val, d_rows.as<int32_t>(), d_bs.as<uint32_t>(), A->d_tloff, shift, pbits, A->d_tlidx, A->d_tlval);
            else
                hipLaunchKernelGGL((ctile_emit4_kernel<int32_t>), dim3(grid), dim3(256), 0, st, k_lo, n, s_lo, (int)P, d_key2.as<uint64_t>(), d_idx2.as<uint32_t>(), d_srow.as<int32_t>(), (const int32_t *)A->d_rowptr, A->d_col, A->d_val, d_rows.as<int32_t>(), d_bs.as<uint32_t>(), A->d_tloff, shift, pbits, A->d_tlidx, A->d_tlval);
            if (!launch_ok()) return give_up();
            continue;
        }
        if (A->rp64) hipLaunchKernelGGL((ctile_bounds_kernel<int64_t>), dim3(gt), dim3(256), 0, st, s_lo, s_hi, (int)P, d_key2.as<uint64_t>(), d_srow.as<int32_t>(), (const int64_t *)A->d_rowptr, shift, pbits, k_lo, d_bs.as<uint32_t>());
        else hipLaunchKernelGGL((ctile_bounds_kernel<int32_t>), dim3(gt), dim3(256), 0, st, s_lo, s_hi, (int)P, d_key2.as<uint64_t>(), d_srow.as<int32_t>(), (const int32_t *)A->d_rowptr, shift, pbits, k_lo, d_bs.as<uint32_t>());
        hipLaunchKernelGGL(ctile_counts_kernel, dim3(gt), dim3(256), 0, st, s_lo, s_hi, (int)P, d_bs.as<uint32_t>(), A->d_tloff);
        hipLaunchKernelGGL(ctile_scan_kernel, dim3((unsigned)((s_hi - s_lo + 63) / 64)), dim3(64), 0, st, s_lo, s_hi, (int)P, A->d_tloff);
        if (A->rp64)
            hipLaunchKernelGGL((ctile_emit_kernel<int64_t>), dim3(grid), dim3(256), 0, st, k_lo, n, s_lo, (int)P, d_key2.as<uint64_t>(), d_idx2.as<uint32_t>(), d_srow.as<int32_t>(), (const int64_t *)A->d_rowptr, A->d_col, A->d_val, d_rows.as<int32_t>(), d_bs.as<uint32_t>(), A->d_tloff, shift, pbits, A->d_tlidx, A->d_tlval);
        else
            hipLaunchKernelGGL((ctile_emit_kernel<int32_t>), dim3(grid), dim3(256), 0, st, k_lo, n, s_lo, (int)P, d_key2.as<uint64_t>(), d_idx2.as<uint32_t>(), d_srow.as<int32_t>(), (const int32_t *)A->d_rowptr, A->d_col, A->d_val, d_rows.as<int32_t>(), d_bs.as<uint32_t>(), A->d_tloff, shift, pbits, A->d_tlidx, A->d_tlval);
        if (!launch_ok()) return give_up();
    }
    SLA_HIP_TRY(hipStreamSynchronize(st));
    lap("keys, sort, offsets, emit (all chunks)");
    *done = true;
    return SLA_OK;
}

}  // namespace sla
