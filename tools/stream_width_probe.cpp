// stream_width_probe.cpp -- does the WIDTH of the per-lane loads bound the CSR-stream SpMV?  (round 3)
//
// PMC on spmv_stream_kernel (profiles/r03_pmc_stream_kernel.txt): the waves wait 85 % of their cycles, the L1s sit in
// TCP_PENDING_STALL (data pending from the L2) half of the time, no FIFO-full, 6 % issue -- memory-side bound at 7.8 B/clk/CU
// where the 16-byte-per-lane vector kernels reach 10.6.  The stream kernel loads 4 B (col) and 8 B (val) per lane, lane-strided:
// 256 / 512 bytes per wave-instruction.  This probe runs the kernel's skeleton (7 entries per row, 146-row blocks of 1022
// entries, products staged in LDS, one lane per row, y stored) with
//   W0: dword col + dwordx2 val, 4 + 4 lane-strided loads per lane        (what spmv_stream_kernel does)
//   W1: dwordx2 col + dwordx4 val, 2 + 2 loads per lane, 16-byte aligned  (512 B / 1 KiB per wave-instruction)
// each without and with the x gather, 8 workgroups per CU, persistent grid of 2048.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/stream_width_probe.cpp -o tools/stream_width_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int kRows = 146, kEnt = 1022;

template <int W, bool GATHER>
__global__ void __launch_bounds__(256, 8) k_probe(const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ x,
                                                  double *__restrict__ y, int nblk) {
    __shared__ double s_prod[2][1024];
    const int tid = threadIdx.x;
    int buf = 0;
    for (int b = blockIdx.x; b < nblk; b += gridDim.x) {
        const long k0 = (long)b * kEnt;
        double *prod = s_prod[buf];
        if constexpr (W == 0) {
            int c[4];
            double v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = min(tid + 256 * j, kEnt - 1);
                c[j] = __builtin_nontemporal_load(col + k0 + i);
                v[j] = __builtin_nontemporal_load(val + k0 + i);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = tid + 256 * j;
                const double xv = GATHER ? x[c[j]] : (double)c[j];
                if (i < kEnt) prod[i] = v[j] * xv;
            }
        } else {
            i32x2 c[2];
            f64x2 v[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int i = min(2 * tid + 512 * j, kEnt - 2);   // (k0 is even: 8-byte aligned col pairs, 16-byte aligned val pairs)
                c[j] = __builtin_nontemporal_load((const i32x2 *)(col + k0 + i));
                v[j] = __builtin_nontemporal_load((const f64x2 *)(val + k0 + i));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int i = 2 * tid + 512 * j;
                const double xa = GATHER ? x[c[j].x] : (double)c[j].x, xb = GATHER ? x[c[j].y] : (double)c[j].y;
                if (i < kEnt) *(f64x2 *)(prod + i) = f64x2{v[j].x * xa, v[j].y * xb};
            }
        }
        __syncthreads();
        if (tid < kRows) {
            double acc = 0.0;
            for (int k = tid * 7; k < tid * 7 + 7; ++k) acc += prod[k];
            y[(long)b * kRows + tid] = acc;
        }
        buf ^= 1;
    }
}


// Structure variants of the W1 skeleton (no gather): which part costs what?
//   M0 loads only (row sums in registers, nothing stored)   M1 + LDS stage + barrier + lane-per-row fold   M2 + y store (= W1 above)
//   M3 = M2 with the next block's loads issued before the barrier (one-deep prefetch, like spmv_stream_kernel)
template <int M, int OCC>
__global__ void __launch_bounds__(256, OCC) k_struct(const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ x,
                                                     double *__restrict__ y, int nblk) {
    __shared__ double s_prod[2][1024];
    const int tid = threadIdx.x;
    int buf = 0;
    double keep = 0.0;
    i32x2 c[2];
    f64x2 v[2];
    auto issue = [&](int b) {
        const long k0 = (long)b * kEnt;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = min(2 * tid + 512 * j, kEnt - 2);
            c[j] = __builtin_nontemporal_load((const i32x2 *)(col + k0 + i));
            v[j] = __builtin_nontemporal_load((const f64x2 *)(val + k0 + i));
        }
    };
    int b = blockIdx.x;
    if (M == 3 && b < nblk) issue(b);
    for (; b < nblk; b += gridDim.x) {
        if (M != 3) issue(b);
        double *prod = s_prod[buf];
        f64x2 p[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) p[j] = f64x2{v[j].x * (double)c[j].x, v[j].y * (double)c[j].y};
        if (M == 0) {
            keep += (p[0].x + p[0].y) + (p[1].x + p[1].y);
            continue;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = 2 * tid + 512 * j;
            if (i < kEnt) *(f64x2 *)(prod + i) = p[j];
        }
        if (M == 3 && b + (int)gridDim.x < nblk) issue(b + gridDim.x);
        __syncthreads();
        if (tid < kRows) {
            double acc = 0.0;
            for (int k = tid * 7; k < tid * 7 + 7; ++k) acc += prod[k];
            if (M >= 2) y[(long)b * kRows + tid] = acc;
            else keep += acc;
        }
        buf ^= 1;
    }
    if (keep == 123.456) y[tid] = keep;
}


// The y store: 8 bytes per lane from `rows` lanes of a block whose y segment starts at block * rows * 8 bytes.  With 146 rows the
// segments are 1168 bytes: every block's first and last 128-byte line is shared with a neighbour block (another workgroup, another
// time).  ST 0: as the kernel does.  ST 1: 16 bytes per lane (row pairs through LDS).  ST 2: non-temporal.  ST 3: no store at all but
// an equally sized LOAD of y (reads instead of writes).  `rows` 144 = whole lines per block.
template <int ST>
__global__ void __launch_bounds__(256, 8) k_store(const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ x,
                                                  double *__restrict__ y, int nblk, int rows) {
    __shared__ double s_prod[2][1024];
    __shared__ double s_y[256];
    const int tid = threadIdx.x;
    const int ent = rows * 7;
    int buf = 0;
    double keep = 0.0;
    // ST 4: like ST 0, but a workgroup takes 4 CONSECUTIVE blocks per turn (the partial lines between them are its own)
    const int CH = ST == 4 ? 4 : 1;
    for (int bb = blockIdx.x * CH; bb < nblk; bb += gridDim.x * CH)
    for (int b = bb; b < min(bb + CH, nblk); ++b) {
        const long k0 = (long)b * ent;
        i32x2 c[2];
        f64x2 v[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = min(2 * tid + 512 * j, ent - 2);
            c[j] = __builtin_nontemporal_load((const i32x2 *)(col + k0 + i));
            v[j] = __builtin_nontemporal_load((const f64x2 *)(val + k0 + i));
        }
        double *prod = s_prod[buf];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = 2 * tid + 512 * j;
            if (i < ent) *(f64x2 *)(prod + i) = f64x2{v[j].x * (double)c[j].x, v[j].y * (double)c[j].y};
        }
        __syncthreads();
        double acc = 0.0;
        if (tid < rows) {
            for (int k = tid * 7; k < tid * 7 + 7; ++k) acc += prod[k];
            if (ST == 0 || ST == 4) y[(long)b * rows + tid] = acc;
            if (ST == 2) __builtin_nontemporal_store(acc, y + (long)b * rows + tid);
            if (ST == 3) keep += acc + y[(long)b * rows + tid];
            if (ST == 1) s_y[tid] = acc;
        }
        if (ST == 1) {
            __syncthreads();
            if (2 * tid < rows) *(f64x2 *)(y + (long)b * rows + 2 * tid) = f64x2{s_y[2 * tid], s_y[2 * tid + 1]};
        }
        buf ^= 1;
    }
    if (keep == 123.456) y[tid] = keep;
}


// Wave-private row blocks (round 3, late): every wavefront owns blocks of 64 rows = 448 entries, stages their products in its own
// 4 KiB of LDS and folds one row per lane -- no workgroup barrier, each wavefront its own pipeline (the next block's pairs are
// issued in front of the fold when PF).  Same loads as W1 (dwordx2 col + dwordx4 val per lane).
template <bool GATHER, bool PF>
__global__ void __launch_bounds__(256, 8) k_wave(const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ x,
                                                 double *__restrict__ y, int nblk146) {
    __shared__ double s_prod[4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nrows = (long)nblk146 * kRows, nb = nrows / 64;            // whole 64-row blocks (the tail rows are left out: a probe)
    double *prod = s_prod[wave];
    const long W = (long)gridDim.x * 4;
    i32x2 c[4];
    f64x2 v[4];
    auto issue = [&](long b) {
        const long k0 = b * 448;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = min(2 * lane + 128 * j, 446);
            c[j] = __builtin_nontemporal_load((const i32x2 *)(col + k0 + i));
            v[j] = __builtin_nontemporal_load((const f64x2 *)(val + k0 + i));
        }
    };
    long b = (long)blockIdx.x * 4 + wave;
    if (b < nb) issue(b);
    for (; b < nb; b += W) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 2 * lane + 128 * j;
            const double xa = GATHER ? x[c[j].x] : (double)c[j].x, xb = GATHER ? x[c[j].y] : (double)c[j].y;
            if (i < 448) *(f64x2 *)(prod + i) = f64x2{v[j].x * xa, v[j].y * xb};
        }
        if (PF && b + W < nb) issue(b + W);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wavefront's LDS stores are in place
        double acc = 0.0;
        for (int k = lane * 7; k < lane * 7 + 7; ++k) acc += prod[k];
        y[b * 64 + lane] = acc;
        __builtin_amdgcn_wave_barrier();
        if (!PF && b + W < nb) issue(b + W);
    }
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 216;
    const long n0 = (long)N * N * N, nblk = n0 / kRows, n = nblk * kRows, nnz = n * 7;
    std::vector<int> hc((size_t)nnz);
    std::vector<double> hv((size_t)nnz);
    const long off[7] = {-(long)N * N, -N, -1, 0, 1, N, (long)N * N};
    for (long r = 0; r < n; ++r)
        for (int t = 0; t < 7; ++t) {
            hc[(size_t)(7 * r + t)] = (int)std::min<long>(std::max<long>(r + off[t], 0), n - 1);
            hv[(size_t)(7 * r + t)] = t == 3 ? 6.0 : -1.0;
        }
    int *col;
    double *val, *x, *y;
    hipMalloc(&col, 4 * (size_t)nnz + 64);
    hipMalloc(&val, 8 * (size_t)nnz + 64);
    hipMalloc(&x, 8 * (size_t)n);
    hipMalloc(&y, 8 * (size_t)n);
    hipMemcpy(col, hc.data(), 4 * (size_t)nnz, hipMemcpyHostToDevice);
    hipMemcpy(val, hv.data(), 8 * (size_t)nnz, hipMemcpyHostToDevice);
    std::vector<double> hx((size_t)n, 1.0);
    hipMemcpy(x, hx.data(), 8 * (size_t)n, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](const char *name, auto kern, double bytes) {
        std::vector<float> t;
        for (int i = 0; i < 13; ++i) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, col, val, x, y, (int)nblk);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (i >= 3) t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("%-58s %8.1f us  %7.0f GB/s\n", name, t[t.size() / 2] * 1e3, bytes / t[t.size() / 2] / 1e6);
    };
    printf("rows %ld, entries %ld, %ld blocks of %d rows\n", n, nnz, nblk, kRows);
    const double bs = 12.0 * nnz + 8.0 * n, bg = bs + 8.0 * n;
    run("W0 dword col + dwordx2 val, no gather", k_probe<0, false>, bs);
    run("W1 dwordx2 col + dwordx4 val, no gather", k_probe<1, false>, bs);
    run("W0 dword col + dwordx2 val, x gathered", k_probe<0, true>, bg);
    run("W1 dwordx2 col + dwordx4 val, x gathered", k_probe<1, true>, bg);
    run("wave-private 64-row blocks, no gather", k_wave<false, false>, bs);
    run("wave-private 64-row blocks, no gather, next pairs ahead", k_wave<false, true>, bs);
    run("wave-private 64-row blocks, x gathered", k_wave<true, false>, bg);
    run("wave-private 64-row blocks, x gathered, next pairs ahead", k_wave<true, true>, bg);
    if (getenv("PROBE_WAVE")) return 0;
    auto run2 = [&](const char *name, auto kern, int grid, double bytes) {
        std::vector<float> t;
        for (int i = 0; i < 13; ++i) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, col, val, x, y, (int)nblk);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (i >= 3) t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("%-58s %8.1f us  %7.0f GB/s\n", name, t[t.size() / 2] * 1e3, bytes / t[t.size() / 2] / 1e6);
    };
    const double bl = 12.0 * nnz;
    run2("M0 loads only, 8 WG/CU", k_struct<0, 8>, 2048, bl);
    run2("M1 + LDS + barrier + fold, 8 WG/CU", k_struct<1, 8>, 2048, bl);
    run2("M2 + y store, 8 WG/CU", k_struct<2, 8>, 2048, bs);
    run2("M3 + one-deep prefetch, 8 WG/CU", k_struct<3, 8>, 2048, bs);
    run2("M0 loads only, 4 WG/CU", k_struct<0, 4>, 1024, bl);
    run2("M2 full, 4 WG/CU", k_struct<2, 4>, 1024, bs);
    run2("M3 prefetch, 4 WG/CU", k_struct<3, 4>, 1024, bs);
    run2("M2 full, 8 WG/CU, grid 4096", k_struct<2, 8>, 4096, bs);
    run2("M2 full, 8 WG/CU, grid 16384", k_struct<2, 8>, 16384, bs);
    auto run3 = [&](const char *name, auto kern, int rows) {
        const int nb = (int)(n / rows);
        std::vector<float> t;
        for (int i = 0; i < 13; ++i) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, col, val, x, y, nb, rows);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (i >= 3) t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("%-50s rows %3d %8.1f us  %7.0f GB/s\n", name, rows, t[t.size() / 2] * 1e3, (84.0 + 8.0) * nb * rows / t[t.size() / 2] / 1e6);
    };
    for (int rows : {146, 144, 128}) {
        run3("ST0 8 B per lane", k_store<0>, rows);
        run3("ST1 16 B per lane via LDS", k_store<1>, rows);
        run3("ST2 8 B per lane, non-temporal", k_store<2>, rows);
        run3("ST3 y LOADED instead of stored", k_store<3>, rows);
        run3("ST4 8 B per lane, 4 consecutive blocks per turn", k_store<4>, rows);
    }
    double s = 0;
    std::vector<double> hy((size_t)n);
    hipMemcpy(hy.data(), y, 8 * (size_t)n, hipMemcpyDeviceToHost);
    for (long i = 0; i < n; i += 9973) s += hy[(size_t)i];
    printf("checksum %.1f\n", s);
    return 0;
}
