// sweep_probe.cpp -- what bounds the fused K4+K5 sweep (bicg_k45_kernel: 5 vectors read, 3 written, x and p in place)?  (round 3)
// The library's own stream probe (sla_stream_probe, 5 reads + 3 writes on 8 distinct buffers) runs at 7.3 TB/s where the sweep
// reaches 6.5.  Variants of the bare loop, 10 077 696 elements, rotating over 3 vector sets (7 x 80 MB each: nothing stays cached):
//   distinct / in-place (x, p updated in place like the solver) x plain / non-temporal loads x plain / non-temporal stores of x, r
//   x one-deep loop / next iteration's loads issued in front of this iteration's stores
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/sweep_probe tools/sweep_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NTL> __device__ __forceinline__ d2 ld(const double *p, long i) {
    return NTL ? __builtin_nontemporal_load((const d2 *)p + i) : ((const d2 *)p)[i];
}
template <bool NTS> __device__ __forceinline__ void st(double *p, long i, d2 v) {
    if (NTS) __builtin_nontemporal_store(v, (d2 *)p + i); else ((d2 *)p)[i] = v;
}

// x' = (x + alpha p) + omega s ; r' = s - omega as ; p' = r' + beta (p - omega ap)
template <bool NTL, bool NTS, bool PIPE>
__global__ void __launch_bounds__(256) sweep(long n2, const double *s, const double *as, const double *ap, const double *xin, const double *pin,
                                             double *x, double *r, double *p, double alpha, double omega, double beta) {
    const long gs = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (!PIPE) {
        for (; i < n2; i += gs) {
            d2 sv = ld<NTL>(s, i), av = ld<NTL>(as, i), vv = ld<NTL>(ap, i), pv = ld<NTL>(pin, i), xv = ld<NTL>(xin, i);
            xv = (xv + alpha * pv) + omega * sv;
            st<NTS>(x, i, xv);
            const d2 rv = sv - omega * av;
            st<NTS>(r, i, rv);
            st<false>(p, i, rv + beta * (pv - omega * vv));
        }
    } else {
        if (i >= n2) return;
        d2 sv = ld<NTL>(s, i), av = ld<NTL>(as, i), vv = ld<NTL>(ap, i), pv = ld<NTL>(pin, i), xv = ld<NTL>(xin, i);
        for (; i < n2; i += gs) {
            const long j = i + gs < n2 ? i + gs : i;
            const d2 sn = ld<NTL>(s, j), an = ld<NTL>(as, j), vn = ld<NTL>(ap, j), pn = ld<NTL>(pin, j), xn = ld<NTL>(xin, j);
            xv = (xv + alpha * pv) + omega * sv;
            st<NTS>(x, i, xv);
            const d2 rv = sv - omega * av;
            st<NTS>(r, i, rv);
            st<false>(p, i, rv + beta * (pv - omega * vv));
            sv = sn; av = an; vv = vn; pv = pn; xv = xn;
        }
    }
}


// each lane takes W consecutive pairs (16 W bytes per stream): a wavefront touches 1 KB x W of every stream per iteration
template <int W, bool NTS>
__global__ void __launch_bounds__(256) sweep_wide(long n2, const double *s, const double *as, const double *ap, const double *xin, const double *pin,
                                                  double *x, double *r, double *p, double alpha, double omega, double beta) {
    const long gs = (long)gridDim.x * 256 * W;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * W; i + W <= n2; i += gs) {
        d2 sv[W], av[W], vv[W], pv[W], xv[W];
#pragma unroll
        for (int j = 0; j < W; ++j) { sv[j] = ld<true>(s, i + j); av[j] = ld<true>(as, i + j); vv[j] = ld<true>(ap, i + j); pv[j] = ld<true>(pin, i + j); xv[j] = ld<true>(xin, i + j); }
#pragma unroll
        for (int j = 0; j < W; ++j) {
            st<NTS>(x, i + j, (xv[j] + alpha * pv[j]) + omega * sv[j]);
            const d2 rv = sv[j] - omega * av[j];
            st<NTS>(r, i + j, rv);
            st<false>(p, i + j, rv + beta * (pv[j] - omega * vv[j]));
        }
    }
}
// each WORKGROUP takes a contiguous block of B iterations (B x 4 KB of every stream) instead of striding by the grid
template <int B, bool NTS>
__global__ void __launch_bounds__(256) sweep_blocked(long n2, const double *s, const double *as, const double *ap, const double *xin, const double *pin,
                                                     double *x, double *r, double *p, double alpha, double omega, double beta) {
    const long chunks = (n2 + 256 * B - 1) / (256 * B);
    for (long c = blockIdx.x; c < chunks; c += gridDim.x) {
#pragma unroll
        for (int j = 0; j < B; ++j) {
            const long i = (c * B + j) * 256 + threadIdx.x;
            if (i < n2) {
                const d2 sv = ld<true>(s, i), av = ld<true>(as, i), vv = ld<true>(ap, i), pv = ld<true>(pin, i), xv = ld<true>(xin, i);
                st<NTS>(x, i, (xv + alpha * pv) + omega * sv);
                const d2 rv = sv - omega * av;
                st<NTS>(r, i, rv);
                st<false>(p, i, rv + beta * (pv - omega * vv));
            }
        }
    }
}

int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 10077696;
    const int sets = 3, reps = 30;
    std::vector<std::vector<double *>> v(sets, std::vector<double *>(8));
    // SWEEP_SKEW=<bytes>: vector k of a set starts k * skew bytes into its allocation (do the eight streams of the sweep, all at the same
    // offset of 2 MiB-aligned allocations, meet in the same HBM channels?)
    const long skew = getenv("SWEEP_SKEW") ? atol(getenv("SWEEP_SKEW")) : 0;
    for (auto &set : v) { int k = 0; for (auto &p : set) { char *q; CK(hipMalloc(&q, n * 8 + 8 * skew + 256)); CK(hipMemset(q, 0, n * 8 + 8 * skew + 256)); p = (double *)(q + k * skew); ++k; } }
    if (skew) printf("# vector k of a set skewed by k * %ld bytes\n", skew);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, int grid, bool inplace, auto kern) {
        auto launch = [&](int k) {
            auto &b = v[k % sets];
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, n / 2, b[0], b[1], b[2], b[3], b[4], inplace ? b[3] : b[5], b[6], inplace ? b[4] : b[7], 0.3, 0.7, 0.1);
        };
        for (int k = 0; k < 3; ++k) launch(k);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int k = 0; k < reps; ++k) launch(k);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("%-52s grid %5d  %7.1f us  %5.2f TB/s\n", name, grid, us, 64.0 * n / us * 1e-6);
    };
    for (int grid : {2048, 1024, 4096}) {
        run("distinct  plain loads  plain stores", grid, false, sweep<false, false, false>);
        run("distinct  nt loads     plain stores", grid, false, sweep<true, false, false>);
        run("in place  nt loads     plain stores", grid, true, sweep<true, false, false>);
        run("in place  nt loads     nt stores x r", grid, true, sweep<true, true, false>);
        run("in place  nt loads     nt stores x r   pipelined", grid, true, sweep<true, true, true>);
        run("in place  nt loads     plain stores    pipelined", grid, true, sweep<true, false, true>);
        run("distinct  nt loads     plain stores    pipelined", grid, false, sweep<true, false, true>);
        run("in place  nt loads     nt stores x r   2 pairs per lane", grid, true, sweep_wide<2, true>);
        run("in place  nt loads     nt stores x r   4 pairs per lane", grid, true, sweep_wide<4, true>);
        run("in place  nt loads     nt stores x r   2 iterations per workgroup block", grid, true, sweep_blocked<2, true>);
        run("in place  nt loads     nt stores x r   4 iterations per workgroup block", grid, true, sweep_blocked<4, true>);
    }
    return 0;
}
