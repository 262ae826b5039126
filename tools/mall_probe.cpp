// mall_probe.cpp -- does the 256 MB memory-side cache keep what one streaming kernel wrote for the next kernel, and in which
// sweep direction?  (MI355X, round 3)
// Producer P: v[i] = a[i] + b[i] over 80 MB vectors, forward (3 x 80 MB of traffic: by its end an LRU-like cache holds the
// tail of each stream).  Consumer C: s[i] = r[i] - alpha v[i], forward or BACKWARD, r read non-temporally or not.
// Rotating over 3 vector sets so that nothing survives from the previous round.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mall_probe tools/mall_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NT_IN, bool NT_OUT>
__global__ void __launch_bounds__(256) producer(long n2, const double *a, const double *b, double *v) {
    const long gs = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += gs) {
        const d2 x = NT_IN ? __builtin_nontemporal_load((const d2 *)a + i) : ((const d2 *)a)[i];
        const d2 y = NT_IN ? __builtin_nontemporal_load((const d2 *)b + i) : ((const d2 *)b)[i];
        if (NT_OUT) __builtin_nontemporal_store(x + y, (d2 *)v + i); else ((d2 *)v)[i] = x + y;
    }
}
// producer with more allocating traffic than the cache holds: 4 vectors read + 1 written, all cacheable (400 MB)
__global__ void __launch_bounds__(256) producer5(long n2, const double *a, const double *b, const double *c, const double *d, double *v) {
    const long gs = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += gs)
        ((d2 *)v)[i] = (((const d2 *)a)[i] + ((const d2 *)b)[i]) + (((const d2 *)c)[i] + ((const d2 *)d)[i]);
}
template <bool BWD, bool NT_R, bool NT_V, bool NT_S>
__global__ void __launch_bounds__(256) consumer(long n2, const double *r, const double *v, double *s, double alpha) {
    const long gs = (long)gridDim.x * 256;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n2; k += gs) {
        const long i = BWD ? n2 - 1 - k : k;
        const d2 x = NT_R ? __builtin_nontemporal_load((const d2 *)r + i) : ((const d2 *)r)[i];
        const d2 y = NT_V ? __builtin_nontemporal_load((const d2 *)v + i) : ((const d2 *)v)[i];
        if (NT_S) __builtin_nontemporal_store(x - alpha * y, (d2 *)s + i); else ((d2 *)s)[i] = x - alpha * y;
    }
}

int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 10077696;
    const int sets = 3, reps = 12;
    std::vector<std::vector<double *>> b(sets, std::vector<double *>(5));
    for (auto &set : b) for (auto &p : set) { CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); }
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    auto run = [&](const char *name, auto prod, auto cons) {
        double tp = 0, tc = 0;
        for (int k = 0; k < reps + 2; ++k) {
            auto &q = b[k % sets];
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(prod, dim3(1024), dim3(256), 0, 0, n / 2, q[0], q[1], q[2]);
            CK(hipEventRecord(e1));
            hipLaunchKernelGGL(cons, dim3(1024), dim3(256), 0, 0, n / 2, q[3], q[2], q[4], 0.5);
            CK(hipEventRecord(e2));
            CK(hipEventSynchronize(e2));
            float a, c; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&c, e1, e2));
            if (k >= 2) { tp += a; tc += c; }
        }
        printf("%-78s producer %6.1f us   consumer %6.1f us (%.2f TB/s of its 3 streams)\n", name, tp * 1e3 / reps, tc * 1e3 / reps, 24.0 * n / (tc * 1e3 / reps) * 1e-6);
    };
    auto run5 = [&](const char *name, auto cons) {
        double tp = 0, tc = 0;
        for (int k = 0; k < reps + 2; ++k) {
            auto &q = b[k % sets]; auto &q2 = b[(k + 1) % sets];
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(producer5, dim3(1024), dim3(256), 0, 0, n / 2, q[0], q[1], q2[0], q2[1], q[2]);
            CK(hipEventRecord(e1));
            hipLaunchKernelGGL(cons, dim3(1024), dim3(256), 0, 0, n / 2, q[3], q[2], q[4], 0.5);
            CK(hipEventRecord(e2));
            CK(hipEventSynchronize(e2));
            float a, c; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&c, e1, e2));
            if (k >= 2) { tp += a; tc += c; }
        }
        printf("%-78s producer %6.1f us   consumer %6.1f us (%.2f TB/s of its 3 streams)\n", name, tp * 1e3 / reps, tc * 1e3 / reps, 24.0 * n / (tc * 1e3 / reps) * 1e-6);
    };
    run5("P5 (400 MB cacheable)   C forward   r nt    v plain  s nt", consumer<false, true, false, true>);
    run5("P5 (400 MB cacheable)   C BACKWARD  r nt    v plain  s nt", consumer<true, true, false, true>);
    run5("P5 (400 MB cacheable)   C forward   r nt    v plain  s plain", consumer<false, true, false, false>);
    run5("P5 (400 MB cacheable)   C BACKWARD  r nt    v plain  s plain", consumer<true, true, false, false>);
    run("P plain in/out          C forward   r plain v plain", producer<false, false>, consumer<false, false, false, false>);
    run("P plain in/out          C BACKWARD  r plain v plain", producer<false, false>, consumer<true, false, false, false>);
    run("P nt in, plain out      C forward   r nt    v plain", producer<true, false>, consumer<false, true, false, false>);
    run("P nt in, plain out      C BACKWARD  r nt    v plain", producer<true, false>, consumer<true, true, false, false>);
    run("P nt in, plain out      C BACKWARD  r nt    v plain  s nt", producer<true, false>, consumer<true, true, false, true>);
    run("P nt in, nt out         C BACKWARD  r nt    v nt     s nt   (nothing cached)", producer<true, true>, consumer<true, true, true, true>);
    run("P nt in, nt out         C forward   r nt    v nt     s nt   (nothing cached)", producer<true, true>, consumer<false, true, true, true>);
    run("P plain in/out          C BACKWARD  r nt    v plain", producer<false, false>, consumer<true, true, false, false>);
    run("P nt in, plain out      C forward   r nt    v NT     s plain  (cached v read non-temporally)", producer<true, false>, consumer<false, true, true, false>);
    run("P nt in, plain out      C forward   r nt    v plain  s plain", producer<true, false>, consumer<false, true, false, false>);
    run("P plain in, plain out   C forward   r nt    v NT     s plain", producer<false, false>, consumer<false, true, true, false>);
    return 0;
}
