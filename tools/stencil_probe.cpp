// stencil_probe.cpp -- would an LDS-staged x window pay for the wave-sliced stencil SpMV?  (MI355X, round 2)
//
// spmv_wdia_kernel gathers every (offset, value) record of a 128-row slice straight from global memory: 7 wave-wide
// 16-byte loads per slice for the 7-pt Laplacian, all of them touching nearly the same lines.  Cache-resident it takes
// 38 us at 10 M rows where the HBM time of x + y is 26 us.  This probe times the bare access pattern both ways on a
// 216^3 grid (constant coefficients, no boundary handling: x is padded by one plane on both sides):
//   direct   two rows per lane, 7 global 16-byte gathers, fold, 16-byte store                      (the product's pattern)
//   lds      a workgroup stages the three x windows of its 512-row step in LDS (in-plane window of 512 + 2 * 216
//            elements, the planes behind / ahead: 4 global 16-byte loads per lane instead of 7, next step's loads in
//            flight during the fold), the 7 operands come from LDS
//   lds+axpy K2 folded into K3's staging (windows of r - alpha v computed on the fly, own rows of s written) against the axpy kernel
//            followed by the staged SpMV: slower (last output line)
//   hipcc --offload-arch=gfx950 -O3 -o tools/stencil_probe tools/stencil_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <utility>
#include <algorithm>
#include <cmath>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ d2 ld2(const double *p) {   // 16 bytes at 8-byte alignment
    d2 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

__device__ __forceinline__ d2 fold7(d2 a, d2 b, d2 c, d2 d, d2 e, d2 f, d2 g) {
#pragma clang fp contract(off)
    d2 s = -1.0 * a;
    s = s + -1.0 * b;
    s = s + -1.0 * c;
    s = s + 6.0 * d;
    s = s + -1.0 * e;
    s = s + -1.0 * f;
    s = s + -1.0 * g;
    return s;
}

// step i of this workgroup: plain = grid-stride over the row order; sched = XCD x (= blockIdx.x % 8) owns T / 8 in-plane tiles
// and walks them plane by plane (the product's sched[] order): the plane behind / ahead of a step is in the XCD's own L2
// ord: SCHED 2 = the product's visiting order (sla_api.cpp: steps keyed by position inside the plane / tile); SCHED 3 = plane
// sweep: XCD x owns the steps whose position inside the plane falls into its eighth, 8 lists of `per` entries (padded with -1)
template <int SCHED>
__device__ __forceinline__ int step_of(int i, int nsteps, int T, const int *__restrict__ ord, int per) {
    if constexpr (SCHED == 3) {
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        const int b = local + i * (gridDim.x >> 3);
        return b < per ? ord[xcd * per + b] : -1;
    } else if constexpr (SCHED == 2) {   // XCD-contiguous walk of the order array, like rb_walk
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, per8 = (nsteps + 7) >> 3;
        const int b = xcd * per8 + local + i * (gridDim.x >> 3);
        return b < min((xcd + 1) * per8, nsteps) ? ord[b] : -1;
    } else if constexpr (SCHED == 0) {
        const int s = blockIdx.x + i * gridDim.x;
        return s < nsteps ? s : -1;
    } else {
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nloc = gridDim.x >> 3, tx = T >> 3;
        const int q = local + nloc * i;
        const int s = (q / tx) * T + xcd * tx + q % tx;
        return s < nsteps ? s : -1;
    }
}

template <int OCC, int SCHED>
__global__ void __launch_bounds__(256, OCC) direct_kernel(const double *__restrict__ x, double *__restrict__ y, int nsteps, int D1, int D2, const int *__restrict__ ord, int per) {
    const int tid = threadIdx.x;
    const int T = D2 / 512;
    for (int i = 0;; ++i) {
        const int s = step_of<SCHED>(i, nsteps, T, ord, per);
        if (s < 0) break;
        const double *b = x + (size_t)s * 512 + 2 * tid;
        const d2 a = ld2(b - D2), bb = ld2(b - D1), c = ld2(b - 1), d = ld2(b), e = ld2(b + 1), f = ld2(b + D1), g = ld2(b + D2);
        const d2 r = fold7(a, bb, c, d, e, f, g);
        __builtin_nontemporal_store(r, (d2 *)(y + (size_t)s * 512 + 2 * tid));
    }
}

// D1 <= 256, even; x + 512 s - D1 and x + 512 s +- D2 16-byte aligned
template <int OCC, int SCHED>
__global__ void __launch_bounds__(256, OCC) lds_kernel(const double *__restrict__ x, double *__restrict__ y, int nsteps, int D1, int D2, const int *__restrict__ ord, int per) {
    __shared__ d2 buf[2][1024];   // [0, 512): in-plane window (512 + 2 D1 elements), [512, 768): plane behind, [768, 1024): plane ahead
    const int tid = threadIdx.x;
    const int h1 = D1 >> 1;       // the in-plane window is 256 + D1 pairs: lanes < D1 load a second pair
    d2 r0, r1, r2, r3;
    auto load = [&](int s) {
        const double *b = x + (size_t)s * 512 + 2 * tid;
        r0 = *(const d2 *)(b - D1);
        r1 = tid < D1 ? *(const d2 *)(b - D1 + 512) : d2{0.0, 0.0};
        r2 = *(const d2 *)(b - D2);
        r3 = *(const d2 *)(b + D2);
    };
    auto stage = [&](int p) {
        buf[p][tid] = r0;
        if (tid < D1) buf[p][256 + tid] = r1;
        buf[p][512 + tid] = r2;
        buf[p][768 + tid] = r3;
    };
    const int T = D2 / 512;
    int s = step_of<SCHED>(0, nsteps, T, ord, per), p = 0;
    if (s < 0) return;
    load(s);
    stage(0);
    __syncthreads();
    for (int i = 1; s >= 0; ++i, p ^= 1) {
        const int sn = step_of<SCHED>(i, nsteps, T, ord, per);
        if (sn >= 0) load(sn);
        const double *w = (const double *)buf[p] + D1 + 2 * tid;
        const d2 a = buf[p][512 + tid], g = buf[p][768 + tid];
        const d2 bb = *(const d2 *)(w - D1), f = *(const d2 *)(w + D1);
        const d2 d = *(const d2 *)w;
        const double lo = w[-1], hi = w[2];
        const d2 c = d2{lo, d.x}, e = d2{d.y, hi};
        const d2 r = fold7(a, bb, c, d, e, f, g);
        __builtin_nontemporal_store(r, (d2 *)(y + (size_t)s * 512 + 2 * tid));
        if (sn >= 0) stage(p ^ 1);
        __syncthreads();
        s = sn;
        (void)h1;
    }
}


// the same with the windows of step i + 2 in flight while step i is folded (two register sets): a workgroup's step no longer
// waits for a full memory round trip
template <int OCC, int SCHED>
__global__ void __launch_bounds__(256, OCC) lds2_kernel(const double *__restrict__ x, double *__restrict__ y, int nsteps, int D1, int D2, const int *__restrict__ ord, int per) {
    __shared__ d2 buf[2][1024];
    const int tid = threadIdx.x;
    const int T = D2 / 512;
    d2 ra[4], rb[4];
    auto load = [&](int s, d2 *r) {
        const double *b = x + (size_t)s * 512 + 2 * tid;
        r[0] = *(const d2 *)(b - D1);
        r[1] = *(const d2 *)(b - D1 + (tid < D1 ? 512 : 0));
        r[2] = *(const d2 *)(b - D2);
        r[3] = *(const d2 *)(b + D2);
    };
    auto stage = [&](int p, const d2 *r) {
        buf[p][tid] = r[0];
        if (tid < D1) buf[p][256 + tid] = r[1];
        buf[p][512 + tid] = r[2];
        buf[p][768 + tid] = r[3];
    };
    auto fold = [&](int p, int s) {
        const double *w = (const double *)buf[p] + D1 + 2 * tid;
        const d2 a = buf[p][512 + tid], g = buf[p][768 + tid];
        const d2 bb = *(const d2 *)(w - D1), f = *(const d2 *)(w + D1);
        const d2 d = *(const d2 *)w;
        const double lo = w[-1], hi = w[2];
        const d2 c = d2{lo, d.x}, e = d2{d.y, hi};
        const d2 r = fold7(a, bb, c, d, e, f, g);
        __builtin_nontemporal_store(r, (d2 *)(y + (size_t)s * 512 + 2 * tid));
    };
    int s0 = step_of<SCHED>(0, nsteps, T, ord, per);
    if (s0 < 0) return;
    int s1 = step_of<SCHED>(1, nsteps, T, ord, per);
    load(s0, ra);
    if (s1 >= 0) load(s1, rb);
    if (s1 >= 0) __builtin_amdgcn_s_waitcnt(0x0f74); else __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(4) / vmcnt(0): the first set is here
    stage(0, ra);
    __syncthreads();
    // steps i, i + 1 per trip: ra holds the windows of step i + 2 while i is folded, rb those of i + 3 while i + 1 is folded
    for (int i = 0;; i += 2) {
        const int s2 = step_of<SCHED>(i + 2, nsteps, T, ord, per);
        if (s2 >= 0) load(s2, ra);
        fold(0, s0);
        if (s1 < 0) break;
        if (s2 >= 0) __builtin_amdgcn_s_waitcnt(0x0f74); else __builtin_amdgcn_s_waitcnt(0x0f70);
        stage(1, rb);
        __syncthreads();
        const int s3 = step_of<SCHED>(i + 3, nsteps, T, ord, per);
        if (s3 >= 0) load(s3, rb);
        fold(1, s1);
        if (s2 < 0) break;
        if (s3 >= 0) __builtin_amdgcn_s_waitcnt(0x0f74); else __builtin_amdgcn_s_waitcnt(0x0f70);
        stage(0, ra);
        __syncthreads();
        s0 = s2;
        s1 = s3;
    }
}

// K2 folded into K3's staging: the windows hold s = r - alpha v computed on the fly (8 staging loads per lane instead of 4), the
// workgroup's own 512 rows of s are written out (K4+K5 need s), then the 7-point fold as before.  Against lds_kernel on a
// precomputed s plus a separate kernel that forms s (3 vector passes).
template <int OCC, int SCHED>
__global__ void __launch_bounds__(256, OCC) lds_axpy_kernel(const double *__restrict__ r, const double *__restrict__ v, double alpha,
                                                            double *__restrict__ sout, double *__restrict__ y, int nsteps, int D1, int D2,
                                                            const int *__restrict__ ord, int per) {
    __shared__ d2 buf[2][1024];
    const int tid = threadIdx.x;
    const int T = D2 / 512;
    d2 a0, a1, a2, a3, b0, b1, b2, b3;
    auto load = [&](int s) {
        const size_t o = (size_t)s * 512 + 2 * tid;
        const int second = tid < D1 ? 512 : 0;
        a0 = *(const d2 *)(r + o - D1); b0 = *(const d2 *)(v + o - D1);
        a1 = *(const d2 *)(r + o - D1 + second); b1 = *(const d2 *)(v + o - D1 + second);
        a2 = *(const d2 *)(r + o - D2); b2 = *(const d2 *)(v + o - D2);
        a3 = *(const d2 *)(r + o + D2); b3 = *(const d2 *)(v + o + D2);
    };
    auto stage = [&](int p, int s) {
#pragma clang fp contract(off)
        const d2 s0 = a0 - alpha * b0, s1 = a1 - alpha * b1, s2 = a2 - alpha * b2, s3 = a3 - alpha * b3;
        buf[p][tid] = s0;
        if (tid < D1) buf[p][256 + tid] = s1;
        buf[p][512 + tid] = s2;
        buf[p][768 + tid] = s3;
    };
    int s = step_of<SCHED>(0, nsteps, T, ord, per), p = 0;
    if (s < 0) return;
    load(s);
    stage(0, s);
    __syncthreads();
    for (int i = 1; s >= 0; ++i, p ^= 1) {
        const int sn = step_of<SCHED>(i, nsteps, T, ord, per);
        if (sn >= 0) load(sn);
        const double *w = (const double *)buf[p] + D1 + 2 * tid;
        const d2 a = buf[p][512 + tid], g = buf[p][768 + tid];
        const d2 bb = *(const d2 *)(w - D1), f = *(const d2 *)(w + D1);
        const d2 d = *(const d2 *)w;
        const double lo = w[-1], hi = w[2];
        const d2 c = d2{lo, d.x}, e = d2{d.y, hi};
        const d2 res = fold7(a, bb, c, d, e, f, g);
        *(d2 *)(sout + (size_t)s * 512 + 2 * tid) = d;                       // this workgroup's own rows of s (read again by K4+K5)
        __builtin_nontemporal_store(res, (d2 *)(y + (size_t)s * 512 + 2 * tid));
        if (sn >= 0) stage(p ^ 1, sn);
        __syncthreads();
        s = sn;
    }
}

__global__ void __launch_bounds__(256) axpy_kernel(const double *__restrict__ r, const double *__restrict__ v, double alpha, double *__restrict__ s, size_t n2) {
#pragma clang fp contract(off)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        const d2 a = __builtin_nontemporal_load((const d2 *)r + i), b = __builtin_nontemporal_load((const d2 *)v + i);
        ((d2 *)s)[i] = a - alpha * b;
    }
}


// z-march (round 3): a workgroup owns an in-plane tile of 512 contiguous elements and walks a run of planes.  Per plane it loads
// ONE window (512 + 2 D1 elements of the plane ahead) instead of three: the plane behind is the centre pair kept in a register,
// the plane ahead is the centre of the window staged for the next step.  944 instead of 1974 elements through L1 / L2 per 512
// rows (216^3).  Three LDS buffers in rotation, one barrier per step, DIST planes in flight in registers.
// XCD x (= blockIdx.x % 8) owns TX consecutive tiles; a run is PS planes.
template <int OCC, int DIST>
__global__ void __launch_bounds__(256, OCC) march_kernel(const double *__restrict__ x, double *__restrict__ y, int D1, int D2, int G, int T,
                                                         int TX, int PS) {
    __shared__ d2 buf[3][512];
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tile = xcd * TX + local % TX, seg = local / TX;
    if (tile >= T || local % TX >= TX) return;
    const int k0 = seg * PS, k1 = min(k0 + PS, G);
    if (k0 >= k1) return;
    const int valid = min(512, D2 - tile * 512);
    const int h1 = D1 >> 1;
    const double *xt = x + (size_t)tile * 512 - D1 + 2 * tid;
    auto ld = [&](int k, d2 &r0, d2 &r1) {
        const double *b = xt + (ptrdiff_t)min(k, G) * D2;
        r0 = *(const d2 *)b;
        r1 = *(const d2 *)(b + (tid < D1 ? 512 : 0));
    };
    auto st = [&](int q, const d2 &r0, const d2 &r1) {
        buf[q][tid] = r0;
        if (tid < D1) buf[q][256 + tid] = r1;
    };
    d2 ra[DIST], rb[DIST], p0, p1;
    ld(k0 - 1, p0, p1);
    st(0, p0, p1);
    ld(k0, p0, p1);
    st(1, p0, p1);
#pragma unroll
    for (int i = 0; i < DIST; ++i) ld(k0 + 1 + i, ra[i], rb[i]);
    __syncthreads();
    d2 a = buf[0][h1 + tid];
    int jb = 1;                                      // buffer of plane k
    for (int k = k0; k < k1; k += DIST) {
#pragma unroll
        for (int i = 0; i < DIST; ++i) {
            const int kk = k + i;
            if (kk >= k1) break;
            const int jn = jb == 2 ? 0 : jb + 1;     // buffer of plane kk + 1
            st(jn, ra[i], rb[i]);
            ld(kk + 1 + DIST, ra[i], rb[i]);
            __syncthreads();
            const double *w = (const double *)buf[jb] + D1 + 2 * tid;
            const d2 g = buf[jn][h1 + tid];
            const d2 bb = *(const d2 *)(w - D1), f = *(const d2 *)(w + D1);
            const d2 d = *(const d2 *)w;
            const double lo = w[-1], hi = w[2];
            const d2 c = d2{lo, d.x}, e = d2{d.y, hi};
            const d2 r = fold7(a, bb, c, d, e, f, g);
            if (2 * tid < valid) __builtin_nontemporal_store(r, (d2 *)(y + (size_t)kk * D2 + (size_t)tile * 512 + 2 * tid));
            a = d;
            jb = jn;
        }
    }
}

// The same march with the windows staged by direct-to-LDS loads (round 5 probe: global_load_lds, 16 B per lane -- the window layout is
// lane-contiguous already): no staging registers, no ds_write, and DIST planes in flight cost LDS (DIST + 3 buffers of 256 + D1 pairs)
// instead of registers.  One raw s_barrier per step; the wait is counted (vmcnt(2 DIST): the planes further ahead stay in flight -- stores
// share the counter, so the count is the conservative one).  Buffer of plane p: (p - (k0 - 1)) mod (DIST + 3); the buffer written at the
// top of step kk held plane kk - 2, which every wavefront finished reading before it passed the barrier of step kk - 1.
template <int OCC, int DIST>
__global__ void __launch_bounds__(256, OCC) march_glds_kernel(const double *__restrict__ x, double *__restrict__ y, int D1, int D2, int G, int T,
                                                              int TX, int PS) {
    constexpr int NB = DIST + 3;
    extern __shared__ d2 gbuf[];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int stride = 256 + D1;                          // pairs per buffer
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tile = xcd * TX + local % TX, seg = local / TX;
    if (tile >= T || local % TX >= TX) return;
    const int k0 = seg * PS, k1 = min(k0 + PS, G);
    if (k0 >= k1) return;
    const int valid = min(512, D2 - tile * 512);
    const int h1 = D1 >> 1;
    const double *xt = x + (size_t)tile * 512 - D1 + 2 * tid;
    auto glds = [&](int k, int q) {
        const double *b = xt + (ptrdiff_t)min(k, G) * D2;
        d2 *dst = gbuf + (size_t)q * stride;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)b, (void __attribute__((address_space(3))) *)(dst + wave * 64), 16, 0, 0);
        if (tid < D1)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(b + 512), (void __attribute__((address_space(3))) *)(dst + 256 + wave * 64), 16, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < DIST + 2; ++i) glds(k0 - 1 + i, i);          // planes k0 - 1 .. k0 + DIST
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DIST));             // plane k0 - 1 and k0 are in
    __builtin_amdgcn_s_barrier();
    d2 a = gbuf[h1 + tid];                                            // centre pair of plane k0 - 1 (buffer 0)
    int jb = 1, jw = DIST + 2;                                        // buffer of plane kk; buffer the next glds goes to
    for (int kk = k0; kk < k1; ++kk) {
        const int jn = jb + 1 == NB ? 0 : jb + 1;                     // buffer of plane kk + 1
        glds(kk + 1 + DIST, jw);
        jw = jw + 1 == NB ? 0 : jw + 1;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DIST) : "memory");   // plane kk + 1 is in (this wavefront's part)
        __builtin_amdgcn_s_barrier();
        const d2 *cur = gbuf + (size_t)jb * stride, *nxt = gbuf + (size_t)jn * stride;
        const double *w = (const double *)cur + D1 + 2 * tid;
        const d2 g = nxt[h1 + tid];
        const d2 bb = *(const d2 *)(w - D1), f = *(const d2 *)(w + D1);
        const d2 d = *(const d2 *)w;
        const double lo = w[-1], hi = w[2];
        const d2 c = d2{lo, d.x}, e = d2{d.y, hi};
        const d2 r = fold7(a, bb, c, d, e, f, g);
        if (2 * tid < valid) __builtin_nontemporal_store(r, (d2 *)(y + (size_t)kk * D2 + (size_t)tile * 512 + 2 * tid));
        a = d;
        jb = jn;
    }
}

// z-march with K2 folded into the staging (round 3): the window of r - alpha v is formed on the fly (two staging loads per vector and
// lane instead of two), own rows of s written for K4+K5.  Product structure: one round trip per step, the wait in front of the stores.
// FUSED = false: the same kernel on a precomputed s (what spmv_wdia_march_kernel does).
template <int OCC, bool FUSED>
__global__ void __launch_bounds__(256, OCC) march2_kernel(const double *__restrict__ r, const double *__restrict__ v, double alpha,
                                                          double *__restrict__ sout, double *__restrict__ y, int D1, int D2, int G, int T, int TX, int PS) {
    __shared__ d2 buf[4][512];
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tile = xcd * TX + local % TX, seg = local / TX;
    if (tile >= T) return;
    const int k0 = seg * PS, k1 = min(k0 + PS, G);
    if (k0 >= k1) return;
    const int valid = min(512, D2 - tile * 512);
    const int h1 = D1 >> 1;
    const size_t xo = (size_t)tile * 512 - D1 + 2 * tid;
    const int second = tid < D1 ? 512 : 0;
    d2 a0, a1, b0, b1;
    auto ld = [&](int k) {
        const ptrdiff_t o = (ptrdiff_t)min(k, G) * D2 + (ptrdiff_t)xo;
        a0 = *(const d2 *)(r + o);
        a1 = *(const d2 *)(r + o + second);
        if (FUSED) { b0 = *(const d2 *)(v + o); b1 = *(const d2 *)(v + o + second); }
    };
    // own rows of the window: pairs [h1, h1 + 256): lane tid >= h1 holds one as its first pair, lane tid < h1 as its second
    auto st = [&](int q, int k, bool own) {
#pragma clang fp contract(off)
        d2 s0 = a0, s1 = a1;
        if (FUSED) { s0 = a0 - alpha * b0; s1 = a1 - alpha * b1; }
        buf[q][tid] = s0;
        if (tid < D1) buf[q][256 + tid] = s1;
        if (FUSED && own) {
            const int pr = tid >= h1 ? tid - h1 : tid + 256 - h1;       // this lane's own pair inside the tile
            if (2 * pr < valid) *(d2 *)(sout + (size_t)k * D2 + (size_t)tile * 512 + 2 * pr) = tid >= h1 ? s0 : s1;
        }
    };
    ld(k0 - 1); st(3, k0 - 1, false);
    ld(k0); st(0, k0, true);
    ld(k0 + 1); st(1, k0 + 1, k0 + 1 < k1);
    __syncthreads();
    for (int k = k0; k < k1; k += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kk = k + u;
            if (kk >= k1) break;
            ld(min(kk + 2, k1));
            const double *w = (const double *)buf[u] + D1 + 2 * tid;
            const d2 a = buf[(u + 3) & 3][h1 + tid], g = buf[(u + 1) & 3][h1 + tid];
            const d2 bb = *(const d2 *)(w - D1), f = *(const d2 *)(w + D1);
            const d2 d = *(const d2 *)w;
            const double lo = w[-1], hi = w[2];
            const d2 c = d2{lo, d.x}, e = d2{d.y, hi};
            const d2 res = fold7(a, bb, c, d, e, f, g);
            __builtin_amdgcn_s_waitcnt(0x0f70);
            st((u + 2) & 3, kk + 2, kk + 2 < k1);
            if (2 * tid < valid) __builtin_nontemporal_store(res, (d2 *)(y + (size_t)kk * D2 + (size_t)tile * 512 + 2 * tid));
            __syncthreads();
        }
    }
}

// z-march with tiles of RP * 512 rows (RP row pairs per lane): the in-plane halo of 2 D1 elements is shared by more rows
// (216^3: 944 / 512 = 1.84 x for RP = 1, 1456 / 1024 = 1.42 x for RP = 2).  Product structure (one round trip per step).
template <int OCC, int RP>
__global__ void __launch_bounds__(256, OCC) march3_kernel(const double *__restrict__ x, double *__restrict__ y, int D1, int D2, int G, int T, int TX, int PS) {
    constexpr int TILE = 512 * RP, NL = RP + 1;          // staging loads per lane: RP * 256 + D1 pairs <= NL * 256
    __shared__ d2 buf[4][256 * NL];
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tile = xcd * TX + local % TX, seg = local / TX;
    if (tile >= T) return;
    const int k0 = seg * PS, k1 = min(k0 + PS, G);
    if (k0 >= k1) return;
    const int valid = min(TILE, D2 - tile * TILE);
    const int h1 = D1 >> 1;
    const size_t xo = (size_t)tile * TILE - D1 + 2 * tid;
    d2 r[NL];
    auto ld = [&](int k) {
        const ptrdiff_t o = (ptrdiff_t)min(k, G) * D2 + (ptrdiff_t)xo;
#pragma unroll
        for (int j = 0; j < NL; ++j) r[j] = *(const d2 *)(x + o + ((j < RP || tid < D1) ? 512 * j : 0));
    };
    auto st = [&](int q) {
#pragma unroll
        for (int j = 0; j < NL; ++j)
            if (j < RP || tid < D1) buf[q][tid + 256 * j] = r[j];
    };
    ld(k0 - 1); st(3);
    ld(k0); st(0);
    ld(k0 + 1); st(1);
    __syncthreads();
    for (int k = k0; k < k1; k += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kk = k + u;
            if (kk >= k1) break;
            ld(min(kk + 2, k1));
            d2 res[RP];
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                const double *w = (const double *)buf[u] + D1 + 2 * tid + 512 * j;
                const d2 a = buf[(u + 3) & 3][h1 + tid + 256 * j], g = buf[(u + 1) & 3][h1 + tid + 256 * j];
                const d2 bb = *(const d2 *)(w - D1), f = *(const d2 *)(w + D1);
                const d2 d = *(const d2 *)w;
                const double lo = w[-1], hi = w[2];
                res[j] = fold7(a, bb, d2{lo, d.x}, d, d2{d.y, hi}, f, g);
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);
            st((u + 2) & 3);
#pragma unroll
            for (int j = 0; j < RP; ++j)
                if (2 * tid + 512 * j < valid) __builtin_nontemporal_store(res[j], (d2 *)(y + (size_t)kk * D2 + (size_t)tile * TILE + 2 * tid + 512 * j));
            __syncthreads();
        }
    }
}

// y = 6 x with the streaming shape of the vector kernels: what a plain pass over x + y costs on the same rotating buffers
__global__ void __launch_bounds__(256) copy_kernel(const double *__restrict__ x, double *__restrict__ y, size_t n2) {
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        const d2 v = ((const d2 *)x)[i];
        __builtin_nontemporal_store(6.0 * v, (d2 *)y + i);
    }
}

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 216;
    const int D1 = G, D2 = G * G;
    const size_t n = (size_t)G * G * G / 512 * 512;
    const int nsteps = (int)(n / 512);
    const int pairs = 4, reps = 40;
    const size_t pad = (size_t)D2 + 512;
    std::vector<double> hx(n + 2 * pad);
    uint64_t z = 88172645463325252ull;
    for (auto &v : hx) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; v = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5; }
    std::vector<double *> dx(pairs), dy(pairs);
    for (int i = 0; i < pairs; ++i) {
        CK(hipMalloc(&dx[i], (n + 2 * pad) * 8));
        CK(hipMalloc(&dy[i], n * 8));
        CK(hipMemcpy(dx[i], hx.data(), (n + 2 * pad) * 8, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<double> ref(n), got(n);
    auto run = [&](const char *name, auto launch, bool check) {
        for (int i = 0; i < 4; ++i) launch(dx[i % pairs] + pad, dy[i % pairs]);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch(dx[i % pairs] + pad, dy[i % pairs]);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch(dx[0] + pad, dy[0]);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms1;
        CK(hipEventElapsedTime(&ms1, e0, e1));
        CK(hipMemcpy(got.data(), dy[0], n * 8, hipMemcpyDeviceToHost));
        const char *ok = "";
        if (!check) ref = got;
        else ok = memcmp(ref.data(), got.data(), n * 8) == 0 ? "  bit-identical to direct" : "  MISMATCH";
        const double us = ms * 1e3 / reps;
        printf("%-28s %7.1f us  %6.2f TB/s of x + y (rotating %d pairs; one pair: %.1f us)%s\n", name, us, 16.0 * n / us * 1e-6, pairs, ms1 * 1e3 / reps, ok);
    };
    printf("# %d^3 grid: %zu rows, %d steps of 512 rows\n", G, n, nsteps);
#define DIRECT(OCC, SC) run(SC == 3 ? "direct sweep occ " #OCC : SC == 2 ? "direct order[] occ " #OCC : SC ? "direct sched occ " #OCC : "direct occ " #OCC, [&](const double *x, double *y) { hipLaunchKernelGGL((direct_kernel<OCC, SC>), dim3(256 * OCC), dim3(256), 0, 0, x, y, nsteps, D1, D2, SC == 3 ? d_sweep : d_order, sweep_per); }, !first); first = false
#define LDSK(OCC, SC) run(SC == 3 ? "lds sweep occ " #OCC : SC == 2 ? "lds order[] occ " #OCC : SC ? "lds sched occ " #OCC : "lds occ " #OCC, [&](const double *x, double *y) { hipLaunchKernelGGL((lds_kernel<OCC, SC>), dim3(256 * OCC), dim3(256), 0, 0, x, y, nsteps, D1, D2, SC == 3 ? d_sweep : d_order, sweep_per); }, true)
    int *d_order = nullptr, *d_sweep = nullptr, sweep_per = 0;
    {   // the product's order: stable sort of the steps by (position inside the plane) / tile
        const double bpp = (double)D2 / 512.0;
        const int tile = (int)std::max(8.0, bpp / 6.0 + 0.5);
        std::vector<int> order(nsteps), key(nsteps);
        for (int b = 0; b < nsteps; ++b) { order[b] = b; key[b] = (int)(fmod((double)b, bpp) / tile); }
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return key[x] < key[y]; });
        int *d;
        CK(hipMalloc(&d, sizeof(int) * nsteps));
        CK(hipMemcpy(d, order.data(), sizeof(int) * nsteps, hipMemcpyHostToDevice));
        d_order = d;
        printf("# order[]: %.3f steps per plane, tiles of %d steps\n", bpp, tile);
    }
    {   // plane sweep, in-plane eighths per XCD
        const double bpp = (double)D2 / 512.0;
        std::vector<std::vector<int>> lists(8);
        for (int b = 0; b < nsteps; ++b) lists[std::min(7, (int)(fmod((double)b, bpp) * 8.0 / bpp))].push_back(b);
        size_t per = 0;
        for (auto &l : lists) per = std::max(per, l.size());
        std::vector<int> flat(8 * per, -1);
        for (int x = 0; x < 8; ++x) std::copy(lists[x].begin(), lists[x].end(), flat.begin() + x * per);
        int *d;
        const int iper = (int)per;
        CK(hipMalloc(&d, sizeof(int) * flat.size()));
        CK(hipMemcpy(d, flat.data(), sizeof(int) * flat.size(), hipMemcpyHostToDevice));
        d_sweep = d;
        sweep_per = iper;
    }
#define LDS2(OCC, SC) run(SC == 3 ? "lds dist-2 sweep occ " #OCC : SC == 2 ? "lds dist-2 order[] occ " #OCC : "lds dist-2 occ " #OCC, [&](const double *x, double *y) { hipLaunchKernelGGL((lds2_kernel<OCC, SC>), dim3(256 * OCC), dim3(256), 0, 0, x, y, nsteps, D1, D2, SC == 3 ? d_sweep : d_order, sweep_per); }, true)
    bool first = true;
    DIRECT(6, 0);
    const bool only_k23 = getenv("PROBE_K23") != nullptr;
    if (getenv("PROBE_MARCH") && !only_k23) {
    run("copy (16 B per lane)", [&](const double *x, double *y) { hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, x, y, n / 2); }, false);
    first = true;
    DIRECT(6, 0);
    {
        const int T = (D2 + 511) / 512, TX = (T + 7) / 8;
#define MARCH(OCC, DIST) do { const int S = std::max(1, (OCC * 256) / (8 * TX)), PS = (G + S - 1) / S; char nm[64]; snprintf(nm, sizeof nm, "march occ %d dist %d runs of %d", OCC, DIST, PS); \
        run(nm, [&](const double *x, double *y) { hipLaunchKernelGGL((march_kernel<OCC, DIST>), dim3(8 * TX * S), dim3(256), 0, 0, x, y, D1, D2, G, T, TX, PS); }, true); } while (0)

#define MARCH3(OCC, RP_) do { const int T3 = (D2 + 512 * RP_ - 1) / (512 * RP_), TX3 = (T3 + 7) / 8; const int S = std::max(1, (OCC * 256) / (8 * TX3)), PS = (G + S - 1) / S; char nm[64]; \
        snprintf(nm, sizeof nm, "march tile %d occ %d runs of %d", 512 * RP_, OCC, PS); \
        run(nm, [&](const double *x, double *y) { hipLaunchKernelGGL((march3_kernel<OCC, RP_>), dim3(8 * TX3 * S), dim3(256), 0, 0, x, y, D1, D2, G, T3, TX3, PS); }, true); } while (0)
        MARCH3(3, 1); MARCH3(4, 1); MARCH3(5, 1);
        MARCH3(2, 2); MARCH3(3, 2); MARCH3(4, 2);
        MARCH3(2, 3); MARCH3(3, 3);
#define MARCHG(OCC, DIST) do { const int S = std::max(1, (OCC * 256) / (8 * TX)), PS = (G + S - 1) / S; char nm[64]; snprintf(nm, sizeof nm, "march glds occ %d dist %d", OCC, DIST); \
        const size_t lds = sizeof(double) * 2 * (size_t)(DIST + 3) * (256 + D1); \
        CK(hipFuncSetAttribute((const void *)march_glds_kernel<OCC, DIST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        run(nm, [&](const double *x, double *y) { hipLaunchKernelGGL((march_glds_kernel<OCC, DIST>), dim3(8 * TX * S), dim3(256), lds, 0, x, y, D1, D2, G, T, TX, PS); }, true); } while (0)
        MARCHG(4, 1); MARCHG(4, 2); MARCHG(3, 2); MARCHG(3, 3); MARCHG(5, 1); MARCHG(6, 1); MARCHG(2, 4);
        MARCH(3, 1); MARCH(3, 2); MARCH(3, 3);
        MARCH(4, 1); MARCH(4, 2); MARCH(4, 3); MARCH(4, 4);
        MARCH(5, 2); MARCH(5, 3);
        MARCH(6, 2); MARCH(6, 3); MARCH(6, 4);
        MARCH(8, 2); MARCH(8, 3);
    }
    LDSK(3, 3);
    LDSK(4, 3);
        return 0;
    }
    if (!only_k23) {
    LDSK(3, 0);
    DIRECT(6, 2);
    LDSK(3, 2);
    LDSK(4, 2);
    LDS2(3, 2);
    LDS2(4, 2);
    LDS2(4, 3);
    DIRECT(6, 3);
    DIRECT(4, 3);
    LDSK(3, 3);
    LDSK(4, 3);
    LDSK(5, 3);
    if (D2 % (512 * 8) == 0) {   // whole tiles per plane and XCD
        DIRECT(6, 1);
        DIRECT(4, 1);
        DIRECT(8, 1);
        LDSK(2, 1);
        LDSK(3, 1);
        LDSK(4, 1);
    }

    }
    {   // K2 + K3: separate (axpy kernel, then the staged SpMV on its result) vs folded into the staging
        double *dv[pairs], *ds[pairs];
        for (int i = 0; i < pairs; ++i) {
            CK(hipMalloc(&dv[i], (n + 2 * pad) * 8));
            CK(hipMalloc(&ds[i], (n + 2 * pad) * 8));
            CK(hipMemcpy(dv[i], hx.data(), (n + 2 * pad) * 8, hipMemcpyHostToDevice));
            CK(hipMemset(ds[i], 0, (n + 2 * pad) * 8));
        }
        const double alpha = 0.37;
        auto both = [&](const char *name, auto launch) {
            for (int i = 0; i < 4; ++i) launch(i % pairs);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) launch(i % pairs);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(got.data(), dy[0], n * 8, hipMemcpyDeviceToHost));
            return std::make_pair(ms * 1e3 / reps, got);
        };
        auto sep = both("separate", [&](int i) {
            hipLaunchKernelGGL(axpy_kernel, dim3(2048), dim3(256), 0, 0, dx[i], dv[i], alpha, ds[i], (n + 2 * pad) / 2);
            hipLaunchKernelGGL((lds_kernel<4, 2>), dim3(1024), dim3(256), 0, 0, ds[i] + pad, dy[i], nsteps, D1, D2, d_order, sweep_per);
        });
        auto fus = both("fused", [&](int i) {
            hipLaunchKernelGGL((lds_axpy_kernel<4, 2>), dim3(1024), dim3(256), 0, 0, dx[i] + pad, dv[i] + pad, alpha, ds[i] + pad, dy[i], nsteps, D1, D2, d_order, sweep_per);
        });
        {
            const int T = (D2 + 511) / 512, TX = (T + 7) / 8;
            const int S = std::max(1, (4 * 256) / (8 * TX)), PS = (G + S - 1) / S;
            auto msep = both("march separate", [&](int i) {
                hipLaunchKernelGGL(axpy_kernel, dim3(2048), dim3(256), 0, 0, dx[i], dv[i], alpha, ds[i], (n + 2 * pad) / 2);
                hipLaunchKernelGGL((march2_kernel<4, false>), dim3(8 * TX * S), dim3(256), 0, 0, ds[i] + pad, dv[i] + pad, alpha, ds[i] + pad, dy[i], D1, D2, G, T, TX, PS);
            });
            auto mfus = both("march fused", [&](int i) {
                hipLaunchKernelGGL((march2_kernel<4, true>), dim3(8 * TX * S), dim3(256), 0, 0, dx[i] + pad, dv[i] + pad, alpha, ds[i] + pad, dy[i], D1, D2, G, T, TX, PS);
            });
            printf("K2 + K3, plane march, rotating %d vector sets: separate kernels %.1f us, K2 folded into the staging %.1f us%s\n", pairs, msep.first,
                   mfus.first, memcmp(msep.second.data(), mfus.second.data(), n * 8) == 0 ? "  (y bit-identical)" : "  (y MISMATCH)");
        }
        printf("K2 + K3, product order, rotating %d vector sets: separate kernels %.1f us, K2 folded into the staging %.1f us%s\n", pairs, sep.first,
               fus.first, memcmp(sep.second.data(), fus.second.data(), n * 8) == 0 ? "  (y bit-identical)" : "  (y MISMATCH)");
    }
    return 0;
}
