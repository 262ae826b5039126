// sla_probe.hip -- measurement hook: what this GPU sustains for the access patterns of the solver's vector kernels.
//
// bench.py prints a "measured ceiling" next to the 8 TB/s spec peak.  Round 2 took it from an axpby triad over
// rotating vectors and the product's own K45 sweep beat it by 23 %: a ceiling the product exceeds is a broken probe.  The probe
// is now the same access shape as the kernels it is compared with -- R vectors read and W vectors written per element,
// 16 bytes per lane, grid of vec_grid(n), non-temporal loads once the footprint overflows the memory-side cache -- for the
// shapes (R, W) = (8, 0) pure read, (5, 3) the K4+K5 sweep, (2, 1) a triad.  Timed with HIP events around every launch on
// the context stream.  No reference counterpart (the reference has no device).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "sla_internal.hpp"
#include "sla_device.hpp"

namespace sla {

namespace {
typedef double probe_d2 __attribute__((ext_vector_type(2)));
struct ProbePtrs {
    const double *in[8];
    double *out[4];
};

template <int R, int W, bool NT>
__global__ void __launch_bounds__(kBlock) stream_probe_kernel(ProbePtrs p, int64_t n2, double *parts) {
    __shared__ double s_red[4];
    double acc = 0.0;
    const int64_t gs = (int64_t)gridDim.x * kBlock;
    for (int64_t i2 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i2 < n2; i2 += gs) {
        probe_d2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const probe_d2 *src = reinterpret_cast<const probe_d2 *>(p.in[r]) + i2;
            v[r] = NT ? __builtin_nontemporal_load(src) : *src;
        }
        probe_d2 s = v[0];
#pragma unroll
        for (int r = 1; r < R; ++r) s += v[r];
        if constexpr (W == 0) {
            acc += s.x + s.y;
        } else {
#pragma unroll
            for (int w = 0; w < W; ++w) reinterpret_cast<probe_d2 *>(p.out[w])[i2] = s * (double)(w + 1);
        }
    }
    if constexpr (W == 0) {
        const double t = block_sum(acc, s_red);
        if (threadIdx.x == 0) parts[blockIdx.x] = t;
    }
}

template <int R, int W>
void launch_probe(sla_ctx *c, const ProbePtrs &p, int64_t n, bool nt) {
    const int grid = vec_grid(n);
    if (nt) hipLaunchKernelGGL((stream_probe_kernel<R, W, true>), dim3(grid), dim3(kBlock), 0, stream_of(c), p, n >> 1, c->d_parts);
    else hipLaunchKernelGGL((stream_probe_kernel<R, W, false>), dim3(grid), dim3(kBlock), 0, stream_of(c), p, n >> 1, c->d_parts);
}
}  // namespace

}  // namespace sla

using namespace sla;

extern "C" int sla_stream_probe(sla_ctx_t c, int reads, int writes, int64_t n, int reps, double *mean_ms, double *min_ms) {
    if (c && !c->kids.empty()) return sla_stream_probe(c->kids[0], reads, writes, n, reps, mean_ms, min_ms);
    return no_throw("sla_stream_probe", [&]() -> int {
        if (!c || n < 2 || reps < 1 || reps > 4096 || !mean_ms) return fail(SLA_ERR_INVALID, "sla_stream_probe: bad argument");
        if (!((reads == 8 && writes == 0) || (reads == 5 && writes == 3) || (reads == 2 && writes == 1)))
            return fail(SLA_ERR_INVALID, "sla_stream_probe: (reads, writes) must be (8, 0), (5, 3) or (2, 1)");
        Bind bind(c);
        n &= ~(int64_t)1;
        struct Bufs {
            std::vector<void *> p;
            ~Bufs() { for (void *q : p) (void)hipFree(q); }
        } bufs;
        ProbePtrs pp{};
        for (int i = 0; i < reads + writes; ++i) {
            void *q = nullptr;
            SLA_HIP_TRY(dev_malloc(c, &q, sizeof(double) * (size_t)n));
            bufs.p.push_back(q);
            SLA_HIP_TRY(hipMemsetAsync(q, 0, sizeof(double) * (size_t)n, stream_of(c)));
            if (i < reads) pp.in[i] = (const double *)q;
            else pp.out[i - reads] = (double *)q;
        }
        const bool nt = (int64_t)(reads + writes) * 8 * n > c->mall_bytes;
        auto launch = [&]() {
            if (reads == 8) launch_probe<8, 0>(c, pp, n, nt);
            else if (reads == 5) launch_probe<5, 3>(c, pp, n, nt);
            else launch_probe<2, 1>(c, pp, n, nt);
        };
        for (int i = 0; i < 3; ++i) launch();
        SLA_HIP_TRY(hipGetLastError());
        std::vector<hipEvent_t> ev((size_t)reps + 1);
        for (auto &e : ev) SLA_HIP_TRY(hipEventCreate(&e));
        SLA_HIP_TRY(hipEventRecord(ev[0], stream_of(c)));
        for (int i = 0; i < reps; ++i) {
            launch();
            SLA_HIP_TRY(hipEventRecord(ev[(size_t)i + 1], stream_of(c)));
        }
        SLA_HIP_TRY(hipStreamSynchronize(stream_of(c)));
        double sum = 0.0, mn = 1e300;
        for (int i = 0; i < reps; ++i) {
            float ms = 0.f;
            SLA_HIP_TRY(hipEventElapsedTime(&ms, ev[(size_t)i], ev[(size_t)i + 1]));
            sum += ms;
            mn = std::min<double>(mn, ms);
        }
        for (auto &e : ev) (void)hipEventDestroy(e);
        *mean_ms = sum / reps;
        if (min_ms) *min_ms = mn;
        return SLA_OK;
    });
}
