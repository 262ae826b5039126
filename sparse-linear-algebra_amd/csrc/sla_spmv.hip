// sla_spmv.hip -- (#>) dispatch (Data/Sparse/Common.hs:242-260): which kernel family runs a launch on a lowered matrix, with what
// grid, and the column-panel pass sequence.  The kernels themselves live with their families:
//   sla_spmv_stream.hip   CSR-stream (values + i32 columns), LDS x window, dual (K1 + residual), one-lane-per-row baseline
//   sla_spmv_pipe.hip     the three-stage pipelined CSR-stream kernel
//   sla_spmv_dict.hip     1-byte column codes (diagonal dictionary) and 1-byte (offset, value) pair codes
//   sla_spmv_wdia.hip     wave-sliced (offset, value) records in SGPRs;  sla_spmv_wdia_lds.hip: uniform records + LDS windows;
//                         sla_spmv_wdia_march.hip: the same walking a 3-D stencil plane by plane
//   sla_spmv_lpanel.hip   dense rows: x panels in LDS;                   sla_spmv_tiles.hip: row-slice x column-panel tiles
// All families share the fused epilogues / prologue of sla_device.hpp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "sla_internal.hpp"

namespace sla {

// row-sharded overlap (sla_api.cpp: spmv_exchanged): interior / boundary launches of the wave-sliced forms
bool overlap_split(const sla_csr *A) {
    return A->ov_nint > 0 && A->ov_nbnd > 0 && A->ctx->overlap >= 0 && A->ctx->collectives && A->use_wdia && wd_on(A) && A->ctx->spmv_algo == 0 && !A->rp64;
}
int overlap_grid(const sla_csr *A, int part) {
    // the two launches write their fused partial sums into consecutive slots of ONE kMaxParts-slot array (interior first):
    // the interior grid leaves room for the boundary launch's (a 299-CU part at 6 workgroups per CU, or SLA_WD_GRID=2048,
    // would otherwise push the boundary partials into the next slot array)
    const int gb = std::max(1, std::min<int>(A->ov_nbnd, 256));
    if (part != 1) return gb;
    const int cap = std::min(A->wd_vv ? A->ctx->wd_grid_max_vv : wd_lds_on(A) ? wd_lds_grid(A) : A->ctx->wd_grid_max, kMaxParts - gb);
    return std::max(1, std::min<int>(A->ov_nint, cap));
}

// does a plain (#>) on A end up on spmv_wave_kernel?  (the forms dispatched in front of it in launch_spmv_rp all have to be out)
bool wave_plain(const sla_csr *A) {
    return wave_on(A) && !A->is_panel_view && !diag_on(A) && !stream_xwin_on(A);
}

// BiCGSTAB's K2 folded into K3 (SpmvLaunch::fs_ap; cgsStep's C2 into C3 on the plane march): would the whole-matrix (slab: whole-slab) launch
// land on a kernel that has the variant -- the plane march, or the gather kernel of the wave-sliced forms?  The conditions of the dispatch
// below, in its order
bool spmv_fuse_s_ok(const sla_csr *A, bool slab) {
    const sla_ctx *c = A->ctx;
    // (slab: the whole-slab launches of the ghost-row flow -- r and Ap are valid on the ghost rows, x is addressed by global column)
    if ((!slab && (c->collectives || A->row_begin != 0)) || A->rp64 || A->is_panel_view) return false;
    if (tiles_on(A) && !lflat_on(A)) return false;
    if (!A->panels.empty() && c->panels && c->spmv_algo == 0) return false;
    if ((A->use_lpanel && c->lpanel && c->spmv_algo == 0) || lflat_on(A)) return false;
    if (!(A->use_wdia && wd_on(A) && c->spmv_algo == 0)) return false;
    if (wd_march_on(A)) return true;
    // the gather kernel of the wave-sliced forms (spmv_wdia_kernel<.., SF>: every gather loads r AND Ap) where the two vectors stay in the
    // caches -- 4 M rows = 2 x 32 MB; beyond that the doubled gathers cost more than the 24 n-byte pass they save (bicg_fuse23 = 2: always)
    if (wd_lds_on(A)) return false;                    // (the three-window kernel has no such variant: round 2's probe, LABNOTES L7)
    return c->bicg_fuse23 == 2 || A->rows <= ((int64_t)1 << 22);
}

int spmv_grid(const sla_csr *A) {
    const sla_ctx *c = A->ctx;
    // with column panels the fused partials are written by the LAST panel pass: its grid is the one that counts
    if (tiles_on(A) && !lflat_on(A)) return tiles_grid(A);
    if (!A->panels.empty() && c->panels && c->spmv_algo == 0) return spmv_grid(A->panels.back());
    int64_t g;
    if (c->spmv_algo == 1 || (A->use_lpanel && c->lpanel) || lflat_on(A)) g = (A->rows + kBlock - 1) / kBlock;   // (lpanel / lflat: the finish kernel)
    else if (A->use_wdia && wd_on(A) && wd_march_on(A)) g = wd_march_grid(A);   // (the whole-matrix launch: overlap_grid sizes the split ones)
    else if (A->use_wdia && wd_on(A)) g = std::min<int64_t>(A->nblk_wd, A->wd_vv ? c->wd_grid_max_vv : wd_lds_on(A) ? wd_lds_grid(A) : c->wd_grid_max);
    else if (A->use_vdict && c->vdict) g = A->nblk_vd;
    else if (wave_plain(A)) return wave_grid(A);
    else g = A->nrb;
    if (g < 1) g = 1;
    if (g > c->spmv_grid_max) g = c->spmv_grid_max;
    return (int)g;
}

template <typename RP>
static int launch_spmv_rp(const sla_csr *A, const SpmvLaunch &l);

// Column-panel SpMV for matrices whose gathers do not fit the XCD-private L2: the matrix is stored a second
// time panel-major (panel p holds the entries with column in [p W, (p+1) W), W * 8 B <= ~3 MB), and y = A x
// is evaluated as P passes y += A_p x in ascending panel order, so every gather of a pass hits a window of
// x that stays L2-resident.  Each pass continues the row's running sum (yinit), i.e. the per-row order is
// still one ascending left fold; the fused epilogue runs in the last pass only.
static int launch_spmv_panels(const sla_csr *A, const SpmvLaunch &l) {
    const size_t P = A->panels.size();
    ProfScope prof(A->ctx, l.kernel_id);  // one timed interval for the whole panel sequence
    double *ytmp = l.y;
    if (!ytmp) ytmp = A->d_panel_y;  // epilogues that never store y (EPI_RES, EPI_AXPY_DOT, ...) still need the running sum
    for (size_t p = 0; p < P; ++p) {
        const bool last = p + 1 == P;
        SpmvLaunch lp = l;
        lp.yinit = p == 0 ? nullptr : ytmp;
        lp.kernel_id = -2;               // never matches: the enclosing scope does the timing
        if (!last) {
            lp.epi = EPI_NONE;
            lp.y = ytmp;
            lp.pres = nullptr;            // prologue checks / step bookkeeping happen once, in the last pass
            lp.step_begin = 0;
            lp.pa = nullptr;
        }
        // (a view holds only its panel's entries: it may use 32-bit row pointers where the parent needs 64-bit ones)
        const sla_csr *V = A->panels[p];
        SLA_TRY(V->rp64 ? launch_spmv_rp<int64_t>(V, lp) : launch_spmv_rp<int32_t>(V, lp));
    }
    return SLA_OK;
}

template <typename RP>
static int launch_spmv_rp(const sla_csr *A, const SpmvLaunch &l) {
    sla_ctx *c = A->ctx;
    if (!A->panels.empty() && c->panels && c->spmv_algo == 0 && !l.x2 && !l.in_panel) {
        SpmvLaunch lp = l;
        lp.in_panel = 1;
        return launch_spmv_panels(A, lp);
    }
    if (l.x2 && l.epi != EPI_DOT) return fail(SLA_ERR_INVALID, "dual SpMV is only defined for the K1 epilogue");
    SpmvArgs<RP> a;
    a.rowptr = (const RP *)A->d_rowptr;
    if (A->canon_lazy && !spmv_value_indexed(A, l.x2 != nullptr)) SLA_TRY(csr_ensure_canon(const_cast<sla_csr *>(A)));   // (a CSR kernel after all)
    a.col = A->d_col;
    a.val = A->d_val;
    a.x = l.x;
    a.y = l.y;
    a.rb = A->d_rb;
    a.rbk = (const RP *)A->d_rbk;
    a.nrb = A->nrb;
    a.rows = (int32_t)A->rows;
    a.w = l.w;
    a.z = l.z;
    a.p1 = l.p1;
    a.p2 = l.p2;
    a.p3 = l.p3;
    a.p4 = l.p4;
    a.acc3 = 0.0;
    a.acc4 = 0.0;
    a.sc = l.sc;
    a.pres = l.pres;
    a.npres = l.npres;
    a.pres_stride = l.pres_stride;
    a.pa = l.pa;
    a.pb = l.pb;
    a.npa = l.npa;
    a.pa_stride = l.pa_stride;
    a.step_begin = l.step_begin;
    a.yinit = l.yinit;
    a.fs_ap = l.fs_ap;
    if (l.fs_ap && !(spmv_fuse_s_ok(A, true) && l.part == 0 && !l.x2 && !l.yinit)) return fail(SLA_ERR_INVALID, "launch_spmv: a fused input vector needs the plane-march or the wave-sliced gather kernel");
    const int grid = spmv_grid(A);
    // (the forms whose launcher is ONE kernel launched through SLA_KLAUNCH carry the profiling events themselves: the gather kernel of the
    // wave-sliced forms, the wavefront-private CSR kernel -- conditions as in the dispatch below)
    bool prof_ext = false;
    if constexpr (std::is_same<RP, int32_t>::value) {
        const bool lp = A->use_lpanel && c->lpanel && c->spmv_algo == 0 && !l.x2 && !l.yinit;
        const bool lf = lflat_on(A) && !l.x2 && !l.yinit;
        const bool wd = A->use_wdia && wd_on(A) && c->spmv_algo == 0 && !l.x2;
        if (!lp && !lf) {
            if (wd) prof_ext = !(wd_march_on(A) && l.part == 0) && !wd_lds_on(A);
            else if (!(A->use_vdict && c->vdict && c->spmv_algo == 0) && !l.x2 && !(c->spmv_algo != 1 && diag_on(A)))
                prof_ext = wave_plain(A) && !l.yinit;
        }
    }
    ProfScope prof(c, l.kernel_id, prof_ext);
    if (A->use_lpanel && c->lpanel && c->spmv_algo == 0 && !l.x2 && !l.yinit) return launch_spmv_lpanel(A, l.epi, a, grid);
    if constexpr (std::is_same<RP, int32_t>::value) {
        if (lflat_on(A) && !l.x2 && !l.yinit) return launch_spmv_lflat(A, l.epi, a, grid);
    }
    if constexpr (std::is_same<RP, int32_t>::value) {   // the value-indexed forms exist with 32-bit row pointers only
        if (A->use_wdia && wd_on(A) && c->spmv_algo == 0 && !l.x2) {
            const int32_t *sched = c->wd_tile != 0 ? A->d_wsched : nullptr;
            int32_t nblk_wd = A->nblk_wd;
            int g = grid;
            if (l.part == 1) { sched = A->d_ov_int; nblk_wd = A->ov_nint; g = overlap_grid(A, 1); }
            else if (l.part == 2) { sched = A->d_ov_bnd; nblk_wd = A->ov_nbnd; g = overlap_grid(A, 2); }
            const int stream_nt = vec_stream_nt(c, A->rows) ? 1 : 0;
            if (wd_march_on(A) && l.part == 0) return launch_wdia_march(A, l.epi, a, g, stream_nt | (c->wd_nt_store ? 2 : 0));
            if (wd_lds_on(A)) return launch_wdia_lds(A, l.epi, a, sched, nblk_wd, g, stream_nt | (c->wd_nt_store ? 2 : 0));
            return launch_spmv_wdia(A, l.epi, a, sched, nblk_wd, g, stream_nt);
        }
        if (A->use_vdict && c->vdict && c->spmv_algo == 0) return launch_spmv_vdict(A, l.epi, a, l.x2, l.b2, grid);
    }
    if (l.x2) return diag_on(A) ? launch_spmv_dual_diag(A, a, l.x2, l.b2, grid) : launch_spmv_dual(A, a, l.x2, l.b2, grid);
    if (c->spmv_algo != 1 && diag_on(A)) return launch_spmv_diag(A, l.epi, a, grid);
    if constexpr (std::is_same<RP, int32_t>::value) {
        if (wave_plain(A) && !l.yinit) return launch_spmv_wave(A, l.epi, a, grid);
    }
    return launch_spmv_stream(A, l.epi, a, grid);   // (also the one-lane-per-row baseline, SLA_SPMV_ALGO=scalar)
}

int launch_spmv(const sla_csr *A, const SpmvLaunch &l) {
    if (l.fs_ap && !spmv_fuse_s_ok(A, true)) return fail(SLA_ERR_INVALID, "launch_spmv: a fused input vector needs the plane-march or the wave-sliced gather kernel");
    if (tiles_on(A) && !lflat_on(A) && !l.x2 && (!l.yinit || l.tv1 >= 0)) return launch_spmv_tiles(A, l);
    return A->rp64 ? launch_spmv_rp<int64_t>(A, l) : launch_spmv_rp<int32_t>(A, l);
}

}  // namespace sla
