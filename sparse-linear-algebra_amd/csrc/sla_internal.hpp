// sla_internal.hpp -- shared declarations of libsla_hip.so (not part of the public C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <map>
#include <memory>
#include <new>
#include <atomic>
#include <chrono>
#include <future>
#include <string>
#include <algorithm>
#include <thread>
#include <vector>

#include "sla_hip.h"

namespace sla {

// ---------------------------------------------------------------------------------------------
// geometry constants
// ---------------------------------------------------------------------------------------------
constexpr int kBlock = 256;          // threads per workgroup (4 wavefronts of 64)
constexpr int kNnzPerRowBlock = 1024; // products staged in LDS per row block (8 KiB)
constexpr int kMaxRowsPerRowBlock = 256;
constexpr int kXWin = 768;           // doubles of x staged in LDS per row block by spmv_xwin_kernel (6 KiB)
constexpr int kXWinHalo = 256;       // the window starts kXWinHalo columns left of the block's first diagonal column
#if !defined(SLA_WD_OCC)
#define SLA_WD_OCC 6
#endif
#if !defined(SLA_WD_STAGES)
#define SLA_WD_STAGES 1
#endif
constexpr int kWdBlocksPerCu = SLA_WD_OCC;      // resident workgroups per CU of spmv_wdia_kernel (its persistent grid = that x CUs)
// ... for the variable-coefficient variant (per-row value blocks): 4 since late round 3 -- at 5 the K1 / K3 / CGS instantiations spilled
// 8-22 VGPRs into scratch to fit 96 and the four-sum one (108 VGPRs, compiled for 4) ran a 5-per-CU grid in two rounds; at 4 nothing
// spills: 2 M-row banded problem 12 490 -> 12 830 (four-sum fix) -> 13 100 it/s same-box, K1 24.2 -> 23.2 us
#if !defined(SLA_WD_OCC_VV)
#define SLA_WD_OCC_VV 4
#endif
#ifndef SLA_WD_OCC_VV4
#define SLA_WD_OCC_VV4 4
#endif
constexpr int kWdBlocksPerCuVV4 = SLA_WD_OCC_VV4;           // (its four-sum instantiation)
constexpr int kWdBlocksPerCuVV = SLA_WD_OCC_VV;
constexpr int kWdMaxSliceRecords = 40;          // wave-sliced forms: more diagonals per 128-row slice than this and the older kernels are used
                                                // (27-point stencil, 128^3: wdia 25.5 us, vdict 42 us, diagdict 167 us)
constexpr int kWdGatherStages = SLA_WD_STAGES;  // 2: the gathers of the next slice are issued before the current one is folded
// LDS-window variant (spmv_wdia_lds_kernel): the diagonal offsets of the matrix are clustered into <= kWdWinMax windows; a
// workgroup stages x[step base + omin_k, ... + span_k + 512] of every window in LDS once per 512-row step
constexpr int kWdWinMax = 8;
constexpr int kWdWinMerge = 512;     // offsets closer than this share a window (a separate window costs 512 more elements)
constexpr int kWdWinMaxPairs = 1536; // 16-byte pairs per LDS buffer (24 KiB; two buffers)
constexpr int kWdWinLoads = kWdWinMaxPairs / 256;   // staging loads per lane and step, at most
struct WdWin {
    int32_t n = 0, pairs = 0;        // windows; 16-byte pairs of one LDS buffer
    int32_t omin[kWdWinMax] = {};    // first offset of window k (even)
    int32_t pb[kWdWinMax + 1] = {};  // its first pair in the buffer; pb[n] = pairs
};
struct WdUni {                       // uniform records: the matrix's <= 8 (offset, value) pairs in table order
    int32_t n = 0;
    int32_t lpos[8] = {};            // element of the staged buffer that x[step base + offset] lands on
    int32_t lpos0 = -1;              // ... that x[step base] itself lands on (-1: no window covers offset 0)
    double val[8] = {};
};
// plane-march variant (spmv_wdia_march_kernel): a 3-D stencil whose far offsets are one pair -D and one pair +D.  A workgroup owns
// an in-plane tile of 512 rows and walks a run of planes; per step it stages ONE window (the in-plane window of the plane ahead),
// the planes behind / ahead of a step are the buffers staged one step earlier / later.
struct WdMarch {
    int32_t D = 0;                   // plane stride = the far offset (even)
    int32_t T = 0;                   // 512-row tiles per plane (the last one partial)
    int32_t planes = 0;              // ceil(rows / D)
    int32_t PS = 0, S = 0;           // planes per run, runs per tile
    int32_t ntasks = 0;              // T * S, tile-major
    int32_t omin = 0;                // first offset of the in-plane window (even, <= 0)
    int32_t pairs = 0;               // its 16-byte pairs (<= 512)
};
constexpr int kVdRows = 256;         // rows per block of spmv_vdict_kernel (one lane per row)
constexpr int kLpW = 16384;          // columns of x one workgroup of spmv_lpanel_kernel keeps in LDS (128 KiB)
constexpr int kLpBlock = 1024;       // its workgroup: 16 wavefronts, one per CU (LDS-bound occupancy)
#ifndef SLA_LP_ROWS
#define SLA_LP_ROWS 2
#endif
constexpr int kLpRowsInFlight = SLA_LP_ROWS;   // (row, panel) segments a wavefront of spmv_lpanel_kernel keeps in flight (2: 0.900 ms, 3: 0.910, 4: 0.926)
constexpr int kLpMinSeg = 16;        // mean entries per (row, panel) segment below which the stream kernel wins (tiny segments waste sectors)
constexpr int kVdMaxRowNnz = 31;     // longest row the value-indexed kernel takes (256 rows x 31 B of codes fit its LDS stage)
constexpr int kWvMaxRow = 128;       // longest row spmv_wave_kernel takes (one lane folds a row from the wavefront's LDS stage)
constexpr int kRowptrPad = 192;      // entries (= nnz) behind the rows + 1 row pointers of the 32-bit array: 128-row blocks read past the last row unclamped
constexpr int kWaveRowMax = 16384;   // rows of 1025..16384 entries: one wavefront each, 4 per row block; longer: whole workgroup
#ifndef SLA_TILE_ROWS
#define SLA_TILE_ROWS 4896
#endif
#ifndef SLA_TILE_OCC
#define SLA_TILE_OCC 1
#endif
#ifndef SLA_TILE_WAVES
#define SLA_TILE_WAVES 4
#endif
constexpr int kTileWaves = SLA_TILE_WAVES;  // wavefronts of its workgroup that walk slices (the others only join the reductions): 4 x 4896 rows or e.g. 2 x 9792
constexpr int kTileRows = SLA_TILE_ROWS;  // rows per slice of spmv_tile_kernel: one wavefront's row sums in LDS (38 KiB; 4 x 4896 x 8 B + the pacing tables = all of the CU's 160 KiB)
constexpr int kTileBlocksPerCu = SLA_TILE_OCC;  // its resident workgroups per CU (1 x 128 KiB of LDS, 4 wavefronts with 12 x 64 gathers in flight each: few, fat
                                                // wavefronts drift apart less and keep more misses in flight than 16 thin ones -- 2.26 -> 1.98 ms at 10 M rows)
#ifndef SLA_CT_ROWS
#define SLA_CT_ROWS 19584
#endif
constexpr int kCtRows = SLA_CT_ROWS;  // rows per slice of spmv_ctile_kernel (round 5): ONE array of row sums shared by the workgroup's four wavefronts (153 KiB of the CU's 160)
constexpr int kCtWaves = 4;           // its wavefronts (= kBlock / 64): the layout deals a tile's 64-entry groups to them
constexpr int kMaxParts = 2048;      // partial-sum slots per reduction (256 CUs x 8)
constexpr int kVecGridMax = 1024;    // grid cap of the streaming BLAS-1 kernels
constexpr int kSpmvGridMax = 2048;   // persistent grid cap of the SpMV kernels
constexpr int kArnGridMax = 1024;    // grid cap (= partials per column) of the Arnoldi kernels
constexpr size_t kArraySlack = 64;              // zeroed bytes behind every lowered matrix array (clamped whole-block loads of the pipelined stream kernel)
constexpr int kTriBlockRows = 16384;            // rows of one block of the block-local triangular solve (128 KiB of LDS)
constexpr size_t kGuardBytes = 256;             // readable slack on both sides of every buffer an SpMV gathers from
constexpr size_t kHaloBytes = (size_t)4 << 20;  // the same for vectors of sharded contexts: room for the neighbours' halo planes
constexpr int kMaxKrylov = 64;       // max Arnoldi basis columns handled by the fused GS kernels

// Device-resident scalars of one solver.  Written only by block 0 / thread 0 of a kernel and
// read by LATER kernels on the same stream (kernel boundaries order the accesses).
struct SolverScalars {
    double rho2[2];  // r . r0hat (CGNE: r . r), double-buffered by step parity: a step reads
                     // rho2[par] everywhere and its last kernel writes rho2[par ^ 1]
    double alpha, omega, beta;
    double resnorm;  // last true residual norm evaluated on the device
    double tol;      // max tolAbs (tolRel * r0norm)
    double r0norm;
    double hnorm;    // Arnoldi: h_{i+1,i}
    int32_t done;    // 1 once resnorm <= tol (or Arnoldi breakdown); later kernels return at once
    int32_t iters;   // steps started while not done
    int32_t flags;
    int32_t kdone;   // Arnoldi: number of H columns produced
    double *hist;    // linSolve0 with a residual trace: hist[j - 1] = true residual norm of the iterate after j steps (null: no trace)
    int32_t hist_cap;
};

// epilogues fused into the SpMV kernels
enum Epi : int {
    EPI_NONE = 0,     // y = A x
    EPI_DOT = 1,      // y = A x ; p1 += y . w                         (K1 / C1)
    EPI_DOT2 = 2,     // y = A x ; p1 += y . w ; p2 += y . y           (K3)
    EPI_RES = 3,      // p1 += (A x - b)^2, y not stored              (true residual)
    EPI_AXPY_DOT = 4, // z = z - alpha * (A x) ; p1 += z . w (w==null: z . z)   (CGS C3, CGNE N1)
    EPI_XPBY_NRM = 5, // z = (A x) + beta * z ; p1 += z . z            (CGNE N3)
    EPI_SUB = 6,      // y = b - A x                                   (init: r0 = b ^-^ A x0)
    EPI_DOT4 = 7      // EPI_DOT2 and p3 += y . z ; p4 += w . z (z read-only)   (K3 of the fused K4+K5 flow).  A separate instantiation, not a
                      // run-time option of EPI_DOT2: carrying the optional operand tripled the scratch of the variable-coefficient K3 (16 -> 44
                      // B per lane) and cost the reference's split flow 4 % on the 2 M-row banded problem even when unused (measured); that
                      // kernel's EPI_DOT4 instantiation is compiled for one workgroup per CU less instead
};

template <typename RP>
struct SpmvArgs {
    const RP *rowptr;        // local rows + 1, rowptr[0] == 0
    const int32_t *col;      // global column indices
    const double *val;
    const double *x;         // gather base (full-length x)
    double *y;               // local rows (may be null for EPI_RES)
    const int32_t *rb;       // row-block starts, nrb + 1 entries
    const RP *rbk;           // rowptr[rb[b]]: entry offset of each row block, nrb + 1 entries
    int32_t nrb;
    int32_t rows;            // local rows
    const double *w;         // epilogue operand (w / b)
    double *z;               // epilogue in-out operand
    double *p1, *p2;         // partial outputs, one slot per block
    double *p3, *p4;         // EPI_DOT4 only: partials of y . z and w . z (z read-only there)
    mutable double acc3, acc4;   // their per-thread accumulators (a kernel parameter is a private copy: the epilogue adds here, so the
                                 // kernels keep their two-accumulator signatures)
    SolverScalars *sc;       // may be null (stand-alone SpMV)
    const double *pres;      // prologue: partials of the previous true-residual evaluation (or null)
    int32_t npres, pres_stride;
    const double *pa, *pb;   // prologue partials feeding alpha / beta for EPI_AXPY_DOT / EPI_XPBY_NRM
    int32_t npa, pa_stride;
    int32_t step_begin;      // bit 0: this launch opens a solver step (iters++); bit 1: step parity
    const double *yinit;     // column-panel passes: running row sums of the previous panels (or null)
    const double *fs_ap;     // plane-march kernel, K2 folded into K3: the gathered vector is x - alpha fs_ap, alpha from pa (or null)
};

}  // namespace sla

namespace sla {
// Exchange plan of the row-sharded SpMV: this rank's rows only reference columns in [cmin, cmax], so it
// needs from peer q the part of that window q owns, and sends q the part of q's window it owns.
struct XPlan {
    bool use_window = false;                 // false: plain all-gather of the padded shards
    std::vector<int64_t> send_begin, send_len;  // per peer, global index / length
    std::vector<int64_t> recv_begin, recv_len;
};
// All-gather-mode matrices on the tile form (BASELINE config 3a sharded): the x all-gather overlapped with the SpMV.
// The gather is issued as G grouped point-to-point exchanges (each ONE ncclGroupStart/End launch that moves a piece of EVERY
// shard to every peer: all xGMI links busy at once, each carrying 1/G of a shard), and the tile launch is cut into PASSES over
// column panels: pass 0 walks the panels this rank already owns, pass k the panels whose columns have arrived with the first
// `need` groups.  The row sums are carried from pass to pass (yinit), so a row is still ONE left fold -- over the panels in the
// plan's visiting order, ascending columns inside a panel.
//   order 0 "arrival": groups = column chunks of every shard; panels visited by (groups needed, panel): own panels first.  Every
//           pass but the first waits for ONE more group, so compute follows the exchange at 1/G granularity.
//   order 1 "ascending": groups = whole shards in source-rank order (rank q's shard goes to everybody in group q); panels visited
//           in ascending order: the reference's fold bit for bit (Common.hs:247-260), but rank r cannot start before shards
//           0 .. r-1 have landed and the groups use one sender's links at a time: the exchange is ~nranks x slower on a full mesh.
struct AgPiece { int src; int64_t b, e; };      // columns [b, e) of rank src's shard
struct AgPlan {
    int order = 0, G = 0, P = 0, shift = 0, nranks = 1, rank = 0;
    std::vector<std::vector<AgPiece>> groups;
    std::vector<int32_t> vis;                    // panel visiting order: a permutation of 0 .. P-1
    std::vector<int32_t> pass_ptr, pass_need;    // pass p = vis[pass_ptr[p] .. pass_ptr[p+1]), launched once pass_need[p] groups have completed
    int32_t *d_vis = nullptr;
    double *d_yrun = nullptr;                    // running row sums between passes
    std::vector<hipEvent_t> ev;                  // group g completed (recorded on the comm stream)
    bool sim = false;                            // single-rank rehearsal of rank `rank` of `nranks` (option ag_sim_ranks): passes without an exchange
};
// pure host planning (also exported as sla_plan_allgather_passes for the CPU tests and the oracle-side restatement of the fold order)
void plan_allgather_passes(int nranks, int rank, int64_t n, int shift, int groups, int order, AgPlan &plan);
// pure host planning (also exported as sla_plan_window_exchange for the CPU tests)
void plan_window_exchange(int nranks, int rank, int64_t n, const int64_t *windows /* [2*nranks] cmin,cmax (cmax<cmin: empty) */,
                          XPlan &plan);
}  // namespace sla

// ---------------------------------------------------------------------------------------------
// handle types of the C ABI
// ---------------------------------------------------------------------------------------------
struct sla_ctx {
    std::vector<sla_ctx *> kids;     // non-empty: the PARENT of a single-process multi-device group (sla_multi.cpp); nothing below is used then
    void *pool = nullptr;            // parent only: its persistent per-rank worker threads
    int device = 0, rank = 0, nranks = 1;
    hipStream_t stream = nullptr;
    void *comm = nullptr;            // ncclComm_t when nranks > 1 (or a 1-rank test comm)
    void *loop = nullptr;            // in-process loopback group (test backend: all ranks are threads on one GPU)
    double *d_parts = nullptr;       // 4 * kMaxParts doubles of scratch partials
    double *d_result = nullptr;      // small device scratch for scalar results / per-rank sums
    double *h_result = nullptr;      // pinned host mirror
    double *d_xfull = nullptr;       // all-gather landing buffer (nranks * shard) when sharded
    int64_t xfull_cap = 0;
    double *d_tfull = nullptr;       // full-length partial result of a sharded transpose SpMV (before the reduce-scatter)
    int64_t tfull_cap = 0;
    int spmv_algo = 0;               // 0 stream, 1 scalar (SLA_SPMV_ALGO)
    int xcd_remap = 1;               // SLA_XCD_REMAP
    int64_t device_coo_min = 1 << 20; // triple lists at least this long are sorted on the GPU (SLA_DEVICE_COO_MIN)
    int rb_nnz = 0;                  // 0: automatic row-block size (SLA_RB_NNZ overrides, <= 1024)
    int row_align = 0;               // > 1: row blocks end on multiples of this many rows (SLA_ROW_ALIGN; measured -1.5 % at 16)
    int step_graph = -1;             // replay solver steps as a captured HIP graph: -1 when the matrix has <= step_graph_max_rows rows, 0 never, 1 always (SLA_STEP_GRAPH)
    int64_t step_graph_max_rows = 2500000;
    int ag_groups = 4;               // all-gather-mode tile matrices on sharded contexts: exchange groups of the overlapped all-gather (SLA_AG_GROUPS; 0: plain ncclAllGather, then one launch)
    int ag_order = 0;                // ... 0 arrival order (own panels first, then by exchange group), 1 ascending panels with source-ordered groups (SLA_AG_ORDER)
    int ag_sim_ranks = 0, ag_sim_rank = 0;   // single-rank contexts: run the pass structure of rank ag_sim_rank of ag_sim_ranks without an exchange (cost of the split; SLA_AG_SIM_RANKS / _RANK)
    int overlap = 1;                 // sharded (#>): 1 interior rows run while the halo exchange is in flight (second stream), 0 same split launches with
                                     // the exchange serialised on the compute stream (A/B, bit-identical), -1 no split at all (SLA_OVERLAP)
    hipStream_t comm_stream = nullptr;   // created on first use
    hipEvent_t ev_x_ready = nullptr, ev_x_done = nullptr;
    int xcd8 = -1;                   // 1: workgroups are dealt round-robin over 8 XCDs (those with equal b % 8 share one; probed once: what the tile kernel's panel pacing relies on), 0: not so
    int tiles_device = 1;            // the tile form's re-ordering as a device sort (sla_tiles_build.hip): 1 from 2^20 entries on, 2 always, 0 host builder (SLA_TILES_DEVICE)
    int tiles = 1;                   // allow the row-slice x column-panel tile SpMV for irregular matrices with x > L2 (SLA_TILES=0: column-panel passes)
    std::future<void> xfer_warmup;   // the copy lanes of this device being built (sla_xfer.cpp)
    std::vector<std::future<void>> deferred;   // host buffers being released off the caller's thread (defer_release below): returning a GB to the system costs ~0.1 s
    int xfer = 1;                    // copies >= 24 MiB from / to pageable host memory: own pinned staging on xfer_lanes threads (0: plain hipMemcpy)
    int xfer_lanes = 4;
    int canon_lazy = 1;              // ... and only when something asks for them (export, transpose, a CSR kernel after the form was peeled off): 843 MB and 12-23 ms of
                                     // first-touch allocation less at 216^3
    int transpose_device = 1;        // transposeSM of a lowered matrix as a device sort (1: from 2^18 entries on, 2: always, 0: host)
    int canon_device = 1;            // value-indexed matrices: canonical col / val written on the device from the 1-byte codes instead of uploaded
    int tile_rowown = -1;            // round 6: CU-wide tiles with every row of a slice owned by ONE wavefront (reproducible row sums, the reference's left fold; a quarter of the
                                     // relaxed dealing's gather density): -1 = whenever tile_relaxed = 0 asks for the exact form, 0 = never (exact = wavefront-private slices), 1 = always
    int tile_relaxed = 0;            // the tile form of irregular matrices.  0 (DEFAULT since the end of round 6; ADVICE r05: reruns must be bit-identical, like the reference's pure
                                     // functions) = the exact forms: CU-wide slices with every row owned by one wavefront (tile_rowown; the reference's left fold bit for bit) and,
                                     // for rows of ~100 entries and more, the LDS-flat form (a fixed regrouping).  1 = OPT-IN: CU-wide slices with the row sums added by LDS atomics in
                                     // relaxed order (spmv_ctile_kernel, round 5) -- within nnz_i eps sum|a_ij x_j| of the reference's fold, NOT reproducible bit for bit, announced by
                                     // sla_csr_get_props().fold and SLA_FLAG_RELAXED_ORDER; config 3a (#>) 1.35 ms against 1.62 exact, power-law rows 170 against 230 us, rows of
                                     // 100 - 500 entries 2 - 8 % (profiles/r06_tile_default_ab.txt)
    int onchip = 1;                  // sla_solver_step on constant-coefficient stencil / banded matrices that fit the chip's registers + LDS: the whole step loop as ONE persistent
                                     // launch (sla_onchip.hip; SLA_ONCHIP: 0 never, 1 when the plan says it fits, 2 the same and a plan failure is an error: tests)
    int onchip_sync = 0;             // ... its grid-wide synchronisation: 0 = XCD-hierarchical arrival counters (default), 1 = one epoch word per workgroup polled by everybody
                                     // (one hop on paper; measured SLOWER: 17.4 against 14.6 us per BiCGSTAB step at 1 M rows -- 256 x 256 scoped polling loads per round)
    int onchip_fault = 0;            // test hook: 1 = the last workgroup of an on-chip launch leaves at once (the others' barrier times out after 2 s: a lost CU, rehearsed)
    int arn_orth = 1;                // one Arnoldi step's Gram-Schmidt (dots | update | normalisation) as ONE persistent launch with w in registers and six basis columns of
                                     // the block kept on chip between the passes (sla_arnoldi_orth.hip; single-rank, <= 8192 rows per CU): 0 = the three launches
    int arn_orth_fault = 0;          // ... test hook: the last workgroup of a fused step leaves at once (the timeout / fallback path, tests/test_gpu_arnoldi_orth.py)
    int arn_orth_state = 0;          // ... 0 not asked yet, 1 the kernel is resident with one workgroup per CU, -1 it is not (launch flow)
    long arn_orth_launches = 0;      // (read-only)
    long arn_orth_fallbacks = 0;     // (read-only) Arnoldi runs repeated on the launch flow after a fused step reported SLA_FLAG_SYNC_TIMEOUT
    long onchip_fallbacks = 0;       // (read-only) linSolve0 calls re-run on the launch flow after an on-chip launch reported SLA_FLAG_SYNC_TIMEOUT
    int onchip_grid = 0;             // ... its workgroups at most (0: one per CU; tests use small grids)
    int onchip_rows = 0;             // ... rows per workgroup at most (0: what the registers hold: 12 x 512; tests force short blocks)
    int onchip_bricks = 1;           // ... 1: bricks where consecutive rows do not fit (3-D stencils), 2: bricks wherever the stencil allows them, 0: never
    long onchip_launches = 0;        // (read-only) persistent launches so far (sla_ctx_get_option "onchip_launches")
    double onchip_plan_ms = 0.0;     // (read-only) what the last on-chip plan cost to build
    std::string onchip_note;         // (read-only) the last plan's shape or the reason there is none ("onchip_plan")
    int tri_block_rows = 16384;      // tri_syncfree = 2: rows per block (<= kTriBlockRows: the block's x sits in LDS; small values are for the tests)
    int tri_syncfree = 3;            // triLowerSolve / triUpperSolve: 0 one launch per dependency level, 1 one persistent launch whose rows poll x in memory, 2 the block-local persistent launch (sla_tri.hip), 3 pick 2 or 0 by the schedule's shape (sla_precond.cpp)
    int tri_grid = 0;                // ... its workgroups (0: 256 for the row-polling kernel, as many as are co-resident for the block-local one)
    std::string tri_plan_note;       // ... and the shape of its plan (sla_ctx_get_option "tri_plan")
    int tri_mode_used = 0;           // what the last triangular solve ran as (sla_ctx_get_option "tri_mode_used")
    long tri_fallbacks = 0;          // solves that left the persistent kernel for the level schedule (sla_ctx_get_option "tri_fallbacks")
    int tri_spin = 200000;           // ... polls without progress after which a lane gives up and the host runs the level schedule instead
    int tile_depth = 0;              // CU-wide tile kernel, 64-entry groups per chunk: 0 by the tiles' density (20 below 4 entries per x line, else 12), 1 always 12, 2 always 20 (A/B)
    int tile_prefetch = 0;           // x-panel prefetch distance of spmv_tile_kernel in visit steps (0: demand misses only; measured: never a gain)
    int tile_poll = 1;               // 1: pacing slots polled one step ahead, 0: dependent poll in front of every tile (rounds 2-3)
    int tile_slack = 3;              // panel pacing: a wavefront starts panel step q once its XCD has finished step q - slack (SLA_TILE_SLACK, 0: no pacing)
    int tile_shift = 0;              // log2 of its panel width in columns (SLA_TILE_SHIFT; 0: 17 from 6 M columns on, 16 below -- at 10 M rows slack 3 / shift 17: 1.98 ms, slack 4: 2.18, slack 2: 2.1, shift 16: 2.03-2.28, shift 18: +15 %)
    int panels = 1;                  // allow the column-panel SpMV for irregular matrices (SLA_PANELS=0 disables)
    int64_t panel_cols = 384 * 1024; // panel width in columns (SLA_PANEL_COLS): 3 MiB of x per pass
    int diag = 1;                    // allow the dictionary-compressed-index SpMV kernel (SLA_DIAG=0 disables)
    int diag_lazy = 1;               // skip building its 1-byte codes for matrices that take the constant-coefficient wave-sliced form (SLA_DIAG_LAZY=0: always build)
    int vec_policy = 0x2bff;         // per-stream cache policy of K2 and the K4+K5 sweep when vec_nt applies (sla_vec_kernels.hip: ldpol; default: every BiCGSTAB stream but the p store, and CGS's q and u stores, past the caches)
    int vec_nt = -1;                 // non-temporal loads in the BiCGSTAB vector kernels: -1 when the vectors overflow the memory-side cache, 0 / 1 (SLA_VEC_NT)
    int64_t mall_bytes = 256ll << 20; // capacity of the memory-side cache (MI355X: 256 MiB)
    int wdia_vv = 1;                 // allow the variable-coefficient wave-sliced form (SLA_WDIA_VV=0 disables)
    int wd_grid_max_vv = sla::kWdBlocksPerCuVV * 256;
    int bicg_fuse23 = 1;             // ... and K2 folded into K3 on the wave-sliced forms (plane march: s = r - alpha Ap built in the staged windows; gather kernel up to 4 M rows: r and Ap gathered), never stored (SLA_BICG_FUSE23; 2: the gather kernel at any size)
    int bicg_fuse45 = 1;             // single-rank BiCGSTAB: K4 + K5 in one sweep, rho through K3's extra sums (SLA_BICG_FUSE45)
    int wd_lds = 1;                  // stencils with <= 8 (offset, value) pairs: uniform records + x windows staged in LDS (SLA_WD_LDS; 2: at any size)
    int wd_nt_store = 0;             // its y / z stores past the caches (SLA_WD_NT_STORE)
    int wd_march = 1;                // 3-D stencils (whole-matrix / whole-slab launches): plane-march walk of the LDS-window form (SLA_WD_MARCH; 2: at any size, 0: never)
    int wd_march_occ = 4;            // its workgroups per CU (32 KiB of LDS each)
    int wd_lds_occ = 0;              // its workgroups per CU (SLA_WD_LDS_OCC; 0: as many as the LDS holds, at most 4)
    int wd_tile = -1;                // plane tiling of the wave-sliced walk: -1 automatic, 0 off, > 0 steps per tile (SLA_WD_TILE)
    int wdia = 1;                    // allow the wave-sliced SpMV kernel (SLA_WDIA=0 disables)
    int vdict = 1;                   // allow the value-indexed SpMV kernel (SLA_VDICT=0 disables)
    int n_cu = 256;                  // compute units of the device (persistent grids)
    int bicg_ghost = 1;              // sharded BiCGSTAB keeps ghost rows: 3 grouped exchanges per step instead of 5 (SLA_BICG_GHOST=0: plain flow)
    int lp_attr = 0;                 // spmv_lpanel_kernel<i32 / i64> had its dynamic-LDS limit raised on this device (bits 0 / 1)
    int lp_min_seg = sla::kLpMinSeg, lp_tasks = 32, lp_cfg = -1, lp_rowcost = 256;   // its tuning knobs (SLA_LP_MINSEG / _TASKS / _CFG / _ROWCOST)
    int force_rp64 = 0;              // test hook: 64-bit row pointers at any size (SLA_FORCE_RP64)
    int lp_copy = 1;                 // LDS-panel form: stream a panel-major second copy of col / val (SLA_LP_COPY=0: the row-major arrays)
    int lpanel = 1;                  // allow the LDS-panel SpMV kernel for matrices with dense rows (SLA_LPANEL=0 disables)
    int lflat = 1;                   // allow its flat variant for segments of 1.5 .. 16 entries (SLA_LFLAT: 0 off, 2: also above the LDS-panel threshold)
    int lf_min_seg10 = 15;           // ... from this mean (panel, row) segment length on, in tenths (SLA_LF_MIN_SEG10)
    int xwin = 1;                    // LDS x windows: 1 = where they pay (the pair-code kernel), 2 = also the dictionary-code kernels, 0 = nowhere (SLA_XWIN); the plain CSR-stream kernel takes its window form only with stream_wide = 0
    int stream_wide = 1;             // spmv_stream_kernel: pairs of entries per load (8-byte col / 16-byte val loads) instead of one (SLA_STREAM_WIDE=0)
    int stream_wave = 1;             // plain CSR (#>): wavefront-private 128-row blocks with row-pair stores (sla_spmv_wave.hip) instead of spmv_stream_kernel when no row
                                     // exceeds kWvMaxRow entries (SLA_STREAM_WAVE: 0 off, 1 on = 4 entry pairs per lane and chunk, 4 / 7 force that chunk size)
                                     // (SLA_STREAM_PIPE=1; OFF by default: measured 7-12 % SLOWER than the one-deep prefetch)
    int wave_flat = 0;               // ... 1: the prefetching instantiation as a flat chunk walk over two register sets (no `cur = nxt` drain per chunk, counted waits, unconditional
                                     // store; round 6): bit-identical, measured 3-6 % SLOWER (profiles/r06_ab_wave_flat.txt) -- like round 3's deeper ring: the waits were not the bound
    int wave_sync = 0;               // ... EXPERIMENT (round 6): a raw workgroup barrier per run so that the four wavefronts' requests for shared x lines meet in the L1
    int wave_over = 0;               // ... grid of the plain-CSR wave kernel: 0 = automatic (short launches are oversubscribed, see wave_grid), k >= 1 = k resident rounds
    int wave_cc = 0;                 // ... EXPERIMENT (round 6): continuous chunks over runs of wave_cc consecutive blocks (one partial chunk per run instead of per block); 0 off
    int wave_run = 1;                // ... consecutive 128-row blocks a wavefront of spmv_wave_kernel walks before it jumps (round 6; 1 = the round-4 walk)
    int dual_spmv = 1;               // linSolve0: fuse the true-residual SpMV into the next K1 (SLA_DUAL_SPMV=0 disables)
    int x_exchange = 0;              // 0 auto (window exchange when it pays), 1 always all-gather, 2 always window (SLA_X_EXCHANGE=allgather|window)
    int halo_inplace = 1;            // window exchange straight into the slack around the vector (SLA_HALO_INPLACE=0: via the landing buffer)
    size_t vec_guard = sla::kGuardBytes;  // slack on both sides of every pooled vector (kHaloBytes once the context is sharded)
    bool collectives = false;        // nranks > 1, or SLA_FORCE_COLLECTIVES=1 on a 1-rank communicator (test hook)
    int spmv_grid_max = sla::kSpmvGridMax;
    int wd_grid_max = sla::kWdBlocksPerCu * 256;  // persistent grid of spmv_wdia_kernel: kWdBlocksPerCu x CUs (a multiple of 8)
    // Device-vector pool: a pure `linSolve0` call allocates ~10 vectors and frees them again; hipMalloc /
    // hipFree of 80 MB blocks cost milliseconds each, so freed vector buffers are kept (by exact size) and
    // handed out again.  Reuse is ordered by the context stream, so no synchronisation is needed.
    std::multimap<size_t, void *> vec_pool;
    size_t vec_pool_bytes = 0;
    // profiling
    int opt_gen = 0;                 // bumped by every sla_ctx_set_option: a solver's captured step graph is only replayed under the options it was captured with
    int prof_kernel = -2, prof_max = 0;   // -2: not recording, SLA_KERNEL_ALL (-1): every kernel id, else one id
    std::vector<int> prof_ids;            // kernel id of each recorded launch
    std::vector<float> prof_ms;           // durations of the last recording (filled by sla_prof_stop)
    std::vector<hipEvent_t> prof_ev;
    int prof_count = 0;
    int prof_pending = -1;           // index of the event pair an `ext` ProfScope has handed to the next SLA_KLAUNCH (-1: none)
};

struct sla_vec {
    std::vector<sla_vec *> kids;     // non-empty: a bundle of per-rank vectors of a multi-device context
    sla_ctx *ctx = nullptr;
    int64_t n = 0;        // global dimension
    int64_t n_local = 0;  // this rank's entries
    int64_t shard = 0;    // allocation length (= ceil(n / nranks), zero padded)
    int64_t begin = 0;    // global index of the first local entry
    double *d = nullptr;
};

// level schedule of a triangular solve with one triangle of a matrix (sla_tri_solve)
struct sla_tri_plan {
    int64_t nlevels = 0, widest = 0;
    std::vector<int64_t> level_ptr;   // rows of level l: order[level_ptr[l] .. level_ptr[l + 1])
    int32_t *d_order = nullptr;       // rows sorted by (level, row)
    // the triangle re-stored in that order (a lane's row is then found without the order -> rowptr indirection and
    // neighbouring lanes read neighbouring entries): strictly-triangular entries of schedule slot t are
    // tcol/tval[tptr[t] .. tptr[t + 1]), its diagonal entry tdiag[t]
    int64_t *d_tptr = nullptr;
    int32_t *d_tcol = nullptr;
    double *d_tval = nullptr, *d_tdiag = nullptr;
    hipGraphExec_t graph = nullptr;   // the level launches captured for (gb, gx)
    const double *gb = nullptr;
    double *gx = nullptr;
    // the block-local form (option tri_syncfree = 2, tri_blocks_kernel in sla_tri.hip): the sweep order cut into blocks of brows consecutive
    // rows, one workgroup per block with the block's x in LDS; blocks stored in the order the workgroups take them ((block level, block)),
    // rows of a block by (level inside the block, position); slot s: row bl_row[s], diagonal bl_diag[s], strictly-triangular entries
    // bl_col / bl_val[bl_ptr[s] .. bl_ptr[s + 1]) in ascending column order, bl_col >= 0: x index of another block's row, < 0: ~(LDS cell =
    // slot inside the block) of a row of the same block
    int64_t nb = 0;
    int32_t brows = 0;
    bool bricks = false;              // the block order is the stencil-brick order, not the sweep's
    double bl_cross = 0.0;            // share of the triangle's entries that read a row of another block (polled in memory)
    bool bl_rejected = false;         // the automatic mode measured bl_cross > 1/4 once: the block arrays were freed, the level schedule stays
    int64_t *d_bl_slots = nullptr;    // [2 * p]: first slot of the p-th block taken, [2 * p + 1]: its first sweep position; [2 * nb]: n
    int32_t *d_bl_row = nullptr, *d_bl_col = nullptr;
    int64_t *d_bl_ptr = nullptr;
    double *d_bl_val = nullptr, *d_bl_diag = nullptr;
};

// On-chip solver plan (sla_onchip.hip, round 6): constant-coefficient stencil / banded matrices of <= 8 (offset, value) pairs whose rows are
// dealt to one 512-thread workgroup per CU -- consecutive rows (2-D / banded) or bricks (3-D) -- so that a whole BiCGSTAB / CGS state lives in
// registers + LDS and sla_solver_step(k) is ONE persistent launch.  Built lazily by the first step on a matrix, kept with the matrix.
namespace sla {
constexpr int kOcThreads = 512;      // 8 wavefronts of 256 VGPRs, one workgroup per CU
constexpr int kOcMaxPairs = 8;
struct OcPlan {
    bool ok = false;
    std::string note;                // the plan's shape, or why there is none (sla_ctx_get_option "onchip_plan")
    int G = 0, L = 0, rpt = 0, hpt = 0, np = 0, mode = 0;   // workgroups, local cells, own rows / halo cells per thread (instantiation classes), pairs, 0 rows / 1 bricks
    int bx = 0, by = 0, bz = 0;
    int64_t own_max = 0, halo_max = 0, nbound = 0;
    int loff[kOcMaxPairs] = {};      // local cell offset of pair k
    double val[kOcMaxPairs] = {};
    uint32_t *d_own_cm = nullptr;    // [G][rpt * 512]: local cell | pair mask << 16 | boundary << 24 | valid << 25
    int32_t *d_own_row = nullptr;    // [G][rpt * 512]: global row
    uint32_t *d_halo_cell = nullptr; // [G][hpt * 512]: local cell (slots without one: the dummy cell L)
    int32_t *d_halo_row = nullptr;   // [G][hpt * 512]: the row the cell mirrors
    int32_t *d_halo_src = nullptr;   // [G][hpt * 512]: ... and the slot its owner publishes it in
    double *d_pubA = nullptr, *d_pubS = nullptr;   // boundary rows of the two SpMV results of a step, by the owner's slot
    double *d_parts = nullptr;       // [4][G] partial sums
    unsigned *d_bar = nullptr;       // barrier words (zeroed in front of every launch)
    size_t lds_bytes = 0;
    double build_ms = 0.0;           // what planning cost (host threads + uploads); sla_ctx_get_option "onchip_plan_ms"
    int kstate[4] = {1, 0, 0, 0};    // per kernel (bicgstabStep -- asked by the plan builder --, linSolve0 BICGSTAB_, cgsStep, linSolve0 CGS_): 0 not asked yet, 1 resident, -1 declined
    std::string knote[4];            // ... and why
};
void onchip_plan_free(OcPlan *p);
}  // namespace sla

struct sla_csr {
    std::vector<sla_csr *> kids;     // non-empty: a bundle of per-rank row blocks of a multi-device context
    sla_ctx *ctx = nullptr;
    int64_t m = 0, n = 0;            // global dims
    int64_t row_begin = 0, rows = 0; // local row block
    int64_t nnz = 0;                 // local nnz
    bool rp64 = false;
    void *d_rowptr = nullptr;        // int32 or int64
    int32_t *d_col = nullptr;
    double *d_val = nullptr;
    int32_t *d_rb = nullptr;
    void *d_rbk = nullptr;           // int32 or int64, like d_rowptr
    int32_t *d_rbw = nullptr;        // first column of each row block's LDS x window
    uint8_t *d_code = nullptr;       // dictionary-compressed column indices: code[k] indexes d_dict (col - row offsets)
    int32_t *d_dict = nullptr;       // 256 sorted diagonal offsets (unused slots repeat the last one)
    bool use_diag = false;           // <= 256 distinct (col - row) values: spmv_diag_kernel streams 9 B per entry
    int ndiag = 0;
    uint8_t *d_vcode = nullptr;      // value-indexed form: code[k] indexes the (offset, value) pair table (padded to dwords)
    int32_t *d_vdoff = nullptr;      // 256 pair offsets (col - row), sorted by (offset, value bits)
    double *d_vdval = nullptr;       // 256 pair values
    bool canon_lazy = false;         // value-indexed matrix whose canonical col / val arrays have not been materialised (csr_ensure_canon writes them from the codes)
    bool use_vdict = false;          // <= 256 distinct (col - row, value) pairs, rows <= kVdMaxRowNnz: spmv_vdict_kernel streams 1 B per entry
    int npairs = 0;
    int32_t nblk_vd = 0;             // ceil(rows / kVdRows)
    int32_t *d_wptr = nullptr;       // wave-sliced form: first record of each 128-row slice, nslices + 1 entries
    unsigned long long *d_wme = nullptr;    // per record: lanes whose EVEN row (2 lane) of the slice holds the entry ...
    unsigned long long *d_wmo = nullptr;    // ... lanes whose ODD row (2 lane + 1) holds it ...
    double *d_wval = nullptr;               // ... its value ...
    int32_t *d_woff = nullptr;              // ... and its diagonal offset (col - row); all padded by 8 records
    unsigned long long *d_wum = nullptr;    // LDS-window variant (<= 8 pairs): per slice 8 even-row masks then 8 odd-row masks, in table order
    bool wd_lds = false;
    sla::WdWin wd_win;
    sla::WdUni wd_uni;
    bool wd_march = false;                  // plane-march variant of it (spmv_wdia_march_kernel): geometry, pair positions, masks in march order
    sla::WdMarch wd_mg;
    sla::WdUni wd_muni;
    unsigned long long *d_wum_m = nullptr;  // per (tile, plane, wavefront): 8 even-row masks then 8 odd-row masks
    int32_t wd_col_lo = 0, wd_col_hi = -1;  // smallest / largest column these rows reference: what the staged windows may read
    // LDS-panel form (rows with many entries per 16384-column panel): per (panel, row) entry ranges into col / val
    void *d_lpp = nullptr;           // (P + 1) x rows, RP-typed, panel-major: pp[p][i] = first entry of row i with col >= p * lp_W
    int32_t lp_col_lo = 0, lp_col_hi = -1;   // smallest / largest column any of these rows references
    int32_t *d_lpt = nullptr;        // lp_G + 1 task boundaries: workgroup g runs tasks [lpt[g], lpt[g+1]) (equal entries each)
    int32_t lp_G = 0;
    uint16_t *d_lpcol = nullptr;     // panel-major second copy of the entries (segments of a panel contiguous, rows ascending; columns as 16-bit offsets into the panel): then
    double *d_lpval = nullptr;       //   d_lpp holds P x rows + 1 segment starts into it instead of the (P + 1) x rows table
    double *d_lpy = nullptr;         // P x rows partial sums, summed in ascending panel order by lpanel_finish_kernel
    bool use_lpanel = false;
    // flat LDS-panel form (sla_spmv_lflat.hip: one lane per (panel, row) segment; mean segment of 1.5 .. 16 entries): panel-major copy + segment starts;
    // shares lp_P / lp_W / lp_C / lp_chunk / lp_G, d_lpt and d_lpy with the LDS-panel form (the two exclude each other)
    bool use_lflat = false;
    uint32_t *d_lfq = nullptr;       // P x rows + 1 segment starts into the copy
    uint16_t *d_lfcol = nullptr;     // columns as 16-bit offsets into the panel
    double *d_lfval = nullptr;
    int32_t lp_cfg = 0;              // lane-group shape of spmv_lpanel_kernel (0: 64 lanes per segment ... 3: 8 lanes)
    int32_t lp_P = 0, lp_W = 0, lp_C = 0, lp_chunk = 0;   // panels, columns per panel, row chunks per panel, rows per chunk
    bool use_wdia = false;
    int32_t nslices = 0;
    double *d_wvblk = nullptr;       // variable-coefficient variant: 128 values per record, laid out like the slice's rows
    bool wd_vv = false;
    int32_t *d_wsched = nullptr;     // visiting order of the 512-row steps (null: ascending); see csr_upload
    int32_t nblk_wd = 0;             // ceil(nslices / 4): workgroup steps of spmv_wdia_kernel
    int64_t nwent = 0;
    bool use_xwin = false;           // enough entries fall inside the windows for spmv_xwin_kernel to pay
    double xwin_fraction = 0.0;
    int32_t nrb = 0;
    bool is_diagonal = false;        // global isDiagonalSM
    sla_csr *transposed = nullptr;   // built lazily (single-rank only)
    std::vector<sla_csr *> panels;   // column-panel views (irregular matrices whose x does not fit the L2), else empty
    double *d_panel_y = nullptr;     // running row sums for epilogues that do not store y
    bool is_panel_view = false;
    // row-slice x column-panel tiles (irregular matrices whose x does not fit one XCD's L2; sla_lower_tiles.cpp): the entries
    // re-ordered by (slice, panel, row, column); a wavefront keeps a slice's row sums in LDS while it walks the panels
    bool use_tiles = false;
    int32_t tl_S = 0, tl_P = 0, tl_shift = 17;   // slices, panels, log2(columns per panel)
    int32_t *d_tlrow = nullptr;      // tl_S + 1 slice row starts
    uint32_t *d_tloff = nullptr;     // tl_S x (tl_P + 1): first entry of tile (s, j), relative to the slice's first entry rowptr[tlrow[s]]
    uint32_t *d_tlidx = nullptr;     // per entry: (row - tlrow[s]) << tl_shift | (col - j * W)
    double *d_tlval = nullptr;
    unsigned *d_tlprog = nullptr;    // per-XCD (round, panel) arrival counters of the launch in flight (panel pacing)
    size_t tlprog_bytes = 0;
    int64_t tl_maxseg = 0;           // longest (row, panel) segment = layers of the deepest tile
    double *d_tldummy = nullptr;     // one all-zero panel (2^tl_shift doubles): what the tile kernels' empty pipeline-drain chunks gather from
    bool tl_cu = false;              // CU-wide slices, relaxed order (sla_spmv_ctiles.hip): entries [slice][wavefront][panel], d_tloff = tl_S x 4 x (tl_P + 1)
    bool tl_rowown = false;          // ... with every row owned by one wavefront: exact, reproducible (round 6)
    sla_tri_plan *tri[2] = {nullptr, nullptr};  // [0] lower, [1] upper triangle schedules (built on first use)
    sla::OcPlan *oc = nullptr;       // on-chip solver plan (built by the first sla_solver_step that could use it; ok = false: tried, not eligible)
    // comm / compute overlap of the sharded (#>) (wave-sliced forms): the 512-row steps whose rows reference own columns only
    // (interior: they can run while the halo exchange is in flight) and the rest, each in the visiting order of d_wsched
    int32_t *d_ov_int = nullptr, *d_ov_bnd = nullptr;
    int32_t ov_nint = 0, ov_nbnd = 0;
    std::vector<int32_t> h_wsched;   // host copy of d_wsched (empty: ascending order)
    sla::XPlan *xplan = nullptr;     // sharded only: which x entries this rank exchanges with each peer
    sla::AgPlan *ag = nullptr;       // tile form in all-gather mode: exchange groups + panel passes of the overlapped all-gather (sla_dist.cpp)
    int64_t max_row_nnz = 0;
    std::string lower_log;           // "phase=milliseconds;..." of this matrix's lowering (csr_upload; sla_csr_lower_info)
};

struct sla_solver {
    std::vector<sla_solver *> kids;  // non-empty: a bundle of per-rank states of a multi-device context
    sla_ctx *ctx = nullptr;
    sla_csr *A = nullptr;
    int method = 0;
    sla_vec *x = nullptr, *r = nullptr, *p = nullptr, *u = nullptr;
    sla_vec *r0hat = nullptr, *b = nullptr;
    sla_vec *t1 = nullptr, *t2 = nullptr, *t3 = nullptr;  // Ap / s / As  (CGS: aap / q / uq)
    double *d_parts = nullptr;  // 6 * kMaxParts
    double *d_gath = nullptr;   // per-rank sums, 8 slots * 2 * nranks
    sla::SolverScalars *d_sc = nullptr;
    sla::SolverScalars *h_sc = nullptr;  // pinned
    bool have_res = false;               // d_parts[RES] holds the residual of the current x
    bool step_graph_failed = false;      // capture / instantiation failed once: this state record keeps to plain stream launches
    hipGraphExec_t step_graph = nullptr; // two consecutive steps (even, odd parity) captured for replay (sla_solver_step, launch-bound sizes)
    int step_graph_gen = -1;             // ctx->opt_gen at capture time (flow-selecting knobs -- bicg_fuse45, vec_policy ... -- may change between calls)
    // sharded BiCGSTAB with ghost rows (sla_solvers.cpp): r, p, Ap and s are kept valid on the ghl / ghr rows this
    // rank's SpMV reads from its neighbours, so a step needs 3 grouped exchanges instead of 5
    bool ghost = false;
    int64_t ghl = 0, ghr = 0;
    double *d_hist = nullptr;            // linSolve0's residual trace (sla_solve_opts.history), one slot per iteration
    int32_t hist_cap = 0;
    alignas(8) char ctl_storage[128];    // driver-private step bookkeeping (sla_solvers.cpp)
};

// single-process multi-device bundles (sla_multi.cpp): every m_* runs the per-rank entry point on all ranks concurrently
namespace sla {
int m_ctx_destroy(sla_ctx *p);
int m_ctx_sync(sla_ctx *p);
int m_csr_from_coo(sla_ctx *p, int64_t m, int64_t n, int64_t nnz, const int64_t *row, const int64_t *col, const double *val, int dup, sla_csr_t *out);
int m_csr_from_csr(sla_ctx *p, int64_t m, int64_t n, const int64_t *rp, const int64_t *ci, const double *va, sla_csr_t *out);
int m_csr_from_matrix_market(sla_ctx *p, const char *path, int dup, sla_csr_t *out);
int m_csr_destroy(sla_csr *A);
int m_csr_export(sla_csr *A, int64_t *rowptr, int64_t *colidx, double *val);
int m_vec_create(sla_ctx *p, int64_t n, const double *host, sla_vec_t *out);
int m_vec_from_matrix_market(sla_ctx *p, const char *path, sla_vec_t *out);
int m_vec_destroy(sla_vec *v);
int m_vec_to_host(sla_vec *v, double *host);
int m_vec_copy(sla_vec *s, sla_vec *d);
int m_spmv(sla_csr *A, sla_vec *x, sla_vec *y, bool transposed);
int m_dot(sla_vec *x, sla_vec *y, double *out);
int m_nrm2(sla_vec *x, double *out);
int m_axpby(double a, sla_vec *x, double b, sla_vec *y);
int m_scal(double a, sla_vec *x);
int m_solver_init(int method, sla_csr *A, sla_vec *b, sla_vec *x0, sla_solver_t *out);
int m_solver_step(sla_solver *S, int k);
int m_solver_get(sla_solver *S, int field, sla_vec *out);
int m_solver_clone(sla_solver *S, sla_solver_t *out);
int m_solver_set_shadow(sla_solver *S, sla_vec *r0hat);
int m_solver_destroy(sla_solver *S);
int m_linsolve0(int method, sla_csr *A, sla_vec *b, sla_vec *x0, const sla_solve_opts *o, sla_vec *xo, sla_solve_info *info);
int m_gmres(sla_csr *A, sla_vec *b, sla_vec *x0, int restart, const sla_solve_opts *o, sla_vec *xo, sla_solve_info *info);
int m_linsolve(sla_csr *A, sla_vec *b, sla_vec *xo, sla_solve_info *info);
int m_arnoldi(sla_csr *A, sla_vec *b, int kn, double *Q, double *H, int *k_done);
int multi_unsupported(const char *what);
}  // namespace sla

namespace sla {
// is the wave-sliced SpMV form of A enabled by the context's knobs?
// (its workgroups need a few steps each to amortise the staging pipeline's fill: below that the gather kernel is faster -- 1 M-row
// Poisson: 22 500 vs 21 300 it/s; SLA_WD_LDS=2 forces it)
// (LDS windows: from 8 steps per workgroup on -- 16 before the y store moved behind the end-of-step wait, round 3)
inline bool wd_lds_on(const sla_csr *A) { return A->wd_lds && !A->wd_vv && (A->ctx->wd_lds == 2 || (A->ctx->wd_lds == 1 && A->nblk_wd >= 8 * 4 * A->ctx->n_cu)); }
// (plane march: from ~2 M rows and 48 planes on -- measured on 216 x 216 x k: k = 27: 17.9 k against 18.4 k it/s for the gather
// kernel, k = 54: 12.9-13.0 k against 12.5 k, k = 108: 7.6 k against 7.3 k, profiles/r03_ab_slab_forms.txt)
inline bool wd_march_on(const sla_csr *A) {
    if (!A->wd_march || A->wd_vv || A->ctx->wd_march < 1 || A->ctx->wd_lds < 1 || A->ctx->spmv_algo != 0) return false;
    return A->ctx->wd_march == 2 || (A->wd_mg.planes >= 48 && A->nblk_wd >= 4 * 4 * A->ctx->n_cu);
}
// does the plain CSR-stream form of A stage an x window in LDS (spmv_xwin_kernel)?  Only without the paired loads of
// spmv_stream_kernel (stream_wide, default): with them the plain kernel is the faster one (round 3: K1 222-231 vs 232-241 us,
// K3 -- four sums since the fused K4+K5 flow -- 228-238 vs 258 us on the 216^3 Laplacian), so the window form is an A/B knob now
inline bool stream_xwin_on(const sla_csr *A) { return A->use_xwin && A->ctx->xwin && !A->ctx->stream_wide; }
// ... of the dictionary-code kernels (spmv_diag_kernel): only with xwin = 2.  Since the paired loads and the operands issued with the
// gathers its window variant loses (216^3 Laplacian on 9 B per entry: K1 202 vs 188 us, the four-sum K3 -- 17 spilled VGPRs -- 271 vs
// 210 us, 1640 vs 1870 it/s); the pair-code kernel (spmv_vdict_kernel) keeps its window at xwin = 1 (K1 85 vs 90 us).
inline bool diag_xwin_on(const sla_csr *A) { return A->use_xwin && A->ctx->xwin >= 2; }
// ... is the dictionary-code kernel (spmv_diag_kernel: 9 B per entry) the one to run?  Only for short rows: its row phase decodes and
// gathers per lane, and from ~12 entries per row on the plain CSR kernels win (round 4, e05r0000 tiled to 1 M rows, 25 per row:
// dictionary codes 4380 it/s, plain CSR 5190-5330; 5-7 per row: dictionary codes +0..10 %); diag = 2 takes it at any row length.
inline bool diag_on(const sla_csr *A) { return A->use_diag && (A->ctx->diag == 2 || (A->ctx->diag == 1 && A->nnz <= 12 * A->rows)); }
inline bool wd_on(const sla_csr *A) { return A->wd_vv ? (A->ctx->wdia && A->ctx->wdia_vv) : A->ctx->wdia != 0; }
// a (#>) of this matrix runs one of the value-indexed kernels, which never touch col / val (the dispatch of launch_spmv_t, sla_spmv.hip)
inline bool spmv_value_indexed(const sla_csr *A, bool dual) {
    const sla_ctx *c = A->ctx;
    if (A->rp64 || c->spmv_algo != 0) return false;
    return (A->use_wdia && wd_on(A) && !dual) || (A->use_vdict && c->vdict);
}
int csr_ensure_canon(sla_csr *A);   // sla_lower.cpp: materialise col / val of a canon_lazy matrix (allocation + one kernel); no-op otherwise

// ---------------------------------------------------------------------------------------------------------------
// Device binding.  HIP's current device is a per-THREAD setting and a new thread starts on device 0, so every C-ABI entry
// point that can allocate, launch or call RCCL opens a `Bind` on its context first (hipSetDevice + a thread-local "this
// thread is working for context c" token): the rank contexts of sla_ctx_create_multi are driven from worker threads, and
// a caller may hop between contexts on different devices.  SLA_DEBUG_BINDING=1 makes the class of bug testable on a
// one-GPU box, where every device id is 0 and a wrong current device is invisible: stream_of() (every launch, copy and
// collective takes its stream through it) and dev_malloc() then require the calling thread to be inside an entry point
// bound to THAT context; violations are counted (sla_debug_binding_violations) and reported on stderr, =2 aborts.
struct Bind {
    const sla_ctx *prev;
    int prev_device;
    explicit Bind(const sla_ctx *c);
    ~Bind();
    Bind(const Bind &) = delete;
    Bind &operator=(const Bind &) = delete;
};
extern int g_debug_binding;                                   // SLA_DEBUG_BINDING, read once
void binding_violation(const sla_ctx *c, const char *what);   // out of line: count, report, maybe abort
const sla_ctx *bound_ctx();
void unbind_destroyed(const sla_ctx *c);
inline void check_bound(const sla_ctx *c, const char *what) {
    if (g_debug_binding && bound_ctx() != c) binding_violation(c, what);
}
inline hipStream_t stream_of(const sla_ctx *c) {
    check_bound(c, "stream use");
    return c->stream;
}
inline hipError_t dev_malloc(const sla_ctx *c, void **p, size_t bytes) {
    check_bound(c, "hipMalloc");
    return hipMalloc(p, bytes);
}

// Guarded device allocation for everything an SpMV may gather from (vectors, the exchange landing buffer, the
// Arnoldi basis): kGuardBytes of readable slack on both sides.  spmv_wdia_kernel gathers row PAIRS with one
// 16-byte load; when only one row of a pair holds an entry, the other half may lie one element outside
// [0, n) -- inside the slack, never used (the fold is masked per row).
// On sharded contexts the vectors get kHaloBytes instead: the window (halo) exchange then receives the neighbours' planes
// straight into the slack around the rank's own rows, and the SpMV gathers from `vector - first_row` with no copy at all.
inline hipError_t guard_malloc(const sla_ctx *c, void **p, size_t bytes, size_t guard = kGuardBytes) {
    void *raw = nullptr;
    const hipError_t e = dev_malloc(c, &raw, bytes + 2 * guard);
    if (e == hipSuccess) *p = (char *)raw + guard;
    return e;
}
inline hipError_t guard_free(void *p, size_t guard = kGuardBytes) { return p ? hipFree((char *)p - guard) : hipSuccess; }

// error plumbing ---------------------------------------------------------------------------------
void set_error(const std::string &msg);
int fail(int code, const std::string &msg);
int mixed_handles(const char *what);
long binding_violations();
// No C++ exception may cross the C ABI: a failed host allocation inside an entry point becomes SLA_ERR_ALLOC.
template <class F>
inline int no_throw(const char *what, F body) {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(SLA_ERR_ALLOC, std::string(what) + ": out of host memory");
    } catch (const std::exception &e) {
        return fail(SLA_ERR_INVALID, std::string(what) + ": " + e.what());
    } catch (...) {
        return fail(SLA_ERR_INVALID, std::string(what) + ": unknown C++ exception");
    }
}
#define SLA_HIP_TRY(expr)                                                                        \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return ::sla::fail(SLA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)
#define SLA_TRY(expr)                  \
    do {                               \
        int _rc = (expr);              \
        if (_rc != SLA_OK) return _rc; \
    } while (0)

// sla_solve_opts / sla_solve_info carry the caller's struct size (include/sla_hip.h): read / write only what lies inside it
int read_solve_opts(const sla_solve_opts *in, sla_solve_opts *out, const char *who);      // *out = defaults overlaid with the caller's members
int info_begin(const sla_solve_info *user, sla_solve_info *local, const char *who);       // validates user->struct_size; *local = "nothing evaluated yet"
void info_commit(sla_solve_info *user, const sla_solve_info &local);                      // copies the members the caller's struct has

// profiling scope: brackets a launch with events when the context is recording that kernel id
struct ProfScope {
    sla_ctx *c;
    bool on;
    bool deferred = false;   // the launch site takes the events itself (SLA_KLAUNCH): nothing recorded here unless it did not
    // ext: the scope holds exactly ONE kernel launch made through SLA_KLAUNCH.  The kernel then carries the two events itself
    // (hipExtLaunchKernelGGL: its own start / stop stamps) instead of standing between two marker packets -- round 4: the markers
    // around the SpMV of a 137 us Arnoldi step cost 3 us of it.
    ProfScope(sla_ctx *ctx, int kernel_id, bool ext = false);
    ~ProfScope();
};
// the pending event pair of an `ext` ProfScope, taken (once) by the launch it was opened for
inline bool prof_take(sla_ctx *c, hipEvent_t *e0, hipEvent_t *e1) {
    if (c->prof_pending < 0) return false;
    *e0 = c->prof_ev[2 * (size_t)c->prof_pending];
    *e1 = c->prof_ev[2 * (size_t)c->prof_pending + 1];
    c->prof_pending = -1;
    return true;
}
#define SLA_KLAUNCH(c_, kernel, grid, block, shmem, stream, ...)                                                           \
    do {                                                                                                                   \
        hipEvent_t e0_ = nullptr, e1_ = nullptr;                                                                           \
        if (prof_take((c_), &e0_, &e1_)) hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, e0_, e1_, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                          \
    } while (0)

// host CSR builder (sla_csr_build.cpp) -------------------------------------------------------------
// vector whose resize() leaves the new elements uninitialised (assign(n, v) and push_back still initialise): the host CSR arrays are
// written in full right after they are sized -- zero-filling 2 x 0.56 GB first cost 0.2 s at 70 M entries
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    template <class U, class... Args>
    void construct(U *p, Args &&...args) {
        if constexpr (sizeof...(Args) == 0) ::new ((void *)p) U;
        else ::new ((void *)p) U(std::forward<Args>(args)...);
    }
};
template <class T> using raw_vector = std::vector<T, NoInitAlloc<T>>;
struct HostCsr {
    int64_t m = 0, n = 0;
    raw_vector<int64_t> rowptr;
    raw_vector<int64_t> col;
    raw_vector<double> val;
};
int build_csr_from_coo(int64_t m, int64_t n, int64_t nnz, const int64_t *row, const int64_t *col,
                       const double *val, int dup_policy, HostCsr &out);
void transpose_csr(const HostCsr &a, HostCsr &t);
// device-side sort / dedupe of large triple lists (sla_coo_sort.hip); bit-identical to build_csr_from_coo
bool device_coo_supported(int64_t m, int64_t n, int64_t nnz);
int device_coo_to_csr(sla_ctx *c, int64_t m, int64_t n, int64_t nnz, const int64_t *row, const int64_t *col,
                      const double *val, int dup_policy, HostCsr &out);
int device_transpose_to_host(sla_csr *A, HostCsr &t, bool *done);
int validate_columns_device(sla_csr *A, int64_t n, int *verdict);   // sla_coo_sort.hip: canonical-CSR check of uploaded columns (0 fine, 1 out of bounds, 2 not ascending)   // sla_coo_sort.hip: transposeSM of a lowered row block by a device sort
bool host_is_diagonal(int64_t rows, int64_t row_begin, const int64_t *rowptr, const int64_t *col);
void build_row_blocks(int64_t rows, const int64_t *rowptr, std::vector<int32_t> &rb, int64_t &max_row_nnz, int row_align,
                      int nnz_target);

// distributed plumbing (sla_dist.cpp) ---------------------------------------------------------------
int dist_unique_id(void *out128);
int dist_comm_init(sla_ctx *ctx, const void *unique_id);
int dist_comm_destroy(sla_ctx *ctx);
int dist_loopback_join(sla_ctx *ctx, int group_key);
int dist_allgather_f64(sla_ctx *ctx, const double *send, double *recv, int64_t count);
int dist_reduce_scatter_f64(sla_ctx *ctx, const double *send, double *recv, int64_t recvcount);
// xfull == xlocal - my_begin: in-place (the own rows are where they belong already, nothing is copied)
int dist_exchange_window(sla_ctx *ctx, const XPlan &plan, const double *xlocal, int64_t my_begin, int64_t n_local, double *xfull);
int dist_allgather_p2p_f64(sla_ctx *ctx, const double *send, double *recv, int64_t count);   // all-gather as grouped send/recv
// one group of the overlapped all-gather: this rank's pieces go to every peer, the peers' pieces land at xfull + their global column
int dist_exchange_group(sla_ctx *ctx, const std::vector<AgPiece> &pieces, const double *xlocal, int64_t my_begin, double *xfull);
int build_ag_plan(sla_csr *A, bool failed);                // after build_tiles: the plan, its device visit list, running-sum buffer and events
void ag_plan_free(AgPlan *p);
bool ag_split(const sla_csr *A);              // does (#>) on A run as panel passes behind a grouped all-gather?
// the collectives issued between the two calls go out as ONE grouped launch (ncclGroupStart / ncclGroupEnd)
int dist_group_begin(sla_ctx *ctx);
int dist_group_end(sla_ctx *ctx);
// does the window (halo) exchange of A land in place around vectors of x's shape?  If so: the ghost rows on either side
bool halo_inplace_extents(const sla_csr *A, const sla_vec *x, int64_t *left, int64_t *right);
int dist_allreduce_max_i32(sla_ctx *ctx, int *value_host);
int dist_comm_count(sla_ctx *ctx, int *nranks);
int dist_p2p_selftest(sla_ctx *ctx, int64_t count, int pieces, double *max_abs_err);
int dist_preflight(sla_ctx *ctx, int phase, int64_t count, double *max_abs_err, double *ms);   // one checked collective across the real ranks (sla_dist_preflight)   // grouped ncclSend / ncclRecv with this rank as its own peer

// shared helpers of sla_api.cpp --------------------------------------------------------------------------
// full-length gather base for an SpMV with matrix A (null: plain all-gather) whose input is `x`
int gather_x(const sla_csr *A, sla_vec *x, const double **base);
int gather_raw(sla_ctx *c, const sla_csr *A, const double *local, int64_t shard, const double **base);
int reduce_to_host(sla_ctx *c, const double *p1, const double *p2, int np, double *out);
int vec_alloc(sla_ctx *c, int64_t n, sla_vec **out);
int csr_transposed(sla_csr *A, sla_csr **out);
// Host-side set-up work (lowering analyses, COO validation) is row-parallel: par_rows runs fn(t, lo, hi) over T contiguous row ranges
// whose boundaries are multiples of `align` rows, on T host threads (SLA_HOST_THREADS, default <= 16); returns T.
// sla_xfer.cpp: large copies between the caller's pageable arrays and the device through pinned slots on several host threads
typedef void (*xfer_stage_fn)(void *slot, size_t off, size_t len, const void *ctx);
hipError_t xfer_copy(sla_ctx *c, void *dst, const void *src, size_t bytes, hipMemcpyKind kind, const std::atomic<int> *stop = nullptr,
                     size_t *done = nullptr, xfer_stage_fn stage = nullptr, const void *stage_ctx = nullptr, bool ordered = false);
void xfer_warm(int device, int lanes);
// Background threads that outlive the call that started them (the lane warm-up, deferred releases) are counted; a process-exit handler
// registered on the first context waits for the count to reach zero BEFORE the HIP runtime's own exit handler runs (a context alive at
// exit -- a caller that never destroys it -- had the warm-up inside hipHostMalloc while the runtime unmapped itself: a segfault at exit).
void bg_begin();
void bg_end();
void bg_exit_handler_once();
struct BgTask {   // bg_begin() is the starter's (before the thread exists); the thread owns the matching bg_end()
    BgTask() = default;
    BgTask(const BgTask &) = delete;
    ~BgTask() { bg_end(); }
};
// Run `release` (the destruction of large host buffers) on a background thread; finished ones are forgotten, the context's destructor
// waits for the rest.
template <class F>
inline void defer_release(sla_ctx *c, F &&release) {
    auto &d = c->deferred;
    for (size_t i = 0; i < d.size();)
        if (d[i].wait_for(std::chrono::seconds(0)) == std::future_status::ready) { d[i] = std::move(d.back()); d.pop_back(); }
        else ++i;
    auto r = std::make_shared<typename std::decay<F>::type>(std::forward<F>(release));
    bg_begin();
    try {
        d.push_back(std::async(std::launch::async, [r] {
            BgTask task;
            (*r)();
        }));
    } catch (...) {   // no thread to be had: release here (the count must not stay up, the exit handler waits for it)
        bg_end();
        (*r)();
    }
}
int host_threads();   // sla_lower.cpp
template <class F>
int par_rows(int64_t rows, int64_t align, F fn, int64_t serial_below = 200000) {
    int T = host_threads();
    const int64_t units = (rows + align - 1) / align;
    if (units < 64 || rows < serial_below) T = 1;
    T = (int)std::min<int64_t>(T, std::max<int64_t>(units, 1));
    auto range = [&](int t, int64_t &lo, int64_t &hi) {
        lo = std::min<int64_t>(rows, units * t / T * align);
        hi = std::min<int64_t>(rows, units * (t + 1) / T * align);
    };
    if (T == 1) { fn(0, (int64_t)0, rows); return 1; }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) {
        int64_t lo, hi;
        range(t, lo, hi);
        th.emplace_back([=, &fn] { fn(t, lo, hi); });
    }
    for (auto &x : th) x.join();
    return T;
}
// sla_lower.cpp: validate, upload and lower one rank's row block (panel_view: a column-panel view of a parent, lowered plainly);
// csr_reject: a rank whose input failed validation still joins the agreement collective of csr_upload before it reports
int csr_upload(sla_ctx *c, int64_t m, int64_t n, int64_t row_begin, int64_t rows, const int64_t *rowptr, const int64_t *col,
               const double *val, sla_csr **out, bool panel_view = false, bool validate_cols = false);
int csr_reject(sla_ctx *c, int rc);
void tri_plan_free(sla_tri_plan *p);   // sla_precond.cpp (sla_csr_destroy releases a matrix's level schedules)
int spmv_transposed(sla_csr *A, const double *x_local, double *y_local, int64_t y_shard);

// kernel launchers (sla_spmv.hip dispatch + the family files, sla_vec_kernels.hip, sla_arnoldi.hip, sla_tri.hip) -----------------------------------------------------------------
struct Parts {  // a partial-sum array as seen by a consumer prologue: p[i * stride], i < n
    const double *p; int n; int stride;
};
struct SpmvLaunch {
    int epi = EPI_NONE;
    const double *x = nullptr;  // gather base
    double *y = nullptr;
    const double *w = nullptr;
    double *z = nullptr;
    double *p1 = nullptr, *p2 = nullptr;
    double *p3 = nullptr, *p4 = nullptr;       // EPI_DOT4: y . z and w . z (z read-only)
    SolverScalars *sc = nullptr;
    const double *pres = nullptr; int npres = 0, pres_stride = 1;
    const double *pa = nullptr, *pb = nullptr; int npa = 0, pa_stride = 1;
    int step_begin = 0;
    const double *x2 = nullptr, *b2 = nullptr;  // dual SpMV: also leave partials of ||A x2 - b2||^2 in p2
    const double *yinit = nullptr;              // column-panel pass: continue these running row sums
    int in_panel = 0;
    int kernel_id = SLA_KERNEL_SPMV;
    int part = 0;                               // row-sharded overlap: 0 all rows, 1 the interior steps only, 2 the boundary steps only
    const int32_t *tvis = nullptr; int tv0 = 0, tv1 = -1;   // tile form, one PASS of an overlapped all-gather: the panels tvis[tv0 .. tv1) (tv1 < 0: all panels, ascending)
    int tlast = 1;                              // ... 0: not the last pass -- the running row sums go to y, no epilogue
    const double *fs_ap = nullptr;              // BiCGSTAB's K2 folded into K3 (spmv_fuse_s_ok): gather from s = x - alpha fs_ap, alpha = rho / sum(pa)
};
bool spmv_fuse_s_ok(const sla_csr *A, bool slab = false);   // would a whole-matrix (slab: whole-slab) (#>) on A run the plane-march kernel?
int spmv_grid(const sla_csr *A);  // number of blocks (= partial slots written) of an SpMV launch on A
int launch_spmv(const sla_csr *A, const SpmvLaunch &l);
// exchange the input vector of a (#>) and launch it; on row-sharded contexts the interior rows run while the halo is in flight.
// *np (may be null) = partial slots the launch(es) wrote (spmv_grid(A) unless the launch was split)
int spmv_exchanged(sla_csr *A, sla_vec *x, SpmvLaunch l, int *np);
bool overlap_split(const sla_csr *A);   // does (#>) on A run as interior + boundary launches?
int overlap_grid(const sla_csr *A, int part);
bool tiles_on(const sla_csr *A);                               // is the tile form of A in use?
int launch_spmv_tiles(const sla_csr *A, const SpmvLaunch &l);   // sla_spmv_tiles.hip
// the kernel families behind launch_spmv (sla_spmv.hip picks; `a` = the kernel argument block it assembled, `grid` = spmv_grid(A))
int launch_spmv_stream(const sla_csr *A, int epi, const SpmvArgs<int32_t> &a, int grid);                     // sla_spmv_stream.hip
int launch_spmv_stream(const sla_csr *A, int epi, const SpmvArgs<int64_t> &a, int grid);
int launch_spmv_dual(const sla_csr *A, const SpmvArgs<int32_t> &a, const double *x2, const double *b2, int grid);
int launch_spmv_dual(const sla_csr *A, const SpmvArgs<int64_t> &a, const double *x2, const double *b2, int grid);
int launch_spmv_diag(const sla_csr *A, int epi, const SpmvArgs<int32_t> &a, int grid);                       // sla_spmv_dict.hip
int launch_spmv_diag(const sla_csr *A, int epi, const SpmvArgs<int64_t> &a, int grid);
int launch_spmv_dual_diag(const sla_csr *A, const SpmvArgs<int32_t> &a, const double *x2, const double *b2, int grid);
int launch_spmv_dual_diag(const sla_csr *A, const SpmvArgs<int64_t> &a, const double *x2, const double *b2, int grid);
int launch_spmv_vdict(const sla_csr *A, int epi, const SpmvArgs<int32_t> &a, const double *x2, const double *b2, int grid);
int launch_spmv_wdia(const sla_csr *A, int epi, const SpmvArgs<int32_t> &a, const int32_t *sched, int32_t nblk_wd, int grid, int stream_nt);   // sla_spmv_wdia.hip
int launch_lp_reorder(sla_ctx *c, bool rp64, const void *pp, const void *q, const int32_t *col, const double *val, uint16_t *col2, double *val2,
                      int64_t rows, int64_t P, int32_t W);                                                            // sla_spmv_lpanel.hip
int launch_spmv_lpanel(const sla_csr *A, int epi, const SpmvArgs<int32_t> &a, int grid);
int launch_lpanel_finish(const sla_csr *A, int epi, const SpmvArgs<int32_t> &a, int grid);   // its finish kernel alone
bool lflat_on(const sla_csr *A);                                                                             // sla_spmv_lflat.hip
bool lflat_candidate(const sla_csr *A, int64_t n, int64_t rows);
int build_lflat(sla_csr *A, int64_t n, int64_t rows, int64_t col_lo, int64_t col_hi);
int launch_spmv_lflat(const sla_csr *A, int epi, const SpmvArgs<int32_t> &a, int grid);
int launch_spmv_lpanel(const sla_csr *A, int epi, const SpmvArgs<int64_t> &a, int grid);
// do the solver's vectors (7 of n entries for BiCGSTAB) overflow the memory-side cache?  Then stream them past it.
inline bool vec_stream_nt(const sla_ctx *c, int64_t n) { return c->vec_nt < 0 ? 7 * 8 * n > c->mall_bytes : c->vec_nt != 0; }
int launch_wdia_lds(const sla_csr *A, int epi, const SpmvArgs<int32_t> &a, const int32_t *sched, int32_t nblk, int grid, int stream_nt);   // sla_spmv_wdia_lds.hip
int wd_lds_grid(const sla_csr *A);
int wd_march_grid(const sla_csr *A);
void wd_march_prepare();   // (lowering: queries the instantiations' register counts once, outside any stream capture)
int launch_wdia_march(const sla_csr *A, int epi, const SpmvArgs<int32_t> &a, int grid, int stream_nt);
bool wave_on(const sla_csr *A);                                                                              // sla_spmv_wave.hip
int wave_grid(const sla_csr *A);
int fold_kind(const sla_csr *A);           // sla_api.cpp: sla_fold_kind of A's (#>)
bool csr_fold_relaxed(const sla_csr *A);   // sla_api.cpp: is A's (#>) the order-relaxed tile form (SLA_FLAG_RELAXED_ORDER)?
bool wave_plain(const sla_csr *A);   // sla_spmv.hip: does a plain (#>) on A end up on spmv_wave_kernel?
int launch_spmv_wave(const sla_csr *A, int epi, const SpmvArgs<int32_t> &a, int grid);
int tiles_grid(const sla_csr *A);
int launch_tri_syncfree(const sla_csr *T, const sla_tri_plan *p, const double *b, double *x, int *d_fail);   // sla_tri.hip
constexpr int SLA_TRI_NO_FIT = -100;   // launch_tri_blocks: the block kernel cannot be made resident on this device (not an error: the level schedule runs)
int launch_tri_blocks(const sla_csr *T, const sla_tri_plan *p, int upper, const double *b, double *x, int *d_fail);
int launch_col_slack_fill(sla_ctx *c, int32_t *d_col, int64_t nnz);   // sla_spmv_wave.hip: the slack behind the column array repeats the last column
int launch_spmv_ctiles(const sla_csr *A, const SpmvLaunch &l);   // sla_spmv_ctiles.hip (CU-wide slices)
int ctiles_grid(const sla_csr *A);
int probe_xcd_layout(sla_ctx *c);   // sets c->xcd8 (sla_spmv_tiles.hip)
// sla_lower_tiles.cpp: builds the tile form of A when it pays (irregular structure, x larger than the L2); no-op otherwise
int build_tiles(sla_csr *A, int64_t n, int64_t rows, const int64_t *rowptr, const int64_t *col, const double *val);
// sla_tiles_build.hip: the same re-ordering on the device from A's canonical arrays (*done = false: not taken, use the host builder)
int build_tiles_device(sla_csr *A, const std::vector<int32_t> &srow, int shift, int64_t P, int64_t *maxseg_out, int64_t *nbreaks_out, bool *done);
// ... and the CU-wide layout of sla_spmv_ctiles.hip (relaxed: one column-sorted run per tile)
int build_ctiles_device(sla_csr *A, const std::vector<int32_t> &srow, int shift, int64_t P, const int64_t *rowptr_host, bool *done);

// sla_onchip.hip: can `k` steps of S run as one persistent launch (builds the matrix's plan on first use)?  Then run them.
bool onchip_usable(sla_solver *S, bool res = false);   // res: linSolve0's loop (step, true residual, test) inside the launch
int launch_onchip_steps(sla_solver *S, int par, int k, bool res = false);

int vec_grid(int64_t n_local);
// p1[b] = sum x.y over block b's elements (grid = vec_grid)
int launch_dot(sla_ctx *c, int64_t n, const double *x, const double *y, double *p1);
// out[0] = sum(p1[0..np)), out[1] = sum(p2[0..np)) (p2 may be null); one block
int launch_finalize(sla_ctx *c, const double *p1, const double *p2, int np, double *out);
// out[j] = sum_i parts[j * cs + i * stride], i < np, j < ncols; one block
int launch_finalize_cols(sla_ctx *c, const double *parts, int np, int cs, int stride, int ncols, double *out);
int launch_diag_solve(sla_ctx *c, int64_t n, const double *diag, const double *b, double *x);
int launch_axpby(sla_ctx *c, int64_t n, double a, const double *x, double b, double *y);
int launch_scal(sla_ctx *c, int64_t n, double a, double *x);
int launch_fill(sla_ctx *c, int64_t n, double a, double *x);

// BiCGSTAB (Sparse.hs:972-981)
int launch_bicg_k2(sla_ctx *c, int64_t n, SolverScalars *sc, Parts apr, int par, Parts res, int count_iter,
                   const double *r, const double *ap, double *s);
int launch_bicg_k4(sla_ctx *c, int64_t n, SolverScalars *sc, Parts ass, Parts asas, const double *p, const double *s,
                   const double *as, const double *r0hat, double *x, double *r, double *prho);
int launch_bicg_k5(sla_ctx *c, int64_t n, SolverScalars *sc, Parts rhonew, int par, const double *r, const double *ap, double *p);
int launch_cgs_c24(sla_ctx *c, int64_t n, SolverScalars *sc, Parts rhonew, int par, const double *r, const double *aap, double *u, double *p, double *x);
int launch_bicg_k45(sla_ctx *c, int64_t n, SolverScalars *sc, Parts ass, Parts asas, Parts tr0, Parts sr0, int par, const double *s,   // (s == nullptr: rebuilt from r and ap)
                    const double *as, const double *ap, double *x, double *r, double *p);
// CGS (Sparse.hs:928-939)
int launch_cgs_c2(sla_ctx *c, int64_t n, SolverScalars *sc, Parts apr, int par, Parts res, int count_iter,
                  const double *u, const double *aap, double *q, double *uq, double *x);
int launch_cgs_c4(sla_ctx *c, int64_t n, SolverScalars *sc, Parts rhonew, int par, const double *r, const double *q,
                  double *u, double *p);
// CGNE (Sparse.hs:870-878)
int launch_cgne_n2(sla_ctx *c, int64_t n, SolverScalars *sc, const double *p, double *x);
// unfused N3 for the sharded path: beta ; p1 = t ^+^ beta .* p (t = transpose aa #> r1 after the reduce-scatter) ; p1 . p1
int launch_cgne_n3b(sla_ctx *c, int64_t n, SolverScalars *sc, Parts rr1, int par, const double *t, double *p, double *ppout);
// BCG (extension: the commented bcgStep, Sparse.hs:899-909)
int launch_bcg_b3(sla_ctx *c, int64_t n, SolverScalars *sc, Parts app, int par, const double *p, const double *aap, const double *atp, double *x,
                  double *r, double *rhat, double *rrout);
int launch_bcg_b4(sla_ctx *c, int64_t n, SolverScalars *sc, Parts rr1, int par, const double *r, const double *rhat, double *p, double *phat);
// residual check at the end of a host batch (one block): publishes resnorm / done
int launch_check(sla_ctx *c, SolverScalars *sc, Parts res);
int launch_init_scalars(sla_ctx *c, SolverScalars *sc, Parts rho, Parts r0sq, double tol_abs, double tol_rel, double *hist = nullptr, int hist_cap = 0);
int launch_set_rho(sla_ctx *c, SolverScalars *sc, Parts rho, int par);   // sc->rho2[par] = sum(rho)
// Arnoldi (Sparse.hs:630-667); Q column-major with leading dimension ldq
int launch_arn_dots(sla_ctx *c, int64_t n, const double *Q, int64_t ldq, int ncols, const double *w, double *parts, SolverScalars *sc);
// sla_arnoldi_orth.hip: the three Gram-Schmidt launches of an Arnoldi step as one persistent launch (single-rank contexts that fit)
size_t arn_orth_bar_bytes();
bool arn_orth_usable(sla_ctx *c, int64_t n, int64_t ldq, int ncols_max);
int launch_arn_orth(sla_ctx *c, int64_t n, const double *Q, int64_t ldq, int ncols, const double *w, double *qnext, double *Hcol, double *hsub,
                    SolverScalars *sc, double *parts, unsigned *bar, int first);
int arn_grid(int64_t n);  // grid (= partials per column) of the Arnoldi update / normalise kernels
int arn_dots_grid(int64_t n, int ncols);   // ... of the dots pass over ncols basis columns
// partial i of column j lives at hp[j * cs + i * stride], i < np
int launch_arn_update(sla_ctx *c, int64_t n, const double *Q, int64_t ldq, int ncols, const double *hp, int np, int cs,
                      int stride, double *w, double *pn, double *Hcol, SolverScalars *sc);
int launch_arn_normalize(sla_ctx *c, int64_t n, Parts nrm, const double *w, double *qnext, double *hsub,
                         SolverScalars *sc, int first);
int launch_gemv_accum(sla_ctx *c, int64_t n, const double *Q, int64_t ldq, int ncols, const double *ycoef_dev, double *x);
// one dependency level of a triangular solve: rows order[0..count), one lane per row
int launch_tri_level(const sla_csr *T, const sla_tri_plan *p, int64_t first, int64_t count, const double *b, double *x);
int launch_tri_sparsify(sla_ctx *c, int64_t n, double *x);  // sparsifySV: |x_i| <= 1e-12 -> 0

}  // namespace sla
