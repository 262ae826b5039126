// sla_lower_tiles.cpp -- lowering analysis for the row-slice x column-panel tile form (kernel: sla_spmv_tiles.hip).
//
// Taken for irregular matrices (no diagonal dictionary, entries not clustered around the diagonal, rows not dense
// enough for the LDS-panel form) whose x is larger than two L2-sized panels.  The canonical CSR arrays stay (export,
// transpose, fallback kernels); this adds a second, tile-major copy of the entries (12 B each) plus
//   tlrow[S + 1]        slice row starts: slices of <= kTileRows rows holding about nnz / S entries each, S = a whole
//                       number of rounds of the kernel's persistent grid (every wavefront gets the same number of slices);
//   tloff[S x (P + 1)]  first entry of tile (slice, panel) relative to the slice's first entry (= rowptr[tlrow[s]]: a
//                       slice owns the same entry range in both orders, so slices are built independently, in parallel).
// Round 5, option tile_relaxed = 1 (the default of round 5; an opt-in since the end of round 6): CU-WIDE slices of <= kCtRows rows, every tile's entries sorted by column and dealt to
// the workgroup's four wavefronts in 64-entry groups, stored [slice][wavefront][panel] with tloff[S x 4 x (P + 1)] (kernel:
// sla_spmv_ctiles.hip, relaxed-order row sums); tile_relaxed = 0 (default): the same CU-wide slices with every row owned by one wavefront (round 6, below) or,
// with tile_rowown = 0, the wavefront-private layout above -- both the bit-exact left fold.
#include <algorithm>
#include <cstring>
#include <limits>
#include <thread>
#include <utility>
#include <vector>

#include "sla_internal.hpp"

namespace sla {

namespace {
// The CU-wide layout of sla_spmv_ctiles.hip on the host (small matrices, option tiles_device = 0, and the cross-check of the device
// builder: both produce the same bits).  Per slice: the entries sorted by (panel, column inside the panel; ties: input order =
// ascending rows), every tile's groups of 64 dealt round-robin to the four wavefronts.
template <typename Finish>
int build_ctiles_host(sla_csr *A, const std::vector<int32_t> &srow, int shift, int64_t P, const int64_t *rowptr, const int64_t *col, const double *val,
                      Finish finish) {
    sla_ctx *c = A->ctx;
    const int64_t S = (int64_t)srow.size() - 1, nnz = rowptr[srow.back()];
    const size_t rowlen = (size_t)(P + 1);
    std::vector<uint32_t> toff((size_t)S * kCtWaves * rowlen);
    std::vector<uint32_t> tidx((size_t)nnz);
    std::vector<double> tval((size_t)nnz);
    const uint32_t cmask = (uint32_t)(((int64_t)1 << shift) - 1);
    int T = (int)std::min<int64_t>(std::max(1u, std::min(16u, std::thread::hardware_concurrency())), S);
    if (const char *e = getenv("SLA_HOST_THREADS")) T = std::max(1, std::min(atoi(e), 64));
    if (nnz < 2000000) T = 1;
    struct Ent { int64_t col, k; uint32_t rl; };
    auto work = [&](int t) {
        std::vector<Ent> ent;
        std::vector<size_t> b((size_t)P + 1);
        for (int64_t s = S * t / T; s < S * (t + 1) / T; ++s) {
            const int64_t r0 = srow[(size_t)s], r1 = srow[(size_t)s + 1], k0 = rowptr[r0];
            ent.clear();
            for (int64_t i = r0; i < r1; ++i)
                for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) ent.push_back({col[k], k, (uint32_t)(i - r0)});
            std::sort(ent.begin(), ent.end(), [](const Ent &x, const Ent &y) { return x.col != y.col ? x.col < y.col : x.k < y.k; });
            {   // tile starts
                size_t o = 0;
                for (int64_t j = 0; j <= P; ++j) {
                    while (o < ent.size() && (ent[o].col >> shift) < j) ++o;
                    b[(size_t)j] = o;
                }
            }
            uint32_t *off = toff.data() + (size_t)s * kCtWaves * rowlen;
            // Round 6 (A->tl_rowown): every ROW of the slice belongs to one wavefront (local row & 3) --
            // a tile's column-sorted entries are split into the four wavefronts' column-sorted sub-runs instead of being dealt in 64-entry
            // groups.  All ds_add_f64 of a row then come from one wavefront in program order (ascending panels, ascending columns): the row
            // sums become reproducible.  The price is the density of a gather instruction: a quarter of the tile's entries per x line.
            const bool rowown = A->tl_rowown;
            for (int64_t j = 0; j < P; ++j) {
                if (rowown) {
                    uint32_t cnt[4] = {0, 0, 0, 0};
                    for (size_t o = b[(size_t)j]; o < b[(size_t)j + 1]; ++o) cnt[ent[o].rl & 3]++;
                    for (uint32_t w = 0; w < 4; ++w) off[w * rowlen + (size_t)j] = cnt[w];
                    continue;
                }
                const uint32_t n0 = (uint32_t)(b[(size_t)j + 1] - b[(size_t)j]), G = (n0 + 63) >> 6, tail = n0 & 63;
                for (uint32_t w = 0; w < 4; ++w) {
                    uint32_t ng = G > w ? (G - w + 3) >> 2 : 0, na = ng * 64;
                    if (tail && G > 0 && ((G - 1) & 3) == w) na -= 64 - tail;
                    off[w * rowlen + (size_t)j] = na;
                }
            }
            uint32_t run = 0;
            for (uint32_t w = 0; w < 4; ++w) {
                for (size_t j = 0; j < (size_t)P; ++j) { const uint32_t cnt = off[w * rowlen + j]; off[w * rowlen + j] = run; run += cnt; }
                off[w * rowlen + (size_t)P] = run;
            }
            std::vector<uint32_t> fill;
            if (rowown) fill.assign((size_t)4 * (size_t)P, 0u);
            for (size_t o = 0; o < ent.size(); ++o) {
                const Ent &e = ent[o];
                const int64_t j = e.col >> shift;
                const uint32_t tt = (uint32_t)(o - b[(size_t)j]), g = tt >> 6, w = rowown ? (e.rl & 3) : (g & 3);
                const size_t dst = rowown ? (size_t)(k0 + off[w * rowlen + (size_t)j] + fill[(size_t)w * (size_t)P + (size_t)j]++)
                                          : (size_t)(k0 + off[w * rowlen + (size_t)j] + ((g >> 2) << 6) + (tt & 63));
                tidx[dst] = (e.rl << shift) | ((uint32_t)e.col & cmask);
                tval[dst] = val[e.k];
            }
        }
    };
    if (T == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    hipError_t err = hipSuccess;
    auto upload = [&](void **dst, const void *src, size_t bytes) {
        if (err != hipSuccess) return;
        err = dev_malloc(c, dst, std::max<size_t>(bytes + 64, 8));
        if (err == hipSuccess && bytes) err = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    };
    upload((void **)&A->d_tloff, toff.data(), sizeof(uint32_t) * toff.size());
    upload((void **)&A->d_tlidx, tidx.data(), sizeof(uint32_t) * tidx.size());
    upload((void **)&A->d_tlval, tval.data(), sizeof(double) * tval.size());
    if (err != hipSuccess) return fail(SLA_ERR_ALLOC, std::string("tile form upload: ") + hipGetErrorString(err));
    A->lower_log += "cu tiles=1;";
    return finish();
}
}  // namespace

int build_tiles(sla_csr *A, int64_t n, int64_t rows, const int64_t *rowptr, const int64_t *col, const double *val) {
    sla_ctx *c = A->ctx;
    const int64_t nnz = rowptr[rows];
    if (!c->tiles || rows == 0 || nnz == 0) return SLA_OK;
    auto skip = [&](const char *why) { A->lower_log += std::string("tile form not taken=") + why + ";"; return SLA_OK; };   // (sla_csr_lower_info)
    const bool force = c->tiles == 2;   // (A/B hook: the tile form wherever it is structurally possible)
    if (!force && (A->use_diag || A->use_wdia || A->use_vdict || A->xwin_fraction >= 0.5)) return SLA_OK;   // stencil / banded structure
    if ((A->use_lpanel && c->lpanel) || (A->use_lflat && c->lflat)) return SLA_OK;               // dense / medium rows: x panels in LDS
    {   // the kernel keeps one slice's row sums per wavefront in static LDS (4 x kTileRows doubles = 153 KiB of the MI355X's 160 KiB)
        int lds = 0;
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device) != hipSuccess ||
            (int64_t)lds < (int64_t)kTileWaves * kTileRows * 8 + 1024 || (int64_t)lds < (int64_t)kCtRows * 8 + 2048)
            return skip("LDS per workgroup");
    }
    // round 5: CU-wide slices (sla_spmv_ctiles.hip) -- one slice of kCtRows rows per WORKGROUP, its row sums shared by the four wavefronts
    // round 6: ... and the same CU-wide slices with every ROW owned by one wavefront (local row & 3): all ds_add_f64 of a row come from one
    // wavefront in program order -- ascending panels, ascending columns -- so the row sums are reproducible and (measured on every case of the
    // suite and of tools/fuzz_tiles.py) the reference's left fold bit for bit.  It is what tile_relaxed = 0 selects now (tile_rowown = -1, auto):
    // config 3a (#>) 1.58 ms against 1.81 for the wavefront-private slices (tile_rowown = 0) and 1.32 for the relaxed dealing.
    const bool rowown = c->tile_rowown == 1 || (c->tile_rowown < 0 && c->tile_relaxed == 0);
    const bool cu = c->tile_relaxed != 0 || rowown;
    A->tl_rowown = cu && rowown;
    const int64_t slice_rows = cu ? kCtRows : kTileRows;
    // (the slices come first: they do not depend on the panel width, and the widest one decides how many bits a local row takes)
    // slices: whole rounds of the persistent grid (kTileBlocksPerCu workgroup(s) of 4 wavefronts per CU)
    const int64_t waves = cu ? (int64_t)c->n_cu : (int64_t)kTileBlocksPerCu * c->n_cu * kTileWaves;   // (CU-wide: slice owners = workgroups)
    const int64_t rounds = std::max<int64_t>(1, (rows + slice_rows * waves - 1) / (slice_rows * waves));
    int64_t S0 = rounds * waves;
    S0 = std::max<int64_t>(1, std::min<int64_t>(S0, (rows + 63) / 64));
    const int64_t target = (nnz + S0 - 1) / S0;
    std::vector<int32_t> srow;
    srow.reserve((size_t)S0 + 64);
    srow.push_back(0);
    {
        int64_t r = 0;
        while (r < rows) {
            // close the slice at kTileRows rows or once it holds `target` entries (a row is never split)
            const int64_t rcap = std::min<int64_t>(rows, r + slice_rows);
            const int64_t want = rowptr[r] + target;
            int64_t e = std::upper_bound(rowptr + r + 1, rowptr + rcap + 1, want) - rowptr;   // first row end beyond the target
            e = std::max<int64_t>(r + 1, std::min<int64_t>(e, rcap));
            if (rowptr[e] - rowptr[r] > (int64_t)std::numeric_limits<uint32_t>::max() - 1024) return skip("slice beyond 32-bit offsets");   // 32-bit tile offsets
            srow.push_back((int32_t)e);
            r = e;
        }
    }
    int row_bits = 0;
    {
        int64_t widest = 1;
        for (size_t i = 0; i + 1 < srow.size(); ++i) widest = std::max<int64_t>(widest, srow[i + 1] - srow[i]);
        // CU-wide slices pack (local row, panel column) in 32 bits with no flag: a matrix whose slices are short (1 M rows: 3906 rows per
        // slice = 12 bits) may take panels wider than 2^17 columns (option tile_shift); the wavefront-private layout keeps its fixed 13 bits
        const int64_t span = cu ? widest : slice_rows;
        while (((int64_t)1 << row_bits) < span) ++row_bits;
    }
    // panel width: 2^17 columns (1 MiB of x) at 10 M rows, 2^16 below ~6 M (measured: 7-8 % faster at 0.5 / 1 / 3 M rows, 3-15 % slower at 10 M)
    // (CU-wide slices, round 5: 2^17 at every size -- 100 / 200 per row at 1 M rows 0.49 -> 0.51 / 0.53 -> 0.56 of peak against 2^16, power-law rows equal)
    const int want = c->tile_shift > 0 ? c->tile_shift : cu ? 17 : (n < 6000000 ? 16 : 17);
    const int shift = std::max(10, std::min((cu ? 32 : 31) - row_bits, want));   // (layer flag, slice row, panel column) packed in 32 bits (CU-wide slices: no flag, 15 + 17 bits)
    const int64_t W = (int64_t)1 << shift;
    if (n <= 2 * W || (c->tile_shift <= 0 && !force && n <= ((int64_t)1 << 18))) return SLA_OK;   // x (nearly) fits the L2 already
    const int64_t P = (n + W - 1) / W;
    if (P > 16384) return skip("more than 16384 panels");
    const int64_t S = (int64_t)srow.size() - 1;
    const int64_t ntoff = cu ? S * kCtWaves * (P + 1) : S * (P + 1);
    if (ntoff > ((int64_t)1 << 31) || ntoff * 4 > nnz * 12 / 2) return skip("offset table larger than half the matrix");   // offset table must stay a fraction of the matrix
    auto finish = [&]() -> int {   // what both builders share once d_tlidx / d_tlval / d_tloff exist: slice starts, pacing table, geometry
        hipError_t e2 = dev_malloc(c, (void **)&A->d_tlrow, sizeof(int32_t) * srow.size() + 64);
        if (e2 == hipSuccess) e2 = hipMemcpy(A->d_tlrow, srow.data(), sizeof(int32_t) * srow.size(), hipMemcpyHostToDevice);
        A->tlprog_bytes = sizeof(int) * 8 * 256;   // pacing table: one progress slot per workgroup, 256 per XCD; zeroed before every launch
        if (e2 == hipSuccess) e2 = dev_malloc(c, (void **)&A->d_tlprog, A->tlprog_bytes);
        if (A->d_tldummy) { (void)hipFree(A->d_tldummy); A->d_tldummy = nullptr; }
        if (e2 == hipSuccess) e2 = dev_malloc(c, (void **)&A->d_tldummy, sizeof(double) << shift);
        if (e2 == hipSuccess) e2 = hipMemset(A->d_tldummy, 0, sizeof(double) << shift);
        if (e2 != hipSuccess) return fail(SLA_ERR_ALLOC, std::string("tile form upload: ") + hipGetErrorString(e2));
        SLA_TRY(probe_xcd_layout(c));
        A->tl_S = (int32_t)S;
        A->tl_P = (int32_t)P;
        A->tl_shift = shift;
        A->tl_cu = cu;
        A->use_tiles = true;
        return SLA_OK;
    };
    // Round 4: the re-ordering as a device sort of the canonical arrays that are on the device already (sla_tiles_build.hip) -- at
    // 330 M entries the host builder below took 1.2 s (16 threads) plus 4 GB of PCIe; option tiles_device: 1 from 2^20 entries on,
    // 2 always, 0 never (the two builders are bit-identical: tests/test_gpu_tiles.py).
    if (c->tiles_device == 2 || (c->tiles_device == 1 && nnz >= ((int64_t)1 << 20))) {
        int64_t mseg = 0, nb = 0;
        bool done = false;
        if (cu) {   // (relaxed order: no layers, nothing to step aside for)
            SLA_TRY(build_ctiles_device(A, srow, shift, P, rowptr, &done));
            if (done) {
                A->lower_log += "tile builder on device=1;cu tiles=1;";
                return finish();
            }
        } else {
            SLA_TRY(build_tiles_device(A, srow, shift, P, &mseg, &nb, &done));
        }
        if (done) {
            A->tl_maxseg = mseg;
            if (nb * 8 > nnz) {   // (dense rows: see the host builder's test below)
                (void)hipFree(A->d_tlidx); (void)hipFree(A->d_tlval); (void)hipFree(A->d_tloff);
                A->d_tlidx = nullptr; A->d_tlval = nullptr; A->d_tloff = nullptr;
                return skip("layer boundaries (dense rows)");
            }
            A->lower_log += "tile builder on device=1;";
            return finish();
        }
    }
    if (cu) return build_ctiles_host(A, srow, shift, P, rowptr, col, val, finish);
    std::vector<uint32_t> toff((size_t)(S * (P + 1)));
    std::vector<uint32_t> tidx((size_t)nnz);
    std::vector<double> tval((size_t)nnz);
    const uint32_t cmask = (uint32_t)(W - 1);
    int T = (int)std::min<int64_t>(std::max(1u, std::min(16u, std::thread::hardware_concurrency())), S);
    if (const char *e = getenv("SLA_HOST_THREADS")) T = std::max(1, std::min(atoi(e), 64));
    if (nnz < 2000000) T = 1;
    std::vector<int64_t> maxseg((size_t)T, 0);
    // Inside a tile the entries are ordered by (layer, column, row): layer = rank of the entry inside its (row, panel) segment in
    // ascending column order.  A layer holds every row at most once, so 64 consecutive entries of a layer are 64 DIFFERENT
    // rows -- the kernel adds them to the row sums with plain LDS read-modify-writes, no cross-lane work -- and a row's entries
    // are still added one by one in ascending column order (layer l before layer l + 1).  Bit 31 of an index marks the first entry
    // of a layer >= 1 (where a 64-entry group has to be split in two passes).  Inside a layer the entries are sorted by COLUMN
    // (round 4; rounds 2-3: by row): a wavefront's 64 gathers then fall into a 30-40 KB stretch of the panel instead of all over its
    // 1 MiB -- tools/gather_locality_probe.cpp measures 243 against 179 G gathers/s for exactly this change on the bare pattern.
    std::vector<int64_t> breaks((size_t)T, 0);
    auto work = [&](int t) {
        std::vector<uint32_t> pos((size_t)P + 1);
        std::vector<std::vector<uint32_t>> lay((size_t)P), lay0;  // per panel: rows holding more than l entries, then layer starts
        std::vector<std::pair<uint32_t, double>> tmp;
        int64_t mseg = 0, nbreaks = 0;
        for (int64_t s = S * t / T; s < S * (t + 1) / T; ++s) {
            const int64_t r0 = srow[(size_t)s], r1 = srow[(size_t)s + 1], k0 = rowptr[r0];
            uint32_t *off = toff.data() + (size_t)(s * (P + 1));
            std::fill(pos.begin(), pos.end(), 0u);
            for (auto &v : lay) v.clear();
            for (int64_t i = r0; i < r1; ++i) {
                int64_t k = rowptr[i];
                const int64_t e = rowptr[i + 1];
                while (k < e) {                                   // one (row, panel) segment
                    const int64_t j = col[k] >> shift;
                    int64_t len = 0;
                    while (k < e && (col[k] >> shift) == j) { ++k; ++len; }
                    pos[(size_t)j + 1] += (uint32_t)len;
                    std::vector<uint32_t> &L = lay[(size_t)j];
                    if ((int64_t)L.size() < len) L.resize((size_t)len, 0u);
                    for (int64_t q = 0; q < len; ++q) L[(size_t)q]++;
                    mseg = std::max(mseg, len);
                }
            }
            for (int64_t j = 0; j < P; ++j) pos[(size_t)j + 1] += pos[(size_t)j];
            std::memcpy(off, pos.data(), sizeof(uint32_t) * (size_t)(P + 1));
            for (int64_t j = 0; j < P; ++j) {                     // counts -> first slot of each layer inside the tile
                std::vector<uint32_t> &L = lay[(size_t)j];
                uint32_t run = 0;
                for (size_t q = 0; q < L.size(); ++q) { const uint32_t c = L[q]; L[q] = run; run += c; }
                if (!L.empty()) nbreaks += (int64_t)L.size() - 1;
            }
            lay0 = lay;                                           // (the fill below advances the slots: kept for the per-layer sort)
            for (int64_t i = r0; i < r1; ++i) {
                int64_t k = rowptr[i];
                const int64_t e = rowptr[i + 1];
                while (k < e) {
                    const int64_t j = col[k] >> shift;
                    std::vector<uint32_t> &L = lay[(size_t)j];
                    for (int64_t q = 0; k < e && (col[k] >> shift) == j; ++k, ++q) {
                        const size_t o = (size_t)(k0 + pos[(size_t)j] + L[(size_t)q]++);
                        tidx[o] = ((uint32_t)(i - r0) << shift) | ((uint32_t)col[k] & cmask);
                        tval[o] = val[k];
                    }
                }
            }
            for (int64_t j = 0; j < P; ++j) {                     // every layer of every tile: by (column, row); first entry of a layer >= 1 flagged
                const std::vector<uint32_t> &L0 = lay0[(size_t)j], &L1 = lay[(size_t)j];
                for (size_t q = 0; q < L0.size(); ++q) {
                    const size_t b = (size_t)(k0 + pos[(size_t)j] + L0[q]), e = (size_t)(k0 + pos[(size_t)j] + L1[q]);
                    tmp.resize(e - b);
                    for (size_t o = b; o < e; ++o) tmp[o - b] = {tidx[o], tval[o]};
                    std::sort(tmp.begin(), tmp.end(), [&](const std::pair<uint32_t, double> &x, const std::pair<uint32_t, double> &y) {
                        const uint32_t cx = x.first & cmask, cy = y.first & cmask;
                        return cx != cy ? cx < cy : x.first < y.first;      // (equal columns: ascending rows, like the stable device sort)
                    });
                    for (size_t o = b; o < e; ++o) { tidx[o] = tmp[o - b].first; tval[o] = tmp[o - b].second; }
                    if (q > 0 && e > b) tidx[b] |= 0x80000000u;
                }
            }
        }
        maxseg[(size_t)t] = mseg;
        breaks[(size_t)t] = nbreaks;
    };
    if (T == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    A->tl_maxseg = *std::max_element(maxseg.begin(), maxseg.end());
    {   // every layer boundary inside a 64-entry group costs one more LDS pass: matrices made of long (row, panel) segments
        // (dense rows) belong to the LDS-panel / stream kernels
        int64_t nb = 0;
        for (int64_t b : breaks) nb += b;
        if (nb * 8 > nnz) return skip("layer boundaries (dense rows)");
    }
    hipError_t err = hipSuccess;
    auto upload = [&](void **dst, const void *src, size_t bytes) {
        if (err != hipSuccess) return;
        err = dev_malloc(c, dst, std::max<size_t>(bytes + 64, 8));   // (+64: the streams are read in whole dwords / qwords only, slack for safety)
        if (err == hipSuccess && bytes) err = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    };
    upload((void **)&A->d_tloff, toff.data(), sizeof(uint32_t) * toff.size());
    upload((void **)&A->d_tlidx, tidx.data(), sizeof(uint32_t) * tidx.size());
    upload((void **)&A->d_tlval, tval.data(), sizeof(double) * tval.size());
    if (err != hipSuccess) return fail(SLA_ERR_ALLOC, std::string("tile form upload: ") + hipGetErrorString(err));
    return finish();
}

}  // namespace sla
