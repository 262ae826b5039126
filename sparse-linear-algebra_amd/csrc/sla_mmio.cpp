// sla_mmio.cpp -- MatrixMarket ingestion with the loader semantics of the reference's test/Perf.hs:20-45:
// coordinate real general files, 1-based indices shifted to 0-based (toD3), entries fed to fromListSM in
// file order (so a repeated (i,j) resolves to the LAST one), no symmetric expansion; array files give the
// dense right-hand side (buildSparse only drops |x| <= 1e-12, which is a no-op for a dense device vector).
#include <cctype>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "sla_internal.hpp"

using namespace sla;

namespace {

struct File {
    FILE *f;
    explicit File(const char *p) : f(fopen(p, "r")) {}
    ~File() { if (f) fclose(f); }
};

// reads the banner + skips comments; returns the first non-comment line in `line`
int read_header(FILE *f, const char *path, std::string &banner, char *line, size_t cap) {
    if (!fgets(line, (int)cap, f)) return fail(SLA_ERR_INVALID, std::string("empty MatrixMarket file: ") + path);
    banner = line;
    // (the format is case-insensitive: `%%MatrixMarket MATRIX Coordinate Real General` is legal)
    for (size_t i = 14; i < banner.size(); ++i) banner[i] = (char)tolower((unsigned char)banner[i]);
    if (banner.compare(0, 14, "%%MatrixMarket") != 0) return fail(SLA_ERR_INVALID, std::string("missing %%MatrixMarket banner: ") + path);
    do {
        if (!fgets(line, (int)cap, f)) return fail(SLA_ERR_INVALID, std::string("truncated MatrixMarket header: ") + path);
    } while (line[0] == '%' || line[0] == '\n' || line[0] == '\r');
    return SLA_OK;
}

}  // namespace

extern "C" {

int sla_csr_from_matrix_market(sla_ctx_t c, const char *path, int dup_policy, sla_csr_t *out) {
    if (c && !c->kids.empty()) return sla::m_csr_from_matrix_market(c, path, dup_policy, out);
    return no_throw("sla_csr_from_matrix_market", [&]() -> int {
        if (!c || !path || !out) return fail(SLA_ERR_INVALID, "sla_csr_from_matrix_market: null argument");
        File fl(path);
        if (!fl.f) return fail(SLA_ERR_INVALID, std::string("cannot open ") + path + ": " + strerror(errno));
        std::string banner;
        char line[1024];
        SLA_TRY(read_header(fl.f, path, banner, line, sizeof line));
        if (banner.find("coordinate") == std::string::npos || banner.find("general") == std::string::npos ||
            (banner.find("real") == std::string::npos && banner.find("integer") == std::string::npos))
            return fail(SLA_ERR_INVALID, "only `matrix coordinate real|integer general` files are supported (test/Perf.hs reads RMatrix)");
        long long m = 0, n = 0, nz = 0;
        if (sscanf(line, "%lld %lld %lld", &m, &n, &nz) != 3 || m < 0 || n < 0 || nz < 0)
            return fail(SLA_ERR_INVALID, std::string("bad size line in ") + path);
        std::vector<int64_t> r((size_t)nz), cc((size_t)nz);
        std::vector<double> v((size_t)nz);
        for (long long k = 0; k < nz; ++k) {
            long long i, j;
            double x;
            if (fscanf(fl.f, "%lld %lld %lf", &i, &j, &x) != 3) return fail(SLA_ERR_INVALID, std::string("truncated entry list in ") + path);
            r[(size_t)k] = i - 1;  // toD3 (i, j, x) = (i - 1, j - 1, toRealFloat x)
            cc[(size_t)k] = j - 1;
            v[(size_t)k] = x;
        }
        return sla_csr_from_coo(c, m, n, nz, r.data(), cc.data(), v.data(), dup_policy, out);
    });
}

int sla_vec_from_matrix_market(sla_ctx_t c, const char *path, sla_vec_t *out) {
    if (c && !c->kids.empty()) return sla::m_vec_from_matrix_market(c, path, out);
    return no_throw("sla_vec_from_matrix_market", [&]() -> int {
        if (!c || !path || !out) return fail(SLA_ERR_INVALID, "sla_vec_from_matrix_market: null argument");
        File fl(path);
        if (!fl.f) return fail(SLA_ERR_INVALID, std::string("cannot open ") + path + ": " + strerror(errno));
        std::string banner;
        char line[1024];
        SLA_TRY(read_header(fl.f, path, banner, line, sizeof line));
        if (banner.find("array") == std::string::npos) return fail(SLA_ERR_INVALID, "expected a `matrix array` file");
        long long m = 0, n = 0;
        if (sscanf(line, "%lld %lld", &m, &n) != 2 || m < 0 || n < 0) return fail(SLA_ERR_INVALID, std::string("bad size line in ") + path);
        std::vector<double> v((size_t)(m * n));
        for (size_t k = 0; k < v.size(); ++k)
            if (fscanf(fl.f, "%lf", &v[k]) != 1) return fail(SLA_ERR_INVALID, std::string("truncated array in ") + path);
        return sla_vec_create(c, m * n, v.data(), out);
    });
}

}  // extern "C"
