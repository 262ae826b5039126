/*
 * sla_wlgen.c -- fast host-side assembly of BASELINE config 3's synthetic matrix:
 * the symmetrised random pattern of sla_amd/workloads.py::random_spd.  Bench / test INPUT generation only
 * (plain C, no GPU, not part of libsla_hip.so): numpy's two global argsorts of 3.2e8 keys take ~6 minutes at
 * n = 10 M, a counting sort by row + tiny per-row sorts takes seconds.  The result is bit-identical to the
 * numpy construction (tests/test_cabi_and_host.py compares them).
 *
 * Input: the numpy PCG64 draws c[i] (column pick of row i / k) and v[i] (U(-1,1)), i < n * k.
 * Matrix: A = (R + R^T) / 2 on the union pattern (duplicates summed in list order: all (r, c) picks first, then
 * the mirrored (c, r) ones), diagonal = 1 + sum_j |a_ij| accumulated in ascending column order.
 */
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <omp.h>

typedef struct { int64_t col; double val; } ent_t;

static void sort_row(ent_t *e, int64_t cnt, ent_t *tmp) {
    /* stable by column: insertion sort on runs of 32, then bottom-up merges (rows hold ~2k entries: 33 ... 2000) */
    for (int64_t lo = 0; lo < cnt; lo += 32) {
        const int64_t hi = lo + 32 < cnt ? lo + 32 : cnt;
        for (int64_t i = lo + 1; i < hi; ++i) {
            ent_t t = e[i];
            int64_t j = i - 1;
            while (j >= lo && e[j].col > t.col) { e[j + 1] = e[j]; --j; }
            e[j + 1] = t;
        }
    }
    ent_t *src = e, *dst = tmp;
    for (int64_t w = 32; w < cnt; w *= 2) {
        for (int64_t lo = 0; lo < cnt; lo += 2 * w) {
            const int64_t mid = lo + w < cnt ? lo + w : cnt, hi = lo + 2 * w < cnt ? lo + 2 * w : cnt;
            int64_t a = lo, b = mid, o = lo;
            while (a < mid && b < hi) dst[o++] = (src[b].col < src[a].col) ? src[b++] : src[a++];
            while (a < mid) dst[o++] = src[a++];
            while (b < hi) dst[o++] = src[b++];
        }
        ent_t *sw = src; src = dst; dst = sw;
    }
    if (src != e) memcpy(e, src, (size_t)cnt * sizeof(ent_t));
}

/* returns nnz, or -1 on allocation failure.  rowptr: n + 1; col / val: capacity 2 * n * k + n. */
int64_t sla_wl_random_spd(int64_t n, int64_t k, const int64_t *c, const double *v, int64_t *rowptr, int64_t *col, double *val) {
    const int64_t picks = n * k;
    const int dbg = getenv("SLA_WL_DEBUG") != NULL;
    double t0 = omp_get_wtime();
#define PHASE(name) do { if (dbg) { double t1 = omp_get_wtime(); fprintf(stderr, "wlgen %s %.2f s\n", name, t1 - t0); t0 = t1; } } while (0)
    int64_t *start = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    if (!start) return -1;
    /* The mirrored halves land in random rows: a single thread would pay one DRAM round trip per pick.  Every
     * thread therefore owns a contiguous range of destination rows and scans the whole pick list for them (the
     * list is read sequentially; the random accesses of the threads overlap). */
#pragma omp parallel
    {
        const int nt = omp_get_num_threads(), t = omp_get_thread_num();
        const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
        for (int64_t r = lo; r < hi; ++r)
            for (int64_t i = r * k; i < (r + 1) * k; ++i) start[r + 1] += (c[i] != r);
        for (int64_t i = 0; i < picks; ++i) {
            const int64_t d = c[i];
            if (d >= lo && d < hi && d != i / k) start[d + 1]++;
        }
    }
    PHASE("count");
    for (int64_t i = 0; i < n; ++i) start[i + 1] += start[i];
    const int64_t total = start[n];
    ent_t *e = (ent_t *)malloc((size_t)(total ? total : 1) * sizeof(ent_t));
    int64_t *fill = (int64_t *)malloc((size_t)(n ? n : 1) * sizeof(int64_t));
    if (!e || !fill) { free(start); free(e); free(fill); return -1; }
    memcpy(fill, start, (size_t)n * sizeof(int64_t));
#pragma omp parallel
    {
        const int nt = omp_get_num_threads(), t = omp_get_thread_num();
        const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
        for (int64_t r = lo; r < hi; ++r)           /* the (r, c) halves, list order */
            for (int64_t i = r * k; i < (r + 1) * k; ++i)
                if (c[i] != r) { ent_t w = {c[i], 0.5 * v[i]}; e[fill[r]++] = w; }
        for (int64_t i = 0; i < picks; ++i) {       /* then the mirrored (c, r) halves, list order */
            const int64_t d = c[i];
            if (d >= lo && d < hi) {
                const int64_t r = i / k;
                if (d != r) { ent_t w = {r, 0.5 * v[i]}; e[fill[d]++] = w; }
            }
        }
    }
    free(fill);
    PHASE("fill");
    /* per row: stable sort by column, merge duplicates (left-to-right sums), diagonal = 1 + sum |a_ij| */
    int64_t *cnt = (int64_t *)malloc((size_t)(n ? n : 1) * sizeof(int64_t));
    double *diag = (double *)malloc((size_t)(n ? n : 1) * sizeof(double));
    if (!cnt || !diag) { free(start); free(e); free(cnt); free(diag); return -1; }
    int alloc_failed = 0;
    int64_t longest = 0;
    for (int64_t r = 0; r < n; ++r) if (start[r + 1] - start[r] > longest) longest = start[r + 1] - start[r];
#pragma omp parallel
    {
    ent_t *tmp = (ent_t *)malloc((size_t)(longest ? longest : 1) * sizeof(ent_t));   /* NULL: rows <= 32 still sort */
#pragma omp for schedule(static, 4096)
    for (int64_t r = 0; r < n; ++r) {
        ent_t *row = e + start[r];
        const int64_t m = start[r + 1] - start[r];
        if (m > 32 && !tmp) { alloc_failed = 1; continue; }
        sort_row(row, m, tmp);
        int64_t w = 0;
        for (int64_t i = 0; i < m;) {
            int64_t j = i + 1;
            double s = row[i].val;
            while (j < m && row[j].col == row[i].col) s += row[j++].val;
            row[w].col = row[i].col;
            row[w].val = s;
            ++w;
            i = j;
        }
        double a = 0.0;
        for (int64_t i = 0; i < w; ++i) a += fabs(row[i].val);
        cnt[r] = w;
        diag[r] = 1.0 + a;
    }
    free(tmp);
    }
    if (alloc_failed) { free(start); free(e); free(cnt); free(diag); return -1; }
    PHASE("sort");
    /* emit rows with the diagonal entry merged in at its sorted place */
    rowptr[0] = 0;
    for (int64_t r = 0; r < n; ++r) rowptr[r + 1] = rowptr[r] + cnt[r] + 1;
#pragma omp parallel for schedule(static, 4096)
    for (int64_t r = 0; r < n; ++r) {
        const ent_t *row = e + start[r];
        int64_t o = rowptr[r];
        int64_t i = 0;
        for (; i < cnt[r] && row[i].col < r; ++i, ++o) { col[o] = row[i].col; val[o] = row[i].val; }
        col[o] = r; val[o] = diag[r]; ++o;
        for (; i < cnt[r]; ++i, ++o) { col[o] = row[i].col; val[o] = row[i].val; }
    }
    PHASE("emit");
    const int64_t nnz = rowptr[n];
    free(start); free(e); free(cnt); free(diag);
    return nnz;
}


void sla_wl_free(void *p) { free(p); }

/* The same matrix, rows [rb, re) only (a rank of the row-sharded bench builds its own slab; global column ids):
 * rowptr (caller's, re - rb + 1 entries) starts at 0; *col_out / *val_out are malloc'ed here (release with sla_wl_free).
 * Returns the slab's nnz, -1 on allocation failure.  `threads` > 0 sets the OpenMP team (N ranks of one node must share the
 * host's cores: an oversubscribed libgomp team spins in its barriers and takes minutes).
 * Bit-identical to rows [rb, re) of sla_wl_random_spd (tests/test_cabi_and_host.py). */
int64_t sla_wl_random_spd_rows(int64_t n, int64_t k, const int64_t *c, const double *v, int64_t rb, int64_t re, int64_t *rowptr,
                               int64_t **col_out, double **val_out, int threads) {
    const int64_t picks = n * k, rows = re - rb;
    if (threads > 0) omp_set_num_threads(threads);
    int64_t *start = (int64_t *)calloc((size_t)rows + 1, sizeof(int64_t));
    if (!start) return -1;
#pragma omp parallel
    {
        const int nt = omp_get_num_threads(), t = omp_get_thread_num();
        const int64_t lo = rb + rows * t / nt, hi = rb + rows * (t + 1) / nt;
        for (int64_t r = lo; r < hi; ++r)
            for (int64_t i = r * k; i < (r + 1) * k; ++i) start[r - rb + 1] += (c[i] != r);
        for (int64_t i = 0; i < picks; ++i) {
            const int64_t d = c[i];
            if (d >= lo && d < hi && d != i / k) start[d - rb + 1]++;
        }
    }
    for (int64_t i = 0; i < rows; ++i) start[i + 1] += start[i];
    const int64_t total = start[rows];
    ent_t *e = (ent_t *)malloc((size_t)(total ? total : 1) * sizeof(ent_t));
    int64_t *fill = (int64_t *)malloc((size_t)(rows ? rows : 1) * sizeof(int64_t));
    int64_t *cnt = (int64_t *)malloc((size_t)(rows ? rows : 1) * sizeof(int64_t));
    double *diag = (double *)malloc((size_t)(rows ? rows : 1) * sizeof(double));
    if (!e || !fill || !cnt || !diag) { free(start); free(e); free(fill); free(cnt); free(diag); return -1; }
    memcpy(fill, start, (size_t)rows * sizeof(int64_t));
#pragma omp parallel
    {
        const int nt = omp_get_num_threads(), t = omp_get_thread_num();
        const int64_t lo = rb + rows * t / nt, hi = rb + rows * (t + 1) / nt;
        for (int64_t r = lo; r < hi; ++r)           /* the (r, c) halves, list order */
            for (int64_t i = r * k; i < (r + 1) * k; ++i)
                if (c[i] != r) { ent_t w = {c[i], 0.5 * v[i]}; e[fill[r - rb]++] = w; }
        for (int64_t i = 0; i < picks; ++i) {       /* then the mirrored (c, r) halves, list order */
            const int64_t d = c[i];
            if (d >= lo && d < hi) {
                const int64_t r = i / k;
                if (d != r) { ent_t w = {r, 0.5 * v[i]}; e[fill[d - rb]++] = w; }
            }
        }
    }
    free(fill);
    int alloc_failed = 0;
    int64_t longest = 0;
    for (int64_t r = 0; r < rows; ++r) if (start[r + 1] - start[r] > longest) longest = start[r + 1] - start[r];
#pragma omp parallel
    {
    ent_t *tmp = (ent_t *)malloc((size_t)(longest ? longest : 1) * sizeof(ent_t));
#pragma omp for schedule(static, 4096)
    for (int64_t r = 0; r < rows; ++r) {
        ent_t *row = e + start[r];
        const int64_t m = start[r + 1] - start[r];
        if (m > 32 && !tmp) { alloc_failed = 1; continue; }
        sort_row(row, m, tmp);
        int64_t w = 0;
        for (int64_t i = 0; i < m;) {
            int64_t j = i + 1;
            double s = row[i].val;
            while (j < m && row[j].col == row[i].col) s += row[j++].val;
            row[w].col = row[i].col;
            row[w].val = s;
            ++w;
            i = j;
        }
        double a = 0.0;
        for (int64_t i = 0; i < w; ++i) a += fabs(row[i].val);
        cnt[r] = w;
        diag[r] = 1.0 + a;
    }
    free(tmp);
    }
    if (alloc_failed) { free(start); free(e); free(cnt); free(diag); return -1; }
    rowptr[0] = 0;
    for (int64_t r = 0; r < rows; ++r) rowptr[r + 1] = rowptr[r] + cnt[r] + 1;
    int64_t *col = (int64_t *)malloc((size_t)(rowptr[rows] ? rowptr[rows] : 1) * sizeof(int64_t));
    double *val = (double *)malloc((size_t)(rowptr[rows] ? rowptr[rows] : 1) * sizeof(double));
    if (!col || !val) { free(start); free(e); free(cnt); free(diag); free(col); free(val); return -1; }
    *col_out = col;
    *val_out = val;
#pragma omp parallel for schedule(static, 4096)
    for (int64_t r = 0; r < rows; ++r) {
        const ent_t *row = e + start[r];
        const int64_t g = rb + r;
        int64_t o = rowptr[r];
        int64_t i = 0;
        for (; i < cnt[r] && row[i].col < g; ++i, ++o) { col[o] = row[i].col; val[o] = row[i].val; }
        col[o] = g; val[o] = diag[r]; ++o;
        for (; i < cnt[r]; ++i, ++o) { col[o] = row[i].col; val[o] = row[i].val; }
    }
    const int64_t nnz = rowptr[rows];
    free(start); free(e); free(cnt); free(diag);
    return nnz;
}
