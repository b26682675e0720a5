/*
 * oracle/nns_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, scalar, one thread) of the reference's descriptor
 * matching path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path
 * (cslam_amd/ -> libcslam_hip.so) never does.
 *
 * What it restates (paths relative to the reference checkout):
 *   cslam/nns_matching.py:42-61   NearestNeighborsMatching.search
 *       for i < n: sim[i] = 1 - scipy.spatial.distance.cosine(query, data[i])
 *       ns = argsort(sim)[::-1][:k]
 *   scipy.spatial.distance.cosine -> correlation(u, v, centered=False)
 *   (third-party, scipy 1.15.3 in the build container; not under the reference):
 *       uv = u.v ; uu = u.u ; vv = v.v
 *       dist = clip(1 - uv / sqrt(uu * vv), 0, 2)
 *   cslam/nns_matching.py:63-76   search_best = search(q, 1)
 *   cslam/loop_closure_sparse_matching.py:74-92  causal intra search (row limit)
 *
 * Parity pin: checked against tests/golden/nns_*.npz, which were produced by
 * importing the real reference in the build container (oracle/gen_golden.py).
 *
 * Arithmetic contract of THIS restatement (and of the HIP product):
 *   - bank rows are float32 (the reference stores float32: nns_matching.py:21,39);
 *   - queries are float32 or float64;
 *   - all three dot products are accumulated in float64 from the exact
 *     float32/float64 inputs, in index order 0..d-1.  The reference accumulates
 *     in float32 when the query is float32 (numpy sdot; order is BLAS-defined) and
 *     on numpy >= 2 also does the divide/subtract in float32, so its scores differ
 *     from these by <= ~3e-7; with float64 queries it is float64 throughout and the
 *     difference is ~1e-16.  north_star's gate is 1e-5 on scores;
 *   - ordering: descending similarity; NaN (zero-norm vector) ranks first, as it
 *     does after the reference's argsort()[::-1]; exact ties are broken towards
 *     the LARGER bank row index (what a stable argsort followed by [::-1] gives;
 *     the reference's own tie order is numpy-sort-implementation defined).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

static double sim_from_dots(double uv, double uu, double vv) {
    double dist = 1.0 - uv / sqrt(uu * vv);
    /* np.clip(dist, 0, 2): NaN propagates */
    if (dist < 0.0) dist = 0.0;
    if (dist > 2.0) dist = 2.0;
    return 1.0 - dist;
}

/* similarity of one f32 query against one f32 row */
ORACLE_API double oracle_cosine_sim_f32(const float *q, const float *row, int d) {
    double uv = 0.0, uu = 0.0, vv = 0.0;
    for (int i = 0; i < d; ++i) {
        double a = (double)q[i], b = (double)row[i];
        uv += a * b; uu += a * a; vv += b * b;
    }
    return sim_from_dots(uv, uu, vv);
}

ORACLE_API double oracle_cosine_sim_f64(const double *q, const float *row, int d) {
    double uv = 0.0, uu = 0.0, vv = 0.0;
    for (int i = 0; i < d; ++i) {
        double a = q[i], b = (double)row[i];
        uv += a * b; uu += a * a; vv += b * b;
    }
    return sim_from_dots(uv, uu, vv);
}

typedef struct { double s; int64_t i; } cand_t;

/* "a ranks before b": NaN first, then larger score, then larger index */
static int ranks_before(double sa, int64_t ia, double sb, int64_t ib) {
    int na = isnan(sa), nb = isnan(sb);
    if (na || nb) {
        if (na && nb) return ia > ib;
        return na;
    }
    if (sa != sb) return sa > sb;
    return ia > ib;
}

static int cmp_cand(const void *pa, const void *pb) {
    const cand_t *a = (const cand_t *)pa, *b = (const cand_t *)pb;
    if (a->i == b->i) return 0;
    return ranks_before(a->s, a->i, b->s, b->i) ? -1 : 1;
}

/*
 * All similarities of nq queries against rows [0, limit_q) of the bank, then the
 * first k of the descending order.  q_is_f64: 0 -> queries are float32, 1 -> float64.
 * row_limit: NULL -> every query sees all n rows; else query j sees rows < row_limit[j]
 * (the causal mask of gdlcd.py:157-160: a keyframe is searched before it is added).
 * out_idx [nq,k] int64 (row numbers, -1 padded), out_sim [nq,k] f64 (NaN padded),
 * out_cnt[nq] = min(k, limit).
 * Returns 0, or -1 on allocation failure.
 */
ORACLE_API int oracle_nns_search(const float *bank, int64_t n, int d,
                                 const void *queries, int q_is_f64, int64_t nq, int k,
                                 const int64_t *row_limit,
                                 int64_t *out_idx, double *out_sim, int32_t *out_cnt) {
    cand_t *c = (cand_t *)malloc(sizeof(cand_t) * (size_t)(n > 0 ? n : 1));
    if (!c) return -1;
    for (int64_t j = 0; j < nq; ++j) {
        int64_t lim = row_limit ? row_limit[j] : n;
        if (lim > n) lim = n;
        if (lim < 0) lim = 0;
        for (int64_t i = 0; i < lim; ++i) {
            const float *row = bank + (size_t)i * d;
            c[i].i = i;
            c[i].s = q_is_f64
                ? oracle_cosine_sim_f64((const double *)queries + (size_t)j * d, row, d)
                : oracle_cosine_sim_f32((const float *)queries + (size_t)j * d, row, d);
        }
        /* the reference does a full argsort (nns_matching.py:60); so do we */
        qsort(c, (size_t)lim, sizeof(cand_t), cmp_cand);
        int cnt = (int)(lim < k ? lim : k);
        out_cnt[j] = cnt;
        for (int t = 0; t < k; ++t) {
            out_idx[(size_t)j * k + t] = t < cnt ? c[t].i : -1;
            out_sim[(size_t)j * k + t] = t < cnt ? c[t].s : NAN;
        }
    }
    free(c);
    return 0;
}

/*
 * Faithful-order variant used as bench.py's cpu_baseline ("port"): same per-row
 * loop as nns_matching.py:55-58 (three dots per row, one row at a time), full
 * score vector, full sort -- i.e. the reference's algorithm without the Python
 * interpreter.  Identical results to oracle_nns_search.
 */
ORACLE_API int oracle_nns_search_faithful(const float *bank, int64_t n, int d,
                                          const float *queries, int64_t nq, int k,
                                          int64_t *out_idx, double *out_sim) {
    int32_t *cnt = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nq > 0 ? nq : 1));
    if (!cnt) return -1;
    int rc = oracle_nns_search(bank, n, d, queries, 0, nq, k, NULL, out_idx, out_sim, cnt);
    free(cnt);
    return rc;
}
