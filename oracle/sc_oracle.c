/* oracle/sc_oracle.c -- TEST INFRASTRUCTURE, not product code.
 *
 * Plain-C restatement of the reference's lidar ScanContext matcher, used only by tests/,
 * __graft_entry__.smoke() and tools/ baselines as the checker of the HIP path.
 *   oracle_sc_ringkey   <- cslam/lidar_pr/scancontext_utils.py:78-79  (sc2rk = np.mean(sc, axis=1))
 *   oracle_sc_distance  <- cslam/lidar_pr/scancontext_utils.py:81-113 (distance_sc)
 *   oracle_sc_search    <- cslam/lidar_pr/scancontext_matching.py:44-87 (search: KD-tree k-NN on ring
 *                          keys, then the best shifted-cosine distance among the candidates)
 * Pinned by tests/golden/sc_g9.npz (oracle/gen_golden_sc.py runs the real reference):
 * ring keys bit-identical (numpy's pairwise summation order is restated), candidate sets, best
 * item and yaw identical, distances within 1e-12 (the reference's dots go through BLAS ddot whose
 * summation order is not specified; here every dot is a left-to-right fma chain, the order the
 * HIP kernels also use so that GPU == oracle bit for bit).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define API __attribute__((visibility("default")))

/* numpy add.reduce over a contiguous run of n doubles (pairwise_sum, n <= 128 branch) */
static double np_sum(const double *a, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r[8];
    int i;
    for (int k = 0; k < 8; k++) r[k] = a[k];
    for (i = 8; i < n - (n % 8); i += 8)
        for (int k = 0; k < 8; k++) r[k] += a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

API int oracle_sc_ringkey(const double *sc, int R, int S, double *rk) {
    if (S > 128) return -1;
    for (int r = 0; r < R; r++) rk[r] = (0.0 + np_sum(sc + (int64_t)r * S, S)) / (double)S;
    return 0;
}

static void col_norms(const double *sc, int R, int S, double *nrm, int *any) {
    for (int j = 0; j < S; j++) {
        double s = 0.0;
        int a = 0;
        for (int r = 0; r < R; r++) {
            double v = sc[(int64_t)r * S + j];
            s = fma(v, v, s);
            a |= (v != 0.0);
        }
        nrm[j] = sqrt(s);
        any[j] = a;
    }
}

/* sc1 = candidate (the one that is rolled), sc2 = query */
API int oracle_sc_distance(const double *sc1, const double *sc2, int R, int S, double *dist, int *yaw) {
    double *n1 = malloc(sizeof(double) * S), *n2 = malloc(sizeof(double) * S);
    int *a1 = malloc(sizeof(int) * S), *a2 = malloc(sizeof(int) * S);
    col_norms(sc1, R, S, n1, a1);
    col_norms(sc2, R, S, n2, a2);
    double best = -INFINITY;
    int best_i = 0;
    for (int i = 0; i < S; i++) {
        int s = (i + 1) % S;                         /* cumulative np.roll by one per iteration */
        double sum = 0.0;
        int engaged = 0;
        for (int j = 0; j < S; j++) {
            int c = ((j - s) % S + S) % S;           /* rolled[:, j] = sc1[:, j - s] */
            if (!a1[c] || !a2[j]) continue;
            double d = 0.0;
            for (int r = 0; r < R; r++) d = fma(sc1[(int64_t)r * S + c], sc2[(int64_t)r * S + j], d);
            sum = sum + d / (n1[c] * n2[j]);
            engaged++;
        }
        double sim = engaged ? sum / (double)engaged : 0.0;
        if (sim > best) { best = sim; best_i = i; }  /* np.argmax: first maximum */
    }
    *dist = 1.0 - best;
    *yaw = best_i + 1;
    free(n1); free(n2); free(a1); free(a2);
    return 0;
}

/* bank [n, R*S]; queries [nq, R*S]; row_limit[nq] or NULL (query j sees rows < row_limit[j]).
 * cand [nq, ncand] (-1 = missing), cdist/cyaw [nq, ncand]; best_idx -1 when no candidate beats
 * distance 1.0 (the reference then answers items[0] with similarity 0.0). */
API int oracle_sc_search(const double *bank, int64_t n, int R, int S, const double *q, int64_t nq, int ncand,
                         const int64_t *row_limit, int64_t *best_idx, double *best_sim, int *best_yaw,
                         int64_t *cand, double *cdist, int *cyaw) {
    int L = R * S;
    double *rk = malloc(sizeof(double) * (size_t)(n > 0 ? n : 1) * R);
    double *qrk = malloc(sizeof(double) * R);
    double *d2 = malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; i++) oracle_sc_ringkey(bank + i * L, R, S, rk + i * R);
    for (int64_t j = 0; j < nq; j++) {
        int64_t lim = row_limit ? row_limit[j] : n;
        if (lim > n) lim = n;
        oracle_sc_ringkey(q + j * L, R, S, qrk);
        for (int64_t i = 0; i < lim; i++) {
            double s = 0.0;
            for (int r = 0; r < R; r++) {
                double t = rk[i * R + r] - qrk[r];
                s = fma(t, t, s);
            }
            d2[i] = s;
        }
        double nn_dist = 1.0;
        int64_t nn_idx = -1;
        int nn_yaw = 0;
        for (int c = 0; c < ncand; c++) {            /* selection of the c-th nearest ring key */
            int64_t arg = -1;
            for (int64_t i = 0; i < lim; i++)
                if (d2[i] >= 0.0 && (arg < 0 || d2[i] < d2[arg])) arg = i;
            if (cand) cand[j * ncand + c] = arg;
            if (arg < 0) {
                if (cdist) { cdist[j * ncand + c] = 1.0; cyaw[j * ncand + c] = 0; }
                continue;
            }
            d2[arg] = -1.0;                          /* taken */
            double dist;
            int yaw;
            oracle_sc_distance(bank + arg * L, q + j * L, R, S, &dist, &yaw);
            if (cdist) { cdist[j * ncand + c] = dist; cyaw[j * ncand + c] = yaw; }
            if (dist < nn_dist) { nn_dist = dist; nn_idx = arg; nn_yaw = yaw; }
        }
        best_idx[j] = nn_idx;
        best_sim[j] = nn_idx >= 0 ? 1.0 - nn_dist : 0.0;
        best_yaw[j] = nn_idx >= 0 ? nn_yaw : 0;
    }
    free(rk); free(qrk); free(d2);
    return 0;
}

/* ---- descriptor: oracle_ptcloud2sc <- cslam/lidar_pr/scancontext_utils.py:10-75 (xy2theta, pt2rs,
 * ptcloud2sc) for float64 clouds [n,3].  Every bin keeps the maximum of point[2] + 2.0 over the first
 * 500 points that fall in it (in cloud order; the reference's `enough_large` storage) and 0.0 for the
 * unused storage slots.  np.divmod is restated from numpy's npy_divmod. */
static double np_floordiv(double a, double b) {
    double mod = fmod(a, b);
    double div = (a - mod) / b;
    if (mod != 0.0 && ((b < 0) != (mod < 0))) div -= 1.0;
    if (div != 0.0) {
        double fl = floor(div);
        if (div - fl > 0.5) fl += 1.0;
        return fl;
    }
    return copysign(0.0, a / b);
}

API int oracle_ptcloud2sc(const double *pts, int64_t n, int R, int S, double max_length, double *sc) {
    const int cap = 500;
    const double gap_ring = max_length / R, gap_sector = 360.0 / S;
    const double k = 180.0 / 3.141592653589793;
    int *cnt = calloc((size_t)R * S, sizeof(int));
    for (int i = 0; i < R * S; i++) sc[i] = 0.0;
    for (int64_t i = 0; i < n; i++) {
        double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        if (isnan(x) || isnan(y) || isnan(z)) continue;
        double h = z + 2.0;
        if (x == 0.0) x = 0.001;
        if (y == 0.0) y = 0.001;
        double theta;
        if (x >= 0 && y >= 0) theta = k * atan(y / x);
        else if (x < 0 && y >= 0) theta = 180.0 - (k * atan(y / (-x)));
        else if (x < 0 && y < 0) theta = 180.0 + (k * atan(y / x));
        else theta = 360.0 - (k * atan((-y) / x));
        double far = sqrt(x * x + y * y);
        double ring = np_floordiv(far, gap_ring), sector = np_floordiv(theta, gap_sector);
        if (ring >= R) ring = R - 1;
        int ir = (int)ring, is = (int)sector;
        if (is < 0 || is >= S) { free(cnt); return -2; }     /* the reference raises IndexError */
        int b = ir * S + is;
        if (cnt[b] >= cap) continue;
        if (cnt[b] == 0 || h > sc[b]) sc[b] = (cnt[b] == 0) ? h : (h > sc[b] ? h : sc[b]);
        cnt[b]++;
    }
    for (int b = 0; b < R * S; b++)
        if (cnt[b] < cap && sc[b] < 0.0) sc[b] = 0.0;       /* an unused storage slot holds 0.0 */
    free(cnt);
    return 0;
}
