/* Plain-C client of include/cslam_hip.h: proves the drop-in boundary is a real C ABI (no Python,
 * no torch, no C++ types).  Built and run by tests/test_c_client_gpu.py on the GPU box:
 *   gcc -std=c11 -Iinclude tests/c_abi/abi_smoke.c -Lcslam_amd -lcslam_hip -lm -o abi_smoke
 * Fills a 2000 x 96 bank, queries perturbed copies of 40 rows (float32 and float64), checks that
 * every query finds its source row first with similarity > 0.99, in scan and MFMA mode. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "cslam_hip.h"

#define N 2000
#define D 96
#define NQ 40
#define K 3

static unsigned long long s = 88172645463325252ULL;
static double rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0 - 0.5; }

int main(void) {
    int ndev = 0;
    if (cslam_device_count(&ndev) != CSLAM_OK || ndev < 1) { printf("no device: %s\n", cslam_last_error()); return 2; }
    float *bank = malloc(sizeof(float) * N * D);
    float *q32 = malloc(sizeof(float) * NQ * D);
    double *q64 = malloc(sizeof(double) * NQ * D);
    for (int i = 0; i < N * D; ++i) bank[i] = (float)rnd();
    for (int j = 0; j < NQ; ++j)
        for (int c = 0; c < D; ++c) {
            double v = bank[(size_t)(j * 37 % N) * D + c] * 2.5 + 0.01 * rnd();   /* scaled: cosine ignores norms */
            q32[j * D + c] = (float)v; q64[j * D + c] = v;
        }
    cslam_bank_t *b = NULL;
    if (cslam_bank_create(0, D, 0, &b) != CSLAM_OK) { printf("create: %s\n", cslam_last_error()); return 1; }
    if (cslam_bank_add_host(b, bank, CSLAM_F32, N) != CSLAM_OK) { printf("add: %s\n", cslam_last_error()); return 1; }
    int64_t n = 0; int dim = 0;
    cslam_bank_size(b, &n, &dim);
    if (n != N || dim != D) { printf("size mismatch\n"); return 1; }
    int64_t idx[NQ * K]; double sim[NQ * K]; int32_t cnt[NQ];
    int modes[2] = {CSLAM_MODE_SCAN, CSLAM_MODE_MFMA};
    for (int m = 0; m < 2; ++m)
        for (int f64 = 0; f64 < 2; ++f64) {
            int rc = cslam_bank_search_host(b, f64 ? (const void *)q64 : (const void *)q32, f64 ? CSLAM_F64 : CSLAM_F32,
                                            NQ, K, NULL, modes[m], idx, sim, cnt);
            if (rc != CSLAM_OK) { printf("search: %s\n", cslam_last_error()); return 1; }
            for (int j = 0; j < NQ; ++j) {
                if (cnt[j] != K || idx[j * K] != j * 37 % N || !(sim[j * K] > 0.99) || !(sim[j * K] >= sim[j * K + 1])) {
                    printf("mode %d f64 %d query %d: idx %lld sim %f cnt %d\n", modes[m], f64, j, (long long)idx[j * K],
                           sim[j * K], cnt[j]);
                    return 1;
                }
            }
        }
    /* error path: dimension mismatch must be reported, not crash */
    if (cslam_bank_search_host(b, q32, 7, NQ, K, NULL, 0, idx, sim, cnt) != CSLAM_E_INVALID) { printf("bad dtype accepted\n"); return 1; }
    cslam_bank_destroy(b);
    free(bank); free(q32); free(q64);
    /* MAC's Fiedler pair in one call (mac.py:35-59) on a cycle of RING nodes = an odometry chain closed by one loop edge:
     * lambda_2 = 2 - 2 cos(2 pi / RING) in closed form */
    {
        enum { RING = 500 };
        int64_t indptr[RING + 1]; int32_t indices[3 * RING]; double data[3 * RING], v[RING], lam = 0.0; int iters = 0;
        int64_t p = 0;
        for (int i = 0; i < RING; ++i) {
            int nb[3] = {(i + RING - 1) % RING, i, (i + 1) % RING};
            for (int a = 0; a < 3; ++a) for (int c = a + 1; c < 3; ++c) if (nb[c] < nb[a]) { int t = nb[a]; nb[a] = nb[c]; nb[c] = t; }
            indptr[i] = p;
            for (int a = 0; a < 3; ++a) { indices[p] = nb[a]; data[p] = nb[a] == i ? 2.0 : -1.0; ++p; }
        }
        indptr[RING] = p;
        int rc = cslam_fiedler(RING, indptr, indices, data, NULL, 7u, 1e-8, 0, &lam, v, &iters, NULL);
        if (rc != CSLAM_OK) { printf("fiedler: %s\n", cslam_last_error()); return 1; }
        double want = 2.0 - 2.0 * cos(2.0 * 3.14159265358979323846 / RING), nrm = 0.0, sum = 0.0;
        for (int i = 0; i < RING; ++i) { nrm += v[i] * v[i]; sum += v[i]; }
        if (fabs(lam - want) > 1e-7 * want || fabs(nrm - 1.0) > 1e-9 || fabs(sum) > 1e-9 || iters < 1) {
            printf("fiedler: lambda_2 %.12e, expected %.12e (|v|^2 %.12f, sum %.3e, %d iterations)\n", lam, want, nrm, sum, iters);
            return 1;
        }
        /* the sparsifier's Frank-Wolfe loop (mac.py:191-233) on the same ring as fixed edges + 8 candidate chords: 3 chosen */
        {
            int64_t fi[RING], fj[RING], ci[8], cj[8]; double fw[RING], cw[8], w0[8], sel[8], up = 0.0; int its = 0, chosen = 0;
            for (int i = 0; i < RING; ++i) { fi[i] = i; fj[i] = (i + 1) % RING; fw[i] = 1.0; }
            for (int c = 0; c < 8; ++c) { ci[c] = 7 * c; cj[c] = (7 * c + 60 + 25 * c) % RING; cw[c] = 0.3 + 0.1 * c; w0[c] = c < 3 ? 1.0 : 0.0; }
            rc = cslam_mac_fw_subset(RING, RING, fi, fj, fw, 8, ci, cj, cw, w0, 3, 5, 1e-8, 1e-8, sel, NULL, &up, &its, NULL);
            if (rc != CSLAM_OK) { printf("fw_subset: %s\n", cslam_last_error()); return 1; }
            for (int c = 0; c < 8; ++c) chosen += sel[c] == 1.0;
            if (chosen != 3 || its < 1 || !(up > 0.0)) { printf("fw_subset: %d chosen, %d iterations, bound %g\n", chosen, its, up); return 1; }
        }
        cslam_fiedler_release();
    }
    printf("C ABI smoke ok (version %d)\n", cslam_version());
    return 0;
}
