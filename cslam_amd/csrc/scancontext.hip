// scancontext.hip -- lidar ScanContext bank in HBM + the two-stage search (gfx950).
//
// Replaces cslam/lidar_pr/scancontext_matching.py:5-104 (ScanContextMatching) and the two helpers
// it calls, scancontext_utils.py:78-79 (sc2rk) and :81-113 (distance_sc):
//   storage      : growable float64 [cap, R*S] scan contexts + [cap, R] ring keys (reference
//                  scancontext_matching.py:18-19,31-41) + [cap, S] column norms (negative = the
//                  column is all zero, the reference's "~np.any" skip at scancontext_utils.py:96).
//   sc_prep      : ring key with numpy's pairwise summation order (bit-identical to np.mean) and
//                  column norms, one wave per scan context.
//   sc_knn       : stage 1, brute-force k-NN over the ring keys (the reference rebuilds a KD-tree
//                  per query, scancontext_matching.py:56-61): one lane per bank row, float64
//                  squared distance, wave-resident sorted candidate list.  HBM-bound: R*8 bytes/row.
//   sc_distance  : stage 2, one workgroup per (query, candidate): both scan contexts in LDS, the
//                  S x S column-cosine matrix once (R fma per entry), then each of the S yaw shifts
//                  is a wrapped diagonal sum -- 60x fewer dot products than the reference's
//                  roll-and-recompute loop, same sums in the same order.
//   sc_pick      : first strict minimum over the candidates in ring-key order, from 1.0
//                  (scancontext_matching.py:64-75).
// All arithmetic is float64 in the order of oracle/sc_oracle.c (left-to-right fma chains).
#include <new>
#include "common.h"

#define SC_MAX_CAND 64
#define SC_MAX_R 64
#define SC_MAX_S 128

struct cslam_scbank {
    int device, R, S, L;
    int64_t n, cap;
    double *sc;   // [cap, L]
    double *rk;   // [cap, R]
    double *cn;   // [cap, S]
    char *ws;     // search workspace
    size_t ws_bytes;
    char *stage;  // host-API staging
    size_t stage_bytes;
    int num_cu;
};

static int sc_reserve(char **p, size_t *have, size_t bytes) {
    if (*have >= bytes) return CSLAM_OK;
    if (*p) HIP_TRY(hipFree(*p));
    *p = nullptr;
    *have = 0;
    size_t want = bytes + bytes / 2;
    if (hipMalloc((void **)p, want) != hipSuccess) {
        (void)hipGetLastError();
        cslam_set_error("out of device memory reserving %zu bytes", want);
        return CSLAM_E_NOMEM;
    }
    *have = want;
    return CSLAM_OK;
}

static int sc_grow(cslam_scbank *b, int64_t need) {
    if (need <= b->cap) return CSLAM_OK;
    int64_t cap = b->cap > 0 ? b->cap : 1000;        // reference starts at 1000 and doubles
    while (cap < need) cap *= 2;
    double *sc = nullptr, *rk = nullptr, *cn = nullptr;
    if (hipMalloc((void **)&sc, (size_t)cap * b->L * 8) != hipSuccess ||
        hipMalloc((void **)&rk, (size_t)cap * b->R * 8) != hipSuccess ||
        hipMalloc((void **)&cn, (size_t)cap * b->S * 8) != hipSuccess) {
        (void)hipGetLastError();
        if (sc) (void)hipFree(sc);
        if (rk) (void)hipFree(rk);
        if (cn) (void)hipFree(cn);
        cslam_set_error("out of device memory growing the scan-context bank to %lld items", (long long)cap);
        return CSLAM_E_NOMEM;
    }
    if (b->n > 0) {
        HIP_TRY(hipMemcpy(sc, b->sc, (size_t)b->n * b->L * 8, hipMemcpyDeviceToDevice));
        HIP_TRY(hipMemcpy(rk, b->rk, (size_t)b->n * b->R * 8, hipMemcpyDeviceToDevice));
        HIP_TRY(hipMemcpy(cn, b->cn, (size_t)b->n * b->S * 8, hipMemcpyDeviceToDevice));
    }
    if (b->sc) HIP_TRY(hipFree(b->sc));
    if (b->rk) HIP_TRY(hipFree(b->rk));
    if (b->cn) HIP_TRY(hipFree(b->cn));
    b->sc = sc; b->rk = rk; b->cn = cn; b->cap = cap;
    return CSLAM_OK;
}

CSLAM_API int cslam_scbank_create(int device, int rings, int sectors, int64_t capacity_hint,
                                  cslam_scbank_t **out) {
    ARG_CHECK(out, "out is NULL");
    ARG_CHECK(rings >= 1 && rings <= SC_MAX_R, "rings must be in [1, 64]");
    ARG_CHECK(sectors >= 1 && sectors <= SC_MAX_S, "sectors must be in [1, 128]");
    DeviceGuard _dev_guard(device);
    if (!_dev_guard.ok) { cslam_set_error("hipSetDevice(%d) failed", device); return CSLAM_E_HIP; }
    cslam_scbank *b = new (std::nothrow) cslam_scbank();
    if (!b) { cslam_set_error("host allocation failed"); return CSLAM_E_NOMEM; }
    b->device = device; b->R = rings; b->S = sectors; b->L = rings * sectors;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    b->num_cu = prop.multiProcessorCount;
    int rc = sc_grow(b, capacity_hint > 0 ? capacity_hint : 1000);
    if (rc) { delete b; return rc; }
    *out = b;
    return CSLAM_OK;
}

CSLAM_API int cslam_scbank_destroy(cslam_scbank_t *b) {
    if (!b) return CSLAM_OK;
    DeviceGuard _dev_guard(b->device);
    (void)hipDeviceSynchronize();
    if (b->sc) (void)hipFree(b->sc);
    if (b->rk) (void)hipFree(b->rk);
    if (b->cn) (void)hipFree(b->cn);
    if (b->ws) (void)hipFree(b->ws);
    if (b->stage) (void)hipFree(b->stage);
    delete b;
    return CSLAM_OK;
}

CSLAM_API int cslam_scbank_size(const cslam_scbank_t *b, int64_t *n, int *rings, int *sectors) {
    ARG_CHECK(b, "bank is NULL");
    if (n) *n = b->n;
    if (rings) *rings = b->R;
    if (sectors) *sectors = b->S;
    return CSLAM_OK;
}

CSLAM_API int cslam_scbank_clear(cslam_scbank_t *b) {
    ARG_CHECK(b, "bank is NULL");
    b->n = 0;
    return CSLAM_OK;
}

// ------------------------------------------------------------------- prep ----
// numpy add.reduce over a contiguous run (pairwise_sum, n <= 128 branch): 8 running sums over
// whole blocks of 8, combined as a balanced tree, then the tail added left to right.
__device__ __forceinline__ double np_sum(const double *a, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r = __dadd_rn(r, a[i]);
        return r;
    }
    double r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = a[k];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = __dadd_rn(r[k], a[i + k]);
    }
    double res = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])),
                           __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
    for (; i < n; ++i) res = __dadd_rn(res, a[i]);
    return res;
}

// one wave per scan context; blockDim = 128 (covers S <= 128 columns and R <= 64 rings)
__global__ __launch_bounds__(128) void sc_prep_kernel(const double *__restrict__ sc, int64_t count, int R, int S,
                                                      double *__restrict__ rk, double *__restrict__ cn) {
    extern __shared__ double s_sc[];
    const int64_t it = blockIdx.x;
    const int L = R * S;
    const double *src = sc + it * L;
    for (int e = threadIdx.x; e < L; e += blockDim.x) s_sc[e] = src[e];
    __syncthreads();
    const int t = threadIdx.x;
    if (t < S) {
        double s = 0.0;
        bool any = false;
        for (int r = 0; r < R; ++r) {
            double v = s_sc[r * S + t];
            s = fma(v, v, s);
            any |= (v != 0.0);
        }
        cn[it * S + t] = any ? sqrt(s) : -1.0;
    }
    if (t < R) rk[it * R + t] = __ddiv_rn(__dadd_rn(0.0, np_sum(s_sc + t * S, S)), (double)S);
}

static int sc_prep_launch(const double *d_sc, int64_t count, int R, int S, double *d_rk, double *d_cn,
                          hipStream_t st) {
    if (count == 0) return CSLAM_OK;
    hipLaunchKernelGGL(sc_prep_kernel, dim3((unsigned)count), dim3(128), (size_t)R * S * 8, st, d_sc, count, R, S,
                       d_rk, d_cn);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

CSLAM_API int cslam_scbank_add_dev(cslam_scbank_t *b, const double *d_sc, int64_t n, void *stream) {
    ARG_CHECK(b && (d_sc || n == 0), "NULL argument");
    ARG_CHECK(n >= 0 && b->n + n < (1LL << 31), "item count must stay below 2^31");
    BANK_DEVICE(b);
    hipStream_t st = (hipStream_t)stream;
    if (b->n + n > b->cap) {
        HIP_TRY(hipStreamSynchronize(st));
        int rc = sc_grow(b, b->n + n);
        if (rc) return rc;
    }
    if (n == 0) return CSLAM_OK;
    HIP_TRY(hipMemcpyAsync(b->sc + b->n * b->L, d_sc, (size_t)n * b->L * 8, hipMemcpyDeviceToDevice, st));
    int rc = sc_prep_launch(b->sc + b->n * b->L, n, b->R, b->S, b->rk + b->n * b->R, b->cn + b->n * b->S, st);
    if (rc) return rc;
    b->n += n;
    return CSLAM_OK;
}

CSLAM_API int cslam_scbank_add_host(cslam_scbank_t *b, const double *sc, int64_t n) {
    ARG_CHECK(b && (sc || n == 0), "NULL argument");
    ARG_CHECK(n >= 0 && b->n + n < (1LL << 31), "item count must stay below 2^31");
    BANK_DEVICE(b);
    int rc = sc_grow(b, b->n + n);
    if (rc) return rc;
    if (n == 0) return CSLAM_OK;
    HIP_TRY(hipMemcpy(b->sc + b->n * b->L, sc, (size_t)n * b->L * 8, hipMemcpyHostToDevice));
    rc = sc_prep_launch(b->sc + b->n * b->L, n, b->R, b->S, b->rk + b->n * b->R, b->cn + b->n * b->S, 0);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(0));
    b->n += n;
    return CSLAM_OK;
}

CSLAM_API int cslam_scbank_read_host(const cslam_scbank_t *b, int64_t first, int64_t count, double *sc_out,
                                     double *rk_out) {
    ARG_CHECK(b, "bank is NULL");
    ARG_CHECK(first >= 0 && count >= 0 && first + count <= b->n, "range outside the bank");
    BANK_DEVICE(b);
    HIP_TRY(hipDeviceSynchronize());
    if (count == 0) return CSLAM_OK;
    if (sc_out) HIP_TRY(hipMemcpy(sc_out, b->sc + first * b->L, (size_t)count * b->L * 8, hipMemcpyDeviceToHost));
    if (rk_out) HIP_TRY(hipMemcpy(rk_out, b->rk + first * b->R, (size_t)count * b->R * 8, hipMemcpyDeviceToHost));
    return CSLAM_OK;
}

// ---------------------------------------------------------------- stage 1 ----
// candidate order: smaller squared distance first, ties -> smaller row.  Encoded for WaveList
// (larger key first, ties -> larger idx) as key = -d2, idx = INT_MAX - row.
#define SC_ENC(row) (0x7fffffff - (row))

__device__ __forceinline__ void wavelist_offer(WaveList &wl, double key, int idx, bool valid, int C, int lane) {
    double tk = wl.key_at(C - 1);
    int ti = wl.idx_at(C - 1);
    unsigned long long m = __ballot(valid && ranks_before(key, idx, tk, ti));
    while (m) {
        int src = __ffsll((long long)m) - 1;
        double ck = __shfl(key, src, 64);
        int ci = __shfl(idx, src, 64);
        wl.insert(ck, ci, lane);
        m &= m - 1;
        tk = wl.key_at(C - 1);
        ti = wl.idx_at(C - 1);
        m &= __ballot(valid && ranks_before(key, idx, tk, ti));
    }
}

// grid (G, nq), 256 threads.  Rows are dealt to workgroups in contiguous chunks.
__global__ __launch_bounds__(256) void sc_knn_kernel(const double *__restrict__ rk, int64_t n, int R,
                                                     const double *__restrict__ qrk, const int64_t *__restrict__ row_limit,
                                                     int C, double *__restrict__ part_key, int *__restrict__ part_idx) {
    __shared__ double s_q[SC_MAX_R];
    __shared__ double s_key[4][SC_MAX_CAND];
    __shared__ int s_idx[4][SC_MAX_CAND];
    const int q = blockIdx.y, g = blockIdx.x, G = gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t lim = row_limit ? row_limit[q] : n;
    lim = lim < n ? lim : n;
    lim = lim < 0 ? 0 : lim;
    if (threadIdx.x < R) s_q[threadIdx.x] = qrk[(int64_t)q * R + threadIdx.x];
    __syncthreads();
    const int64_t per = (n + G - 1) / G;
    const int64_t r0 = (int64_t)g * per;
    int64_t r1 = r0 + per;
    r1 = r1 < lim ? r1 : lim;
    WaveList wl;
    wl.init();
    for (int64_t base = r0 + wave * 64; base < r1; base += 256) {
        const int64_t row = base + lane;
        const bool valid = row < r1;
        double d2 = 0.0;
        if (valid) {
            const double *p = rk + row * R;
            for (int r = 0; r < R; ++r) {
                double t = __dsub_rn(p[r], s_q[r]);
                d2 = fma(t, t, d2);
            }
        }
        wavelist_offer(wl, -d2, SC_ENC((int)row), valid, C, lane);
    }
    if (lane < C) { s_key[wave][lane] = wl.key; s_idx[wave][lane] = wl.idx; }
    __syncthreads();
    if (wave == 0) {
        for (int w = 1; w < 4; ++w) {
            double k = lane < C ? s_key[w][lane] : -INFINITY;
            int i = lane < C ? s_idx[w][lane] : -1;
            wavelist_offer(wl, k, i, i >= 0, C, lane);
        }
        if (lane < C) {
            size_t o = ((size_t)q * G + g) * C + lane;
            part_key[o] = wl.key;
            part_idx[o] = wl.idx;
        }
    }
}

// Batch form of stage 1 for the default 20 rings: QT queries share every bank row read (the row sits in
// registers, the query ring keys in LDS), so the ring-key array is streamed once per QT queries instead of
// once per query.  grid (G, ceil(nq / QT)), 256 threads.  Same distances, same order, same lists.
template <int RR, int QT>
__global__ __launch_bounds__(256) void sc_knn_tile_kernel(const double *__restrict__ rk, int64_t n,
                                                          const double *__restrict__ qrk,
                                                          const int64_t *__restrict__ row_limit, int nq, int C,
                                                          double *__restrict__ part_key, int *__restrict__ part_idx) {
    __shared__ double s_q[QT][RR];
    __shared__ int64_t s_lim[QT];
    __shared__ double s_key[4][SC_MAX_CAND];
    __shared__ int s_idx[4][SC_MAX_CAND];
    const int q0 = blockIdx.y * QT, g = blockIdx.x, G = gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < QT * RR; e += 256) {
        const int t = e / RR, r = e - t * RR;
        s_q[t][r] = q0 + t < nq ? qrk[(int64_t)(q0 + t) * RR + r] : 0.0;
    }
    if (threadIdx.x < QT) {
        const int q = q0 + threadIdx.x;
        int64_t lim = (q < nq) ? (row_limit ? row_limit[q] : n) : 0;
        lim = lim < n ? lim : n;
        s_lim[threadIdx.x] = lim < 0 ? 0 : lim;
    }
    __syncthreads();
    const int64_t per = (n + G - 1) / G;
    const int64_t r0 = (int64_t)g * per;
    int64_t r1 = r0 + per;
    r1 = r1 < n ? r1 : n;
    WaveList wl[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) wl[t].init();
    for (int64_t base = r0 + wave * 64; base < r1; base += 256) {
        const int64_t row = base + lane;
        const bool in = row < r1;
        double v[RR];
        const double *p = rk + (in ? row : r0) * RR;
#pragma unroll
        for (int r = 0; r < RR; ++r) v[r] = p[r];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            double d2 = 0.0;
#pragma unroll
            for (int r = 0; r < RR; ++r) {
                const double d = __dsub_rn(v[r], s_q[t][r]);
                d2 = fma(d, d, d2);
            }
            wavelist_offer(wl[t], -d2, SC_ENC((int)row), in && row < s_lim[t], C, lane);
        }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        __syncthreads();
        if (lane < C) { s_key[wave][lane] = wl[t].key; s_idx[wave][lane] = wl[t].idx; }
        __syncthreads();
        if (wave == 0 && q0 + t < nq) {
            for (int w = 1; w < 4; ++w) {
                double k = lane < C ? s_key[w][lane] : -INFINITY;
                int i = lane < C ? s_idx[w][lane] : -1;
                wavelist_offer(wl[t], k, i, i >= 0, C, lane);
            }
            if (lane < C) {
                size_t o = ((size_t)(q0 + t) * G + g) * C + lane;
                part_key[o] = wl[t].key;
                part_idx[o] = wl[t].idx;
            }
        }
    }
}

// one wave per query: merge the G partial lists; cand[q][c] = bank row or -1
__global__ __launch_bounds__(64) void sc_knn_merge_kernel(const double *__restrict__ part_key,
                                                          const int *__restrict__ part_idx, int G, int C,
                                                          int64_t *__restrict__ cand) {
    const int q = blockIdx.x, lane = threadIdx.x;
    WaveList wl;
    wl.init();
    const size_t base = (size_t)q * G * C;
    const int total = G * C;
    for (int e0 = 0; e0 < total; e0 += 64) {
        int e = e0 + lane;
        bool valid = e < total;
        double k = valid ? part_key[base + e] : -INFINITY;
        int i = valid ? part_idx[base + e] : -1;
        wavelist_offer(wl, k, i, valid && i >= 0, C, lane);
    }
    if (lane < C) cand[(size_t)q * C + lane] = wl.idx >= 0 ? (int64_t)(0x7fffffff - wl.idx) : -1;
}

// ---------------------------------------------------------------- stage 2 ----
// grid (C, nq), 256 threads.  LDS: candidate and query scan contexts [R][S], their column norms,
// the cosine matrix [S][S+1], the per-shift similarities.
__global__ __launch_bounds__(256) void sc_distance_kernel(const double *__restrict__ bank_sc,
                                                          const double *__restrict__ bank_cn,
                                                          const double *__restrict__ q_sc,
                                                          const double *__restrict__ q_cn,
                                                          const int64_t *__restrict__ cand, int R, int S, int C,
                                                          double *__restrict__ cdist, int *__restrict__ cyaw) {
    extern __shared__ double lds[];
    const int L = R * S, P = S + 1;
    double *A = lds;               // candidate  [R][S]
    double *B = A + L;             // query      [R][S]
    double *n1 = B + L;            // [S]
    double *n2 = n1 + S;           // [S]
    double *M = n2 + S;            // [S][P]
    double *sims = M + S * P;      // [S]
    const int c = blockIdx.x, q = blockIdx.y, t = threadIdx.x;
    const int64_t row = cand[(size_t)q * C + c];
    if (row < 0) {                 // fewer bank items than candidates: the reference reads an all-zero
        if (t == 0) {              // slot there (scancontext_matching.py:68), distance 1, never chosen
            cdist[(size_t)q * C + c] = 1.0;
            cyaw[(size_t)q * C + c] = 0;
        }
        return;
    }
    const double *a = bank_sc + row * L, *b = q_sc + (int64_t)q * L;
    for (int e = t; e < L; e += 256) { A[e] = a[e]; B[e] = b[e]; }
    if (t < S) { n1[t] = bank_cn[row * S + t]; n2[t] = q_cn[(int64_t)q * S + t]; }
    __syncthreads();
    for (int e = t; e < S * S; e += 256) {
        const int ca = e / S, cb = e - ca * S;
        double v = 0.0;
        if (n1[ca] >= 0.0 && n2[cb] >= 0.0) {
            double d = 0.0;
            for (int r = 0; r < R; ++r) d = fma(A[r * S + ca], B[r * S + cb], d);
            v = __ddiv_rn(d, __dmul_rn(n1[ca], n2[cb]));
        }
        M[ca * P + cb] = v;
    }
    __syncthreads();
    if (t < S) {
        const int s = (t + 1) % S;             // iteration t of the reference has rolled by t+1
        double sum = 0.0;
        int engaged = 0;
        int ca = (S - s) % S;                  // column of the candidate under query column 0
        for (int j = 0; j < S; ++j) {
            if (n1[ca] >= 0.0 && n2[j] >= 0.0) {
                sum = __dadd_rn(sum, M[ca * P + j]);
                ++engaged;
            }
            ca = ca + 1 == S ? 0 : ca + 1;
        }
        sims[t] = engaged ? __ddiv_rn(sum, (double)engaged) : 0.0;
    }
    __syncthreads();
    if (t == 0) {
        double best = sims[0];
        int bi = 0;
        for (int i = 1; i < S; ++i)
            if (sims[i] > best) { best = sims[i]; bi = i; }      // np.argmax: first maximum
        cdist[(size_t)q * C + c] = __dsub_rn(1.0, best);
        cyaw[(size_t)q * C + c] = bi + 1;
    }
}

__global__ void sc_pick_kernel(const int64_t *__restrict__ cand, const double *__restrict__ cdist,
                               const int *__restrict__ cyaw, int nq, int C, int64_t *__restrict__ best_idx,
                               double *__restrict__ best_sim, int *__restrict__ best_yaw) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    double nn = 1.0;
    int64_t idx = -1;
    int yaw = 0;
    for (int c = 0; c < C; ++c) {
        const int64_t row = cand[(size_t)q * C + c];
        const double d = cdist[(size_t)q * C + c];
        if (row >= 0 && d < nn) { nn = d; idx = row; yaw = cyaw[(size_t)q * C + c]; }
    }
    best_idx[q] = idx;
    best_sim[q] = idx >= 0 ? __dsub_rn(1.0, nn) : 0.0;    // the reference returns 1 - (1 - sim)
    best_yaw[q] = yaw;
}

static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// d_q: [nq, L] float64 in HBM.  Outputs are device pointers; d_cand/d_cdist/d_cyaw may be NULL.
CSLAM_API int cslam_scbank_search_dev(cslam_scbank_t *b, const double *d_q, int64_t nq, int num_candidates,
                                      const int64_t *d_row_limit, int64_t *d_best_idx, double *d_best_sim,
                                      int32_t *d_best_yaw, int64_t *d_cand, double *d_cdist, int32_t *d_cyaw,
                                      void *stream) {
    ARG_CHECK(b && (d_q || nq == 0) && d_best_idx && d_best_sim && d_best_yaw, "NULL argument");
    ARG_CHECK(num_candidates >= 1 && num_candidates <= SC_MAX_CAND, "num_candidates must be in [1, 64]");
    ARG_CHECK(nq >= 0 && nq <= 65535, "nq must be in [0, 65535] per call");
    if (nq == 0) return CSLAM_OK;
    BANK_DEVICE(b);
    hipStream_t st = (hipStream_t)stream;
    const int C = num_candidates, R = b->R, S = b->S;
    // row chunks per query (tile): enough workgroups to fill the chip, no more -- every chunk pays the
    // warm-up of its candidate lists (~C ln(rows) insertions), so long chunks are cheaper than many
    const bool tiled = (R == 20 && nq >= 8);
    const int64_t qgroups = tiled ? ceil_div64(nq, 8) : nq;
    int G = (int)ceil_div64((int64_t)b->num_cu * 4, qgroups);
    const int64_t gcap = ceil_div64(b->n > 0 ? b->n : 1, 2048);
    if (G > gcap) G = (int)gcap;
    if (G < 1) G = 1;
    size_t o_qrk = 0, o_qcn = o_qrk + al256((size_t)nq * R * 8), o_pk = o_qcn + al256((size_t)nq * S * 8);
    size_t o_pi = o_pk + al256((size_t)nq * G * C * 8), o_cand = o_pi + al256((size_t)nq * G * C * 4);
    size_t o_cd = o_cand + al256((size_t)nq * C * 8), o_cy = o_cd + al256((size_t)nq * C * 8);
    size_t total = o_cy + al256((size_t)nq * C * 4);
    int rc = sc_reserve(&b->ws, &b->ws_bytes, total);
    if (rc) return rc;
    double *qrk = (double *)(b->ws + o_qrk), *qcn = (double *)(b->ws + o_qcn);
    double *pk = (double *)(b->ws + o_pk);
    int *pi = (int *)(b->ws + o_pi);
    int64_t *cand = d_cand ? d_cand : (int64_t *)(b->ws + o_cand);
    double *cd = d_cdist ? d_cdist : (double *)(b->ws + o_cd);
    int *cy = d_cyaw ? d_cyaw : (int *)(b->ws + o_cy);
    rc = sc_prep_launch(d_q, nq, R, S, qrk, qcn, st);
    if (rc) return rc;
    if (tiled) {
        constexpr int QT = 8;
        hipLaunchKernelGGL((sc_knn_tile_kernel<20, QT>), dim3((unsigned)G, (unsigned)ceil_div64(nq, QT)), dim3(256),
                           0, st, b->rk, b->n, qrk, d_row_limit, (int)nq, C, pk, pi);
    } else {
        hipLaunchKernelGGL(sc_knn_kernel, dim3((unsigned)G, (unsigned)nq), dim3(256), 0, st, b->rk, b->n, R, qrk,
                           d_row_limit, C, pk, pi);
    }
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(sc_knn_merge_kernel, dim3((unsigned)nq), dim3(64), 0, st, pk, pi, G, C, cand);
    HIP_TRY(hipGetLastError());
    const size_t lds = ((size_t)2 * R * S + 3 * S + (size_t)S * (S + 1)) * 8;
    HIP_TRY(hipFuncSetAttribute((const void *)sc_distance_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    hipLaunchKernelGGL(sc_distance_kernel, dim3((unsigned)C, (unsigned)nq), dim3(256), lds, st, b->sc, b->cn, d_q,
                       qcn, cand, R, S, C, cd, cy);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(sc_pick_kernel, dim3((unsigned)ceil_div64(nq, 128)), dim3(128), 0, st, cand, cd, cy, (int)nq,
                       C, d_best_idx, d_best_sim, d_best_yaw);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

CSLAM_API int cslam_scbank_search_host(cslam_scbank_t *b, const double *queries, int64_t nq, int num_candidates,
                                       const int64_t *row_limit, int64_t *best_idx, double *best_sim,
                                       int32_t *best_yaw, int64_t *cand, double *cdist, int32_t *cyaw) {
    ARG_CHECK(b && (queries || nq == 0) && best_idx && best_sim && best_yaw, "NULL argument");
    ARG_CHECK(num_candidates >= 1 && num_candidates <= SC_MAX_CAND, "num_candidates must be in [1, 64]");
    ARG_CHECK(nq >= 0, "nq must be >= 0");
    BANK_DEVICE(b);
    const int C = num_candidates;
    const int64_t CH = 16384;
    for (int64_t q0 = 0; q0 < nq; q0 += CH) {
        const int64_t m = nq - q0 < CH ? nq - q0 : CH;
        size_t o_q = 0, o_lim = o_q + al256((size_t)m * b->L * 8), o_bi = o_lim + al256((size_t)m * 8);
        size_t o_bs = o_bi + al256((size_t)m * 8), o_by = o_bs + al256((size_t)m * 8);
        size_t o_c = o_by + al256((size_t)m * 4), o_cd = o_c + al256((size_t)m * C * 8);
        size_t o_cy = o_cd + al256((size_t)m * C * 8), total = o_cy + al256((size_t)m * C * 4);
        int rc = sc_reserve(&b->stage, &b->stage_bytes, total);
        if (rc) return rc;
        char *s = b->stage;
        HIP_TRY(hipMemcpy(s + o_q, queries + q0 * b->L, (size_t)m * b->L * 8, hipMemcpyHostToDevice));
        if (row_limit) HIP_TRY(hipMemcpy(s + o_lim, row_limit + q0, (size_t)m * 8, hipMemcpyHostToDevice));
        rc = cslam_scbank_search_dev(b, (const double *)(s + o_q), m, C, row_limit ? (const int64_t *)(s + o_lim) : nullptr,
                                     (int64_t *)(s + o_bi), (double *)(s + o_bs), (int32_t *)(s + o_by),
                                     (int64_t *)(s + o_c), (double *)(s + o_cd), (int32_t *)(s + o_cy), 0);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(0));
        HIP_TRY(hipMemcpy(best_idx + q0, s + o_bi, (size_t)m * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(best_sim + q0, s + o_bs, (size_t)m * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(best_yaw + q0, s + o_by, (size_t)m * 4, hipMemcpyDeviceToHost));
        if (cand) HIP_TRY(hipMemcpy(cand + q0 * C, s + o_c, (size_t)m * C * 8, hipMemcpyDeviceToHost));
        if (cdist) HIP_TRY(hipMemcpy(cdist + q0 * C, s + o_cd, (size_t)m * C * 8, hipMemcpyDeviceToHost));
        if (cyaw) HIP_TRY(hipMemcpy(cyaw + q0 * C, s + o_cy, (size_t)m * C * 4, hipMemcpyDeviceToHost));
    }
    return CSLAM_OK;
}

// ------------------------------------------------------------ descriptor ----
// cslam/lidar_pr/scancontext_utils.py:10-75 (xy2theta, pt2rs, ptcloud2sc): polar binning of a float64
// point cloud, each bin keeping max(point height + 2) over the FIRST 500 points that fall in it in cloud
// order (the reference's `enough_large` storage; later points are dropped) and 0.0 for unused slots.
// One workgroup of 16 waves per frame; bin counters and running maxima live in LDS.  The order-dependent
// cap is honoured without serialising the points: per 1024-point chunk every wave ranks its lanes within
// equal-bin groups (ballot), per-wave bin totals are prefix-summed across the 16 waves per bin, and a
// point takes part iff (points of its bin before it) < 500.
#define SCD_WAVES 16
#define SCD_CAP 500
#define SCD_MAX_BINS 2048

__device__ __forceinline__ double np_floordiv_pos(double a, double b) {   // numpy npy_divmod, a >= 0, b > 0
    double mod = fmod(a, b);
    double div = __ddiv_rn(__dsub_rn(a, mod), b);
    if (div != 0.0) {
        double fl = floor(div);
        if (__dsub_rn(div, fl) > 0.5) fl += 1.0;
        return fl;
    }
    return 0.0;
}

__device__ __forceinline__ unsigned long long f64_ordered(double v) {
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double f64_unordered(unsigned long long u) {
    u = (u >> 63) ? (u & 0x7fffffffffffffffull) : ~u;
    return __longlong_as_double((long long)u);
}

__global__ __launch_bounds__(SCD_WAVES * 64) void sc_from_cloud_kernel(
    const double *__restrict__ pts, const int64_t *__restrict__ offsets, int R, int S, double gap_ring,
    double gap_sector, double *__restrict__ out, int *__restrict__ bad_sector) {
    __shared__ int s_cnt[SCD_MAX_BINS];
    __shared__ unsigned long long s_max[SCD_MAX_BINS];
    __shared__ unsigned short s_hist[SCD_WAVES][SCD_MAX_BINS];
    const int f = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int nb = R * S;
    const int64_t p0 = offsets[f], p1 = offsets[f + 1];
    for (int b = t; b < nb; b += blockDim.x) { s_cnt[b] = 0; s_max[b] = 0ull; }
    const double k = 180.0 / 3.141592653589793;
    for (int64_t base = p0; base < p1; base += SCD_WAVES * 64) {
        for (int e = t; e < SCD_WAVES * nb; e += blockDim.x) s_hist[e / nb][e % nb] = 0;
        __syncthreads();
        const int64_t i = base + t;
        int bin = -1;
        double h = 0.0;
        if (i < p1) {
            double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
            if (!(x != x || y != y || z != z)) {
                h = __dadd_rn(z, 2.0);
                if (x == 0.0) x = 0.001;
                if (y == 0.0) y = 0.001;
                double theta;
                if (x >= 0 && y >= 0) theta = __dmul_rn(k, atan(__ddiv_rn(y, x)));
                else if (x < 0 && y >= 0) theta = __dsub_rn(180.0, __dmul_rn(k, atan(__ddiv_rn(y, -x))));
                else if (x < 0 && y < 0) theta = __dadd_rn(180.0, __dmul_rn(k, atan(__ddiv_rn(y, x))));
                else theta = __dsub_rn(360.0, __dmul_rn(k, atan(__ddiv_rn(-y, x))));
                const double far = sqrt(__dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y)));
                double ring = np_floordiv_pos(far, gap_ring);
                const double sector = np_floordiv_pos(theta, gap_sector);
                if (ring >= (double)R) ring = (double)(R - 1);
                const int is = (int)sector;
                if (is < 0 || is >= S) atomicOr(bad_sector, 1);      // the reference raises IndexError
                else bin = (int)ring * S + is;
            }
        }
        // rank among the equal-bin lanes of this wave, and the wave's total per bin
        int rank = 0;
        unsigned long long remaining = __ballot(bin >= 0);
        while (remaining) {
            const int leader = __ffsll((long long)remaining) - 1;
            const int b = __shfl(bin, leader, 64);
            const unsigned long long m = __ballot(bin == b);
            if (bin == b) {
                rank = __popcll(m & ((1ull << lane) - 1ull));
                if (lane == leader) s_hist[w][b] = (unsigned short)__popcll(m);
            }
            remaining &= ~m;
        }
        __syncthreads();
        for (int b = t; b < nb; b += blockDim.x) {          // exclusive prefix over the waves, per bin
            int running = s_cnt[b];
            for (int ww = 0; ww < SCD_WAVES; ++ww) {
                const int c = s_hist[ww][b];
                s_hist[ww][b] = (unsigned short)(running < SCD_CAP ? running : SCD_CAP);
                running += c;
            }
            s_cnt[b] = running;
        }
        __syncthreads();
        if (bin >= 0 && (int)s_hist[w][bin] + rank < SCD_CAP) atomicMax(&s_max[bin], f64_ordered(h));
        __syncthreads();
    }
    __syncthreads();
    for (int b = t; b < nb; b += blockDim.x) {
        double v = 0.0;
        if (s_cnt[b] > 0) {
            v = f64_unordered(s_max[b]);
            if (s_cnt[b] < SCD_CAP && v < 0.0) v = 0.0;     // an unused storage slot holds 0.0
        }
        out[(int64_t)f * nb + b] = v;
    }
}

CSLAM_API int cslam_scancontext_from_cloud_dev(const double *d_points, const int64_t *d_offsets, int n_frames,
                                               int rings, int sectors, double max_length, double *d_out,
                                               int32_t *d_status, void *stream) {
    PTR_DEVICE(d_points);
    ARG_CHECK(d_offsets && d_out && d_status, "NULL argument");   // d_points may be NULL when every frame is empty
    ARG_CHECK(n_frames >= 0, "n_frames must be >= 0");
    ARG_CHECK(rings >= 1 && sectors >= 1 && rings * sectors <= SCD_MAX_BINS, "rings * sectors must be in [1, 2048]");
    ARG_CHECK(max_length > 0.0, "max_length must be > 0");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(d_status, 0, sizeof(int32_t), st));
    if (n_frames == 0) return CSLAM_OK;
    hipLaunchKernelGGL(sc_from_cloud_kernel, dim3((unsigned)n_frames), dim3(SCD_WAVES * 64), 0, st, d_points,
                       d_offsets, rings, sectors, max_length / rings, 360.0 / sectors, d_out, d_status);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}
