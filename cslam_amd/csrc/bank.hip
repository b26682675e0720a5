// bank.hip -- descriptor bank in HBM + the exact float64 scan search (gfx950).
//
// Replaces cslam/nns_matching.py:10-76 (NearestNeighborsMatching):
//   storage   : growable float32 [cap, ld] array in HBM (amortised doubling like
//               nns_matching.py:31-37), + per-row float64 sum of squares.
//   scan_exact: one wave per bank row, lanes stride the columns with 16-byte loads
//               (coalesced 1 KiB per wave instruction), float64 accumulate, butterfly
//               reduce, wave-resident sorted top-k list.  HBM-bound: reads n*ld*4
//               bytes per pass over up to QT queries.  This is the online (nq small)
//               path and the fallback of the MFMA batch path.
#include <stdarg.h>
#include <new>
#include <hip/hip_fp16.h>
#include "bank.h"

// ------------------------------------------------------------------ errors ----
static thread_local char g_err[512] = "";
void cslam_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
CSLAM_API const char *cslam_last_error(void) { return g_err; }
CSLAM_API int cslam_version(void) { return 100; }

int cslam_visible_devices() {
    static std::atomic<int> n{-1};          // every thread computes the same value
    int v = n.load(std::memory_order_relaxed);
    if (v < 0) {
        int c = 0;
        v = (hipGetDeviceCount(&c) == hipSuccess) ? c : 0;
        if (v == 0) (void)hipGetLastError();
        n.store(v, std::memory_order_relaxed);
    }
    return v;
}

int cslam_cu_count() {
    static std::atomic<int> cu[64];         // zero-initialised; per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (dev >= 0 && dev < 64) { const int c = cu[dev].load(std::memory_order_relaxed); if (c > 0) return c; }
    int c = 0;
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (dev >= 0 && dev < 64) cu[dev].store(c, std::memory_order_relaxed);
    return c;
}

CSLAM_API int cslam_device_count(int *count) {
    ARG_CHECK(count, "count is NULL");
    HIP_TRY(hipGetDeviceCount(count));
    return CSLAM_OK;
}

CSLAM_API int cslam_device_info(int device, char *name, int name_len, int64_t *hbm_bytes, int *cu_count) {
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    if (name && name_len > 0) { strncpy(name, p.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    if (cu_count) *cu_count = p.multiProcessorCount;
    return CSLAM_OK;
}

// ------------------------------------------------------------------ storage ----
int bank_ws_reserve(cslam_bank *b, int slot, size_t bytes) {
    if (bytes <= b->ws_bytes[slot]) return CSLAM_OK;
    if (b->ws[slot]) HIP_TRY(hipFree(b->ws[slot]));   // hipFree synchronises the device
    b->ws[slot] = nullptr; b->ws_bytes[slot] = 0;
    size_t want = bytes + bytes / 4;
    HIP_TRY(hipMalloc((void **)&b->ws[slot], want));
    b->ws_bytes[slot] = want;
    return CSLAM_OK;
}

static int bank_grow(cslam_bank *b, int64_t need) {
    if (need <= b->cap) return CSLAM_OK;
    int64_t ncap = b->cap > 0 ? b->cap : 1000;   // reference starts at 1000 rows (nns_matching.py:21)
    while (ncap < need) ncap *= 2;               // and doubles (nns_matching.py:36)
    float *rows = nullptr; double *vv = nullptr; float *invn = nullptr, *invs = nullptr;
    char *rows2 = nullptr, *rowsh = nullptr;
    HIP_TRY(hipMalloc((void **)&rows, (size_t)ncap * b->ld * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&rowsh, (size_t)ncap * b->ldh));
    if (b->rows2) HIP_TRY(hipMalloc((void **)&rows2, (size_t)ncap * b->ld2));
    HIP_TRY(hipMalloc((void **)&invs, (size_t)ncap * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&vv, (size_t)ncap * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&invn, (size_t)ncap * sizeof(float)));
    if (b->n > 0) {
        // rows appended just before this call may still be in flight on a non-blocking stream, which the
        // null-stream copies below are not ordered against: drain the device first (a grow happens log2(n) times)
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(rows, b->rows, (size_t)b->n * b->ld * sizeof(float), hipMemcpyDeviceToDevice));
        HIP_TRY(hipMemcpy(vv, b->vv, (size_t)b->n * sizeof(double), hipMemcpyDeviceToDevice));
        HIP_TRY(hipMemcpy(invn, b->invn, (size_t)b->n * sizeof(float), hipMemcpyDeviceToDevice));
        HIP_TRY(hipMemcpy(rowsh, b->rowsh, (size_t)b->n * b->ldh, hipMemcpyDeviceToDevice));
        if (rows2) HIP_TRY(hipMemcpy(rows2, b->rows2, (size_t)b->n * b->ld2, hipMemcpyDeviceToDevice));
        HIP_TRY(hipMemcpy(invs, b->invs, (size_t)b->n * sizeof(float), hipMemcpyDeviceToDevice));
    }
    if (b->rows) HIP_TRY(hipFree(b->rows));
    if (b->vv) HIP_TRY(hipFree(b->vv));
    if (b->invn) HIP_TRY(hipFree(b->invn));
    if (b->rows2) HIP_TRY(hipFree(b->rows2));
    if (b->rowsh) HIP_TRY(hipFree(b->rowsh));
    if (b->invs) HIP_TRY(hipFree(b->invs));
    b->rows = rows; b->vv = vv; b->invn = invn; b->rows2 = rows2; b->rowsh = rowsh; b->invs = invs; b->cap = ncap;
    return CSLAM_OK;
}

CSLAM_API int cslam_bank_create(int device, int dim, int64_t capacity_hint, cslam_bank_t **out) {
    ARG_CHECK(out, "out is NULL");
    ARG_CHECK(dim > 0, "dim must be > 0");
    DeviceGuard _dev_guard(device);
    if (!_dev_guard.ok) { cslam_set_error("hipSetDevice(%d) failed", device); return CSLAM_E_HIP; }
    cslam_bank *b = new (std::nothrow) cslam_bank();
    if (!b) { cslam_set_error("out of host memory"); return CSLAM_E_NOMEM; }
    b->device = 0; b->dim = 0; b->kd = 0; b->ld = 0; b->n = 0; b->cap = 0;
    b->rows = nullptr; b->vv = nullptr; b->invn = nullptr; b->rows2 = nullptr; b->rowsh = nullptr; b->invs = nullptr; b->ld2 = 0;
    b->kh = 0; b->ldh = 0;
    for (int s = 0; s < 3; ++s) { b->ws[s] = nullptr; b->ws_bytes[s] = 0; }
    b->stage = nullptr; b->stage_bytes = 0; b->last_stream = nullptr; b->ev_valid = false;
    b->side = nullptr; b->ev_fork = nullptr; b->ev_side = nullptr;
    b->h_nflag = nullptr; b->pending_flag_list = nullptr; b->pending_flag_count = nullptr; b->pending_dbg = 0;
    b->last_nprod = 1; b->f32_backoff = 0; b->f32_backoff_len = 8; b->stage_pinned = false;
    b->pend.active = false; b->pend.deferred = false; b->ev_flag = nullptr;
    b->dbg_part_key = nullptr; b->dbg_part_idx = nullptr; b->dbg_nseg = 0; b->dbg_nq = 0; b->dbg_err_bound = 0.0;
    for (int s = 0; s < 4; ++s) { b->stats[s] = 0; b->item_map_key[s] = -1; }
    b->num_cu = 0;
    b->ring_sched.nqt = -1; b->ring_sched.n_rows = -1; b->ring_sched.n_xcd = 0; b->ring_sched.wpx = 0; b->dbg_ring = false;
    b->device = device; b->dim = dim; b->kd = (int)round_up64(dim, 32);
    // Row pitch: a power-of-two pitch (4096 floats = 16 KiB) maps the same K offset of every row to
    // the same L2 set -- measured 17 % L2 hit rate, 2.1 TB fetched per 100k-query pass; one extra
    // 128-byte line per row spreads the rows of a tile over the sets.
    b->ld = b->kd + ((b->kd % 256 == 0) ? 32 : 0);
    b->ld2 = (int64_t)b->kd * 4 + ((b->kd % 256 == 0) ? 128 : 0);     // the pair copy: same bytes per value, same padding rule
    b->kh = (int)round_up64(dim, 64);
    b->ldh = (int64_t)b->kh * 2 + ((b->kh % 512 == 0) ? 128 : 0);     // the fp16 copy: half the bytes, same padding rule
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) == hipSuccess) b->num_cu = p.multiProcessorCount;
    if (b->num_cu <= 0) b->num_cu = 256;
    int rc = bank_grow(b, capacity_hint > 0 ? capacity_hint : 1000);
    if (rc != CSLAM_OK) { delete b; return rc; }
    if (hipEventCreate(&b->ev0) == hipSuccess && hipEventCreate(&b->ev1) == hipSuccess) b->ev_valid = true;
    *out = b;
    return CSLAM_OK;
}

CSLAM_API int cslam_bank_destroy(cslam_bank_t *b) {
    if (!b) return CSLAM_OK;
    DeviceGuard _dev_guard(b->device);
    (void)hipDeviceSynchronize();
    if (b->rows) (void)hipFree(b->rows);
    if (b->vv) (void)hipFree(b->vv);
    if (b->invn) (void)hipFree(b->invn);
    if (b->rows2) (void)hipFree(b->rows2);
    if (b->rowsh) (void)hipFree(b->rowsh);
    if (b->invs) (void)hipFree(b->invs);
    for (int s = 0; s < 3; ++s) if (b->ws[s]) (void)hipFree(b->ws[s]);
    if (b->stage) (void)hipFree(b->stage);
    if (b->h_nflag) (void)hipHostFree(b->h_nflag);
    if (b->ev_valid) { (void)hipEventDestroy(b->ev0); (void)hipEventDestroy(b->ev1); }
    if (b->ev_flag) (void)hipEventDestroy(b->ev_flag);
    if (b->side) { (void)hipStreamDestroy(b->side); (void)hipEventDestroy(b->ev_fork); (void)hipEventDestroy(b->ev_side); }
    delete b;
    return CSLAM_OK;
}

CSLAM_API int cslam_bank_size(const cslam_bank_t *b, int64_t *n, int *dim) {
    ARG_CHECK(b, "bank is NULL");
    if (n) *n = b->n;
    if (dim) *dim = b->dim;
    return CSLAM_OK;
}

#define NO_PENDING(b) ARG_CHECK(!(b)->pend.active, "a search of this bank is enqueued and not finished: call cslam_bank_search_finish first")

CSLAM_API int cslam_bank_clear(cslam_bank_t *b) {
    ARG_CHECK(b, "bank is NULL");
    NO_PENDING(b);
    b->n = 0;
    return CSLAM_OK;
}

// one wave per appended row: copy (with float64->float32 cast when SRC is double, the
// numpy assignment cast of nns_matching.py:39), zero the padding, float64 sum of squares; then the same row once more, times
// its power-of-two scale, for the candidate stages on the fp16 matrix pipe (sim_topk_pair.hip): rounded to fp16 (rowsh) and,
// when the bank keeps the pair copy, split exactly into hi + lo (rows2).  STATS = false: rows / vv / invn / invs exist already
// and only rows2 is (re)built from the float32 rows (bank_pairs_ensure).
template <typename SRC, bool STATS>
__global__ __launch_bounds__(256) void bank_append_kernel(const SRC *__restrict__ src, int64_t ld_src,
                                                          int64_t nrows, int dim, int ld, int kd, int kh,
                                                          float *__restrict__ rows, double *__restrict__ vv,
                                                          float *__restrict__ invn, char *__restrict__ rowsh, int64_t ldh,
                                                          char *__restrict__ rows2, int64_t ld2,
                                                          float *__restrict__ invs, int64_t row0) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const SRC *s = src + r * ld_src;
    double acc = 0.0;
    float amax = 0.0f;
    bool finite = true;
    if (STATS) {
        float *d = rows + (row0 + r) * (int64_t)ld;
        for (int c = lane; c < ld; c += 64) {
            float v = c < dim ? (float)s[c] : 0.0f;
            d[c] = v;
            acc += (double)v * (double)v;
            finite &= (v - v) == 0.0f;
            amax = fmaxf(amax, fabsf(v));
        }
        acc = wave_allreduce_sum(acc);
    } else {
        for (int c = lane; c < dim; c += 64) {
            const float v = (float)s[c];
            finite &= (v - v) == 0.0f;
            amax = fmaxf(amax, fabsf(v));
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    finite = __all(finite);
    // power-of-two scale: max |v| s in [2^14, 2^15)  (fp16 holds 65504; the split keeps 22 bits of every value that is not
    // below 2^-39 of the row's largest).  Rows the scale cannot serve (non-finite entries, magnitudes outside
    // [2^-100, 2^100]) get invs = NaN: their key is NaN in the candidate stage, which makes them contenders of every query --
    // the float64 rescoring then gives them the score the reference gives them.
    int e = 0;
    if (amax > 0.0f) (void)frexpf(amax, &e);               // amax = m 2^e, m in [0.5, 1)
    const bool servable = finite && (amax == 0.0f || (e > -100 && e < 100));
    const float sc = (amax > 0.0f && servable) ? ldexpf(1.0f, 15 - e) : 1.0f;
    if (STATS && lane == 0) {
        const float inv_n = (float)(1.0 / sqrt(acc));     // zero row -> +inf (NaN score, like the reference)
        vv[row0 + r] = acc;
        invn[row0 + r] = inv_n;
        invs[row0 + r] = servable ? inv_n / sc : NAN;     // sc is a power of two: exact
    }
    // lane = 4 consecutive channels; second pass over the SOURCE row (not over the float32 copy other lanes have just stored)
    char *dh = STATS ? rowsh + (row0 + r) * ldh : nullptr;
    char *d2 = rows2 ? rows2 + (row0 + r) * ld2 : nullptr;
    for (int c = 4 * lane; c < kh; c += 256) {
        unsigned hi[2], lo[2];
        float w[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) w[t] = (c + t < dim ? (float)s[c + t] : 0.0f) * sc;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const __half2 h = __floats2half2_rn(servable ? w[2 * t] : 0.0f, servable ? w[2 * t + 1] : 0.0f);
            const float2 f = __half22float2(h);
            const __half2 l = __floats2half2_rn(servable ? w[2 * t] - f.x : 0.0f, servable ? w[2 * t + 1] - f.y : 0.0f);
            hi[t] = *(const unsigned *)&h; lo[t] = *(const unsigned *)&l;
        }
        if (STATS) *(uint2 *)(dh + c * 2) = make_uint2(hi[0], hi[1]);
        if (d2 && c < kd) {                                // pairs: block kb = [hi 32 | lo 32] halfs
            char *blk = d2 + (c >> 5) * 128 + (c & 31) * 2;
            *(uint2 *)blk = make_uint2(hi[0], hi[1]);
            *(uint2 *)(blk + 64) = make_uint2(lo[0], lo[1]);
        }
    }
}

template <typename SRC>
static int bank_append_launch(cslam_bank *b, const SRC *d_src, int64_t ld_src, int64_t n, hipStream_t st) {
    if (n == 0) return CSLAM_OK;
    dim3 grid((unsigned)ceil_div64(n, 4));
    hipLaunchKernelGGL((bank_append_kernel<SRC, true>), grid, dim3(256), 0, st, d_src, ld_src, n, b->dim, b->ld, b->kd, b->kh,
                       b->rows, b->vv, b->invn, b->rowsh, b->ldh, b->rows2, b->ld2, b->invs, b->n);
    HIP_TRY(hipGetLastError());
    b->n += n;
    b->last_stream = st;
    return CSLAM_OK;
}

// The pair copy on demand (CSLAM_MFMA_STAGE1=pair): allocate it at the bank's capacity and fill rows [0, n) from the float32
// rows, on the caller's stream (ordered behind the appends the caller has ordered its search behind).
int bank_pairs_ensure(cslam_bank *b, hipStream_t st) {
    if (b->rows2) return CSLAM_OK;
    HIP_TRY(hipMalloc((void **)&b->rows2, (size_t)b->cap * b->ld2));
    if (b->n > 0) {
        hipLaunchKernelGGL((bank_append_kernel<float, false>), dim3((unsigned)ceil_div64(b->n, 4)), dim3(256), 0, st,
                           (const float *)b->rows, (int64_t)b->ld, b->n, b->dim, b->ld, b->kd, b->kh, b->rows, b->vv, b->invn,
                           b->rowsh, b->ldh, b->rows2, b->ld2, b->invs, (int64_t)0);
        HIP_TRY(hipGetLastError());
    }
    return CSLAM_OK;
}

CSLAM_API int cslam_bank_add_host(cslam_bank_t *b, const void *vecs, int dtype, int64_t n) {
    ARG_CHECK(b && (vecs || n == 0), "NULL argument");
    ARG_CHECK(dtype == CSLAM_F32 || dtype == CSLAM_F64, "dtype must be CSLAM_F32 or CSLAM_F64");
    ARG_CHECK(n >= 0, "n < 0");
    NO_PENDING(b);
    if (n == 0) return CSLAM_OK;
    BANK_DEVICE(b);
    int rc = bank_grow(b, b->n + n);
    if (rc) return rc;
    size_t esz = dtype == CSLAM_F32 ? 4 : 8;
    const int64_t chunk_rows = 16384;
    for (int64_t r0 = 0; r0 < n; r0 += chunk_rows) {
        int64_t m = n - r0 < chunk_rows ? n - r0 : chunk_rows;
        size_t bytes = (size_t)m * b->dim * esz;
        if (bytes > b->stage_bytes) {
            if (b->stage) HIP_TRY(hipFree(b->stage));
            b->stage = nullptr; b->stage_bytes = 0;
            HIP_TRY(hipMalloc((void **)&b->stage, bytes));
            b->stage_bytes = bytes;
        }
        HIP_TRY(hipMemcpy(b->stage, (const char *)vecs + (size_t)r0 * b->dim * esz, bytes, hipMemcpyHostToDevice));
        if (dtype == CSLAM_F32) rc = bank_append_launch<float>(b, (const float *)b->stage, b->dim, m, 0);
        else rc = bank_append_launch<double>(b, (const double *)b->stage, b->dim, m, 0);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(0));
    }
    return CSLAM_OK;
}

CSLAM_API int cslam_bank_add_dev(cslam_bank_t *b, const float *d_vecs, int64_t ld, int64_t n, void *stream) {
    ARG_CHECK(b && (d_vecs || n == 0), "NULL argument");
    ARG_CHECK(n >= 0 && ld >= b->dim, "bad n / ld");
    NO_PENDING(b);
    BANK_DEVICE(b);
    int rc = bank_grow(b, b->n + n);
    if (rc) return rc;
    return bank_append_launch<float>(b, d_vecs, ld, n, (hipStream_t)stream);
}

CSLAM_API int cslam_bank_read_host(const cslam_bank_t *b, int64_t row0, int64_t nrows, float *out) {
    ARG_CHECK(b && (out || nrows == 0), "NULL argument");
    ARG_CHECK(row0 >= 0 && nrows >= 0 && row0 + nrows <= b->cap, "row range outside the bank");
    if (nrows == 0) return CSLAM_OK;
    BANK_DEVICE(b);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy2D(out, (size_t)b->dim * 4, b->rows + row0 * b->ld, (size_t)b->ld * 4,
                        (size_t)b->dim * 4, (size_t)nrows, hipMemcpyDeviceToHost));
    return CSLAM_OK;
}

CSLAM_API int cslam_bank_device_ptr(const cslam_bank_t *b, const float **d_rows, int64_t *ld) {
    ARG_CHECK(b, "bank is NULL");
    if (d_rows) *d_rows = b->rows;
    if (ld) *ld = b->ld;
    return CSLAM_OK;
}

// ---------------------------------------------------------------- exact scan ----
// workgroup size: 8 waves for single-query tiles (4 workgroups/CU fit), 16 waves for 4-query tiles
// (their 64-128 KiB of LDS allows one workgroup per CU, so the waves must come from inside it)
#define SCAN_MAX_WAVES 16
#define LIST_MAX 64

// Query tile in LDS as QS (float or double), [QT][ld]; list merge area after it.
template <typename QS, int QT, bool Q_IN_LDS>
__global__ __launch_bounds__(QT == 1 ? 512 : 1024) void scan_exact_kernel(
    const float *__restrict__ rows, int64_t pitch, int ld, const double *__restrict__ vv, int64_t n_rows,
    const QS *__restrict__ q, int64_t ldq, int dim,
    const int *__restrict__ qsel, int nsel, int sel0,
    int kk, const int64_t *__restrict__ row_limit,
    const double *__restrict__ bound_key, const int *__restrict__ bound_idx,
    double *__restrict__ part_key, int *__restrict__ part_idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    QS *qlds = (QS *)smem;
    const size_t q_bytes = Q_IN_LDS ? (size_t)QT * ld * sizeof(QS) : 0;
    double *mkey = (double *)(smem + q_bytes);                                    // [8][QT][64]
    const int SCAN_WAVES = blockDim.x >> 6, SCAN_THREADS = blockDim.x;
    int *midx = (int *)(smem + q_bytes + (size_t)SCAN_WAVES * QT * LIST_MAX * 8); // [waves][QT][64]

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int G = gridDim.x;
    const int tile = blockIdx.y;
    const int s_base = sel0 + tile * QT;   // first selected-query slot of this tile

    int qn[QT];           // query number (row of q / outputs)
    bool qvalid[QT];
    int64_t lim[QT];
    double bk[QT]; int bi[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        int s = s_base + t;
        qvalid[t] = s < nsel;
        qn[t] = qvalid[t] ? (qsel ? qsel[s] : s) : 0;
        lim[t] = (qvalid[t] && row_limit) ? row_limit[qn[t]] : n_rows;
        if (lim[t] > n_rows) lim[t] = n_rows;
        bk[t] = (bound_key && qvalid[t]) ? bound_key[s] : INFINITY;
        bi[t] = (bound_key && qvalid[t]) ? bound_idx[s] : 0x7fffffff;
    }
    if (Q_IN_LDS) {
#pragma unroll
        for (int t = 0; t < QT; ++t)
            for (int c = threadIdx.x; c < ld; c += SCAN_THREADS)
                qlds[(size_t)t * ld + c] = (qvalid[t] && c < dim) ? q[(size_t)qn[t] * ldq + c] : (QS)0;
        __syncthreads();
    }
    const QS *qp[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) qp[t] = Q_IN_LDS ? (qlds + (size_t)t * ld) : (q + (size_t)qn[t] * ldq);

    // uu = q.q in float64 (per wave, redundantly)
    double uu[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        double a = 0.0;
        for (int c = lane; c < dim; c += 64) { double x = (double)qp[t][c]; a += x * x; }
        uu[t] = wave_allreduce_sum(a);
    }

    WaveList list[QT];
    double thr_k[QT]; int thr_i[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) { list[t].init(); thr_k[t] = -INFINITY; thr_i[t] = -1; }

    const int nchunk = ld >> 2;   // float4 chunks per row
    for (int64_t row = (int64_t)blockIdx.x * SCAN_WAVES + wave; row < n_rows; row += (int64_t)G * SCAN_WAVES) {
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        const f32x4_t *rp = (const f32x4_t *)(rows + row * pitch);
        double acc[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) acc[t] = 0.0;
        // 8 independent 16-byte loads per lane in flight (8 KiB per wave) to cover HBM latency
#pragma unroll 8
        for (int c = lane; c < nchunk; c += 64) {
            f32x4_t b = __builtin_nontemporal_load(rp + c);   // streamed once: keep it out of the way of L2
            // columns >= dim are zero in the bank, so the (possibly unpadded, global) query
            // values there are never multiplied by anything but 0 -- but avoid reading them
            const int col = c * 4;
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                double q0, q1, q2, q3;
                if (Q_IN_LDS || col + 3 < dim) {
                    q0 = (double)qp[t][col]; q1 = (double)qp[t][col + 1];
                    q2 = (double)qp[t][col + 2]; q3 = (double)qp[t][col + 3];
                } else {
                    q0 = col < dim ? (double)qp[t][col] : 0.0;
                    q1 = col + 1 < dim ? (double)qp[t][col + 1] : 0.0;
                    q2 = col + 2 < dim ? (double)qp[t][col + 2] : 0.0;
                    q3 = 0.0;
                }
                acc[t] += (double)b.x * q0;
                acc[t] += (double)b.y * q1;
                acc[t] += (double)b.z * q2;
                acc[t] += (double)b.w * q3;
            }
        }
        const double vvr = vv[row];
        const int irow = (int)row;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            double uv = wave_allreduce_sum(acc[t]);
            if (!qvalid[t] || row >= lim[t]) continue;
            double key = rank_key(sim_from_dots(uv, uu[t], vvr));
            if (!ranks_before(bk[t], bi[t], key, irow)) continue;        // already reported in an earlier pass
            if (!ranks_before(key, irow, thr_k[t], thr_i[t])) continue;  // not in the current top-kk
            list[t].insert(key, irow, lane);
            thr_k[t] = list[t].key_at(kk - 1);
            thr_i[t] = list[t].idx_at(kk - 1);
        }
    }

    // block merge: 8 wave lists -> one list per query, written to part_*[slot][block][kk]
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        mkey[(wave * QT + t) * LIST_MAX + lane] = list[t].key;
        midx[(wave * QT + t) * LIST_MAX + lane] = list[t].idx;
    }
    __syncthreads();
    if (wave < QT && (s_base + wave) < nsel) {
        const int t = wave;
        WaveList m; m.init();
        double tk = -INFINITY; int ti = -1;
        for (int w = 0; w < SCAN_WAVES; ++w) {
            for (int pos = 0; pos < kk; ++pos) {
                double ck = mkey[(w * QT + t) * LIST_MAX + pos];
                int ci = midx[(w * QT + t) * LIST_MAX + pos];
                if (ci < 0 || !ranks_before(ck, ci, tk, ti)) break;   // lists are sorted
                m.insert(ck, ci, lane);
                tk = m.key_at(kk - 1); ti = m.idx_at(kk - 1);
            }
        }
        if (lane < kk) {
            size_t o = ((size_t)(tile * QT + t) * G + blockIdx.x) * kk + lane;
            part_key[o] = m.key;
            part_idx[o] = m.idx;
        }
    }
}

// ---------------------------------------------------------------- exact search, many queries ----
// scan_exact_kernel streams the bank once per 4 queries: the right shape for the online case and for a handful of
// uncertified queries, a cliff for hundreds of them (k > 16 batches, or degenerate data -- near-duplicate descriptors
// -- that defeats the fp32 certificate: 100k queries would re-read a 1.64 GB bank 25 000 times).  This kernel
// computes the same float64 scores for a 64-row x 64-query tile per workgroup: row and query chunks of 64 columns
// transposed into LDS, each thread accumulating a 4 x 4 block of float64 dots in column order, then the wave-resident
// top-k lists (one lane per row of the tile).  The bank is read once per 64 queries and the work is bound by the
// float64 FMA rate.  Same partial-list format as scan_exact_kernel, merged by scan_merge_kernel.
#define XT_TR 64
#define XT_TQ 64
#define XT_KC 64
template <typename QS>
__global__ __launch_bounds__(256) void exact_tile_kernel(
    const float *__restrict__ rows, int64_t pitch, int kd, const double *__restrict__ vv, int64_t n_rows,
    const QS *__restrict__ q, int64_t ldq, int dim,
    const int *__restrict__ qsel, int nsel, int sel0,
    int kk, const int64_t *__restrict__ row_limit,
    const double *__restrict__ bound_key, const int *__restrict__ bound_idx,
    double *__restrict__ part_key, int *__restrict__ part_idx) {
    // operand chunks and the tile's keys share one buffer (the keys are written after the last chunk was consumed)
    __shared__ __attribute__((aligned(16))) char s_buf[XT_KC * XT_TR * 4 + XT_KC * XT_TQ * sizeof(QS) > XT_TQ * XT_TR * 8
                                                        ? XT_KC * XT_TR * 4 + XT_KC * XT_TQ * sizeof(QS) : XT_TQ * XT_TR * 8];
    float (*s_r)[XT_TR] = (float (*)[XT_TR])s_buf;                               // [column][row]
    QS (*s_q)[XT_TQ] = (QS (*)[XT_TQ])(s_buf + XT_KC * XT_TR * 4);                // [column][query]
    double (*s_key)[XT_TR] = (double (*)[XT_TR])s_buf;                           // [query][row]
    __shared__ double s_uu[XT_TQ];
    __shared__ int s_qn[XT_TQ];
    __shared__ int64_t s_lim[XT_TQ];
    __shared__ double s_bk[XT_TQ];
    __shared__ int s_bi[XT_TQ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x, tile = blockIdx.y;
    const int s_base = sel0 + tile * XT_TQ;
    if (tid < XT_TQ) {
        const int s = s_base + tid;
        const bool ok = s < nsel;
        const int qn = ok ? (qsel ? qsel[s] : s) : -1;
        s_qn[tid] = qn;
        int64_t lim = (ok && row_limit) ? row_limit[qn] : n_rows;
        s_lim[tid] = ok ? (lim > n_rows ? n_rows : lim) : 0;
        s_bk[tid] = (bound_key && ok) ? bound_key[s] : INFINITY;
        s_bi[tid] = (bound_key && ok) ? bound_idx[s] : 0x7fffffff;
    }
    __syncthreads();
    constexpr int QW = XT_TQ / 4;                                // queries (and lists) per wave
    for (int t = wave * QW; t < wave * QW + QW; ++t) {           // uu = q.q in float64
        const int qn = s_qn[t];
        double a = 0.0;
        if (qn >= 0)
            for (int c = lane; c < dim; c += 64) { double x = (double)q[(size_t)qn * ldq + c]; a += x * x; }
        a = wave_allreduce_sum(a);
        if (lane == 0) s_uu[t] = a;
    }
    WaveList list[QW];
#pragma unroll
    for (int t = 0; t < QW; ++t) list[t].init();
    const int ri = tid & 15, qi = tid >> 4;                      // thread: rows 4ri .. 4ri+3 x queries 4qi .. 4qi+3
    const int lrow = tid & 63, lk = (tid >> 6) * 16;             // loader: row / query of the tile, 16 columns of the chunk
    const int ntiles = (int)((n_rows + XT_TR - 1) / XT_TR);
    for (int rt = blockIdx.x; rt < ntiles; rt += G) {
        const int64_t row0 = (int64_t)rt * XT_TR;
        double acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
        for (int kc = 0; kc < kd; kc += XT_KC) {
            __syncthreads();
            {
                const int64_t r = row0 + lrow < n_rows ? row0 + lrow : n_rows - 1;
                const float *src = rows + r * pitch + kc + lk;
                const int qn = s_qn[lrow];
#pragma unroll
                for (int j = 0; j < 16; j += 4) {
                    float4 v = (kc + lk + j < kd) ? *(const float4 *)(src + j) : make_float4(0.f, 0.f, 0.f, 0.f);
                    s_r[lk + j][lrow] = v.x; s_r[lk + j + 1][lrow] = v.y;
                    s_r[lk + j + 2][lrow] = v.z; s_r[lk + j + 3][lrow] = v.w;
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int c = kc + lk + j;
                    s_q[lk + j][lrow] = (qn >= 0 && c < dim) ? q[(size_t)qn * ldq + c] : (QS)0;
                }
            }
            __syncthreads();
#pragma unroll 4
            for (int k = 0; k < XT_KC; ++k) {
                const float4 rv = *(const float4 *)&s_r[k][4 * ri];
                const double r[4] = {(double)rv.x, (double)rv.y, (double)rv.z, (double)rv.w};
                const double qv[4] = {(double)s_q[k][4 * qi], (double)s_q[k][4 * qi + 1], (double)s_q[k][4 * qi + 2],
                                      (double)s_q[k][4 * qi + 3]};
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] += r[a] * qv[b];
            }
        }
        __syncthreads();                                         // last chunk consumed: the buffer becomes s_key
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int64_t row = row0 + 4 * ri + a;
            const double vvr = row < n_rows ? vv[row] : 1.0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int t = 4 * qi + b;
                double key = -INFINITY;                          // masked out
                if (row < s_lim[t]) {
                    key = rank_key(sim_from_dots(acc[a][b], s_uu[t], vvr));
                    if (!ranks_before(s_bk[t], s_bi[t], key, (int)row)) key = -INFINITY;   // reported by an earlier pass
                }
                s_key[t][4 * ri + a] = key;
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < QW; ++t) {
            const double key = s_key[wave * QW + t][lane];
            const int idx = (int)(row0 + lane);
            double tk = list[t].key_at(kk - 1);
            int ti = list[t].idx_at(kk - 1);
            unsigned long long m = __ballot(key != -INFINITY && ranks_before(key, idx, tk, ti));
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const double ck = __shfl(key, src, 64);
                const int ci = __shfl(idx, src, 64);
                if (ranks_before(ck, ci, tk, ti)) {
                    list[t].insert(ck, ci, lane);
                    tk = list[t].key_at(kk - 1);
                    ti = list[t].idx_at(kk - 1);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < QW; ++t) {
        const int slot = tile * XT_TQ + wave * QW + t;
        if (s_base + wave * QW + t < nsel && lane < kk) {
            const size_t o = ((size_t)slot * G + blockIdx.x) * kk + lane;
            part_key[o] = list[t].key;
            part_idx[o] = list[t].idx;
        }
    }
}

template <typename QS>
static int exact_tile_launch(cslam_bank *b, const QS *d_q, int64_t ldq, const int *d_qsel, int nsel, int sel0,
                             int nchunk, int kk, const int64_t *d_row_limit, const double *bkey, const int *bidx,
                             double *part_key, int *part_idx, int G, hipStream_t st) {
    dim3 grid((unsigned)G, (unsigned)ceil_div64(nchunk, XT_TQ));
    hipLaunchKernelGGL(exact_tile_kernel<QS>, grid, dim3(256), 0, st, b->rows, (int64_t)b->ld, b->kd, b->vv, b->n, d_q,
                       ldq, b->dim, d_qsel, nsel, sel0, kk, d_row_limit, bkey, bidx, part_key, part_idx);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

// One wave per selected query: merge the G per-block lists, write the results.
__global__ __launch_bounds__(64) void scan_merge_kernel(
    const double *__restrict__ part_key, const int *__restrict__ part_idx, int G, int kk,
    const int *__restrict__ qsel, int sel0, int k_total, int k_off,
    int64_t *__restrict__ out_idx, double *__restrict__ out_sim, int32_t *__restrict__ out_cnt,
    double *__restrict__ bound_key, int *__restrict__ bound_idx) {
    const int lane = threadIdx.x;
    const int slot = blockIdx.x;            // slot within this launch's chunk
    const int s = sel0 + slot;
    const int qn = qsel ? qsel[s] : s;
    const double *pk = part_key + (size_t)slot * G * kk;
    const int *pi = part_idx + (size_t)slot * G * kk;
    WaveList m; m.init();
    double tk = -INFINITY; int ti = -1;
    const int total = G * kk;
    for (int base = 0; base < total; base += 64) {
        int e = base + lane;
        double ck = e < total ? pk[e] : -INFINITY;
        int ci = e < total ? pi[e] : -1;
        unsigned long long mask = __ballot(ci >= 0 && ranks_before(ck, ci, tk, ti));
        while (mask) {
            int src = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            double k2 = __shfl(ck, src, 64);
            int i2 = __shfl(ci, src, 64);
            if (ranks_before(k2, i2, tk, ti)) {
                m.insert(k2, i2, lane);
                tk = m.key_at(kk - 1); ti = m.idx_at(kk - 1);
            }
        }
    }
    unsigned long long vmask = __ballot(lane < kk && m.idx >= 0);
    const int cnt = __popcll(vmask);
    const double lastk = m.key_at(cnt > 0 ? cnt - 1 : 0);
    const int lasti = m.idx_at(cnt > 0 ? cnt - 1 : 0);
    if (lane < kk) {
        size_t o = (size_t)qn * k_total + k_off + lane;
        out_idx[o] = m.idx >= 0 ? (int64_t)m.idx : -1;
        out_sim[o] = (m.idx >= 0 && m.key != INFINITY) ? m.key : NAN;
    }
    if (lane == 0) {
        out_cnt[qn] = (k_off == 0 ? 0 : out_cnt[qn]) + cnt;
        // the next pass (k > 64) reports entries ranking strictly after the last one written here
        if (bound_key && cnt > 0) { bound_key[s] = lastk; bound_idx[s] = lasti; }
    }
}

__global__ void fill_bounds_kernel(double *bk, int *bi, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { bk[i] = INFINITY; bi[i] = 0x7fffffff; }
}

template <typename QS>
static int scan_launch(cslam_bank *b, const QS *d_q, int64_t ldq, const int *d_qsel, int nsel, int sel0,
                       int nchunk, int kk, const int64_t *d_row_limit, const double *bkey, const int *bidx,
                       double *part_key, int *part_idx, int G, hipStream_t st) {
    const int wide = 16, narrow = 8;       // waves per workgroup for QT = 4 / QT = 1
    size_t merge_bytes = (size_t)wide * LIST_MAX * 12;
    const size_t lds_budget = 150 * 1024;
    int qt = 4;
    if ((size_t)4 * b->kd * sizeof(QS) + 4 * merge_bytes > lds_budget || nchunk == 1) qt = 1;
    const int waves = qt == 4 ? wide : narrow;
    merge_bytes = (size_t)waves * LIST_MAX * 12;
    bool in_lds = (size_t)qt * b->kd * sizeof(QS) + qt * merge_bytes <= lds_budget;
    size_t lds = (in_lds ? (size_t)qt * b->kd * sizeof(QS) : 0) + qt * merge_bytes;
    dim3 grid((unsigned)G, (unsigned)ceil_div64(nchunk, qt));
#define SCAN_GO(QT, INL)                                                                              \
    do {                                                                                              \
        auto kern = scan_exact_kernel<QS, QT, INL>;                                                   \
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                    (int)lds));                                                       \
        hipLaunchKernelGGL(kern, grid, dim3(waves * 64), lds, st, b->rows, (int64_t)b->ld, b->kd, b->vv, b->n, d_q, \
                           ldq, b->dim, d_qsel, nsel, sel0, kk, d_row_limit, bkey, bidx, part_key,    \
                           part_idx);                                                                 \
    } while (0)
    if (qt == 4) SCAN_GO(4, true);
    else if (in_lds) SCAN_GO(1, true);
    else SCAN_GO(1, false);
#undef SCAN_GO
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

int scan_search(cslam_bank *b, const void *d_q, int q_dtype, int64_t ldq, const int *d_qsel,
                int64_t nsel, int k, const int64_t *d_row_limit, int64_t *d_out_idx,
                double *d_out_sim, int32_t *d_out_cnt, hipStream_t st) {
    if (nsel == 0) return CSLAM_OK;
    // grid: enough waves to cover the HBM latency, never more blocks than 8-row groups
    int G = b->num_cu * 2;
    int64_t need = ceil_div64(b->n > 0 ? b->n : 1, 8);
    if (G > need) G = (int)need;
    if (G < 1) G = 1;
    const int CH = 1024;   // selected queries per launch (bounds the partial-list workspace)
    const int passes = (int)ceil_div64(k, LIST_MAX);
    // many queries: 64-row x 32-query float64 tiles (exact_tile_kernel) instead of one bank pass per 4 queries
    const bool tiled = nsel >= 16;
    if (tiled) {
        const int64_t qtiles = ceil_div64(nsel < CH ? nsel : CH, XT_TQ), rtiles = ceil_div64(b->n > 0 ? b->n : 1, XT_TR);
        G = (int)ceil_div64((int64_t)b->num_cu * 2, qtiles);
        if (G > rtiles) G = (int)rtiles;
        if (G < 1) G = 1;
    }
    // partial lists: (queries per launch, rounded up to whole query tiles) x G x entries per list
    const size_t ch_eff = (size_t)round_up64(nsel < CH ? nsel : CH, tiled ? XT_TQ : 4);
    size_t part_elems = ch_eff * G * (size_t)(k < LIST_MAX ? k : LIST_MAX);
    size_t off_pk = 0, off_pi = off_pk + part_elems * 8, off_bk = off_pi + part_elems * 4;
    size_t off_bi = off_bk + (size_t)nsel * 8, total = off_bi + (size_t)nsel * 4;
    int rc = bank_ws_reserve(b, 1, total);
    if (rc) return rc;
    double *part_key = (double *)(b->ws[1] + off_pk);
    int *part_idx = (int *)(b->ws[1] + off_pi);
    double *bkey = passes > 1 ? (double *)(b->ws[1] + off_bk) : nullptr;
    int *bidx = passes > 1 ? (int *)(b->ws[1] + off_bi) : nullptr;
    if (passes > 1) {
        hipLaunchKernelGGL(fill_bounds_kernel, dim3((unsigned)ceil_div64(nsel, 256)), dim3(256), 0, st, bkey,
                           bidx, (int)nsel);
        HIP_TRY(hipGetLastError());
    }
    for (int p = 0; p < passes; ++p) {
        int kk = k - p * LIST_MAX < LIST_MAX ? k - p * LIST_MAX : LIST_MAX;
        for (int64_t s0 = 0; s0 < nsel; s0 += CH) {
            int nchunk = (int)(nsel - s0 < CH ? nsel - s0 : CH);
            if (tiled && q_dtype == CSLAM_F32)
                rc = exact_tile_launch<float>(b, (const float *)d_q, ldq, d_qsel, (int)nsel, (int)s0, nchunk, kk,
                                              d_row_limit, bkey, bidx, part_key, part_idx, G, st);
            else if (tiled)
                rc = exact_tile_launch<double>(b, (const double *)d_q, ldq, d_qsel, (int)nsel, (int)s0, nchunk, kk,
                                               d_row_limit, bkey, bidx, part_key, part_idx, G, st);
            else if (q_dtype == CSLAM_F32)
                rc = scan_launch<float>(b, (const float *)d_q, ldq, d_qsel, (int)nsel, (int)s0, nchunk, kk,
                                        d_row_limit, bkey, bidx, part_key, part_idx, G, st);
            else
                rc = scan_launch<double>(b, (const double *)d_q, ldq, d_qsel, (int)nsel, (int)s0, nchunk, kk,
                                         d_row_limit, bkey, bidx, part_key, part_idx, G, st);
            if (rc) return rc;
            hipLaunchKernelGGL(scan_merge_kernel, dim3((unsigned)nchunk), dim3(64), 0, st, part_key, part_idx, G,
                               kk, d_qsel, (int)s0, k, p * LIST_MAX, d_out_idx, d_out_sim, d_out_cnt, bkey, bidx);
            HIP_TRY(hipGetLastError());
        }
    }
    return CSLAM_OK;
}

// ------------------------------------------------------------ search entry ----
// A search = enqueue (every kernel of it, on the caller's stream, no host synchronisation) + finish (wait for the 4-byte
// count of uncertified queries -- an EVENT recorded right behind its copy, not the stream: work the caller enqueued later,
// the next step's extraction, keeps running -- and enqueue the exact-scan fallback for them, rare).  Results are valid in
// stream order after finish.  One search per bank may be in flight; queries, row limits and outputs must stay alive and
// the bank unchanged (no add / clear) until it is finished.
static int pick_mode(const cslam_bank *b, int mode, int64_t nq, int k) {
    int use = mode;
    // the MFMA path keeps 16 merged candidates per (query, segment): k <= 16 (the reference's default
    // nb_best_matches is 10); larger k, tiny banks and single queries use the exact scan
    if (use == CSLAM_MODE_AUTO) use = (nq <= 8 || k > 16 || b->n < 256) ? CSLAM_MODE_SCAN : CSLAM_MODE_MFMA;
    if (use == CSLAM_MODE_MFMA && (k > 16 || b->n < 1)) use = CSLAM_MODE_SCAN;   // empty bank: the scan writes cnt = 0
    return use;
}

// st_call: the stream the kernels go to; st_join: the stream the results are ordered on (= st_call unless the searches of a
// bank list run side by side on per-bank streams that are joined back into it)
static int bank_enqueue(cslam_bank *b, const void *d_queries, int q_dtype, int64_t ldq, int64_t nq, int k,
                        const int64_t *d_row_limit, int mode, int64_t *d_out_idx, double *d_out_sim, int32_t *d_out_cnt,
                        hipStream_t st_call, hipStream_t st_join) {
    b->last_stream = st_join;
    b->stats[0] = 0; b->stats[2] = 0; b->stats[3] = 0;
    const int use = pick_mode(b, mode, nq, k);
    b->stats[1] = use;
    cslam_bank::Pending &p = b->pend;
    p.deferred = use == CSLAM_MODE_MFMA;
    p.q = d_queries; p.q_dtype = q_dtype; p.ldq = ldq; p.k = k; p.lim = d_row_limit;
    p.oi = d_out_idx; p.os = d_out_sim; p.oc = d_out_cnt; p.st = st_join;
    int rc;
    if (!p.deferred) {
        rc = scan_search(b, d_queries, q_dtype, ldq, nullptr, nq, k, d_row_limit, d_out_idx, d_out_sim, d_out_cnt, st_call);
    } else {
        if (!b->ev_flag) HIP_TRY(hipEventCreateWithFlags(&b->ev_flag, hipEventDisableTiming));
        rc = mfma_search_enqueue(b, d_queries, q_dtype, ldq, nq, k, d_row_limit, d_out_idx, d_out_sim, d_out_cnt, st_call);
        if (rc == CSLAM_OK) HIP_TRY(hipEventRecord(b->ev_flag, st_call));
    }
    p.active = rc == CSLAM_OK;
    return rc;
}

static int bank_finish(cslam_bank *b, int64_t *n_uncertified) {
    if (n_uncertified) *n_uncertified = 0;
    cslam_bank::Pending &p = b->pend;
    if (!p.active) return CSLAM_OK;
    p.active = false;
    if (!p.deferred) return CSLAM_OK;
    HIP_TRY(hipEventSynchronize(b->ev_flag));
    int rc = mfma_search_finish(b, p.q, p.q_dtype, p.ldq, p.k, p.lim, p.oi, p.os, p.oc, p.st);
    if (n_uncertified) *n_uncertified = b->stats[0];
    return rc;
}

static int search_args_ok(const cslam_bank *b, const void *d_queries, int q_dtype, int64_t ldq, int64_t nq, int k,
                          const int64_t *d_out_idx, const double *d_out_sim, const int32_t *d_out_cnt) {
    ARG_CHECK(b && (d_queries || nq == 0) && d_out_idx && d_out_sim && d_out_cnt, "NULL argument");
    ARG_CHECK(q_dtype == CSLAM_F32 || q_dtype == CSLAM_F64, "q_dtype must be CSLAM_F32 or CSLAM_F64");
    ARG_CHECK(nq >= 0 && k >= 1, "nq must be >= 0 and k >= 1");
    ARG_CHECK(ldq >= b->dim, "ldq < dim");
    ARG_CHECK(nq < (1LL << 31) && b->n < (1LL << 31), "nq / bank rows must be < 2^31");
    NO_PENDING(b);
    return CSLAM_OK;
}

CSLAM_API int cslam_bank_search_enqueue_dev(cslam_bank_t *b, const void *d_queries, int q_dtype, int64_t ldq,
                                            int64_t nq, int k, const int64_t *d_row_limit, int mode,
                                            int64_t *d_out_idx, double *d_out_sim, int32_t *d_out_cnt, void *stream) {
    int rc = search_args_ok(b, d_queries, q_dtype, ldq, nq, k, d_out_idx, d_out_sim, d_out_cnt);
    if (rc) return rc;
    BANK_DEVICE(b);
    hipStream_t st = (hipStream_t)stream;
    if (nq == 0) { b->last_stream = st; b->stats[0] = 0; b->stats[2] = 0; b->stats[3] = 0; return CSLAM_OK; }
    return bank_enqueue(b, d_queries, q_dtype, ldq, nq, k, d_row_limit, mode, d_out_idx, d_out_sim, d_out_cnt, st, st);
}

/* The uncertified-query count of the enqueued search, handed on INSIDE the stream: a 4-byte device-to-device copy enqueued on
 * `stream` (the stream of the enqueue) into d_count -- 0 for a search that needs no certificate (exact scan).  A sharded step
 * sends it along with its provisional lists, so that every rank learns in stream order, without a host round trip, whether some
 * shard still owes an exact re-scan (cslam_amd/sharded.py: step_begin / finish). */
CSLAM_API int cslam_bank_search_flag_copy_dev(cslam_bank_t *b, int32_t *d_count, void *stream) {
    ARG_CHECK(b && d_count, "NULL argument");
    BANK_DEVICE(b);
    hipStream_t st = (hipStream_t)stream;
    if (b->pend.active && b->pend.deferred && b->pending_flag_count)
        HIP_TRY(hipMemcpyAsync(d_count, b->pending_flag_count, 4, hipMemcpyDeviceToDevice, st));
    else
        HIP_TRY(hipMemsetAsync(d_count, 0, 4, st));
    return CSLAM_OK;
}

CSLAM_API int cslam_bank_search_finish(cslam_bank_t *b, int64_t *n_uncertified) {
    ARG_CHECK(b, "bank is NULL");
    BANK_DEVICE(b);
    return bank_finish(b, n_uncertified);
}

CSLAM_API int cslam_bank_search_dev(cslam_bank_t *b, const void *d_queries, int q_dtype, int64_t ldq,
                                    int64_t nq, int k, const int64_t *d_row_limit, int mode,
                                    int64_t *d_out_idx, double *d_out_sim, int32_t *d_out_cnt,
                                    void *stream) {
    int rc = cslam_bank_search_enqueue_dev(b, d_queries, q_dtype, ldq, nq, k, d_row_limit, mode, d_out_idx, d_out_sim,
                                           d_out_cnt, stream);
    if (rc) return rc;
    return cslam_bank_search_finish(b, nullptr);
}

// One batch of queries against SEVERAL banks of one device (a robot's local bank and its copies of the other robots'
// banks: lcsm.py:21-31, searched one after the other by lcsm.py:45-53 / gdlcd.py:157-160): every bank's kernels are
// enqueued first (multi_enqueue), the uncertified-query counts are waited for bank by bank and the (rare) exact-scan
// fallbacks run (multi_finish).  Results are those of nb separate cslam_bank_search_dev calls.
CSLAM_API int cslam_bank_search_multi_enqueue_dev(cslam_bank_t *const *banks, int nb, const void *d_queries, int q_dtype,
                                                  int64_t ldq, int64_t nq, const int *k, const int64_t *const *d_row_limit,
                                                  int mode, int64_t *const *d_out_idx, double *const *d_out_sim,
                                                  int32_t *const *d_out_cnt, void *stream) {
    ARG_CHECK(banks && nb >= 1 && nb <= 64 && k && d_out_idx && d_out_sim && d_out_cnt, "bad bank list");
    ARG_CHECK((d_queries || nq == 0) && nq >= 0, "NULL queries");
    ARG_CHECK(q_dtype == CSLAM_F32 || q_dtype == CSLAM_F64, "q_dtype must be CSLAM_F32 or CSLAM_F64");
    for (int i = 0; i < nb; ++i) {
        ARG_CHECK(banks[i] && d_out_idx[i] && d_out_sim[i] && d_out_cnt[i] && k[i] >= 1, "NULL bank / output or k < 1");
        ARG_CHECK(banks[i]->device == banks[0]->device && banks[i]->dim == banks[0]->dim, "banks must share device and dimension");
        ARG_CHECK(ldq >= banks[i]->dim, "ldq smaller than the descriptor dimension");
        ARG_CHECK(nq < (1LL << 31) && banks[i]->n < (1LL << 31), "nq / bank rows must be < 2^31");
        NO_PENDING(banks[i]);
        // every bank has ONE workspace, side stream and pending slot: the same handle twice would race on them
        for (int j = 0; j < i; ++j) ARG_CHECK(banks[j] != banks[i], "the same bank appears twice in the list");
    }
    if (nq == 0) return CSLAM_OK;
    BANK_DEVICE(banks[0]);
    hipStream_t st = (hipStream_t)stream;
    // The searches of the list are independent and, on chunk-sized query batches against banks of a few thousand rows, far
    // too small to fill the chip one after the other (250 queries x 12 500 rows = 49 workgroups): each bank enqueues on a
    // stream of its own, forked from and joined back into the caller's stream (one after the other: C5 rehearsal's local
    // matching 2.5 instead of 2.0 s, profiles/r02_v50_perf_c5_drain_multistream.log).
    const bool fork = nb > 1;
    if (fork) {
        for (int i = 0; i < nb; ++i) {
            cslam_bank *b = banks[i];
            if (!b->side) {
                HIP_TRY(hipStreamCreateWithFlags(&b->side, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&b->ev_side, hipEventDisableTiming));
            }
        }
        HIP_TRY(hipEventRecord(banks[0]->ev_fork, st));
    }
    int rc = CSLAM_OK;
    for (int i = 0; i < nb && rc == CSLAM_OK; ++i) {
        cslam_bank *b = banks[i];
        hipStream_t st_call = st;
        if (fork) { HIP_TRY(hipStreamWaitEvent(b->side, banks[0]->ev_fork, 0)); st_call = b->side; }
        rc = bank_enqueue(b, d_queries, q_dtype, ldq, nq, k[i], d_row_limit ? d_row_limit[i] : nullptr, mode, d_out_idx[i],
                          d_out_sim[i], d_out_cnt[i], st_call, st);
        if (fork) {                                                 // joined even when the enqueue failed: the caller's stream stays ordered
            (void)hipEventRecord(b->ev_side, b->side);
            (void)hipStreamWaitEvent(st, b->ev_side, 0);
        }
    }
    if (rc != CSLAM_OK) {
        // a bank in the middle failed: nothing of this call stays pending (the earlier banks' kernels run to completion, their
        // results are not used)
        char keep[512];
        snprintf(keep, sizeof keep, "%s", g_err);
        (void)hipStreamSynchronize(st);
        for (int i = 0; i < nb; ++i) banks[i]->pend.active = false;
        cslam_set_error("%s", keep);
    }
    return rc;
}

CSLAM_API int cslam_bank_search_multi_finish(cslam_bank_t *const *banks, int nb, int64_t *n_uncertified) {
    ARG_CHECK(banks && nb >= 1 && nb <= 64, "bad bank list");
    for (int i = 0; i < nb; ++i) ARG_CHECK(banks[i], "NULL bank");
    BANK_DEVICE(banks[0]);
    int first = CSLAM_OK;
    int64_t total = 0;
    for (int i = 0; i < nb; ++i) {                                  // every bank is finished even when one fallback fails
        int64_t n = 0;
        const int rc = bank_finish(banks[i], &n);
        total += n;
        if (rc != CSLAM_OK && first == CSLAM_OK) first = rc;
    }
    if (n_uncertified) *n_uncertified = total;
    return first;
}

CSLAM_API int cslam_bank_search_multi_dev(cslam_bank_t *const *banks, int nb, const void *d_queries, int q_dtype,
                                          int64_t ldq, int64_t nq, const int *k, const int64_t *const *d_row_limit,
                                          int mode, int64_t *const *d_out_idx, double *const *d_out_sim,
                                          int32_t *const *d_out_cnt, void *stream) {
    int rc = cslam_bank_search_multi_enqueue_dev(banks, nb, d_queries, q_dtype, ldq, nq, k, d_row_limit, mode, d_out_idx,
                                                 d_out_sim, d_out_cnt, stream);
    if (rc || nq == 0) return rc;
    return cslam_bank_search_multi_finish(banks, nb, nullptr);
}

CSLAM_API int cslam_bank_search_host(cslam_bank_t *b, const void *queries, int q_dtype, int64_t nq,
                                     int k, const int64_t *row_limit, int mode,
                                     int64_t *out_idx, double *out_sim, int32_t *out_cnt) {
    ARG_CHECK(b && (queries || nq == 0) && out_idx && out_sim && out_cnt, "NULL argument");
    ARG_CHECK(q_dtype == CSLAM_F32 || q_dtype == CSLAM_F64, "q_dtype must be CSLAM_F32 or CSLAM_F64");
    ARG_CHECK(nq >= 0 && k >= 1, "nq must be >= 0 and k >= 1");
    if (nq == 0) return CSLAM_OK;
    BANK_DEVICE(b);
    const size_t esz = q_dtype == CSLAM_F32 ? 4 : 8;
    const size_t qb = (size_t)nq * b->dim * esz;
    const size_t lb = row_limit ? (size_t)nq * 8 : 0;
    const size_t ib = (size_t)nq * k * 8, sb = (size_t)nq * k * 8, cb = (size_t)nq * 4;
    size_t off_q = 0, off_l = round_up64(off_q + qb, 256), off_i = round_up64(off_l + lb, 256);
    size_t off_s = round_up64(off_i + ib, 256), off_c = round_up64(off_s + sb, 256);
    size_t total = off_c + cb;
    // persistent per-bank staging (no hipMalloc/hipFree on the per-keyframe path)
    int rc = bank_ws_reserve(b, 2, total);
    if (rc) return rc;
    char *buf = b->ws[2];
    do {
        if (hipMemcpy(buf + off_q, queries, qb, hipMemcpyHostToDevice) != hipSuccess) { rc = CSLAM_E_HIP; break; }
        if (row_limit && hipMemcpy(buf + off_l, row_limit, lb, hipMemcpyHostToDevice) != hipSuccess) { rc = CSLAM_E_HIP; break; }
        rc = cslam_bank_search_dev(b, buf + off_q, q_dtype, b->dim, nq, k,
                                   row_limit ? (const int64_t *)(buf + off_l) : nullptr, mode,
                                   (int64_t *)(buf + off_i), (double *)(buf + off_s), (int32_t *)(buf + off_c), 0);
        if (rc) break;
        if (hipStreamSynchronize(0) != hipSuccess) { rc = CSLAM_E_HIP; break; }
        if (hipMemcpy(out_idx, buf + off_i, ib, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(out_sim, buf + off_s, sb, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(out_cnt, buf + off_c, cb, hipMemcpyDeviceToHost) != hipSuccess) { rc = CSLAM_E_HIP; break; }
    } while (0);
    if (rc == CSLAM_E_HIP && g_err[0] == 0) cslam_set_error("HIP copy failed: %s", hipGetErrorString(hipGetLastError()));
    return rc;
}

CSLAM_API int cslam_bank_last_stats(cslam_bank_t *b, int64_t stats[4]) {
    ARG_CHECK(b && stats, "NULL argument");
    BANK_DEVICE(b);
    HIP_TRY(hipStreamSynchronize(b->last_stream));
    for (int i = 0; i < 4; ++i) stats[i] = b->stats[i];
    return CSLAM_OK;
}

CSLAM_API int cslam_bank_last_stage(cslam_bank_t *b, int32_t info[4]) {
    ARG_CHECK(b && info, "NULL argument");
    info[0] = b->last_nprod; info[1] = b->f32_backoff; info[2] = b->f32_backoff_len; info[3] = b->stage_pinned ? 1 : 0;
    return CSLAM_OK;
}

// ---- merge of per-shard top-k lists (one bank row-sharded over several GPUs) ----------------
// The global top-k of a query is contained in the union of its per-shard top-k lists; shard s holds the
// global rows [row_offset[s], row_offset[s] + n_s).  One thread per query picks the k best of the <= S*k
// entries in the shared ranking order (larger key, NaN first, ties -> larger GLOBAL row): k rounds of
// "best entry ranking after the previous pick".  Rows are unique across shards, so the order is strict.
struct MergeOffsets { int64_t off[64]; };

__global__ __launch_bounds__(64) void topk_merge_kernel(const int64_t *__restrict__ idx, const double *__restrict__ sim,
                                                        const int32_t *__restrict__ cnt, MergeOffsets offs, int S,
                                                        int64_t nq, int k, int64_t *__restrict__ out_idx,
                                                        double *__restrict__ out_sim, int32_t *__restrict__ out_cnt) {
    int64_t q = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (q >= nq) return;
    double last_key = 0.0;
    int64_t last_row = 0;
    int produced = 0;
    for (int j = 0; j < k; ++j) {
        bool have = false;
        double bkey = 0.0, bsim = 0.0;
        int64_t brow = -1;
        for (int s = 0; s < S; ++s) {
            int c = cnt[(int64_t)s * nq + q];
            c = c < 0 ? 0 : (c > k ? k : c);
            const int64_t *li = idx + ((int64_t)s * nq + q) * k;
            const double *ls = sim + ((int64_t)s * nq + q) * k;
            for (int e = 0; e < c; ++e) {
                double v = ls[e];
                double key = rank_key(v);
                int64_t row = li[e] + offs.off[s];
                // entry must rank strictly after the previous pick ...
                if (j > 0 && !(last_key > key || (last_key == key && last_row > row))) continue;
                // ... and before the best seen so far in this round
                if (!have || key > bkey || (key == bkey && row > brow)) { have = true; bkey = key; bsim = v; brow = row; }
                else break;   // the list is sorted: nothing further down can beat this shard's first eligible entry
            }
        }
        if (!have) break;
        out_idx[q * k + j] = brow;
        out_sim[q * k + j] = bsim;
        last_key = bkey; last_row = brow;
        ++produced;
    }
    for (int j = produced; j < k; ++j) { out_idx[q * k + j] = -1; out_sim[q * k + j] = NAN; }
    out_cnt[q] = produced;
}

CSLAM_API int cslam_topk_merge_dev(const int64_t *d_idx, const double *d_sim, const int32_t *d_cnt,
                                   const int64_t *row_offset, int shards, int64_t nq, int k,
                                   int64_t *d_out_idx, double *d_out_sim, int32_t *d_out_cnt, void *stream) {
    PTR_DEVICE(d_idx);
    ARG_CHECK(d_idx && d_sim && d_cnt && row_offset && d_out_idx && d_out_sim && d_out_cnt, "NULL argument");
    ARG_CHECK(shards >= 1 && shards <= 64, "shards must be in [1, 64]");
    ARG_CHECK(nq >= 0 && k >= 1, "nq >= 0 and k >= 1 required");
    if (nq == 0) return CSLAM_OK;
    MergeOffsets offs;
    for (int s = 0; s < 64; ++s) offs.off[s] = s < shards ? row_offset[s] : 0;
    hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)ceil_div64(nq, 64)), dim3(64), 0, (hipStream_t)stream,
                       d_idx, d_sim, d_cnt, offs, shards, nq, k, d_out_idx, d_out_sim, d_out_cnt);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}
