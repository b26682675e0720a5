// cslam_fiedler: (lambda_2, v_2) of a pose-graph Laplacian for a host that has no Python -- the whole of
// cslam/mac/mac.py:35-59 (`find_fiedler_pair` -> networkx `algebraic_connectivity(method='tracemin_lu')`, third party)
// behind one C call.  Same TraceMIN-Fiedler iteration as cslam_amd/mac/fiedler.py (start block of numpy's
// RandomState(seed).normal(size=(4, n)).T, projection on 1-perp, Rayleigh-Ritz, stopping rule ||L v - s v||_1 / ||L||_inf < tol),
// with the inner solves by the chain reduction of cslam_amd/mac/chain_solver.py:
//   host (this file)   chain / junction structure from the CSR arrays, O(nnz) loops; connectivity check on the reduced graph
//   device (this file) the 4 x 4 algebra of the TraceMIN loop (Cholesky, inverse, Jacobi) as one-thread kernels between the
//                      streaming passes: one host synchronisation per iteration (the stopping rule)
//   mac_kernels.hip    segmented scans, back substitution, L X, 4-column block products, blocked triangular solves
//   rocBLAS/rocSOLVER  dense float64 Cholesky of the grounded junction Laplacian (diagonal blocks: potrf, panel: trsm,
//                      trailing matrix: gemm over the stored triangle only), resolved at run time like RCCL in comm.hip
// Device memory is a grow-only workspace kept between calls (MAC calls this once per Frank-Wolfe iteration with a slightly
// larger junction system each time; a fresh 8 GB hipMalloc/hipFree per call cost ~0.1 s); cslam_fiedler_release frees it.
#include <dlfcn.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <vector>
#include "common.h"

int block4_residual_devsigma(const double *d_W, const double *d_X, int64_t n, const double *d_y4, const double *d_sigma,
                             double *d_partial, double *d_out1, hipStream_t st);   // mac_kernels.hip

// ---------------------------------------------------------------------------------------------
// numpy RandomState(seed).normal(): MT19937 (init_genrand seeding), 53-bit doubles, polar Box-Muller with the
// second variate cached (numpy/random/src/legacy/legacy-distributions.c `legacy_gauss`, third party; restated)
namespace {
struct LegacyNormal {
    uint32_t mt[624];
    int pos;
    bool has_gauss;
    double gauss;
    explicit LegacyNormal(uint32_t seed) : pos(624), has_gauss(false), gauss(0.0) {
        mt[0] = seed;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    }
    void refill() {
        for (int k = 0; k < 624; ++k) {
            const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        pos = 0;
    }
    uint32_t next32() {
        if (pos >= 624) refill();
        uint32_t y = mt[pos++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
    double next_double() {
        const int32_t a = (int32_t)(next32() >> 5), b = (int32_t)(next32() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    double normal() {
        if (has_gauss) { has_gauss = false; const double t = gauss; gauss = 0.0; return t; }
        double x1, x2, r2;
        do {
            x1 = 2.0 * next_double() - 1.0;
            x2 = 2.0 * next_double() - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        const double f = sqrt(-2.0 * log(r2) / r2);
        gauss = f * x1; has_gauss = true;
        return f * x2;
    }
};
}  // namespace

CSLAM_API int cslam_fiedler_start_block(uint32_t seed, int64_t n, double *h_x0) {
    ARG_CHECK(h_x0 && n >= 1, "bad argument");
    LegacyNormal g(seed);
    for (int c = 0; c < 4; ++c)                           // normal(size=(4, n)) row by row, stored transposed [n][4]
        for (int64_t k = 0; k < n; ++k) h_x0[k * 4 + c] = g.normal();
    return CSLAM_OK;
}

// ---------------------------------------------------------------------------------------------
// rocBLAS / rocSOLVER at run time
namespace {
typedef void *rb_handle;
typedef int (*fn_create)(rb_handle *);
typedef int (*fn_set_stream)(rb_handle, hipStream_t);
typedef int (*fn_dgemm)(rb_handle, int, int, int, int, int, const double *, const double *, int, const double *, int,
                        const double *, double *, int);
typedef int (*fn_dtrsm)(rb_handle, int, int, int, int, int, int, const double *, const double *, int, double *, int);
typedef int (*fn_dpotrf)(rb_handle, int, int, double *, int, int *);
typedef int (*fn_set_atomics)(rb_handle, int);
enum { RB_OP_N = 111, RB_OP_T = 112, RB_UPPER = 121, RB_NON_UNIT = 131, RB_LEFT = 141 };

struct Blas {
    void *hb = nullptr, *hs = nullptr;
    fn_create create = nullptr; fn_set_stream set_stream = nullptr; fn_dgemm dgemm = nullptr; fn_dtrsm dtrsm = nullptr;
    fn_dpotrf dpotrf = nullptr;
    fn_set_atomics set_atomics = nullptr;      // optional
    rb_handle handle = nullptr, handle2 = nullptr, handle3 = nullptr;
    int handle_dev = -1;
} g_blas;

int blas_load() {
    if (g_blas.dpotrf) return CSLAM_OK;
    // the copies the process already has (PyTorch-ROCm ships its own) come first: rocSOLVER must get a handle of the
    // rocBLAS it was linked against
    const char *dirs[] = {"", "/opt/rocm/lib/"};
    for (const char *d : dirs) {
        char nb[256], ns[256];
        for (const char *suf : {".so", ".so.5"}) {
            snprintf(nb, sizeof nb, "%slibrocblas%s", d, suf);
            g_blas.hb = dlopen(nb, RTLD_NOW | RTLD_GLOBAL);
            if (g_blas.hb) break;
        }
        if (!g_blas.hb) continue;
        for (const char *suf : {".so", ".so.0"}) {
            snprintf(ns, sizeof ns, "%slibrocsolver%s", d, suf);
            g_blas.hs = dlopen(ns, RTLD_NOW | RTLD_GLOBAL);
            if (g_blas.hs) break;
        }
        if (!g_blas.hs) { g_blas.hb = nullptr; continue; }
        g_blas.create = (fn_create)dlsym(g_blas.hb, "rocblas_create_handle");
        g_blas.set_stream = (fn_set_stream)dlsym(g_blas.hb, "rocblas_set_stream");
        g_blas.dgemm = (fn_dgemm)dlsym(g_blas.hb, "rocblas_dgemm");
        g_blas.dtrsm = (fn_dtrsm)dlsym(g_blas.hb, "rocblas_dtrsm");
        g_blas.dpotrf = (fn_dpotrf)dlsym(g_blas.hs, "rocsolver_dpotrf");
        g_blas.set_atomics = (fn_set_atomics)dlsym(g_blas.hb, "rocblas_set_atomics_mode");
        if (g_blas.create && g_blas.set_stream && g_blas.dgemm && g_blas.dtrsm && g_blas.dpotrf) return CSLAM_OK;
        g_blas.dpotrf = nullptr;
    }
    cslam_set_error("rocBLAS / rocSOLVER (librocblas.so, librocsolver.so) not found: cslam_fiedler needs them for the dense junction factor");
    return CSLAM_E_UNSUPPORTED;
}

#define RB_TRY(expr)                                                                       \
    do {                                                                                   \
        int _s = (expr);                                                                   \
        if (_s != 0) { cslam_set_error("%s failed: rocblas status %d (%s:%d)", #expr, _s, __FILE__, __LINE__); return CSLAM_E_HIP; } \
    } while (0)

// ---------------------------------------------------------------------------------------------
// workspace kept between calls
struct Buf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes, double grow) {
        if (bytes <= cap) return CSLAM_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = (size_t)(bytes * grow) + 4096;
        if (hipMalloc(&p, want) != hipSuccess) {
            (void)hipGetLastError();
            if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); p = nullptr; cslam_set_error("hipMalloc of %zu bytes failed", bytes); return CSLAM_E_NOMEM; }
            cap = bytes;
        } else cap = want;
        return CSLAM_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
struct Workspace {
    int device = -1;
    Buf arena, dense, inv, x0;
    int64_t x0_n = -1; uint32_t x0_seed = 0;
    hipStream_t stream = nullptr;                              // used when the caller passes no stream (no implicit ordering against the null stream)
    hipStream_t stream2 = nullptr, stream3 = nullptr;          // look-ahead of the blocked factorisation: panels / block inverses
    hipEvent_t ev_row = nullptr, ev_panel = nullptr, ev_inv = nullptr;
    void release() {
        arena.release(); dense.release(); inv.release(); x0.release(); x0_n = -1; device = -1;
        if (stream) { (void)hipStreamDestroy(stream); stream = nullptr; }
        if (stream2) {
            (void)hipStreamDestroy(stream2); (void)hipStreamDestroy(stream3);
            (void)hipEventDestroy(ev_row); (void)hipEventDestroy(ev_panel); (void)hipEventDestroy(ev_inv);
            stream2 = stream3 = nullptr;
        }
    }
} g_ws;
std::mutex g_mu;

struct Bump {                                                  // 256-byte aligned carving of the arena
    char *base; size_t off;
    template <class T> T *take(size_t count) { T *r = (T *)(base + off); off += (count * sizeof(T) + 255) & ~(size_t)255; return r; }
};
static size_t padded(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

// ---------------------------------------------------------------------------------------------
// device helpers of the junction system
__global__ __launch_bounds__(256) void fj_offdiag_kernel(const int32_t *__restrict__ ei, const int32_t *__restrict__ ej,
                                                         const double *__restrict__ ew, int nE, int g, double *__restrict__ A, int64_t ld) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nE) return;
    const int i = ei[e], j = ej[e];
    if (i == g || j == g) return;
    const int64_t fi = i - (i > g), fj = j - (j > g);
    // at most two terms meet in one entry (a chain segment and a loop edge between the same two junctions): the order of
    // two additions onto zero does not change the bits
    unsafeAtomicAdd(A + fi * ld + fj, -ew[e]);
    unsafeAtomicAdd(A + fj * ld + fi, -ew[e]);
}
__global__ __launch_bounds__(256) void fj_diag_kernel(const double *__restrict__ diag, int m, double *__restrict__ A, int64_t ld) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < m) A[(int64_t)i * ld + i] = diag[i];
}
__global__ __launch_bounds__(256) void fj_identity_kernel(double *__restrict__ D, int bs, int bw) {          // [bs][bs], zeros elsewhere
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)bs * bs) return;
    const int r = (int)(i / bs), c = (int)(i % bs);
    D[i] = (r == c && r < bw) ? 1.0 : 0.0;
}
// rows of a [nJ][4] block without row g -> [nJ-1][4] (gather) and back (scatter; row g of the destination is left alone)
__global__ __launch_bounds__(256) void fj_drop_row_kernel(const double *__restrict__ src, int nJ, int g, double *__restrict__ dst, int scatter) {
    const int i = blockIdx.x * 256 + threadIdx.x;                          // index over (nJ - 1) * 4 doubles
    if (i >= (nJ - 1) * 4) return;
    const int row = i >> 2, c = i & 3, full = row + (row >= g);
    if (scatter) dst[full * 4 + c] = src[i]; else dst[i] = src[full * 4 + c];
}
__global__ __launch_bounds__(256) void fj_column0_kernel(const double *__restrict__ X, int64_t n, double *__restrict__ v) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < n) v[k] = X[k * 4];
}

// ---------------------------------------------------------------------------------------------
// 4 x 4 host algebra (row-major)
__host__ __device__ bool chol4_upper(const double *G, double *R) {                 // G = R^T R, R upper
    double L[16] = {0};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = G[i * 4 + j];
            for (int k = 0; k < j; ++k) s -= L[i * 4 + k] * L[j * 4 + k];
            if (i == j) { if (!(s > 0.0)) return false; L[i * 4 + i] = sqrt(s); }
            else L[i * 4 + j] = s / L[j * 4 + j];
        }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) R[i * 4 + j] = L[j * 4 + i];
    return true;
}
__host__ __device__ bool inv4(const double *A, double *Ai) {                       // Gauss-Jordan, partial pivoting
    double a[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { a[i][j] = A[i * 4 + j]; a[i][4 + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r) if (fabs(a[r][c]) > fabs(a[p][c])) p = r;
        if (a[p][c] == 0.0 || a[p][c] != a[p][c]) return false;
        if (p != c) for (int j = 0; j < 8; ++j) { const double t = a[c][j]; a[c][j] = a[p][j]; a[p][j] = t; }
        const double d = 1.0 / a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] *= d;
        for (int r = 0; r < 4; ++r) if (r != c) { const double f = a[r][c]; if (f != 0.0) for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j]; }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) Ai[i * 4 + j] = a[i][4 + j];
    return true;
}
__host__ __device__ void eigh4(const double *H, double *w, double *V) {            // cyclic Jacobi; ascending eigenvalues, eigenvectors in the columns of V
    double a[4][4], v[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { a[i][j] = 0.5 * (H[i * 4 + j] + H[j * 4 + i]); v[i][j] = i == j ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0.0, dia = 0.0;
        for (int i = 0; i < 4; ++i) { dia += a[i][i] * a[i][i]; for (int j = i + 1; j < 4; ++j) off += a[i][j] * a[i][j]; }
        if (off == 0.0 || off < 1e-300 + 1e-34 * dia) break;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; ++k) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq; }
                for (int k = 0; k < 4; ++k) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk; }
                for (int k = 0; k < 4; ++k) { const double vkp = v[k][p], vkq = v[k][q]; v[k][p] = c * vkp - s * vkq; v[k][q] = s * vkp + c * vkq; }
            }
    }
    int order[4] = {0, 1, 2, 3};
    for (int i = 0; i < 4; ++i) for (int j = i + 1; j < 4; ++j) if (a[order[j]][order[j]] < a[order[i]][order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    for (int j = 0; j < 4; ++j) { w[j] = a[order[j]][order[j]]; for (int i = 0; i < 4; ++i) V[i * 4 + j] = v[i][order[j]]; }
}

// The same algebra as one-thread kernels: the TraceMIN loop then needs ONE host synchronisation per iteration (the residual of
// the stopping rule) instead of five -- every synchronisation is a bubble of ~50 us (read-back, host arithmetic, launch latency)
// in an iteration of 0.75-3 ms.  small[]: M16 @0, shift4 @16, Y16 @20, sig4 @36, y0 @40, res @44, fail flag (as double) @45
__global__ void fj_cholinv_kernel(const double *__restrict__ g20, double *__restrict__ small) {
    double R[16];
    if (!chol4_upper(g20, R) || !inv4(R, small)) small[45] = 1.0;
}
__global__ void fj_eigh_kernel(const double *__restrict__ g20, double *__restrict__ small) {
    eigh4(g20, small + 36, small + 20);
    for (int i = 0; i < 4; ++i) small[40 + i] = small[20 + i * 4];
}
__global__ void fj_proj_kernel(const double *__restrict__ g20, double n, double *__restrict__ small) {
    if (!inv4(g20, small)) small[45] = 2.0;
    for (int j = 0; j < 4; ++j) {
        double s = 0.0;
        for (int i = 0; i < 4; ++i) s += g20[16 + i] * small[i * 4 + j];
        small[16 + j] = s / n;
    }
}
__global__ void fj_mean_kernel(const double *__restrict__ g20, double n, double *__restrict__ small) {
    for (int i = 0; i < 16; ++i) small[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int j = 0; j < 4; ++j) small[16 + j] = g20[16 + j] / n;
    small[45] = 0.0;
}

struct Laps {
    bool on; std::chrono::steady_clock::time_point t;
    Laps() : on(getenv("CSLAM_MAC_TIMING") != nullptr), t(std::chrono::steady_clock::now()) {}
    void lap(const char *tag, hipStream_t st) {
        if (!on) return;
        (void)hipStreamSynchronize(st);
        auto n = std::chrono::steady_clock::now();
        fprintf(stderr, " [%s %.1f ms]", tag, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};
}  // namespace

CSLAM_API int cslam_fiedler_release(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_ws.device >= 0) { DeviceGuard guard(g_ws.device); g_ws.release(); }
    return CSLAM_OK;
}

CSLAM_API int cslam_fiedler(int64_t n, const int64_t *h_indptr, const int32_t *h_indices, const double *h_data,
                            const double *h_x0, uint32_t seed, double tol, int max_iters, double *h_lambda2, double *h_v,
                            int *h_iters, void *stream) {
    ARG_CHECK(h_indptr && h_indices && h_data && h_lambda2 && h_v, "NULL argument");
    ARG_CHECK(n > 4 && n < ((int64_t)1 << 31), "n must be in (4, 2^31): tiny graphs go through the host solver");
    ARG_CHECK(tol > 0.0, "tol <= 0");
    std::lock_guard<std::mutex> lock(g_mu);
    Laps laps;
    int rc = blas_load();
    if (rc) return rc;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (g_ws.device != dev) { if (g_ws.device >= 0) { DeviceGuard guard(g_ws.device); g_ws.release(); } g_ws.device = dev; }
    if (!stream && !g_ws.stream) HIP_TRY(hipStreamCreateWithFlags(&g_ws.stream, hipStreamNonBlocking));
    hipStream_t st = stream ? (hipStream_t)stream : g_ws.stream;
    if (!g_blas.handle || g_blas.handle_dev != dev) {
        RB_TRY(g_blas.create(&g_blas.handle));
        g_blas.handle_dev = dev;
        // the pair this call returns is promised bit for bit (tests/test_mac_gpu.py): no kernels that sum through atomics
        if (g_blas.set_atomics) RB_TRY(g_blas.set_atomics(g_blas.handle, 0 /* rocblas_atomics_not_allowed */));
    }
    RB_TRY(g_blas.set_stream(g_blas.handle, st));
    laps.lap("libraries + handle", st);
    const int64_t nnz = h_indptr[n];
    ARG_CHECK(h_indptr[0] == 0 && nnz >= 0, "bad indptr");

    // ---- chain / junction structure (chain_solver.py:30-84) ----
    std::vector<double> c(n - 1, 0.0), absrow(n, 0.0);
    std::vector<int64_t> li, lj; std::vector<double> lw;
    int64_t ground = 0, best_deg = -1;
    for (int64_t row = 0; row < n; ++row) {
        const int64_t p0 = h_indptr[row], p1 = h_indptr[row + 1];
        ARG_CHECK(p1 >= p0 && p1 <= nnz, "bad indptr");
        if (p1 - p0 > best_deg) { best_deg = p1 - p0; ground = row; }         // argmax of the stored row lengths, first maximum
        double s = 0.0;
        for (int64_t p = p0; p < p1; ++p) {
            const int64_t col = h_indices[p];
            ARG_CHECK(col >= 0 && col < n && (p == p0 || col > h_indices[p - 1]), "column indices must be sorted and unique within a row");
            s += fabs(h_data[p]);
            const int64_t d = col - row;
            if (d == 1) c[row] = -h_data[p];
            else if (d >= 2 && h_data[p] != 0.0) { li.push_back(row); lj.push_back(col); lw.push_back(-h_data[p]); }
        }
        absrow[row] = s;
    }
    double Lnorm = 0.0;
    for (int64_t k = 0; k < n; ++k) if (absrow[k] > Lnorm) Lnorm = absrow[k];
    ARG_CHECK(Lnorm > 0.0, "empty Laplacian");
    std::vector<uint8_t> is_j(n, 0);
    for (size_t e = 0; e < li.size(); ++e) { is_j[li[e]] = 1; is_j[lj[e]] = 1; }
    is_j[ground] = 1; is_j[0] = 1; is_j[n - 1] = 1;
    for (int64_t i = 0; i + 1 < n; ++i) if (!(c[i] > 0.0)) { is_j[i] = 1; is_j[i + 1] = 1; }
    std::vector<double> r(n - 1), rcum(n), Rn(n);
    std::vector<int32_t> jid(n), seg_of(n);
    std::vector<int64_t> J;
    rcum[0] = 0.0;
    for (int64_t i = 0; i + 1 < n; ++i) { r[i] = c[i] > 0.0 ? 1.0 / c[i] : 0.0; rcum[i + 1] = rcum[i] + r[i]; }
    for (int64_t k = 0; k < n; ++k) if (is_j[k]) J.push_back(k);
    const int64_t nJ64 = (int64_t)J.size();
    // the dense float64 junction factor: 64000 junctions = 33 GB.  CSLAM_FIEDLER_MAX_JUNCTIONS moves the limit (tests exercise the
    // callers' fallback with a small one; a host with the memory to spare may raise it up to 150000 = 180 GB)
    int64_t max_j = 64000;
    if (const char *e = getenv("CSLAM_FIEDLER_MAX_JUNCTIONS")) { const long long v = atoll(e); if (v >= 2 && v <= 150000) max_j = v; }
    if (nJ64 - 1 > max_j) {
        cslam_set_error("more than %lld junctions (%lld): the dense junction factor does not apply (use the host sparse LU)",
                        (long long)max_j, (long long)(nJ64 - 1));
        return CSLAM_E_LIMIT;
    }
    const int nJ = (int)nJ64, m = nJ - 1;
    std::vector<int64_t> sa, sb; std::vector<double> Rl;
    std::vector<int32_t> seg_start_of(nJ, -1), seg_end_of(nJ, -1);
    for (int t = 0; t + 1 < nJ; ++t)
        if (c[J[t]] > 0.0) {
            const int s = (int)sa.size();
            sa.push_back(J[t]); sb.push_back(J[t + 1]); Rl.push_back(rcum[J[t + 1]] - rcum[J[t]]);
            seg_start_of[t] = s; seg_end_of[t + 1] = s;
        }
    const int nseg = (int)sa.size();
    {
        int64_t start = 0; int t = -1, seg = -1;
        for (int64_t k = 0; k < n; ++k) {
            if (is_j[k]) { ++t; start = k; seg = seg_start_of[t]; jid[k] = t; } else jid[k] = -1;
            Rn[k] = rcum[k] - rcum[start];
            seg_of[k] = seg;
        }
    }
    const int g = jid[ground];
    // reduced graph on the junctions: chain segments (conductance 1 / Rl), then the loop edges in CSR order
    const int nE = nseg + (int)li.size();
    std::vector<int32_t> ei(nE), ej(nE); std::vector<double> ew(nE), diag(m > 0 ? m : 1, 0.0);
    for (int s = 0; s < nseg; ++s) { ei[s] = jid[sa[s]]; ej[s] = jid[sb[s]]; ew[s] = 1.0 / Rl[s]; }
    for (size_t e = 0; e < li.size(); ++e) { ei[nseg + e] = jid[li[e]]; ej[nseg + e] = jid[lj[e]]; ew[nseg + e] = lw[e]; }
    {   // networkx raises on a graph that is not connected (fiedler_vector), and the grounded junction Laplacian of one is
        // singular only up to round-off: decided here, on the reduced graph (connected iff the pose graph is)
        std::vector<int> parent(nJ);
        for (int t = 0; t < nJ; ++t) parent[t] = t;
        auto find = [&](int x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
        int comps = nJ;
        for (int e = 0; e < nE; ++e) {
            if (!(ew[e] > 0.0)) continue;
            const int a = find(ei[e]), b = find(ej[e]);
            if (a != b) { parent[a] = b; --comps; }
        }
        if (comps != 1) { cslam_set_error("graph is not connected (%d components): no Fiedler pair", comps); return CSLAM_E_GRAPH; }
    }
    for (int e = 0; e < nE; ++e) if (ei[e] != g) diag[ei[e] - (ei[e] > g)] += ew[e];      // fixed order: all first ends, then all second ends
    for (int e = 0; e < nE; ++e) if (ej[e] != g) diag[ej[e] - (ej[e] > g)] += ew[e];
    laps.lap("host structure", st);

    // ---- device workspace ----
    const int bs = m > 4096 ? 2048 : 512;
    const int nb = m > 0 ? (m + bs - 1) / bs : 0;
    const int64_t nch = (n + 2047) / 2048;
    size_t need = 0;
    need += 7 * padded((size_t)n * 32);                                          // X, W, pool x2, Bn, Qn, tmp
    need += 2 * padded((size_t)n * 8) + 2 * padded((size_t)n * 4) + padded((size_t)n);   // r, Rn, jid, seg_of, is_j
    need += padded((size_t)(n + 1) * 8) + padded((size_t)nnz * 4 + 4) + padded((size_t)nnz * 8 + 8);
    need += padded((size_t)nJ * 8) + 2 * padded((size_t)nJ * 4) + 3 * padded((size_t)(nseg + 1) * 8);
    need += 3 * padded((size_t)nJ * 32) + padded((size_t)(m + 1) * 8) + 2 * padded((size_t)(nE + 1) * 4) + padded((size_t)(nE + 1) * 8);
    need += padded((size_t)(9 * nch + 64) * 8) + padded(20 * 1024 * 8) + padded(32 * 8) + padded((size_t)bs * 32) + padded((size_t)(nb + 1) * 4) + padded((size_t)n * 8) + padded(64 * 8);
    if ((rc = g_ws.arena.ensure(need, 1.1))) return rc;
    if (m > 0) {
        if ((rc = g_ws.dense.ensure((size_t)m * m * 8, 1.3))) return rc;
        if ((rc = g_ws.inv.ensure((size_t)2 * nb * bs * bs * 8, 1.3))) return rc;
    }
    Bump bump{(char *)g_ws.arena.p, 0};
    double *X = bump.take<double>(n * 4), *W = bump.take<double>(n * 4), *P0 = bump.take<double>(n * 4), *P1 = bump.take<double>(n * 4);
    double *Bn = bump.take<double>(n * 4), *Qn = bump.take<double>(n * 4), *tmp = bump.take<double>(n * 4);
    double *d_r = bump.take<double>(n), *d_Rn = bump.take<double>(n);
    int32_t *d_jid = bump.take<int32_t>(n), *d_seg_of = bump.take<int32_t>(n);
    uint8_t *d_is_j = bump.take<uint8_t>(n);
    int64_t *d_indptr = bump.take<int64_t>(n + 1); int32_t *d_indices = bump.take<int32_t>(nnz + 1); double *d_data = bump.take<double>(nnz + 1);
    int64_t *d_J = bump.take<int64_t>(nJ); int32_t *d_sso = bump.take<int32_t>(nJ), *d_seo = bump.take<int32_t>(nJ);
    int64_t *d_sa = bump.take<int64_t>(nseg + 1), *d_sb = bump.take<int64_t>(nseg + 1); double *d_Rl = bump.take<double>(nseg + 1);
    double *d_bt = bump.take<double>((size_t)nJ * 4), *d_xJ = bump.take<double>((size_t)nJ * 4), *d_rhs = bump.take<double>((size_t)nJ * 4);
    double *d_diag = bump.take<double>(m + 1); int32_t *d_ei = bump.take<int32_t>(nE + 1), *d_ej = bump.take<int32_t>(nE + 1); double *d_ew = bump.take<double>(nE + 1);
    double *d_scratch = bump.take<double>(9 * nch + 64), *d_partial = bump.take<double>(20 * 1024), *d_out20 = bump.take<double>(32);
    double *d_cs_tmp = bump.take<double>((size_t)bs * 4); int *d_info = bump.take<int>(nb + 1); double *d_v = bump.take<double>(n);
    double *d_small = bump.take<double>(64);
    if (bump.off > g_ws.arena.cap) { cslam_set_error("internal: workspace carving exceeds its size"); return CSLAM_E_INVALID; }
#define UP(dst, vec, count) HIP_TRY(hipMemcpyAsync(dst, (vec), (size_t)(count) * sizeof(*(dst)), hipMemcpyHostToDevice, st))
    UP(d_r, r.data(), n - 1); UP(d_Rn, Rn.data(), n); UP(d_jid, jid.data(), n); UP(d_seg_of, seg_of.data(), n); UP(d_is_j, is_j.data(), n);
    UP(d_indptr, h_indptr, n + 1); UP(d_indices, h_indices, nnz); UP(d_data, h_data, nnz);
    UP(d_J, J.data(), nJ); UP(d_sso, seg_start_of.data(), nJ); UP(d_seo, seg_end_of.data(), nJ);
    if (nseg) { UP(d_sa, sa.data(), nseg); UP(d_sb, sb.data(), nseg); UP(d_Rl, Rl.data(), nseg); }
    if (m > 0) { UP(d_diag, diag.data(), m); UP(d_ei, ei.data(), nE); UP(d_ej, ej.data(), nE); UP(d_ew, ew.data(), nE); }
    // start block: the caller's, or numpy's RandomState(seed).normal(size=(4, n)).T (cached on the device: the reference
    // re-seeds for every call, mac.py:56-58, so consecutive calls with one n share it)
    if (h_x0) UP(X, h_x0, n * 4);
    else {
        if (g_ws.x0_n != n || g_ws.x0_seed != seed) {
            if ((rc = g_ws.x0.ensure((size_t)n * 32, 1.0))) return rc;
            std::vector<double> x0((size_t)n * 4);
            cslam_fiedler_start_block(seed, n, x0.data());
            HIP_TRY(hipMemcpyAsync(g_ws.x0.p, x0.data(), (size_t)n * 32, hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
            g_ws.x0_n = n; g_ws.x0_seed = seed;
        }
        HIP_TRY(hipMemcpyAsync(X, g_ws.x0.p, (size_t)n * 32, hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(hipMemsetAsync(d_xJ, 0, (size_t)nJ * 32, st));                           // the grounded junction stays 0
    HIP_TRY(hipStreamSynchronize(st));                                              // the host vectors above may go after this
    laps.lap("upload", st);

    // ---- dense junction Laplacian, its Cholesky factor and the inverted diagonal blocks ----
    double *A = (double *)g_ws.dense.p, *dinv = (double *)g_ws.inv.p, *dinvT = dinv ? dinv + (size_t)nb * bs * bs : nullptr;
    const int64_t ld = m;
    if (m > 0) {
        HIP_TRY(hipMemsetAsync(A, 0, (size_t)m * m * 8, st));
        hipLaunchKernelGGL(fj_offdiag_kernel, dim3((nE + 255) / 256), dim3(256), 0, st, d_ei, d_ej, d_ew, nE, g, A, ld);
        hipLaunchKernelGGL(fj_diag_kernel, dim3((m + 255) / 256), dim3(256), 0, st, d_diag, m, A, ld);
        HIP_TRY(hipGetLastError());
        laps.lap("assemble", st);
        // the row-major lower factor L is the column-major upper factor U = L^T of the same buffer: A = U^T U, right-looking.
        // Look-ahead: as soon as the block row of the NEXT panel has its update, its diagonal factorisation (potrf: a chain of
        // small kernels) and its panel solve run on a second, high-priority stream under the GEMMs that update the rest of
        // the trailing matrix; the inversions of the diagonal blocks (needed only by the solves afterwards) on a third.
        const double one = 1.0, minus = -1.0;
        const int cb = m > 4096 ? bs : m;                                           // small systems: one potrf
        const char *tenv = getenv("CSLAM_MAC_TIMING");
        const bool split = tenv && tenv[0] == '2';                                 // per-category times: one stream, a sync after every call
        const bool la = m > cb && !split;                                           // look-ahead: 308 -> 278 ms at 31.5k junctions (round 2)
        double t_cat[4] = {0, 0, 0, 0};                                             // potrf, panel trsm, trailing gemm, block inverses
        auto tick = [&](int cat, std::chrono::steady_clock::time_point t0) {
            if (!split) return;
            (void)hipStreamSynchronize(st);
            t_cat[cat] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        };
        auto now = [&]() { if (split) (void)hipStreamSynchronize(st); return std::chrono::steady_clock::now(); };
        if (la) {
            if (!g_ws.stream2) {
                int least = 0, greatest = 0;                                            // the panel chain must not queue behind the GEMM's workgroups
                HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
                HIP_TRY(hipStreamCreateWithPriority(&g_ws.stream2, hipStreamNonBlocking, greatest));
                HIP_TRY(hipStreamCreateWithFlags(&g_ws.stream3, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&g_ws.ev_row, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&g_ws.ev_panel, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&g_ws.ev_inv, hipEventDisableTiming));
            }
            if (!g_blas.handle2) {
                RB_TRY(g_blas.create(&g_blas.handle2));
                RB_TRY(g_blas.create(&g_blas.handle3));
                if (g_blas.set_atomics) {
                    RB_TRY(g_blas.set_atomics(g_blas.handle2, 0));
                    RB_TRY(g_blas.set_atomics(g_blas.handle3, 0));
                }
            }
            RB_TRY(g_blas.set_stream(g_blas.handle2, g_ws.stream2));
            RB_TRY(g_blas.set_stream(g_blas.handle3, g_ws.stream3));
        }
        int ib = 0;
        auto panel = [&](rb_handle h, hipStream_t hs, int k) -> int {               // diagonal block k: factor, panel row, block inverses
            const int e = k + cb < m ? k + cb : m, bw = e - k;
            auto t0 = now();
            RB_TRY(g_blas.dpotrf(h, RB_UPPER, bw, A + (size_t)k * ld + k, (int)ld, d_info + ib++));
            tick(0, t0); t0 = now();
            if (e < m)
                RB_TRY(g_blas.dtrsm(h, RB_LEFT, RB_UPPER, RB_OP_T, RB_NON_UNIT, bw, m - e, &one, A + (size_t)k * ld + k, (int)ld,
                                    A + (size_t)e * ld + k, (int)ld));
            tick(1, t0);
            return CSLAM_OK;
        };
        auto inverses = [&](rb_handle h, hipStream_t hs, int k0, int k1) -> int {   // inverted bs-blocks of the diagonal in [k0, k1)
            auto t0 = now();
            for (int k = k0; k < k1; k += bs) {
                const int t = k / bs, e = k + bs < m ? k + bs : m, bw = e - k;
                double *Dt = dinv + (size_t)t * bs * bs, *DtT = dinvT + (size_t)t * bs * bs;
                const unsigned grid = (unsigned)(((int64_t)bs * bs + 255) / 256);
                hipLaunchKernelGGL(fj_identity_kernel, dim3(grid), dim3(256), 0, hs, Dt, bs, bw);
                hipLaunchKernelGGL(fj_identity_kernel, dim3(grid), dim3(256), 0, hs, DtT, bs, bw);
                // column-major U^-1 is the row-major L^-1; column-major U^-T its transpose
                RB_TRY(g_blas.dtrsm(h, RB_LEFT, RB_UPPER, RB_OP_N, RB_NON_UNIT, bw, bw, &one, A + (size_t)k * ld + k, (int)ld, Dt, bs));
                RB_TRY(g_blas.dtrsm(h, RB_LEFT, RB_UPPER, RB_OP_T, RB_NON_UNIT, bw, bw, &one, A + (size_t)k * ld + k, (int)ld, DtT, bs));
            }
            tick(3, t0);
            return CSLAM_OK;
        };
        auto update = [&](int k, int bw, int j, int je) -> int {                   // C[j:je, j:m] -= U[k:k+bw, j:je]^T U[k:k+bw, j:m]
            auto t0 = now();
            RB_TRY(g_blas.dgemm(g_blas.handle, RB_OP_T, RB_OP_N, je - j, m - j, bw, &minus, A + (size_t)j * ld + k, (int)ld,
                                A + (size_t)j * ld + k, (int)ld, &one, A + (size_t)j * ld + j, (int)ld));
            tick(2, t0);
            return CSLAM_OK;
        };
        if ((rc = panel(g_blas.handle, st, 0))) return rc;
        if (!la) { if ((rc = inverses(g_blas.handle, st, 0, cb < m ? cb : m))) return rc; }
        else {                                                                      // the block inverses are needed last: a stream of their own
            HIP_TRY(hipEventRecord(g_ws.ev_row, st)); HIP_TRY(hipStreamWaitEvent(g_ws.stream3, g_ws.ev_row, 0));
            if ((rc = inverses(g_blas.handle3, g_ws.stream3, 0, cb))) return rc;
        }
        for (int k = 0; k + cb < m; k += cb) {
            const int e = k + cb, e2 = e + cb < m ? e + cb : m;
            if ((rc = update(k, cb, e, e2))) return rc;                             // the next panel's block row first
            if (la) HIP_TRY(hipEventRecord(g_ws.ev_row, st));
            const int ub = cb;                                                      // block rows of one panel height: 289 -> 278 ms at 31.5k junctions against two (less of each diagonal block's unused triangle)
            for (int j = e2; j < m; j += ub) {                                      // queued BEFORE the panel calls: should one of
                const int je = j + ub < m ? j + ub : m;                             // them block the host, the GPU already has these
                if ((rc = update(k, cb, j, je))) return rc;
            }
            if (la) {
                HIP_TRY(hipStreamWaitEvent(g_ws.stream2, g_ws.ev_row, 0));
                if ((rc = panel(g_blas.handle2, g_ws.stream2, e))) return rc;
                HIP_TRY(hipEventRecord(g_ws.ev_panel, g_ws.stream2));
                HIP_TRY(hipStreamWaitEvent(g_ws.stream3, g_ws.ev_panel, 0));
                if ((rc = inverses(g_blas.handle3, g_ws.stream3, e, e2))) return rc;
                HIP_TRY(hipStreamWaitEvent(st, g_ws.ev_panel, 0));
            }
            else { if ((rc = panel(g_blas.handle, st, e))) return rc; if ((rc = inverses(g_blas.handle, st, e, e2))) return rc; }
        }
        if (la) {                                                                   // the last inverses join the main stream
            HIP_TRY(hipEventRecord(g_ws.ev_inv, g_ws.stream3)); HIP_TRY(hipStreamWaitEvent(st, g_ws.ev_inv, 0));
        }
        std::vector<int> info(ib);
        HIP_TRY(hipMemcpyAsync(info.data(), d_info, ib * sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (int i = 0; i < ib; ++i)
            if (info[i] != 0) { cslam_set_error("the grounded junction Laplacian is not positive definite (graph not connected?)"); return CSLAM_E_GRAPH; }
        laps.lap("cholesky + block inverses", st);
        if (split) fprintf(stderr, " {potrf %.1f, panel trsm %.1f, trailing gemm %.1f, block inverses %.1f ms}", t_cat[0], t_cat[1], t_cat[2], t_cat[3]);
    }

    // ---- TraceMIN (fiedler.py / chain_solver_gpu.py, same order of operations) ----
    // The 4 x 4 algebra (Cholesky + inverse of X^T X, Jacobi eigen-decomposition of X^T L X, inverse of X^T A^-1 X) runs in
    // one-thread kernels between the streaming passes; the host synchronises once per iteration, for the stopping rule.
    double h1 = 0.0, sigma[4] = {0, 0, 0, 0};
    double *pool[2] = {P0, P1};
    int npool = 2;
    int iters = 0;
    const int cap = max_iters > 0 ? max_iters : 100000;
    {
        auto gram_d = [&](const double *Am, const double *Bm) { return cslam_block4_gram_dev(Am, Bm, n, d_partial, d_out20, st); };
        auto affine_d = [&](double *&Xc, const double *dM, const double *dshift) {
            double *out = pool[--npool];
            const int r2 = cslam_block4_affine_dev(Xc, n, dM, dshift, out, st);
            pool[npool++] = Xc; Xc = out;
            return r2;
        };
        double hb[10];                                                               // sig4, y0, res, flag of an iteration
        if ((rc = gram_d(X, X))) return rc;
        hipLaunchKernelGGL(fj_mean_kernel, dim3(1), dim3(1), 0, st, d_out20, (double)n, d_small);
        if ((rc = affine_d(X, d_small, d_small + 16))) return rc;
        for (;;) {
            ++iters;
            for (int pass = 0; pass < 2; ++pass) {                                   // CholQR2
                if ((rc = gram_d(X, X))) return rc;
                hipLaunchKernelGGL(fj_cholinv_kernel, dim3(1), dim3(1), 0, st, d_out20, d_small);
                if ((rc = affine_d(X, d_small, nullptr))) return rc;
            }
            if ((rc = cslam_csr_spmm4_dev(d_indptr, d_indices, d_data, n, X, W, st))) return rc;
            if ((rc = gram_d(X, W))) return rc;
            hipLaunchKernelGGL(fj_eigh_kernel, dim3(1), dim3(1), 0, st, d_out20, d_small);
            if ((rc = affine_d(X, d_small + 20, nullptr))) return rc;
            if ((rc = block4_residual_devsigma(W, X, n, d_small + 40, d_small + 36, d_partial, d_small + 44, st))) return rc;
            HIP_TRY(hipMemcpyAsync(hb, d_small + 36, sizeof hb, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (hb[9] != 0.0) { cslam_set_error(hb[9] == 1.0 ? "TraceMIN block lost rank (X^T X not positive definite)" : "TraceMIN: singular X^T A^-1 X"); return CSLAM_E_GRAPH; }
            sigma[0] = hb[0]; h1 = hb[8];
            if (h1 / Lnorm < tol) break;
            if (!(h1 == h1) || iters >= cap) { if (h_iters) *h_iters = iters; cslam_set_error("TraceMIN did not reach tol in %d iterations (residual %.3e)", iters, h1 / Lnorm); return CSLAM_E_GRAPH; }
            double *Wi = pool[--npool];
            if ((rc = cslam_chain_forward_dev(X, d_is_j, d_r, n, d_J, nJ, d_sso, d_seo, d_sa, d_sb, d_Rl, Bn, Qn, tmp, d_scratch, d_bt, st))) return rc;
            if (m > 0) {
                hipLaunchKernelGGL(fj_drop_row_kernel, dim3((m * 4 + 255) / 256), dim3(256), 0, st, d_bt, nJ, g, d_rhs, 0);
                if ((rc = cslam_chol_solve4_dev(A, m, ld, 0, dinv, dinvT, bs, d_rhs, d_cs_tmp, st))) return rc;
                hipLaunchKernelGGL(fj_drop_row_kernel, dim3((m * 4 + 255) / 256), dim3(256), 0, st, d_rhs, nJ, g, d_xJ, 1);
            }
            if ((rc = cslam_chain_backward_dev(d_xJ, Bn, Qn, d_r, d_Rn, d_jid, d_seg_of, d_sa, d_sb, d_Rl, n, Wi, st))) return rc;
            if ((rc = gram_d(X, Wi))) return rc;
            hipLaunchKernelGGL(fj_proj_kernel, dim3(1), dim3(1), 0, st, d_out20, (double)n, d_small);
            if ((rc = cslam_block4_affine_dev(Wi, n, d_small, d_small + 16, X, st))) return rc;
            pool[npool++] = Wi;
        }
    }
    hipLaunchKernelGGL(fj_column0_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, st, X, n, d_v);
    HIP_TRY(hipMemcpyAsync(h_v, d_v, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *h_lambda2 = sigma[0];
    if (h_iters) *h_iters = iters;
    if (laps.on) { laps.lap("tracemin", st); fprintf(stderr, " (nJ=%d, %d iterations)\n", nJ, iters); }
    return CSLAM_OK;
}

// ---------------------------------------------------------------------------------------------
// cslam_mac_fw_subset: the Frank-Wolfe loop of the sparsifier (cslam/mac/mac.py:191-233 `MAC.fw_subset`, with
// :61-77 combined Laplacian, :112-130 gradient, :132-147 / :168-189 roundings) around cslam_fiedler, for a host without
// Python: fixed edges + weighted candidate edges in, the rounded selection out.
namespace {
struct Csr {
    int64_t n = 0;
    std::vector<int64_t> indptr;
    std::vector<int32_t> indices;
    std::vector<double> data;
};
// weighted graph Laplacian of an edge list (mac/utils.py:47-126): per edge the triplets (i,i,w) (j,j,w) (i,j,-w) (j,i,-w),
// duplicates summed in edge order, columns sorted within a row
void laplacian_csr(int64_t n, int64_t m, const int64_t *ei, const int64_t *ej, const double *ew, const uint8_t *use, Csr &L) {
    L.n = n;
    std::vector<int64_t> cnt(n + 1, 0);
    for (int64_t e = 0; e < m; ++e) if (!use || use[e]) { cnt[ei[e] + 1] += 2; cnt[ej[e] + 1] += 2; }
    for (int64_t r = 0; r < n; ++r) cnt[r + 1] += cnt[r];
    const int64_t raw = cnt[n];
    std::vector<int32_t> col(raw); std::vector<double> val(raw);
    std::vector<int64_t> pos(cnt.begin(), cnt.end() - 1);
    for (int64_t e = 0; e < m; ++e) {
        if (use && !use[e]) continue;
        const int64_t i = ei[e], j = ej[e]; const double w = ew[e];
        col[pos[i]] = (int32_t)i; val[pos[i]++] = w;
        col[pos[j]] = (int32_t)j; val[pos[j]++] = w;
        col[pos[i]] = (int32_t)j; val[pos[i]++] = -w;
        col[pos[j]] = (int32_t)i; val[pos[j]++] = -w;
    }
    L.indptr.assign(n + 1, 0); L.indices.clear(); L.data.clear();
    L.indices.reserve(raw / 2 + n); L.data.reserve(raw / 2 + n);
    for (int64_t r = 0; r < n; ++r) {
        const int64_t a = cnt[r], b = cnt[r + 1];
        for (int64_t p = a + 1; p < b; ++p) {                                    // stable insertion sort by column (rows are short)
            const int32_t c = col[p]; const double v = val[p];
            int64_t q = p;
            while (q > a && col[q - 1] > c) { col[q] = col[q - 1]; val[q] = val[q - 1]; --q; }
            col[q] = c; val[q] = v;
        }
        for (int64_t p = a; p < b;) {
            int64_t q = p; double s = 0.0;
            while (q < b && col[q] == col[p]) s += val[q++];
            L.indices.push_back(col[p]); L.data.push_back(s);
            p = q;
        }
        L.indptr[r + 1] = (int64_t)L.indices.size();
    }
}
void csr_add(const Csr &A, const Csr &B, Csr &C) {                                // C = A + B, rows merged by column
    const int64_t n = A.n;
    C.n = n; C.indptr.assign(n + 1, 0); C.indices.clear(); C.data.clear();
    C.indices.reserve(A.indices.size() + B.indices.size()); C.data.reserve(A.indices.size() + B.indices.size());
    for (int64_t r = 0; r < n; ++r) {
        int64_t p = A.indptr[r], pe = A.indptr[r + 1], q = B.indptr[r], qe = B.indptr[r + 1];
        while (p < pe || q < qe) {
            if (q >= qe || (p < pe && A.indices[p] < B.indices[q])) { C.indices.push_back(A.indices[p]); C.data.push_back(A.data[p]); ++p; }
            else if (p >= pe || B.indices[q] < A.indices[p]) { C.indices.push_back(B.indices[q]); C.data.push_back(B.data[q]); ++q; }
            else { C.indices.push_back(A.indices[p]); C.data.push_back(A.data[p] + B.data[q]); ++p; ++q; }
        }
        C.indptr[r + 1] = (int64_t)C.indices.size();
    }
}
// indicator of the k largest keys (ties: the larger index wins; numpy's argpartition leaves them arbitrary)
template <class Less>
void top_k_indicator(int64_t m, int64_t k, Less less, double *out) {
    std::vector<int64_t> idx(m);
    for (int64_t i = 0; i < m; ++i) { idx[i] = i; out[i] = 0.0; }
    if (k <= 0) return;
    if (k < m) std::nth_element(idx.begin(), idx.begin() + (m - k), idx.end(), less);
    for (int64_t t = (k < m ? m - k : 0); t < m; ++t) out[idx[t]] = 1.0;
}
}  // namespace

CSLAM_API int cslam_mac_fw_subset(int64_t num_poses, int64_t n_fixed, const int64_t *fixed_i, const int64_t *fixed_j,
                                  const double *fixed_w, int64_t n_cand, const int64_t *cand_i, const int64_t *cand_j,
                                  const double *cand_w, const double *w_init, int64_t k, int max_iters, double duality_gap_tol,
                                  double fiedler_tol, double *h_selected, double *h_w_unrounded, double *h_upper, int *h_iters,
                                  void *stream) {
    ARG_CHECK(num_poses > 4 && n_fixed >= 0 && n_cand >= 0 && k >= 0 && k <= n_cand, "bad sizes");
    ARG_CHECK((n_fixed == 0 || (fixed_i && fixed_j && fixed_w)) && (n_cand == 0 || (cand_i && cand_j && cand_w && w_init)) && h_selected,
              "NULL argument");
    for (int64_t e = 0; e < n_fixed; ++e) ARG_CHECK(fixed_i[e] >= 0 && fixed_i[e] < num_poses && fixed_j[e] >= 0 && fixed_j[e] < num_poses, "fixed edge out of range");
    for (int64_t e = 0; e < n_cand; ++e) ARG_CHECK(cand_i[e] >= 0 && cand_i[e] < num_poses && cand_j[e] >= 0 && cand_j[e] < num_poses, "candidate edge out of range");
    Csr Lfix, Lc, L;
    laplacian_csr(num_poses, n_fixed, fixed_i, fixed_j, fixed_w, nullptr, Lfix);
    std::vector<double> w(w_init, w_init + n_cand), grad(n_cand), s(n_cand), prod(n_cand), v(num_poses);
    std::vector<uint8_t> use(n_cand);
    double u = INFINITY, f = 0.0;
    int it = 0;
    for (; it < max_iters; ++it) {
        for (int64_t e = 0; e < n_cand; ++e) { use[e] = w[e] > 1e-10; prod[e] = w[e] * cand_w[e]; }      // mac.py:61-77
        laplacian_csr(num_poses, n_cand, cand_i, cand_j, prod.data(), use.data(), Lc);
        csr_add(Lfix, Lc, L);
        int fi = 0;
        const int rc = cslam_fiedler(num_poses, L.indptr.data(), L.indices.data(), L.data.data(), nullptr, 7u, fiedler_tol, 0, &f, v.data(),
                                     &fi, stream);
        if (rc) return rc;
        for (int64_t e = 0; e < n_cand; ++e) {                                                          // mac.py:112-130
            const double d = v[cand_i[e]] - v[cand_j[e]];
            grad[e] = (cand_w[e] * d) * d;
        }
        top_k_indicator(n_cand, k, [&](int64_t a, int64_t b) { return grad[a] < grad[b] || (grad[a] == grad[b] && a < b); }, s.data());
        double dot = 0.0;
        for (int64_t e = 0; e < n_cand; ++e) dot += grad[e] * (s[e] - w[e]);
        if (f + dot < u) u = f + dot;
        if (u - f < duality_gap_tol) break;
        const double alpha = 2.0 / (it + 2.0);
        for (int64_t e = 0; e < n_cand; ++e) w[e] = w[e] + alpha * (s[e] - w[e]);
    }
    // rounding with the weight tie-break (mac.py:168-189): top k of (w rounded to 10 decimals, edge weight)
    std::vector<double> wr(n_cand);
    for (int64_t e = 0; e < n_cand; ++e) wr[e] = nearbyint(w[e] * 1e10) / 1e10;
    top_k_indicator(n_cand, k, [&](int64_t a, int64_t b) {
        if (wr[a] != wr[b]) return wr[a] < wr[b];
        if (cand_w[a] != cand_w[b]) return cand_w[a] < cand_w[b];
        return a < b;
    }, h_selected);
    if (h_w_unrounded) for (int64_t e = 0; e < n_cand; ++e) h_w_unrounded[e] = w[e];
    if (h_upper) *h_upper = u;
    if (h_iters) *h_iters = it < max_iters ? it + 1 : max_iters;
    return CSLAM_OK;
}
