// mac_kernels.hip -- device pieces of the candidate sparsifier (cslam/mac/mac.py), float64.
//   mac_grad   grad_from_fiedler  mac.py:112-130   g[k] = w[k] (v[i_k] - v[j_k])^2
//   csr_spmm   L(w) @ X  inside networkx _tracemin_fiedler (called from mac.py:52-58)
// Both are HBM-bound gathers; one thread per output element, no atomics (deterministic).
#include "common.h"

__global__ __launch_bounds__(256) void mac_grad_kernel(const double *__restrict__ v, const int32_t *__restrict__ ei,
                                                       const int32_t *__restrict__ ej, const double *__restrict__ w,
                                                       int64_t m, double *__restrict__ g) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    double vi = v[ei[k]], vj = v[ej[k]];
    double kdelta = w[k] * (vi - vj);      // same operation order as mac.py:127-129
    g[k] = kdelta * (vi - vj);
}

CSLAM_API int cslam_mac_grad_dev(const double *d_fiedler, const int32_t *d_edge_i, const int32_t *d_edge_j,
                                 const double *d_weights, int64_t m, double *d_grad, void *stream) {
    PTR_DEVICE(d_fiedler);
    ARG_CHECK(m >= 0, "m < 0");
    if (m == 0) return CSLAM_OK;
    ARG_CHECK(d_fiedler && d_edge_i && d_edge_j && d_weights && d_grad, "NULL argument");
    hipLaunchKernelGGL(mac_grad_kernel, dim3((unsigned)ceil_div64(m, 256)), dim3(256), 0, (hipStream_t)stream,
                       d_fiedler, d_edge_i, d_edge_j, d_weights, m, d_grad);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

__global__ __launch_bounds__(256) void csr_spmm_kernel(const int64_t *__restrict__ indptr,
                                                       const int32_t *__restrict__ indices,
                                                       const double *__restrict__ data, int64_t n,
                                                       const double *__restrict__ x, int nvec,
                                                       double *__restrict__ y) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t p0 = indptr[r], p1 = indptr[r + 1];
    for (int c = 0; c < nvec; ++c) {
        const double *xc = x + (size_t)c * n;
        double s = 0.0;
        for (int64_t p = p0; p < p1; ++p) s += data[p] * xc[indices[p]];
        y[(size_t)c * n + r] = s;
    }
}

CSLAM_API int cslam_csr_spmm_dev(const int64_t *d_indptr, const int32_t *d_indices, const double *d_data,
                                 int64_t n, const double *d_x, int nvec, double *d_y, void *stream) {
    PTR_DEVICE(d_indptr);
    ARG_CHECK(n >= 0 && nvec >= 1, "bad n / nvec");
    if (n == 0) return CSLAM_OK;
    ARG_CHECK(d_indptr && d_indices && d_data && d_x && d_y, "NULL argument");
    hipLaunchKernelGGL(csr_spmm_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       d_indptr, d_indices, d_data, n, d_x, nvec, d_y);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

// =====================================================================================
// Chain-reduced Laplacian solves (cslam_amd/mac/chain_solver.py states the algorithm):
// the odometry chains of the pose graph are eliminated in closed form with segmented prefix sums,
// leaving a small dense system on the junction nodes.  Vectors are [n][4] float64, row-major
// (the 4 TraceMIN columns of a node share one 32-byte line: one lane = one node).
// =====================================================================================
#define CQ 4
#define SCAN_ITEMS 8
#define SCAN_BLOCK 256
#define SCAN_CHUNK (SCAN_ITEMS * SCAN_BLOCK)

struct Seg4 { double s[CQ]; int f; };   // running sums since the last flagged element; f = flag seen

__device__ __forceinline__ Seg4 seg_combine(const Seg4 &a, const Seg4 &b) {   // a then b
    Seg4 r;
    r.f = a.f | b.f;
#pragma unroll
    for (int c = 0; c < CQ; ++c) r.s[c] = b.f ? b.s[c] : a.s[c] + b.s[c];
    return r;
}

// phase 1: per-chunk segmented inclusive scan (a flagged element contributes 0 and restarts the sum);
// writes the local result and the chunk aggregate.
__global__ __launch_bounds__(SCAN_BLOCK) void segscan_local_kernel(const double *__restrict__ v,
                                                                    const uint8_t *__restrict__ flag, int64_t n,
                                                                    double *__restrict__ out,
                                                                    double *__restrict__ agg_s, int *__restrict__ agg_f) {
    __shared__ double ws[4][CQ];
    __shared__ int wf[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)tid * SCAN_ITEMS;
    double loc[SCAN_ITEMS][CQ];
    int lf[SCAN_ITEMS];
    Seg4 run; run.f = 0;
#pragma unroll
    for (int c = 0; c < CQ; ++c) run.s[c] = 0.0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t k = base + i;
        const int fl = (k < n) ? flag[k] : 0;
        lf[i] = run.f | fl;
#pragma unroll
        for (int c = 0; c < CQ; ++c) {
            double x = (k < n && !fl) ? v[k * CQ + c] : 0.0;
            run.s[c] = fl ? 0.0 : run.s[c] + x;
            loc[i][c] = run.s[c];
        }
        run.f |= fl;
    }
    // inclusive scan of the per-thread aggregates across the wave
    Seg4 inc = run;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        Seg4 o;
        o.f = __shfl_up(inc.f, off, 64);
#pragma unroll
        for (int c = 0; c < CQ; ++c) o.s[c] = __shfl_up(inc.s[c], off, 64);
        if (lane >= off) inc = seg_combine(o, inc);
    }
    if (lane == 63) { wf[wave] = inc.f; for (int c = 0; c < CQ; ++c) ws[wave][c] = inc.s[c]; }
    __syncthreads();
    // exclusive prefix for this thread = (waves before) then (lanes before)
    Seg4 pre; pre.f = 0;
#pragma unroll
    for (int c = 0; c < CQ; ++c) pre.s[c] = 0.0;
    for (int w = 0; w < wave; ++w) {
        Seg4 t; t.f = wf[w];
        for (int c = 0; c < CQ; ++c) t.s[c] = ws[w][c];
        pre = seg_combine(pre, t);
    }
    {
        Seg4 o;
        o.f = __shfl_up(inc.f, 1, 64);
#pragma unroll
        for (int c = 0; c < CQ; ++c) o.s[c] = __shfl_up(inc.s[c], 1, 64);
        if (lane > 0) pre = seg_combine(pre, o);
    }
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t k = base + i;
        if (k < n) {
#pragma unroll
            for (int c = 0; c < CQ; ++c) out[k * CQ + c] = lf[i] ? loc[i][c] : pre.s[c] + loc[i][c];
        }
    }
    if (tid == SCAN_BLOCK - 1) {
        Seg4 tot = seg_combine(pre, run);
        agg_f[blockIdx.x] = tot.f;
        for (int c = 0; c < CQ; ++c) agg_s[(size_t)blockIdx.x * CQ + c] = tot.s[c];
    }
}

// phase 2: one wave walks the chunk aggregates and produces each chunk's carry-in
__global__ void segscan_carry_kernel(const double *__restrict__ agg_s, const int *__restrict__ agg_f, int nchunks,
                                     double *__restrict__ carry) {
    if (threadIdx.x >= CQ) return;
    const int c = threadIdx.x;
    double run = 0.0;
    for (int b = 0; b < nchunks; ++b) {
        carry[(size_t)b * CQ + c] = run;
        run = agg_f[b] ? agg_s[(size_t)b * CQ + c] : run + agg_s[(size_t)b * CQ + c];
    }
}

// phase 3: elements before the first flag of their chunk receive the chunk's carry-in
__global__ __launch_bounds__(SCAN_BLOCK) void segscan_fix_kernel(const uint8_t *__restrict__ flag, int64_t n,
                                                                  const double *__restrict__ carry,
                                                                  double *__restrict__ out) {
    __shared__ int first_flag;
    if (threadIdx.x == 0) first_flag = SCAN_CHUNK;
    __syncthreads();
    const int64_t cbase = (int64_t)blockIdx.x * SCAN_CHUNK;
    int mine = SCAN_CHUNK;
    for (int i = threadIdx.x; i < SCAN_CHUNK; i += SCAN_BLOCK) {
        int64_t k = cbase + i;
        if (k < n && flag[k] && i < mine) mine = i;
    }
    atomicMin(&first_flag, mine);
    __syncthreads();
    const int ff = first_flag;
    for (int i = threadIdx.x; i < ff; i += SCAN_BLOCK) {
        int64_t k = cbase + i;
        if (k < n) {
#pragma unroll
            for (int c = 0; c < CQ; ++c) out[k * CQ + c] += carry[(size_t)blockIdx.x * CQ + c];
        }
    }
}

static int segscan4(const double *d_v, const uint8_t *d_flag, int64_t n, double *d_out, double *d_scratch,
                    hipStream_t st) {
    const int nchunks = (int)ceil_div64(n, SCAN_CHUNK);
    double *agg_s = d_scratch;                                  // [nchunks][4]
    double *carry = d_scratch + (size_t)nchunks * CQ;            // [nchunks][4]
    int *agg_f = (int *)(d_scratch + (size_t)2 * nchunks * CQ);  // [nchunks]
    hipLaunchKernelGGL(segscan_local_kernel, dim3(nchunks), dim3(SCAN_BLOCK), 0, st, d_v, d_flag, n, d_out, agg_s, agg_f);
    hipLaunchKernelGGL(segscan_carry_kernel, dim3(1), dim3(64), 0, st, agg_s, agg_f, nchunks, carry);
    hipLaunchKernelGGL(segscan_fix_kernel, dim3(nchunks), dim3(SCAN_BLOCK), 0, st, d_flag, n, carry, d_out);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

// t[k] = r[k-1] * Bn[k-1]  (the resistance-weighted running injection arriving at node k)
__global__ __launch_bounds__(256) void chain_weighted_shift_kernel(const double *__restrict__ Bn,
                                                                   const double *__restrict__ r, int64_t n,
                                                                   double *__restrict__ t) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
#pragma unroll
    for (int c = 0; c < CQ; ++c) t[k * CQ + c] = k > 0 ? r[k - 1] * Bn[(k - 1) * CQ + c] : 0.0;
}

// reduced right-hand side on the junctions: bt[j] = b[J[j]] (+ Ql/Rl if a segment starts at j)
//                                                   (+ Bl - Ql/Rl if a segment ends at j)
__global__ __launch_bounds__(256) void chain_reduce_rhs_kernel(
    const double *__restrict__ b, const double *__restrict__ Bn, const double *__restrict__ Qn,
    const double *__restrict__ r, const int64_t *__restrict__ J, int nJ,
    const int *__restrict__ seg_start_of, const int *__restrict__ seg_end_of,
    const int64_t *__restrict__ sa, const int64_t *__restrict__ sb, const double *__restrict__ Rl,
    double *__restrict__ bt) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nJ) return;
    double v[CQ];
#pragma unroll
    for (int c = 0; c < CQ; ++c) v[c] = b[J[j] * CQ + c];
    for (int which = 0; which < 2; ++which) {
        const int s = which == 0 ? seg_start_of[j] : seg_end_of[j];
        if (s < 0) continue;
        const int64_t a = sa[s], e = sb[s] - 1;                  // last node before the end junction
#pragma unroll
        for (int c = 0; c < CQ; ++c) {
            const double Bl = Bn[e * CQ + c];                     // 0 when e == a
            const double Ql = (e > a ? Qn[e * CQ + c] : 0.0) + r[e] * Bl;
            const double corr = Ql / Rl[s];
            v[c] += which == 0 ? corr : Bl - corr;
        }
    }
#pragma unroll
    for (int c = 0; c < CQ; ++c) bt[(size_t)j * CQ + c] = v[c];
}

// interior potentials: x_k = x_a - f1 R_k - Q_k, f1 = (x_a - x_b - Ql) / Rl ; junctions copy xJ
__global__ __launch_bounds__(256) void chain_back_subst_kernel(
    const double *__restrict__ xJ, const double *__restrict__ Bn, const double *__restrict__ Qn,
    const double *__restrict__ r, const double *__restrict__ Rn, const int *__restrict__ jid,
    const int *__restrict__ seg_of, const int64_t *__restrict__ sa, const int64_t *__restrict__ sb,
    const double *__restrict__ Rl, int64_t n, double *__restrict__ x) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int jj = jid[k];
    if (jj >= 0) {
#pragma unroll
        for (int c = 0; c < CQ; ++c) x[k * CQ + c] = xJ[(size_t)jj * CQ + c];
        return;
    }
    const int s = seg_of[k];
    const int64_t a = sa[s], bnode = sb[s], e = bnode - 1;
    const int ja = jid[a], jb = jid[bnode];
#pragma unroll
    for (int c = 0; c < CQ; ++c) {
        const double Bl = Bn[e * CQ + c];
        const double Ql = (e > a ? Qn[e * CQ + c] : 0.0) + r[e] * Bl;
        const double xa = xJ[(size_t)ja * CQ + c], xb = xJ[(size_t)jb * CQ + c];
        const double f1 = (xa - xb - Ql) / Rl[s];
        x[k * CQ + c] = xa - f1 * Rn[k] - Qn[k * CQ + c];
    }
}

CSLAM_API int cslam_chain_forward_dev(const double *d_b, const uint8_t *d_is_junction, const double *d_r, int64_t n,
                                      const int64_t *d_J, int nJ, const int *d_seg_start_of, const int *d_seg_end_of,
                                      const int64_t *d_sa, const int64_t *d_sb, const double *d_Rl,
                                      double *d_Bn, double *d_Qn, double *d_tmp, double *d_scratch, double *d_bt,
                                      void *stream) {
    PTR_DEVICE(d_b);
    ARG_CHECK(d_b && d_is_junction && d_r && d_J && d_Bn && d_Qn && d_tmp && d_scratch && d_bt, "NULL argument");
    ARG_CHECK(n >= 2 && nJ >= 1, "bad n / nJ");
    hipStream_t st = (hipStream_t)stream;
    int rc = segscan4(d_b, d_is_junction, n, d_Bn, d_scratch, st);
    if (rc) return rc;
    hipLaunchKernelGGL(chain_weighted_shift_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, st, d_Bn, d_r, n, d_tmp);
    rc = segscan4(d_tmp, d_is_junction, n, d_Qn, d_scratch, st);
    if (rc) return rc;
    hipLaunchKernelGGL(chain_reduce_rhs_kernel, dim3((unsigned)ceil_div64(nJ, 256)), dim3(256), 0, st, d_b, d_Bn, d_Qn,
                       d_r, d_J, nJ, d_seg_start_of, d_seg_end_of, d_sa, d_sb, d_Rl, d_bt);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

CSLAM_API int cslam_chain_backward_dev(const double *d_xJ, const double *d_Bn, const double *d_Qn, const double *d_r,
                                       const double *d_Rn, const int *d_jid, const int *d_seg_of, const int64_t *d_sa,
                                       const int64_t *d_sb, const double *d_Rl, int64_t n, double *d_x, void *stream) {
    PTR_DEVICE(d_xJ);
    ARG_CHECK(d_xJ && d_Bn && d_Qn && d_r && d_Rn && d_jid && d_seg_of && d_x, "NULL argument");
    hipLaunchKernelGGL(chain_back_subst_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       d_xJ, d_Bn, d_Qn, d_r, d_Rn, d_jid, d_seg_of, d_sa, d_sb, d_Rl, n, d_x);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

// y = A x for CSR A and x, y [n][4] row-major float64 (the L @ X of the TraceMIN block iteration)
__global__ __launch_bounds__(256) void csr_spmm4_kernel(const int64_t *__restrict__ indptr,
                                                        const int32_t *__restrict__ indices,
                                                        const double *__restrict__ data, int64_t n,
                                                        const double *__restrict__ x, double *__restrict__ y) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double s[CQ] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t p = indptr[r]; p < indptr[r + 1]; ++p) {
        const double a = data[p];
        const double *xr = x + (int64_t)indices[p] * CQ;
#pragma unroll
        for (int c = 0; c < CQ; ++c) s[c] += a * xr[c];
    }
#pragma unroll
    for (int c = 0; c < CQ; ++c) y[r * CQ + c] = s[c];
}

CSLAM_API int cslam_csr_spmm4_dev(const int64_t *d_indptr, const int32_t *d_indices, const double *d_data,
                                  int64_t n, const double *d_x, double *d_y, void *stream) {
    PTR_DEVICE(d_indptr);
    ARG_CHECK(n >= 0, "n < 0");
    if (n == 0) return CSLAM_OK;
    ARG_CHECK(d_indptr && d_indices && d_data && d_x && d_y, "NULL argument");
    hipLaunchKernelGGL(csr_spmm4_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       d_indptr, d_indices, d_data, n, d_x, d_y);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

// ---- 4-column block operations of the TraceMIN outer loop ([n][4] float64 row-major) ----
// (tall-skinny products through a BLAS dgemm take ~10-30 ms each at n = 1e6; these are one
// streaming pass each)
#define B4_BLOCK 256
// partial[block][16] = sum over the block's rows of a_i^T b_i ; partial[block][16..19] = column sums of b
__global__ __launch_bounds__(B4_BLOCK) void block4_gram_kernel(const double *__restrict__ A, const double *__restrict__ Bm,
                                                               int64_t n, double *__restrict__ partial) {
    __shared__ double red[4][20];
    double acc[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) acc[i] = 0.0;
    for (int64_t k = (int64_t)blockIdx.x * B4_BLOCK + threadIdx.x; k < n; k += (int64_t)gridDim.x * B4_BLOCK) {
        double a[CQ], b[CQ];
#pragma unroll
        for (int c = 0; c < CQ; ++c) { a[c] = A[k * CQ + c]; b[c] = Bm[k * CQ + c]; }
#pragma unroll
        for (int i = 0; i < CQ; ++i)
#pragma unroll
            for (int j = 0; j < CQ; ++j) acc[i * CQ + j] += a[i] * b[j];
#pragma unroll
        for (int j = 0; j < CQ; ++j) acc[16 + j] += b[j];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 20; ++i) {
        double v = acc[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 20)
        partial[(size_t)blockIdx.x * 20 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void block4_gram_finish_kernel(const double *__restrict__ partial, int nblocks, double *__restrict__ out20) {
    const int i = threadIdx.x;
    if (i >= 20) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += partial[(size_t)b * 20 + i];   // fixed order: deterministic
    out20[i] = s;
}

// out[k][:] = A[k][:] * M (4x4, row-major) - shift[:]
__global__ __launch_bounds__(256) void block4_affine_kernel(const double *__restrict__ A, int64_t n,
                                                            const double *__restrict__ M16, const double *__restrict__ shift4,
                                                            double *__restrict__ out) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double a[CQ], o[CQ];
#pragma unroll
    for (int c = 0; c < CQ; ++c) a[c] = A[k * CQ + c];
#pragma unroll
    for (int j = 0; j < CQ; ++j) {
        double s = shift4 ? -shift4[j] : 0.0;
#pragma unroll
        for (int i = 0; i < CQ; ++i) s += a[i] * M16[i * CQ + j];
        o[j] = s;
    }
#pragma unroll
    for (int c = 0; c < CQ; ++c) out[k * CQ + c] = o[c];
}

// partial[block] = sum_k | sum_c W[k][c] y[c]  -  sigma * X[k][0] |   (1-norm of the Ritz residual)
__global__ __launch_bounds__(B4_BLOCK) void block4_residual_kernel(const double *__restrict__ W, const double *__restrict__ X,
                                                                   int64_t n, const double *__restrict__ y4, double sigma,
                                                                   double *__restrict__ partial) {
    __shared__ double red[4];
    double acc = 0.0;
    const double y0 = y4[0], y1 = y4[1], y2 = y4[2], y3 = y4[3];
    for (int64_t k = (int64_t)blockIdx.x * B4_BLOCK + threadIdx.x; k < n; k += (int64_t)gridDim.x * B4_BLOCK) {
        double w = W[k * CQ] * y0 + W[k * CQ + 1] * y1 + W[k * CQ + 2] * y2 + W[k * CQ + 3] * y3;
        acc += fabs(w - sigma * X[k * CQ]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void block4_sum_finish_kernel(const double *__restrict__ partial, int nblocks, double *__restrict__ out1) {
    if (threadIdx.x != 0) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += partial[b];
    out1[0] = s;
}

#define B4_GRID 1024
CSLAM_API int cslam_block4_gram_dev(const double *d_A, const double *d_B, int64_t n, double *d_partial,
                                    double *d_out20, void *stream) {
    PTR_DEVICE(d_A);
    ARG_CHECK(d_A && d_B && d_partial && d_out20 && n >= 1, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    int grid = (int)ceil_div64(n, B4_BLOCK); if (grid > B4_GRID) grid = B4_GRID;
    hipLaunchKernelGGL(block4_gram_kernel, dim3(grid), dim3(B4_BLOCK), 0, st, d_A, d_B, n, d_partial);
    hipLaunchKernelGGL(block4_gram_finish_kernel, dim3(1), dim3(64), 0, st, d_partial, grid, d_out20);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

CSLAM_API int cslam_block4_affine_dev(const double *d_A, int64_t n, const double *d_M16, const double *d_shift4,
                                      double *d_out, void *stream) {
    PTR_DEVICE(d_A);
    ARG_CHECK(d_A && d_M16 && d_out && n >= 1, "bad argument");
    hipLaunchKernelGGL(block4_affine_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       d_A, n, d_M16, d_shift4, d_out);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

CSLAM_API int cslam_block4_residual_dev(const double *d_W, const double *d_X, int64_t n, const double *d_y4,
                                        double sigma, double *d_partial, double *d_out1, void *stream) {
    PTR_DEVICE(d_W);
    ARG_CHECK(d_W && d_X && d_y4 && d_partial && d_out1 && n >= 1, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    int grid = (int)ceil_div64(n, B4_BLOCK); if (grid > B4_GRID) grid = B4_GRID;
    hipLaunchKernelGGL(block4_residual_kernel, dim3(grid), dim3(B4_BLOCK), 0, st, d_W, d_X, n, d_y4, sigma, d_partial);
    hipLaunchKernelGGL(block4_sum_finish_kernel, dim3(1), dim3(64), 0, st, d_partial, grid, d_out1);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}
