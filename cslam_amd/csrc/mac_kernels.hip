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

// phase 2: the chunk aggregates -> each chunk's carry-in.  The walk is serial (one lane per column), but over LDS: staged
// by the whole block in parallel first -- walking global memory cost one L2 round trip per chunk (0.2 ms at 1e6 poses, more
// than all streaming passes of a solve together).
#define CARRY_TILE 1024
__global__ __launch_bounds__(256) void segscan_carry_kernel(const double *__restrict__ agg_s, const int *__restrict__ agg_f, int nchunks,
                                                            double *__restrict__ carry) {
    __shared__ double ts[CARRY_TILE * CQ];
    __shared__ int tf[CARRY_TILE];
    double run = 0.0;                                              // lanes 0..3: the running carry of their column
    for (int b0 = 0; b0 < nchunks; b0 += CARRY_TILE) {
        const int nb = nchunks - b0 < CARRY_TILE ? nchunks - b0 : CARRY_TILE;
        __syncthreads();
        for (int i = threadIdx.x; i < nb * CQ; i += 256) ts[i] = agg_s[(size_t)b0 * CQ + i];
        for (int i = threadIdx.x; i < nb; i += 256) tf[i] = agg_f[b0 + i];
        __syncthreads();
        if (threadIdx.x < CQ) {
            const int c = threadIdx.x;
            for (int b = 0; b < nb; ++b) {
                const double a = ts[b * CQ + c];
                ts[b * CQ + c] = run;
                run = tf[b] ? a : run + a;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nb * CQ; i += 256) carry[(size_t)b0 * CQ + i] = ts[i];
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
    hipLaunchKernelGGL(segscan_carry_kernel, dim3(1), dim3(256), 0, st, agg_s, agg_f, nchunks, carry);
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

// out20[i] = sum over the blocks of partial[b][i]: lane t of a 256-thread block adds the blocks b = t, t + 256, ... in order,
// then a fixed tree over the lanes (deterministic).  One thread per output walking 1024 partials was a chain of 1024 L2 round
// trips: 0.36 ms per call, six calls per TraceMIN iteration.
__global__ __launch_bounds__(256) void block4_gram_finish_kernel(const double *__restrict__ partial, int nblocks, double *__restrict__ out20) {
    // one workgroup per output value (20 of them side by side: 28 -> ~6 us); the summation order of a value is unchanged
    __shared__ double red[256];
    const int i = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[(size_t)b * 20 + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out20[i] = red[0];
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

__global__ __launch_bounds__(256) void block4_sum_finish_kernel(const double *__restrict__ partial, int nblocks, double *__restrict__ out1) {
    __shared__ double red[256];
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[b];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out1[0] = red[0];
}

#define B4_GRID 1024
CSLAM_API int cslam_block4_gram_dev(const double *d_A, const double *d_B, int64_t n, double *d_partial,
                                    double *d_out20, void *stream) {
    PTR_DEVICE(d_A);
    ARG_CHECK(d_A && d_B && d_partial && d_out20 && n >= 1, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    int grid = (int)ceil_div64(n, B4_BLOCK); if (grid > B4_GRID) grid = B4_GRID;
    hipLaunchKernelGGL(block4_gram_kernel, dim3(grid), dim3(B4_BLOCK), 0, st, d_A, d_B, n, d_partial);
    hipLaunchKernelGGL(block4_gram_finish_kernel, dim3(20), dim3(256), 0, st, d_partial, grid, d_out20);
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
    hipLaunchKernelGGL(block4_sum_finish_kernel, dim3(1), dim3(256), 0, st, d_partial, grid, d_out1);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

// residual with sigma read from device memory (cslam_fiedler keeps the 4 x 4 algebra on the device); internal, not exported
__global__ __launch_bounds__(B4_BLOCK) void block4_residual_ds_kernel(const double *__restrict__ W, const double *__restrict__ X,
                                                                      int64_t n, const double *__restrict__ y4,
                                                                      const double *__restrict__ sigma_p, double *__restrict__ partial) {
    __shared__ double red[4];
    double acc = 0.0;
    const double y0 = y4[0], y1 = y4[1], y2 = y4[2], y3 = y4[3], sigma = sigma_p[0];
    for (int64_t k = (int64_t)blockIdx.x * B4_BLOCK + threadIdx.x; k < n; k += (int64_t)gridDim.x * B4_BLOCK) {
        double w = W[k * CQ] * y0 + W[k * CQ + 1] * y1 + W[k * CQ + 2] * y2 + W[k * CQ + 3] * y3;
        acc += fabs(w - sigma * X[k * CQ]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
int block4_residual_devsigma(const double *d_W, const double *d_X, int64_t n, const double *d_y4, const double *d_sigma,
                             double *d_partial, double *d_out1, hipStream_t st) {
    int grid = (int)ceil_div64(n, B4_BLOCK); if (grid > B4_GRID) grid = B4_GRID;
    hipLaunchKernelGGL(block4_residual_ds_kernel, dim3(grid), dim3(B4_BLOCK), 0, st, d_W, d_X, n, d_y4, d_sigma, d_partial);
    hipLaunchKernelGGL(block4_sum_finish_kernel, dim3(1), dim3(256), 0, st, d_partial, grid, d_out1);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

// ---- host-synchronous twins for the TraceMIN outer loop: the 4 x 4 algebra between the streaming passes runs on the host
// (LAPACK, as in the reference), so every pass ends in a tiny read-back or starts from a tiny matrix.  Passing those through
// torch tensors cost ~2 ms per iteration in copies, allocations and synchronisations (of ~0.3 ms of kernels at 1e6 poses): here
// the small operands travel as kernel ARGUMENTS, and a read-back is one 160-byte copy + one stream synchronisation in the call.
struct B4Mat { double m[16]; double s[4]; };

__global__ __launch_bounds__(256) void block4_affine_val_kernel(const double *__restrict__ A, int64_t n, B4Mat mv,
                                                                double *__restrict__ out) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double a[CQ], o[CQ];
#pragma unroll
    for (int c = 0; c < CQ; ++c) a[c] = A[k * CQ + c];
#pragma unroll
    for (int j = 0; j < CQ; ++j) {
        double s = -mv.s[j];
#pragma unroll
        for (int i = 0; i < CQ; ++i) s += a[i] * mv.m[i * CQ + j];
        o[j] = s;
    }
#pragma unroll
    for (int c = 0; c < CQ; ++c) out[k * CQ + c] = o[c];
}

__global__ __launch_bounds__(B4_BLOCK) void block4_residual_val_kernel(const double *__restrict__ W, const double *__restrict__ X,
                                                                       int64_t n, double y0, double y1, double y2, double y3,
                                                                       double sigma, double *__restrict__ partial) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int64_t k = (int64_t)blockIdx.x * B4_BLOCK + threadIdx.x; k < n; k += (int64_t)gridDim.x * B4_BLOCK) {
        double w = W[k * CQ] * y0 + W[k * CQ + 1] * y1 + W[k * CQ + 2] * y2 + W[k * CQ + 3] * y3;
        acc += fabs(w - sigma * X[k * CQ]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

CSLAM_API int cslam_block4_gram_sync(const double *d_A, const double *d_B, int64_t n, double *d_partial, double *d_out20,
                                     double *h_out20, void *stream) {
    PTR_DEVICE(d_A);
    ARG_CHECK(d_A && d_B && d_partial && d_out20 && h_out20 && n >= 1, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    int grid = (int)ceil_div64(n, B4_BLOCK); if (grid > B4_GRID) grid = B4_GRID;
    hipLaunchKernelGGL(block4_gram_kernel, dim3(grid), dim3(B4_BLOCK), 0, st, d_A, d_B, n, d_partial);
    hipLaunchKernelGGL(block4_gram_finish_kernel, dim3(20), dim3(256), 0, st, d_partial, grid, d_out20);
    HIP_TRY(hipMemcpyAsync(h_out20, d_out20, 20 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return CSLAM_OK;
}

CSLAM_API int cslam_block4_affine_host(const double *d_A, int64_t n, const double *h_M16, const double *h_shift4, double *d_out,
                                       void *stream) {
    PTR_DEVICE(d_A);
    ARG_CHECK(d_A && h_M16 && d_out && n >= 1, "bad argument");
    B4Mat mv;
    for (int i = 0; i < 16; ++i) mv.m[i] = h_M16[i];
    for (int i = 0; i < 4; ++i) mv.s[i] = h_shift4 ? h_shift4[i] : 0.0;
    hipLaunchKernelGGL(block4_affine_val_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream, d_A, n, mv, d_out);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

CSLAM_API int cslam_block4_residual_sync(const double *d_W, const double *d_X, int64_t n, const double *h_y4, double sigma,
                                         double *d_partial, double *d_out1, double *h_out1, void *stream) {
    PTR_DEVICE(d_W);
    ARG_CHECK(d_W && d_X && h_y4 && d_partial && d_out1 && h_out1 && n >= 1, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    int grid = (int)ceil_div64(n, B4_BLOCK); if (grid > B4_GRID) grid = B4_GRID;
    hipLaunchKernelGGL(block4_residual_val_kernel, dim3(grid), dim3(B4_BLOCK), 0, st, d_W, d_X, n, h_y4[0], h_y4[1], h_y4[2],
                       h_y4[3], sigma, d_partial);
    hipLaunchKernelGGL(block4_sum_finish_kernel, dim3(1), dim3(256), 0, st, d_partial, grid, d_out1);
    HIP_TRY(hipMemcpyAsync(h_out1, d_out1, sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return CSLAM_OK;
}

// =====================================================================================
// x = (L L^T)^-1 b for the dense lower Cholesky factor of the grounded junction Laplacian, 4 right-hand sides
// (the inner solve of every TraceMIN iteration: networkx `_tracemin_fiedler` calls SuperLU's solve there, cslam/mac/mac.py:52-58).
// Blocked substitution with pre-inverted diagonal blocks: per block column one product with the inverted diagonal block and one
// update of the rows below (forward) / the columns before (backward).  Both are matrix x [bs][4] products that read the factor
// exactly once at HBM speed; the library's thin-right-hand-side GEMMs took 8 ms per solve on 32k junctions (2 x 4.3 GB = 1.4 ms
// of traffic).  No atomics: a fixed summation order per output element (deterministic iterates).
#ifndef CS4_THREADS
#define CS4_THREADS 512                 // 2 workgroups x 8 waves per compute unit next to 64 KB of staged x each: 2.48 -> 2.24 ms at 32k junctions
#endif
#ifndef CS4_RPW
#define CS4_RPW 4
#endif
#ifndef CS4_ROWS_UNROLL
#define CS4_ROWS_UNROLL 4
#endif
#ifndef CS4_ROWS_GRID
#define CS4_ROWS_GRID 512
#endif
#define CS4_CT 512                     // threads of the column form: 8 waves, 8 rows each in flight

// out[r][:] (-)= sum_c M[r][c] xin[c][:]   r < nrows, c < ncols <= bs; xin is staged in LDS.  One wave per RPW rows,
// lanes stride the columns (512 contiguous bytes per wave load), butterfly reduction.  copy_to: if set, workgroup 0 also
// writes xin there (lets the caller keep the substitution in place).
template <bool SUB, int RPW>
__global__ __launch_bounds__(CS4_THREADS) void cs4_rows_kernel(const double *__restrict__ M, int64_t ldm, int64_t nrows, int ncols,
                                                               const double *__restrict__ xin, double *__restrict__ out,
                                                               double *__restrict__ copy_to) {
    extern __shared__ __attribute__((aligned(16))) double cs4_x[];              // [ncols_pad][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ncp = (ncols + 63) & ~63;
    for (int i = tid; i < ncp * 4; i += CS4_THREADS) cs4_x[i] = i < ncols * 4 ? xin[i] : 0.0;
    if (copy_to && blockIdx.x == 0)
        for (int i = tid; i < ncols * 4; i += CS4_THREADS) copy_to[i] = xin[i];
    __syncthreads();
    const int64_t rows_per_pass = (int64_t)gridDim.x * (CS4_THREADS / 64) * RPW;
    for (int64_t r0 = ((int64_t)blockIdx.x * (CS4_THREADS / 64) + wave) * RPW; r0 < nrows; r0 += rows_per_pass) {
        double acc[RPW][4];
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
        const double *mp[RPW];
#pragma unroll
        for (int i = 0; i < RPW; ++i) mp[i] = M + (r0 + i < nrows ? r0 + i : nrows - 1) * ldm;
#pragma unroll CS4_ROWS_UNROLL
        for (int c = lane; c < ncp; c += 64) {
            const bool in = c < ncols;
            double m[RPW];
#pragma unroll
            for (int i = 0; i < RPW; ++i) m[i] = in ? mp[i][c] : 0.0;
            const double x0 = cs4_x[4 * c], x1 = cs4_x[4 * c + 1], x2 = cs4_x[4 * c + 2], x3 = cs4_x[4 * c + 3];
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                acc[i][0] += m[i] * x0; acc[i][1] += m[i] * x1; acc[i][2] += m[i] * x2; acc[i][3] += m[i] * x3;
            }
        }
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double v = acc[i][j];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
                acc[i][j] = v;
            }
        if (lane < RPW * 4) {
            const int i = lane >> 2, j = lane & 3;
            double v = 0.0;
#pragma unroll
            for (int a = 0; a < RPW; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) v = (a == i && b == j) ? acc[a][b] : v;
            if (r0 + i < nrows) {
                double *o = out + (r0 + i) * 4 + j;
                *o = SUB ? *o - v : v;
            }
        }
    }
}

// out[r][:] = sum_c T[r][c] xin[c][:] for a TRIANGULAR bw x bw block T (row stride ldm) -- the products with the inverted
// diagonal blocks.  UPPER = false: only columns <= r are read, true: only columns >= r (the other triangle is zero by
// construction and is not even loaded: half the bytes of the full-block form).  One wave per row, rows dealt round robin so
// that long and short rows mix in every workgroup; xin staged in LDS.
template <bool UPPER>
__global__ __launch_bounds__(CS4_THREADS) void cs4_tri_kernel(const double *__restrict__ T, int64_t ldm, int bw,
                                                              const double *__restrict__ xin, double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double cs4_x[];              // [bw_pad][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ncp = (bw + 63) & ~63;
    for (int i = tid; i < ncp * 4; i += CS4_THREADS) cs4_x[i] = i < bw * 4 ? xin[i] : 0.0;
    __syncthreads();
    const int rows_per_pass = gridDim.x * (CS4_THREADS / 64);
    for (int r = blockIdx.x * (CS4_THREADS / 64) + wave; r < bw; r += rows_per_pass) {
        const double *mp = T + (int64_t)r * ldm;
        const int c_lo = UPPER ? (r & ~63) : 0, c_hi = UPPER ? bw : r + 1;       // columns [c_lo, c_hi) hold the row's non-zeros
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 4
        for (int c = c_lo + lane; c < c_hi; c += 64) {
            const double mv = (!UPPER || c >= r) ? mp[c] : 0.0;
            a0 += mv * cs4_x[4 * c]; a1 += mv * cs4_x[4 * c + 1]; a2 += mv * cs4_x[4 * c + 2]; a3 += mv * cs4_x[4 * c + 3];
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            a0 += __shfl_xor(a0, off, 64); a1 += __shfl_xor(a1, off, 64); a2 += __shfl_xor(a2, off, 64); a3 += __shfl_xor(a3, off, 64);
        }
        if (lane == 0) { double *o = out + (int64_t)r * 4; o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3; }
    }
}

// out[c][:] (-)= sum_r M[r][c] xin[r][:]   (the transposed product)  r < nrows <= bs, c < ncols.  One workgroup per 64 columns,
// its 8 waves take the rows round robin with 8 rows each in flight, lanes = consecutive columns (coalesced), xin broadcast from
// LDS; fixed-order LDS reduction over the waves.  copy_to as above.
template <bool SUB>
__global__ __launch_bounds__(CS4_CT) void cs4_cols_kernel(const double *__restrict__ M, int64_t ldm, int nrows, int64_t ncols,
                                                          const double *__restrict__ xin, double *__restrict__ out,
                                                          double *__restrict__ copy_to) {
    extern __shared__ __attribute__((aligned(16))) double cs4_x[];              // [nrows][4], then [8 waves][64][4] for the reduction
    constexpr int NW = CS4_CT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < nrows * 4; i += CS4_CT) cs4_x[i] = xin[i];
    if (copy_to && blockIdx.x == 0)
        for (int i = tid; i < nrows * 4; i += CS4_CT) copy_to[i] = xin[i];
    __syncthreads();
    const int64_t c = (int64_t)blockIdx.x * 64 + lane;
    const bool in = c < ncols;
    const double *mp = M + (in ? c : ncols - 1);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int r = wave;
    for (; r + 7 * NW < nrows; r += 8 * NW) {
        double m[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) m[u] = mp[(int64_t)(r + u * NW) * ldm];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double *x = cs4_x + 4 * (r + u * NW);
            a0 += m[u] * x[0]; a1 += m[u] * x[1]; a2 += m[u] * x[2]; a3 += m[u] * x[3];
        }
    }
    for (; r < nrows; r += NW) {
        const double m0 = mp[(int64_t)r * ldm];
        const double *x = cs4_x + 4 * r;
        a0 += m0 * x[0]; a1 += m0 * x[1]; a2 += m0 * x[2]; a3 += m0 * x[3];
    }
    __syncthreads();                                                // xin no longer needed: reuse the LDS for the reduction
    double *red = cs4_x;
    red[(wave * 64 + lane) * 4 + 0] = a0; red[(wave * 64 + lane) * 4 + 1] = a1;
    red[(wave * 64 + lane) * 4 + 2] = a2; red[(wave * 64 + lane) * 4 + 3] = a3;
    __syncthreads();
    if (tid < 256) {
        const int cc = tid >> 2, j = tid & 3;
        const int64_t gc = (int64_t)blockIdx.x * 64 + cc;
        if (gc < ncols) {
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += red[(w * 64 + cc) * 4 + j];
            double *o = out + gc * 4 + j;
            *o = SUB ? *o - v : v;
        }
    }
}

CSLAM_API int cslam_chol_solve4_dev(const double *d_L, int64_t m, int64_t ld, int col_major, const double *d_dinv,
                                    const double *d_dinvT, int bs, double *d_x, double *d_tmp, void *stream) {
    PTR_DEVICE(d_L);
    ARG_CHECK(d_L && d_dinv && d_dinvT && d_x && d_tmp, "NULL argument");
    ARG_CHECK(m >= 1 && ld >= m, "bad m / ld");
    ARG_CHECK(bs >= 64 && bs <= 4096 && (bs % 64) == 0, "block size must be a multiple of 64 in [64, 4096]");
    hipStream_t st = (hipStream_t)stream;
    const int lds_rows = bs * 4 * 8, lds_cols = (bs * 4 * 8 > (CS4_CT / 64) * 64 * 4 * 8) ? bs * 4 * 8 : (CS4_CT / 64) * 64 * 4 * 8;
    static DeviceOnce once;                                             // function attributes are per device
    int once_dev;
    if (once.todo(&once_dev)) {
        HIP_TRY(hipFuncSetAttribute((const void *)cs4_tri_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 4096 * 32));
        HIP_TRY(hipFuncSetAttribute((const void *)cs4_tri_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4096 * 32));
        HIP_TRY(hipFuncSetAttribute((const void *)cs4_rows_kernel<true, CS4_RPW>, hipFuncAttributeMaxDynamicSharedMemorySize, 4096 * 32));
        HIP_TRY(hipFuncSetAttribute((const void *)cs4_cols_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4096 * 32));
        once.done(once_dev);
    }
    const int64_t nb = ceil_div64(m, bs);
    auto row_grid = [](int64_t rows, int rpw) { int64_t g = ceil_div64(rows, (CS4_THREADS / 64) * rpw); return (unsigned)(g > CS4_ROWS_GRID ? CS4_ROWS_GRID : (g < 1 ? 1 : g)); };
    // With a column-major factor the memory image is the row-major UPPER factor L^T: the forward update reads it by columns
    // (cols form), the backward update by rows.  The products with the inverted diagonal blocks are always in the row form
    // (dinv forward, its transpose dinvT backward: one row per wave, 4 rows per workgroup, so that even one 2048-row block
    // fills the chip).  The block's result goes to tmp; the update launch that consumes it also copies it back into x.
    // forward: L y = b
    for (int64_t t = 0; t < nb; ++t) {
        const int64_t k = t * bs, e = (k + bs < m) ? k + bs : m;
        const int bw = (int)(e - k);
        hipLaunchKernelGGL(cs4_tri_kernel<false>, dim3(row_grid(bw, 1)), dim3(CS4_THREADS), lds_rows, st,
                           d_dinv + (size_t)t * bs * bs, (int64_t)bs, bw, d_x + k * 4, d_tmp);
        if (e >= m)
            HIP_TRY(hipMemcpyAsync(d_x + k * 4, d_tmp, (size_t)bw * 32, hipMemcpyDeviceToDevice, st));
        else if (!col_major)
            hipLaunchKernelGGL((cs4_rows_kernel<true, CS4_RPW>), dim3(row_grid(m - e, CS4_RPW)), dim3(CS4_THREADS), lds_rows, st,
                               d_L + e * ld + k, ld, m - e, bw, d_tmp, d_x + e * 4, d_x + k * 4);
        else
            hipLaunchKernelGGL(cs4_cols_kernel<true>, dim3((unsigned)ceil_div64(m - e, 64)), dim3(CS4_CT), lds_cols, st,
                               d_L + k * ld + e, ld, bw, m - e, d_tmp, d_x + e * 4, d_x + k * 4);
    }
    // backward: L^T x = y
    for (int64_t t = nb - 1; t >= 0; --t) {
        const int64_t k = t * bs, e = (k + bs < m) ? k + bs : m;
        const int bw = (int)(e - k);
        hipLaunchKernelGGL(cs4_tri_kernel<true>, dim3(row_grid(bw, 1)), dim3(CS4_THREADS), lds_rows, st,
                           d_dinvT + (size_t)t * bs * bs, (int64_t)bs, bw, d_x + k * 4, d_tmp);
        if (k == 0)
            HIP_TRY(hipMemcpyAsync(d_x + k * 4, d_tmp, (size_t)bw * 32, hipMemcpyDeviceToDevice, st));
        else if (!col_major)
            hipLaunchKernelGGL(cs4_cols_kernel<true>, dim3((unsigned)ceil_div64(k, 64)), dim3(CS4_CT), lds_cols, st,
                               d_L + k * ld, ld, bw, k, d_tmp, d_x, d_x + k * 4);
        else
            hipLaunchKernelGGL((cs4_rows_kernel<true, CS4_RPW>), dim3(row_grid(k, CS4_RPW)), dim3(CS4_THREADS), lds_rows, st,
                               d_L + k, ld, k, bw, d_tmp, d_x, d_x + k * 4);
    }
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}
