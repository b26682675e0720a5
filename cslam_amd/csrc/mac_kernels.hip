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
    ARG_CHECK(n >= 0 && nvec >= 1, "bad n / nvec");
    if (n == 0) return CSLAM_OK;
    ARG_CHECK(d_indptr && d_indices && d_data && d_x && d_y, "NULL argument");
    hipLaunchKernelGGL(csr_spmm_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       d_indptr, d_indices, d_data, n, d_x, nvec, d_y);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}
