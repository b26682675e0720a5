// heads.hip -- descriptor heads of the two extractors + the image transform (gfx950).
//
//   l2_normalize      F.normalize(p=2)  layers.py:32-36 / netvlad.py:105-106,126-128,
//                     and sklearn.preprocessing.normalize (netvlad.py:235-236)
//   vlad_aggregate    NetVLADLayer.forward                    cslam/vpr/netvlad.py:94-130
//   gem_fc_head       L2Norm -> GeM -> Flatten -> Linear -> L2Norm
//                                          cslam/vpr/cosplace_utils/network.py:23-29, layers.py:8-36
//   preprocess        CenterCrop -> Resize(bicubic, PIL, antialias, 8-bit) -> ToTensor -> Normalize
//                                          cslam/vpr/netvlad.py:202-208, cosplace.py:73-79
//
// All of these are HBM-bandwidth bound: one workgroup per image / row,
// coalesced reads along the contiguous (pixel) axis, wave64 shuffle reductions, LDS for the
// per-image intermediates.  float32 arithmetic like the reference's torch float32 modules.
#include "common.h"
#include <vector>

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// block-wide sum (blockDim.x multiple of 64, <= 1024); red: >= 16 floats of LDS
__device__ __forceinline__ float block_sum_f32(float v, float *red) {
    v = wave_sum_f32(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.0f;
    for (int w = 0; w < nw; ++w) t += red[w];
    return t;
}

// ------------------------------------------------------------- l2 normalise ----
__global__ __launch_bounds__(256) void l2_normalize_kernel(float *__restrict__ x, int d, int64_t ld, float eps,
                                                           int zero_norm_to_one) {
    __shared__ float red[16];
    float *row = x + (int64_t)blockIdx.x * ld;
    float s = 0.0f;
    for (int c = threadIdx.x; c < d; c += blockDim.x) { float v = row[c]; s += v * v; }
    float nrm = sqrtf(block_sum_f32(s, red));
    float den = zero_norm_to_one ? (nrm == 0.0f ? 1.0f : nrm) : fmaxf(nrm, eps);
    for (int c = threadIdx.x; c < d; c += blockDim.x) row[c] = row[c] / den;
}

CSLAM_API int cslam_l2_normalize_dev(float *d_x, int64_t n, int d, int64_t ld, float eps,
                                     int zero_norm_to_one, void *stream) {
    ARG_CHECK(d_x || n == 0, "NULL argument");
    ARG_CHECK(n >= 0 && d > 0 && ld >= d, "bad n / d / ld");
    if (n == 0) return CSLAM_OK;
    hipLaunchKernelGGL(l2_normalize_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, d_x, d, ld,
                       eps, zero_norm_to_one);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

// ------------------------------------------------------------- NetVLAD head ----
// One workgroup per image.  K = 64 clusters (netvlad.py:176), C <= 512 channels.
// Pixels are processed in chunks of PCH: phase B (one thread per pixel) computes the pixel's
// inverse L2 norm and the 64 soft-assignment logits in ONE pass over the channels
// (logit = inv_norm * sum_c W[k,c] x[c,p], linear), softmax over k -> a[p][k] in LDS;
// phase C (one thread per channel) accumulates V[k,c] += a[p][k] * x[c,p]*inv_norm[p].
#define VK 64
#define VPCH 128
#define VWCH 32
__global__ __launch_bounds__(512) void vlad_kernel(const float *__restrict__ feat, const float *__restrict__ W,
                                                   const float *__restrict__ bias, const float *__restrict__ cent,
                                                   int C, int P, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *a_lds = (float *)smem;                    // [VPCH][VK]
    float *w_lds = a_lds + VPCH * VK;                // [VWCH][VK]  (transposed weight chunk)
    float *invn = w_lds + VWCH * VK;                 // [VPCH]
    float *red = invn + VPCH;                        // [16][VK] + 16
    const int tid = threadIdx.x, nt = blockDim.x;
    const float *x = feat + (size_t)blockIdx.x * C * P;
    const int c_own = tid;                           // channel owned in phase C (tid < C)

    float acc[VK], asum[VK];
#pragma unroll
    for (int k = 0; k < VK; ++k) { acc[k] = 0.0f; asum[k] = 0.0f; }

    for (int p0 = 0; p0 < P; p0 += VPCH) {
        const int pn = P - p0 < VPCH ? P - p0 : VPCH;
        // ---- phase B: thread tid < pn owns pixel p0 + tid
        float lg[VK];
#pragma unroll
        for (int k = 0; k < VK; ++k) lg[k] = 0.0f;
        float ss = 0.0f;
        for (int c0 = 0; c0 < C; c0 += VWCH) {
            __syncthreads();
            for (int e = tid; e < VWCH * VK; e += nt) {        // w_lds[cc][k] = W[k][c0+cc]
                int k = e / VWCH, cc = e - k * VWCH;
                w_lds[cc * VK + k] = (c0 + cc < C) ? W[(size_t)k * C + c0 + cc] : 0.0f;
            }
            __syncthreads();
            if (tid < pn) {
                const int cn = C - c0 < VWCH ? C - c0 : VWCH;
                for (int cc = 0; cc < cn; ++cc) {
                    float xv = x[(size_t)(c0 + cc) * P + p0 + tid];
                    ss += xv * xv;
                    const float4 *wr = (const float4 *)(w_lds + cc * VK);
#pragma unroll
                    for (int k4 = 0; k4 < VK / 4; ++k4) {
                        float4 w4 = wr[k4];
                        lg[4 * k4 + 0] += w4.x * xv; lg[4 * k4 + 1] += w4.y * xv;
                        lg[4 * k4 + 2] += w4.z * xv; lg[4 * k4 + 3] += w4.w * xv;
                    }
                }
            }
        }
        if (tid < pn) {
            float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);           // F.normalize(dim=1), netvlad.py:105-106
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < VK; ++k) { lg[k] = lg[k] * inv + (bias ? bias[k] : 0.0f); mx = fmaxf(mx, lg[k]); }
            float se = 0.0f;
#pragma unroll
            for (int k = 0; k < VK; ++k) { lg[k] = expf(lg[k] - mx); se += lg[k]; }
            float rs = 1.0f / se;
#pragma unroll
            for (int k = 0; k < VK; ++k) a_lds[tid * VK + k] = lg[k] * rs;   // softmax over clusters, :109-110
            invn[tid] = inv;
        }
        __syncthreads();
        // ---- phase C: thread c_own < C accumulates over the chunk's pixels
        if (c_own < C) {
            for (int pp = 0; pp < pn; ++pp) {
                float xv = x[(size_t)c_own * P + p0 + pp] * invn[pp];
                const float4 *ar = (const float4 *)(a_lds + pp * VK);
#pragma unroll
                for (int k4 = 0; k4 < VK / 4; ++k4) {
                    float4 a4 = ar[k4];
                    acc[4 * k4 + 0] += a4.x * xv; acc[4 * k4 + 1] += a4.y * xv;
                    acc[4 * k4 + 2] += a4.z * xv; acc[4 * k4 + 3] += a4.w * xv;
                    asum[4 * k4 + 0] += a4.x; asum[4 * k4 + 1] += a4.y;
                    asum[4 * k4 + 2] += a4.z; asum[4 * k4 + 3] += a4.w;
                }
            }
        }
    }
    // V[k,c] = sum_p a[k,p] (x[c,p] - cent[k,c])  (netvlad.py:115-124)
#pragma unroll
    for (int k = 0; k < VK; ++k)
        acc[k] = (c_own < C) ? acc[k] - asum[k] * cent[(size_t)k * C + c_own] : 0.0f;
    // intra-normalisation over c for every k (netvlad.py:126), then global L2 (:127-128)
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < VK; ++k) {
        float s = wave_sum_f32(acc[k] * acc[k]);
        if (lane == 0) red[wave * VK + k] = s;
    }
    __syncthreads();
    float gsum = 0.0f;
#pragma unroll
    for (int k = 0; k < VK; ++k) {
        float t = 0.0f;
        for (int w = 0; w < nw; ++w) t += red[w * VK + k];
        float nk = sqrtf(t);
        float sc = 1.0f / fmaxf(nk, 1e-12f);
        acc[k] *= sc;
        float nn = nk * sc;
        gsum += nn * nn;
    }
    float gs = 1.0f / fmaxf(sqrtf(gsum), 1e-12f);
    if (c_own < C) {
        float *o = out + (size_t)blockIdx.x * VK * C;
#pragma unroll
        for (int k = 0; k < VK; ++k) o[(size_t)k * C + c_own] = acc[k] * gs;
    }
}

CSLAM_API int cslam_vlad_aggregate_dev(const float *d_feat, const float *d_assign_w, const float *d_assign_b,
                                       const float *d_centroids, int B, int C, int P, int K,
                                       float *d_out, void *stream) {
    ARG_CHECK(d_feat && d_assign_w && d_centroids && d_out, "NULL argument");
    ARG_CHECK(K == VK, "K must be 64 (reference: num_clusters=64, netvlad.py:176)");
    ARG_CHECK(C >= 1 && C <= 512 && P >= 1 && B >= 0, "need 1 <= C <= 512 (reference encoder_dim = 512, netvlad.py:162)");
    if (B == 0) return CSLAM_OK;
    int threads = (int)round_up64(C > VPCH ? C : VPCH, 64);
    size_t lds = (size_t)(VPCH * VK + VWCH * VK + VPCH + 16 * VK + 16) * 4;
    HIP_TRY(hipFuncSetAttribute((const void *)vlad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(vlad_kernel, dim3(B), dim3(threads), lds, (hipStream_t)stream, d_feat, d_assign_w,
                       d_assign_b, d_centroids, C, P, d_out);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

// ------------------------------------------------------------ CosPlace head ----
__global__ __launch_bounds__(256) void gem_fc_kernel(const float *__restrict__ feat, float pw, float eps,
                                                     const float *__restrict__ W, const float *__restrict__ bias,
                                                     int C, int P, int Dout, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *invn = (float *)smem;          // [P]
    float *g = invn + P;                  // [C]
    float *o = g + C;                     // [Dout]
    float *red = o + Dout;                // [16]
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const float *x = feat + (size_t)blockIdx.x * C * P;
    // L2Norm over channels per pixel (layers.py:32-36)
    for (int p = tid; p < P; p += nt) {
        float s = 0.0f;
        for (int c = 0; c < C; ++c) { float v = x[(size_t)c * P + p]; s += v * v; }
        invn[p] = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    }
    __syncthreads();
    // GeM (layers.py:8-9): avg_pool(clamp(x, min=eps)^p)^(1/p)
    const float ip = 1.0f / pw, invP = 1.0f / (float)P;
    for (int c = tid; c < C; c += nt) {
        float s = 0.0f;
        for (int p = 0; p < P; ++p) s += powf(fmaxf(x[(size_t)c * P + p] * invn[p], eps), pw);
        g[c] = powf(s * invP, ip);
    }
    __syncthreads();
    // Linear (network.py:27): one wave per output feature
    for (int d = wave; d < Dout; d += nw) {
        const float *wr = W + (size_t)d * C;
        float s = 0.0f;
        for (int c = lane; c < C; c += 64) s += wr[c] * g[c];
        s = wave_sum_f32(s);
        if (lane == 0) o[d] = s + (bias ? bias[d] : 0.0f);
    }
    __syncthreads();
    float s = 0.0f;
    for (int d = tid; d < Dout; d += nt) s += o[d] * o[d];
    float nrm = sqrtf(block_sum_f32(s, red));
    float sc = 1.0f / fmaxf(nrm, 1e-12f);
    for (int d = tid; d < Dout; d += nt) out[(size_t)blockIdx.x * Dout + d] = o[d] * sc;
}

CSLAM_API int cslam_gem_fc_head_dev(const float *d_feat, float p, float eps, const float *d_W,
                                    const float *d_b, int B, int C, int P, int Dout,
                                    float *d_out, void *stream) {
    ARG_CHECK(d_feat && d_W && d_out, "NULL argument");
    ARG_CHECK(B >= 0 && C >= 1 && P >= 1 && Dout >= 1, "bad sizes");
    size_t lds = (size_t)(P + C + Dout + 16) * 4;
    ARG_CHECK(lds <= 150 * 1024, "C + P + Dout too large for one workgroup's LDS");
    if (B == 0) return CSLAM_OK;
    HIP_TRY(hipFuncSetAttribute((const void *)gem_fc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(gem_fc_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, d_feat, p, eps, d_W, d_b, C,
                       P, Dout, d_out);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

// ------------------------------------------------------------ image transform ----
// Pillow's antialiased resampling (Resample.c, 8 bits per channel), restated:
//   coefficients: precompute_coeffs() in double, bicubic a = -0.5, support = 2 * max(scale, 1),
//                 normalised per output pixel, then fixed point with 22 fractional bits
//                 (normalize_coeffs_8bpc); horizontal pass first, result clipped to uint8,
//                 then the vertical pass on that uint8 image; both round with + 2^21.
// (Pillow is a third-party dependency of the reference via torchvision.transforms; pinned by
//  tests/golden/heads_g.npz produced with Pillow in the build container.)
#define PREC_BITS 22

static double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

static int precompute_coeffs(int in_size, double in0, double in1, int out_size, std::vector<int> &bounds,
                             std::vector<int> &kk) {
    double scale = (in1 - in0) / out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    bounds.assign((size_t)out_size * 2, 0);
    kk.assign((size_t)out_size * ksize, 0);
    std::vector<double> k((size_t)ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        double center = in0 + (xx + 0.5) * scale, ww = 0.0, ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        int x;
        for (x = 0; x < xmax; ++x) {
            double w = bicubic_filter((x + xmin - center + 0.5) * ss);
            k[x] = w; ww += w;
        }
        for (x = 0; x < xmax; ++x) if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; ++x) k[x] = 0.0;
        for (x = 0; x < ksize; ++x)
            kk[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << PREC_BITS))
                                                  : (int)(0.5 + k[x] * (1 << PREC_BITS));
        bounds[xx * 2] = xmin; bounds[xx * 2 + 1] = xmax;
    }
    return ksize;
}

__device__ __forceinline__ int clip8(int v) {
    v >>= PREC_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: tmp[b][y][xo][ch] for the cropped rows y in [0, crop)
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t *__restrict__ img, int H, int W, int crop,
                                                       int top, int left, int out_w, int ksize,
                                                       const int *__restrict__ bounds, const int *__restrict__ kk,
                                                       uint8_t *__restrict__ tmp) {
    const int b = blockIdx.y, y = blockIdx.x;
    const uint8_t *row = img + ((size_t)b * H + top + y) * W * 3 + (size_t)left * 3;
    for (int e = threadIdx.x; e < out_w * 3; e += blockDim.x) {
        int xo = e / 3, ch = e - xo * 3;
        int xmin = bounds[xo * 2], xn = bounds[xo * 2 + 1];
        const int *k = kk + (size_t)xo * ksize;
        int ss = 1 << (PREC_BITS - 1);
        for (int x = 0; x < xn; ++x) ss += (int)row[(xmin + x) * 3 + ch] * k[x];
        tmp[(((size_t)b * crop + y) * out_w + xo) * 3 + ch] = (uint8_t)clip8(ss);
    }
}

// vertical pass + ToTensor + Normalize: out[b][ch][yo][xo]
__global__ __launch_bounds__(256) void resize_v_kernel(const uint8_t *__restrict__ tmp, int crop, int out_hw,
                                                       int ksize, const int *__restrict__ bounds,
                                                       const int *__restrict__ kk, float m0, float m1, float m2,
                                                       float s0, float s1, float s2, float *__restrict__ out) {
    const int b = blockIdx.y, yo = blockIdx.x;
    const int ymin = bounds[yo * 2], yn = bounds[yo * 2 + 1];
    const int *k = kk + (size_t)yo * ksize;
    for (int e = threadIdx.x; e < out_hw * 3; e += blockDim.x) {
        int xo = e / 3, ch = e - xo * 3;
        int ss = 1 << (PREC_BITS - 1);
        for (int y = 0; y < yn; ++y) ss += (int)tmp[(((size_t)b * crop + ymin + y) * out_hw + xo) * 3 + ch] * k[y];
        float v = (float)clip8(ss) / 255.0f;                       // ToTensor
        float mean = ch == 0 ? m0 : (ch == 1 ? m1 : m2), sd = ch == 0 ? s0 : (ch == 1 ? s1 : s2);
        out[(((size_t)b * 3 + ch) * out_hw + yo) * out_hw + xo] = (v - mean) / sd;   // Normalize
    }
}

struct PreprocCache {
    int device, crop, out_hw, ksize;
    int *d_bounds, *d_kk;
    uint8_t *d_tmp; size_t tmp_bytes;
};
static PreprocCache g_pp = {-1, 0, 0, 0, nullptr, nullptr, nullptr, 0};

CSLAM_API int cslam_preprocess_dev(const uint8_t *d_img, int B, int H, int W, int crop, int out_hw,
                                   const float mean[3], const float std_[3], float *d_out, void *stream) {
    ARG_CHECK(d_img && d_out && mean && std_, "NULL argument");
    ARG_CHECK(B >= 0 && crop >= 1 && out_hw >= 1, "bad sizes");
    ARG_CHECK(H >= crop && W >= crop, "image smaller than the crop (CenterCrop padding not supported)");
    if (B == 0) return CSLAM_OK;
    hipStream_t st = (hipStream_t)stream;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (g_pp.device != dev || g_pp.crop != crop || g_pp.out_hw != out_hw) {
        std::vector<int> bounds, kk;
        int ksize = precompute_coeffs(crop, 0.0, (double)crop, out_hw, bounds, kk);
        if (g_pp.d_bounds) { (void)hipFree(g_pp.d_bounds); (void)hipFree(g_pp.d_kk); }
        HIP_TRY(hipMalloc((void **)&g_pp.d_bounds, bounds.size() * 4));
        HIP_TRY(hipMalloc((void **)&g_pp.d_kk, kk.size() * 4));
        HIP_TRY(hipMemcpy(g_pp.d_bounds, bounds.data(), bounds.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(g_pp.d_kk, kk.data(), kk.size() * 4, hipMemcpyHostToDevice));
        g_pp.device = dev; g_pp.crop = crop; g_pp.out_hw = out_hw; g_pp.ksize = ksize;
    }
    size_t need = (size_t)B * crop * out_hw * 3;
    if (need > g_pp.tmp_bytes) {
        if (g_pp.d_tmp) HIP_TRY(hipFree(g_pp.d_tmp));
        g_pp.d_tmp = nullptr; g_pp.tmp_bytes = 0;
        HIP_TRY(hipMalloc((void **)&g_pp.d_tmp, need));
        g_pp.tmp_bytes = need;
    }
    // torchvision CenterCrop: top = round((H - crop) / 2), left = round((W - crop) / 2)
    const int top = (int)lrint((H - crop) / 2.0), left = (int)lrint((W - crop) / 2.0);
    hipLaunchKernelGGL(resize_h_kernel, dim3(crop, B), dim3(256), 0, st, d_img, H, W, crop, top, left, out_hw,
                       g_pp.ksize, g_pp.d_bounds, g_pp.d_kk, g_pp.d_tmp);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(resize_v_kernel, dim3(out_hw, B), dim3(256), 0, st, g_pp.d_tmp, crop, out_hw, g_pp.ksize,
                       g_pp.d_bounds, g_pp.d_kk, mean[0], mean[1], mean[2], std_[0], std_[1], std_[2], d_out);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}
