// winograd.hip -- Winograd F(2x2, 3x3) transforms for the 3x3 / stride 1 / pad 1 convolutions of the
// extractor backbones (gfx950).  The convolution Y = conv(X, g) + bias becomes
//     V_xi = (B^T d B)_xi  for every 4x4 input tile d            [wino_input_kernel]
//     M_xi = V_xi . U_xi,  U = G g G^T, xi = 0..15               [16 plain fp32 GEMMs: rocBLAS via torch.bmm]
//     Y    = A^T M A + bias (+ ReLU) (+ 2x2 max-pool)            [wino_output_kernel]
// 2.25x fewer multiplications than the direct form the backbone otherwise runs (VGG-16's 512-channel
// layers: cslam/vpr/netvlad.py:163-171 builds `vgg16().features[:-2]`).  Both kernels are pure streaming
// work: activations NHWC, one thread owns 4 consecutive channels of one tile, so every load and store of
// a wave is one contiguous run (>= 1 KiB for C >= 256); each activation is read once from HBM (the 4x tile
// overlap is served by L2) and V / M are written / read exactly once.
//   input  bytes per tile-channel: 16 B read (amortised) + 64 B written;   output: 64 B read + 16 B (4 B pooled) written
#include "common.h"
#include <hip/hip_fp16.h>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

// Block order: plain (workgroup b = tiles b*256/(C/2)...).  Two XCD-aware orders were measured and rejected:
// giving each XCD (b % 8) a contiguous eighth of the tile range, or runs of 64 workgroups, brings FETCH_SIZE down
// from 1.65x to 1.03-1.08x the activation (halo re-reads hit the local L2), but scatters the 36-plane V write
// stream over 8 windows and the kernel gets 5-8 % slower: the extra fetches come from the Infinity Cache, the
// writes are what the kernel is bound by.

// x [B][H][W][C] (NHWC), V [16][T][C] with T = B * (H/2) * (W/2), tile t = (b * H/2 + ti) * W/2 + tj.
__global__ __launch_bounds__(256) void wino_input_kernel(const float *__restrict__ x, int B, int H, int W, int C,
                                                         float *__restrict__ V) {
    const int c4n = C >> 2;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int TH = (H + 1) >> 1, TW = (W + 1) >> 1;      // ragged maps: the last tile row / column is partly outside
    const int64_t T = (int64_t)B * TH * TW;
    if (gid >= T * c4n) return;
    const int c4 = (int)(gid % c4n);
    const int64_t t = gid / c4n;
    const int tj = (int)(t % TW);
    const int ti = (int)((t / TW) % TH);
    const int b = (int)(t / ((int64_t)TW * TH));
    const int h0 = 2 * ti - 1, w0 = 2 * tj - 1;
    f4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int h = h0 + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int w = w0 + j;
            const bool in = (h >= 0) & (h < H) & (w >= 0) & (w < W);
            const f4 *p = (const f4 *)(x + (((int64_t)b * H + (in ? h : 0)) * W + (in ? w : 0)) * C) + c4;
            f4 v = *p;
            d[i][j] = in ? v : (f4)(0.0f);
        }
    }
    f4 r[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {          // B^T d
        r[0][j] = d[0][j] - d[2][j];
        r[1][j] = d[1][j] + d[2][j];
        r[2][j] = d[2][j] - d[1][j];
        r[3][j] = d[1][j] - d[3][j];
    }
    const int64_t plane = T * C;
    float *o = V + t * C + 4 * c4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {          // (.) B
        f4 v0 = r[i][0] - r[i][2], v1 = r[i][1] + r[i][2], v2 = r[i][2] - r[i][1], v3 = r[i][1] - r[i][3];
        __builtin_nontemporal_store(v0, (f4 *)(o + (int64_t)(4 * i + 0) * plane));
        __builtin_nontemporal_store(v1, (f4 *)(o + (int64_t)(4 * i + 1) * plane));
        __builtin_nontemporal_store(v2, (f4 *)(o + (int64_t)(4 * i + 2) * plane));
        __builtin_nontemporal_store(v3, (f4 *)(o + (int64_t)(4 * i + 3) * plane));
    }
}

// M [16][T][C]; y [B][H][W][C] (POOL = false) or [B][H/2][W/2][C] (POOL = true: the 2x2 outputs of a tile are
// exactly one window of the MaxPool2d(2, 2) that follows the layer).  Tiles may hang over the map (odd H / W).
template <bool RELU, bool POOL>
__global__ __launch_bounds__(256) void wino_output_kernel(const float *__restrict__ M, const float *__restrict__ bias,
                                                          const float *__restrict__ res, int B, int H, int W, int C,
                                                          float *__restrict__ y) {
    const int c4n = C >> 2;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int TH = (H + 1) >> 1, TW = (W + 1) >> 1;      // ragged maps: the last tile row / column is partly outside
    const int64_t T = (int64_t)B * TH * TW;
    if (gid >= T * c4n) return;
    const int c4 = (int)(gid % c4n);
    const int64_t t = gid / c4n;
    const int64_t plane = T * C;
    const float *p = M + t * C + 4 * c4;
    f4 m[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) m[i][j] = __builtin_nontemporal_load((const f4 *)(p + (int64_t)(4 * i + j) * plane));
    f4 s[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {          // A^T m
        s[0][j] = m[0][j] + m[1][j] + m[2][j];
        s[1][j] = m[1][j] - m[2][j] - m[3][j];
    }
    const f4 bv = bias ? *((const f4 *)bias + c4) : (f4)(0.0f);
    const int tj = (int)(t % TW);
    const int ti = (int)((t / TW) % TH);
    const int b = (int)(t / ((int64_t)TW * TH));
    f4 o[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {          // (.) A, + bias (+ residual), activation
        o[i][0] = s[i][0] + s[i][1] + s[i][2] + bv;
        o[i][1] = s[i][1] - s[i][2] - s[i][3] + bv;
        if (res && 2 * ti + i < H) {       // ResNet shortcut, same NHWC shape as y (never with POOL)
            o[i][0] += *((const f4 *)(res + (((int64_t)b * H + 2 * ti + i) * W + 2 * tj) * C) + c4);
            if (2 * tj + 1 < W) o[i][1] += *((const f4 *)(res + (((int64_t)b * H + 2 * ti + i) * W + 2 * tj + 1) * C) + c4);
        }
        if (RELU) {
            o[i][0] = __builtin_elementwise_max(o[i][0], (f4)(0.0f));
            o[i][1] = __builtin_elementwise_max(o[i][1], (f4)(0.0f));
        }
    }
    if (POOL) {                                // MaxPool2d(2,2) floors: a window partly outside the map has no output
        const int Ho = H >> 1, Wo = W >> 1;
        f4 v = __builtin_elementwise_max(__builtin_elementwise_max(o[0][0], o[0][1]),
                                         __builtin_elementwise_max(o[1][0], o[1][1]));
        if (ti < Ho && tj < Wo) *((f4 *)(y + (((int64_t)b * Ho + ti) * Wo + tj) * C) + c4) = v;
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (2 * ti + i < H && 2 * tj + j < W)
                    *((f4 *)(y + (((int64_t)b * H + 2 * ti + i) * W + 2 * tj + j) * C) + c4) = o[i][j];
    }
}

// Epilogue of the layers that stay on the direct convolution (3 / 64 input channels): bias + ReLU
// (+ MaxPool2d(2,2)) in ONE pass over the NHWC activation instead of torch's three (add, relu, pool).
// x [B][H][W][C]; y = x in place when POOL is false, else [B][H/2][W/2][C].
template <bool RELU, bool POOL>
__global__ __launch_bounds__(256) void bias_act_pool_kernel(const float *__restrict__ x, const float *__restrict__ bias,
                                                            int64_t n_out, int H, int W, int C, float *__restrict__ y) {
    const int c4n = C >> 2;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n_out) return;
    const int c4 = (int)(gid % c4n);
    const f4 bv = bias ? *((const f4 *)bias + c4) : (f4)(0.0f);
    if (!POOL) {
        f4 v = *((const f4 *)x + gid) + bv;
        if (RELU) v = __builtin_elementwise_max(v, (f4)(0.0f));
        *((f4 *)y + gid) = v;
        return;
    }
    const int Wo = W >> 1, Ho = H >> 1;
    const int64_t pix = gid / c4n;
    const int wo = (int)(pix % Wo);
    const int ho = (int)((pix / Wo) % Ho);
    const int64_t b = pix / ((int64_t)Wo * Ho);
    const f4 *p = (const f4 *)(x + ((b * H + 2 * ho) * W + 2 * wo) * C) + c4;
    const int64_t rowstride = (int64_t)W * c4n;
    f4 v = __builtin_elementwise_max(__builtin_elementwise_max(p[0], p[c4n]),
                                     __builtin_elementwise_max(p[rowstride], p[rowstride + c4n])) + bv;
    if (RELU) v = __builtin_elementwise_max(v, (f4)(0.0f));      // max commutes with + bias and with ReLU
    *((f4 *)y + gid) = v;
}

CSLAM_API int cslam_bias_act_pool_dev(const float *d_x, const float *d_bias, int B, int H, int W, int C, int relu,
                                      int pool, float *d_y, void *stream) {
    PTR_DEVICE(d_x);
    ARG_CHECK(d_x && d_y, "NULL argument");
    ARG_CHECK(B >= 1 && H >= 1 && W >= 1 && C >= 4 && (C % 4) == 0, "C must be a multiple of 4");
    ARG_CHECK(!pool || ((H % 2) == 0 && (W % 2) == 0), "pooling needs even H and W");
    ARG_CHECK(pool || d_x == d_y || d_y + (int64_t)B * H * W * C <= d_x || d_x + (int64_t)B * H * W * C <= d_y,
              "y must be x itself or not overlap it");
    const int64_t n = pool ? (int64_t)B * (H / 2) * (W / 2) * (C / 4) : (int64_t)B * H * W * (C / 4);
    ARG_CHECK(ceil_div64(n, 256) < (1LL << 31), "too many elements for one launch");
    dim3 grid((unsigned)ceil_div64(n, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (relu && pool) hipLaunchKernelGGL((bias_act_pool_kernel<true, true>), grid, block, 0, st, d_x, d_bias, n, H, W, C, d_y);
    else if (relu) hipLaunchKernelGGL((bias_act_pool_kernel<true, false>), grid, block, 0, st, d_x, d_bias, n, H, W, C, d_y);
    else if (pool) hipLaunchKernelGGL((bias_act_pool_kernel<false, true>), grid, block, 0, st, d_x, d_bias, n, H, W, C, d_y);
    else hipLaunchKernelGGL((bias_act_pool_kernel<false, false>), grid, block, 0, st, d_x, d_bias, n, H, W, C, d_y);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

// First layer (3 input channels, e.g. VGG-16 conv1_1): direct 3x3 convolution + bias + ReLU written once,
// NCHW planar input (what the preprocessing kernel produces) -> NHWC output.  A workgroup owns 64
// consecutive pixels; lane = pixel (the 27 tap loads of a wave are contiguous runs served by L1), wave w
// computes output channels [16w, 16w+16) (+64, ...) with wave-uniform weights read through the scalar cache
// (v_fmac with an SGPR operand).  The 64 x Cout tile is transposed through LDS so that every store
// instruction writes whole consecutive NHWC pixels (1 KiB runs).  HBM-bound on the output write
// (Cout*4 B per pixel); 2*27*Cout flop per pixel on the VALU.
#define C3_CG 16
__global__ __launch_bounds__(256) void conv3x3_c3_kernel(const float *__restrict__ x, const float *__restrict__ wt,
                                                         const float *__restrict__ bias, int B, int H, int W, int Cout,
                                                         int relu, float *__restrict__ y) {
    extern __shared__ float s_t[];                      // [64][Cout + 4]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t npix = (int64_t)B * H * W;
    const int64_t pix0 = (int64_t)blockIdx.x * 64;
    const int64_t pix = pix0 + lane;
    const bool live = pix < npix;
    const int64_t pc = live ? pix : npix - 1;
    const int w = (int)(pc % W);
    const int h = (int)((pc / W) % H);
    const int64_t b = pc / ((int64_t)W * H);
    const float *xb = x + b * 3 * (int64_t)H * W;
    float v[27];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hh = h + kh - 1;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ww = w + kw - 1;
                const bool in = (hh >= 0) & (hh < H) & (ww >= 0) & (ww < W);
                v[ci * 9 + kh * 3 + kw] = in ? xb[((int64_t)ci * H + hh) * W + ww] : 0.0f;
            }
        }
    const int P = Cout + 4;
    for (int co0 = wave * C3_CG; co0 < Cout; co0 += 4 * C3_CG) {
        float acc[C3_CG];
#pragma unroll
        for (int j = 0; j < C3_CG; ++j) acc[j] = bias ? bias[co0 + j] : 0.0f;
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const float *wk = wt + k * Cout + co0;      // wave-uniform address
#pragma unroll
            for (int j = 0; j < C3_CG; ++j) acc[j] = fmaf(wk[j], v[k], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < C3_CG; j += 4) {
            f4 o = {acc[j], acc[j + 1], acc[j + 2], acc[j + 3]};
            if (relu) o = __builtin_elementwise_max(o, (f4)(0.0f));
            *((f4 *)(s_t + lane * P + co0 + j)) = o;
        }
    }
    __syncthreads();
    const int c4n = Cout >> 2;
    for (int e = threadIdx.x; e < 64 * c4n; e += 256) {
        const int p = e / c4n, c4 = e - p * c4n;
        if (pix0 + p < npix) *((f4 *)(y + (pix0 + p) * Cout) + c4) = *((const f4 *)(s_t + p * P) + c4);
    }
}

// Cout = 64 form of the same layer (VGG-16 conv1_1).  The generic kernel above spends most of its issue slots on
// per-pixel index arithmetic (64-bit div / mod, 27 predicated tap loads repeated by all four waves: ~590 of its 808
// VALU instructions per wave) rather than on its 216 packed FMAs.  Here a workgroup owns a 32-wide x 2-high pixel
// tile addressed by a 3-D grid (no division), the 3 x 4 x 34 input patch with its zero border is staged ONCE in
// LDS, every lane reads its 27 taps from there, wave w computes channels [16w, 16w+16), and the 64 x 64 tile goes
// out through LDS as whole NHWC pixels (8 KiB contiguous per tile row).
#define C3T_W 32
#define C3T_H 2
#define C3T_PW (C3T_W + 2)
__global__ __launch_bounds__(256) void conv3x3_c3_tile64_kernel(const float *__restrict__ x, const float *__restrict__ wt,
                                                                const float *__restrict__ bias, int H, int W, int relu,
                                                                float *__restrict__ y, unsigned *__restrict__ amax_out) {
    constexpr int COUT = 64, P = COUT + 4;
    __shared__ float s_in[3 * (C3T_H + 2) * C3T_PW];            // [ci][row][col], zero border included
    __shared__ __attribute__((aligned(16))) float s_t[C3T_W * C3T_H * P];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w0 = blockIdx.x * C3T_W, h0 = blockIdx.y * C3T_H;
    const int64_t b = blockIdx.z;
    const float *xb = x + b * 3 * (int64_t)H * W;
    for (int e = threadIdx.x; e < 3 * (C3T_H + 2) * C3T_PW; e += 256) {
        const int ci = e / ((C3T_H + 2) * C3T_PW), r = (e / C3T_PW) % (C3T_H + 2), c = e % C3T_PW;   // constant divisors
        const int hh = h0 + r - 1, ww = w0 + c - 1;
        const bool in = (hh >= 0) & (hh < H) & (ww >= 0) & (ww < W);
        s_in[e] = in ? xb[((int64_t)ci * H + hh) * W + ww] : 0.0f;
    }
    __syncthreads();
    const int pr = lane >> 5, pc = lane & 31;                   // this lane's pixel inside the tile
    float v[27];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
                v[ci * 9 + kh * 3 + kw] = s_in[(ci * (C3T_H + 2) + pr + kh) * C3T_PW + pc + kw];
    const int co0 = wave * C3_CG;
    float acc[C3_CG];
#pragma unroll
    for (int j = 0; j < C3_CG; ++j) acc[j] = bias ? bias[co0 + j] : 0.0f;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        const float *wk = wt + k * COUT + co0;                  // wave-uniform address: scalar loads
#pragma unroll
        for (int j = 0; j < C3_CG; ++j) acc[j] = fmaf(wk[j], v[k], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < C3_CG; j += 4) {
        f4 o = {acc[j], acc[j + 1], acc[j + 2], acc[j + 3]};
        if (relu) o = __builtin_elementwise_max(o, (f4)(0.0f));
        *((f4 *)(s_t + lane * P + co0 + j)) = o;
    }
    __syncthreads();
    float vmax = 0.0f;
#pragma unroll
    for (int i = 0; i < (C3T_W * C3T_H * (COUT / 4)) / 256; ++i) {
        const int e = threadIdx.x + 256 * i;
        const int p = e >> 4, c4 = e & 15;                      // COUT / 4 = 16 float4 per pixel
        const int hh = h0 + (p >> 5), ww = w0 + (p & 31);
        if (hh < H && ww < W) {
            const f4 o = *((const f4 *)(s_t + p * P) + c4);
            *((f4 *)(y + ((b * H + hh) * (int64_t)W + ww) * COUT) + c4) = o;
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
    }
    if (amax_out) {
        // max |y| for the scale of a split-fp16 layer behind this one: wave maximum, then one racy look at the slot per wave
        // and a global atomic only if it would raise it (the slot settles after the first few workgroups; an LDS stage
        // with a barrier per workgroup cost 0.15 ms of 0.76 on the 256-frame layer)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
        if (lane == 0 && __float_as_uint(vmax) > *(volatile unsigned *)amax_out) atomicMax(amax_out, __float_as_uint(vmax));
    }
}

CSLAM_API int cslam_conv3x3_c3_dev(const float *d_x, const float *d_wt, const float *d_bias, int B, int H, int W,
                                   int Cout, int relu, float *d_y, void *stream) {
    return cslam_conv3x3_c3_amax_dev(d_x, d_wt, d_bias, B, H, W, Cout, relu, d_y, nullptr, stream);
}

CSLAM_API int cslam_conv3x3_c3_amax_dev(const float *d_x, const float *d_wt, const float *d_bias, int B, int H, int W,
                                        int Cout, int relu, float *d_y, unsigned *d_amax_out, void *stream) {
    PTR_DEVICE(d_x);
    ARG_CHECK(d_x && d_wt && d_y, "NULL argument");
    ARG_CHECK(B >= 1 && H >= 1 && W >= 1, "empty input");
    ARG_CHECK(Cout >= 16 && (Cout % 16) == 0 && Cout <= 512, "Cout must be a multiple of 16, at most 512");
    const int64_t n = (int64_t)B * H * W;
    ARG_CHECK(ceil_div64(n, 256) < (1LL << 31), "too many pixels for one launch");
    ARG_CHECK(ceil_div64(n, 64) < (1LL << 31), "too many pixels for one launch");
    if (Cout == 64 && B <= 65535 && ceil_div64(H, C3T_H) <= 65535) {
        hipLaunchKernelGGL(conv3x3_c3_tile64_kernel, dim3((unsigned)ceil_div64(W, C3T_W), (unsigned)ceil_div64(H, C3T_H), (unsigned)B),
                           dim3(256), 0, (hipStream_t)stream, d_x, d_wt, d_bias, H, W, relu, d_y, d_amax_out);
        HIP_TRY(hipGetLastError());
        return CSLAM_OK;
    }
    ARG_CHECK(!d_amax_out, "max |y| is only delivered by the 64-channel first-layer kernel");
    const size_t lds = (size_t)64 * (Cout + 4) * 4;
    ARG_CHECK(lds <= 160 * 1024, "Cout too large for the LDS tile");
    HIP_TRY(hipFuncSetAttribute((const void *)conv3x3_c3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(conv3x3_c3_kernel, dim3((unsigned)ceil_div64(n, 64)), dim3(256), lds, (hipStream_t)stream, d_x,
                       d_wt, d_bias, B, H, W, Cout, relu, d_y);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

CSLAM_API int cslam_wino_input_dev(const float *d_x, int B, int H, int W, int C, float *d_V, void *stream) {
    PTR_DEVICE(d_x);
    ARG_CHECK(d_x && d_V, "NULL argument");
    ARG_CHECK(B >= 1 && H >= 1 && W >= 1, "empty map");
    ARG_CHECK(C >= 4 && (C % 4) == 0, "C must be a multiple of 4");
    const int64_t n = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
    ARG_CHECK(ceil_div64(n, 256) < (1LL << 31), "too many tiles for one launch");
    hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream, d_x, B,
                       H, W, C, d_V);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

CSLAM_API int cslam_wino_output_dev(const float *d_M, const float *d_bias, const float *d_res, int B, int H, int W,
                                    int C, int relu, int pool, float *d_y, void *stream) {
    PTR_DEVICE(d_M);
    ARG_CHECK(d_M && d_y, "NULL argument");
    ARG_CHECK(!(d_res && pool), "a residual input cannot be combined with pooling");
    ARG_CHECK(B >= 1 && H >= 1 && W >= 1, "empty map");
    ARG_CHECK(C >= 4 && (C % 4) == 0, "C must be a multiple of 4");
    const int64_t n = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
    ARG_CHECK(ceil_div64(n, 256) < (1LL << 31), "too many tiles for one launch");
    dim3 grid((unsigned)ceil_div64(n, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (relu && pool) hipLaunchKernelGGL((wino_output_kernel<true, true>), grid, block, 0, st, d_M, d_bias, d_res, B, H, W, C, d_y);
    else if (relu) hipLaunchKernelGGL((wino_output_kernel<true, false>), grid, block, 0, st, d_M, d_bias, d_res, B, H, W, C, d_y);
    else if (pool) hipLaunchKernelGGL((wino_output_kernel<false, true>), grid, block, 0, st, d_M, d_bias, d_res, B, H, W, C, d_y);
    else hipLaunchKernelGGL((wino_output_kernel<false, false>), grid, block, 0, st, d_M, d_bias, d_res, B, H, W, C, d_y);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

// ---------------------------------------------------------------- F(4x4, 3x3) ----
// Same scheme with 6x6 input tiles and 4x4 output tiles: 36 GEMMs, 4x fewer multiplications than the direct
// form (F(2x2,3x3): 2.25x) and V / M only 2.25x the activation size (F(2x2,3x3): 4x).  The transform
// constants (up to 8) cost about one decimal digit: ~1e-5 relative on a layer output against ~1.5e-6
// for F(2x2,3x3) (measured, tests/test_heads_gpu.py) -- opt-in through `frontend.backbone_conv`.
// One thread owns 2 consecutive channels of one tile (72 live values).
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void wino4_bt(f2 &d0, f2 &d1, f2 &d2, f2 &d3, f2 &d4, f2 &d5) {
    const f2 r0 = 4.0f * d0 - 5.0f * d2 + d4;
    const f2 r1 = -4.0f * (d1 + d2) + d3 + d4;
    const f2 r2 = 4.0f * (d1 - d2) - d3 + d4;
    const f2 r3 = 2.0f * (d3 - d1) - d2 + d4;
    const f2 r4 = 2.0f * (d1 - d3) - d2 + d4;
    const f2 r5 = 4.0f * d1 - 5.0f * d3 + d5;
    d0 = r0; d1 = r1; d2 = r2; d3 = r3; d4 = r4; d5 = r5;
}

__global__ __launch_bounds__(256) void wino4_input_kernel(const float *__restrict__ x, int B, int H, int W, int C,
                                                          float *__restrict__ V) {
    const int c2n = C >> 1;
    // workgroups go to the 8 XCDs round-robin; give each XCD a contiguous run of tiles so that the 2 of 6 rows / columns
    // a tile shares with each neighbour are found in that XCD's own L2 (gridDim.x is a multiple of 8)
    const int64_t bid = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int64_t gid = bid * 256 + threadIdx.x;
    const int TH = (H + 3) >> 2, TW = (W + 3) >> 2;      // ragged maps: the last tile row / column is partly outside
    const int64_t T = (int64_t)B * TH * TW;
    if (gid >= T * c2n) return;
    const int c2 = (int)(gid % c2n);
    const int64_t t = gid / c2n;
    const int tj = (int)(t % TW);
    const int ti = (int)((t / TW) % TH);
    const int b = (int)(t / ((int64_t)TW * TH));
    const int h0 = 4 * ti - 1, w0 = 4 * tj - 1;
    f2 d[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int h = h0 + i;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int w = w0 + j;
            const bool in = (h >= 0) & (h < H) & (w >= 0) & (w < W);
            const f2 v = *((const f2 *)(x + (((int64_t)b * H + (in ? h : 0)) * W + (in ? w : 0)) * C) + c2);
            d[i][j] = in ? v : (f2)(0.0f);
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) wino4_bt(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);
#pragma unroll
    for (int i = 0; i < 6; ++i) wino4_bt(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5]);
    const int64_t plane = T * C;
    float *o = V + t * C + 2 * c2;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) __builtin_nontemporal_store(d[i][j], (f2 *)(o + (int64_t)(6 * i + j) * plane));
}

// ---- split-fp16 form of the 36 GEMMs (opt-in, vpr/winograd.py `split16`) ----
// A float times a power of two splits exactly into an fp16 pair hi + lo (11 + 11 significant bits); fp16 x fp16 products
// are exact in the MFMA's fp32 accumulator, so  V U = vh uh + vl uh + vh ul  (the dropped vl ul is 2^-22 of the product)
// is an fp32-grade product at the fp16 MFMA rate: one GEMM with K' = 3 Cin over  A' = [vh | vl | vh],  B' = [uh ; uh ; ul].
// The scale keeps max |V| (<= 100 max |x| for B^T d B) below 2^15: s = 2^floor(log2(2^15 / (100 amax))).
__device__ __forceinline__ float wino_h3_scale(unsigned amax_bits) {
    const float a = fminf(fmaxf(__uint_as_float(amax_bits), 1e-30f), 1e30f);
    int e;
    (void)frexpf(327.68f / a, &e);                        // r = m 2^e, m in [0.5, 1): floor(log2 r) = e - 1
    return ldexpf(1.0f, e - 1);
}

// max |x| as float bits in *slot (non-negative floats order like their bit patterns); *slot must be 0 before the launch
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ x, int64_t n4, unsigned *__restrict__ slot) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f4 v = __builtin_nontemporal_load((const f4 *)x + i);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(slot, __float_as_uint(m));
}

__global__ __launch_bounds__(256) void wino4_input_h3_kernel(const float *__restrict__ x, int B, int H, int W, int C,
                                                             const unsigned *__restrict__ amax, __half *__restrict__ V3) {
    const int c2n = C >> 1;
    const int64_t bid = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // as wino4_input_kernel
    const int64_t gid = bid * 256 + threadIdx.x;
    const int TH = (H + 3) >> 2, TW = (W + 3) >> 2;
    const int64_t T = (int64_t)B * TH * TW;
    if (gid >= T * c2n) return;
    const int c2 = (int)(gid % c2n);
    const int64_t t = gid / c2n;
    const int tj = (int)(t % TW);
    const int ti = (int)((t / TW) % TH);
    const int b = (int)(t / ((int64_t)TW * TH));
    const int h0 = 4 * ti - 1, w0 = 4 * tj - 1;
    const float sc = wino_h3_scale(*amax);
    f2 d[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int h = h0 + i;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int w = w0 + j;
            const bool in = (h >= 0) & (h < H) & (w >= 0) & (w < W);
            const f2 v = *((const f2 *)(x + (((int64_t)b * H + (in ? h : 0)) * W + (in ? w : 0)) * C) + c2);
            d[i][j] = in ? v * sc : (f2)(0.0f);
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) wino4_bt(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);
#pragma unroll
    for (int i = 0; i < 6; ++i) wino4_bt(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5]);
    const int64_t plane = T * 3 * C;
    __half *o = V3 + t * 3 * C + 2 * c2;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const __half2 hi = __floats2half2_rn(d[i][j].x, d[i][j].y);
            const float2 hf = __half22float2(hi);
            const __half2 lo = __floats2half2_rn(d[i][j].x - hf.x, d[i][j].y - hf.y);
            __half *q = o + (int64_t)(6 * i + j) * plane;
            *(__half2 *)q = hi;
            *(__half2 *)(q + C) = lo;
            *(__half2 *)(q + 2 * C) = hi;
        }
}

typedef float wf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wino4_bt4(wf4 &d0, wf4 &d1, wf4 &d2, wf4 &d3, wf4 &d4, wf4 &d5) {   // wino4_bt on 4 channels
    const wf4 r0 = 4.0f * d0 - 5.0f * d2 + d4;
    const wf4 r1 = -4.0f * (d1 + d2) + d3 + d4;
    const wf4 r2 = 4.0f * (d1 - d2) - d3 + d4;
    const wf4 r3 = 2.0f * (d3 - d1) - d2 + d4;
    const wf4 r4 = 2.0f * (d1 - d3) - d2 + d4;
    const wf4 r5 = 4.0f * d1 - 5.0f * d3 + d5;
    d0 = r0; d1 = r1; d2 = r2; d3 = r3; d4 = r4; d5 = r5;
}

// wino4_input_h3_kernel with 4 consecutive channels per thread: 16-byte loads, 8-byte fp16 stores (C a multiple of 4)
__global__ __launch_bounds__(256) void wino4_input_h3x4_kernel(const float *__restrict__ x, int B, int H, int W, int C,
                                                               const unsigned *__restrict__ amax, __half *__restrict__ V3) {
    const int c4n = C >> 2;
    const int64_t bid = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int64_t gid = bid * 256 + threadIdx.x;
    const int TH = (H + 3) >> 2, TW = (W + 3) >> 2;
    const int64_t T = (int64_t)B * TH * TW;
    if (gid >= T * c4n) return;
    const int c4 = (int)(gid % c4n);
    const int64_t t = gid / c4n;
    const int tj = (int)(t % TW);
    const int ti = (int)((t / TW) % TH);
    const int b = (int)(t / ((int64_t)TW * TH));
    const int h0 = 4 * ti - 1, w0 = 4 * tj - 1;
    const float sc = wino_h3_scale(*amax);
    wf4 d[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int h = h0 + i;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int w = w0 + j;
            const bool in = (h >= 0) & (h < H) & (w >= 0) & (w < W);
            const wf4 v = *((const wf4 *)(x + (((int64_t)b * H + (in ? h : 0)) * W + (in ? w : 0)) * C) + c4);
            d[i][j] = in ? v * sc : (wf4)(0.0f);
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) wino4_bt4(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);
#pragma unroll
    for (int i = 0; i < 6; ++i) wino4_bt4(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5]);
    const int64_t plane = T * 3 * C;
    __half *o = V3 + t * 3 * C + 4 * c4;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const wf4 v = d[i][j];
            const __half2 h0v = __floats2half2_rn(v.x, v.y), h1v = __floats2half2_rn(v.z, v.w);
            const float2 f0 = __half22float2(h0v), f1 = __half22float2(h1v);
            const __half2 l0v = __floats2half2_rn(v.x - f0.x, v.y - f0.y), l1v = __floats2half2_rn(v.z - f1.x, v.w - f1.y);
            uint2 hi, lo;
            hi.x = *(const unsigned *)&h0v; hi.y = *(const unsigned *)&h1v;
            lo.x = *(const unsigned *)&l0v; lo.y = *(const unsigned *)&l1v;
            __half *q = o + (int64_t)(6 * i + j) * plane;
            *(uint2 *)q = hi;
            *(uint2 *)(q + C) = lo;
            *(uint2 *)(q + 2 * C) = hi;
        }
}

__device__ __forceinline__ void wino4_at(const f2 m0, const f2 m1, const f2 m2, const f2 m3, const f2 m4, const f2 m5,
                                         f2 &s0, f2 &s1, f2 &s2, f2 &s3) {
    const f2 a = m1 + m2, bq = m1 - m2, c = m3 + m4, e = m3 - m4;
    s0 = m0 + a + c;
    s1 = bq + 2.0f * e;
    s2 = a + 4.0f * c;
    s3 = bq + 8.0f * e + m5;
}

template <bool RELU, bool POOL>
__global__ __launch_bounds__(256) void wino4_output_kernel(const float *__restrict__ M, const float *__restrict__ bias,
                                                           const float *__restrict__ res, int B, int H, int W, int C,
                                                           float *__restrict__ y, const unsigned *__restrict__ amax,
                                                           float inv_su, unsigned *__restrict__ amax_out) {
    __shared__ unsigned wg_amax;
    if (amax_out) {
        if (threadIdx.x == 0) wg_amax = 0u;
        __syncthreads();
    }
    const int c2n = C >> 1;
    const int64_t gid0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int TH = (H + 3) >> 2, TW = (W + 3) >> 2;      // ragged maps: the last tile row / column is partly outside
    const int64_t T = (int64_t)B * TH * TW;
    // threads past the range (last workgroup only) stay until both barriers: they compute on element 0 and neither
    // contribute to the maximum nor store
    const bool live = gid0 < T * c2n;
    const int64_t gid = live ? gid0 : 0;
    const int c2 = (int)(gid % c2n);
    const int64_t t = gid / c2n;
    const int64_t plane = T * C;
    const float *p = M + t * C + 2 * c2;
    // split-fp16 GEMM (wino4_input_h3_kernel): M arrives times sV sU, both powers of two -> the rescale is exact
    const float inv = amax ? inv_su / wino_h3_scale(*amax) : 1.0f;
    f2 s[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        f2 m[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) m[i] = __builtin_nontemporal_load((const f2 *)(p + (int64_t)(6 * i + j) * plane));
        wino4_at(m[0], m[1], m[2], m[3], m[4], m[5], s[0][j], s[1][j], s[2][j], s[3][j]);
    }
    const f2 bv = bias ? *((const f2 *)bias + c2) : (f2)(0.0f);
    const int tj = (int)(t % TW);
    const int ti = (int)((t / TW) % TH);
    const int b = (int)(t / ((int64_t)TW * TH));
    f2 o[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        wino4_at(s[i][0], s[i][1], s[i][2], s[i][3], s[i][4], s[i][5], o[i][0], o[i][1], o[i][2], o[i][3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[i][j] = amax ? o[i][j] * inv + bv : o[i][j] + bv;
            if (res && 4 * ti + i < H && 4 * tj + j < W)
                o[i][j] += *((const f2 *)(res + (((int64_t)b * H + 4 * ti + i) * W + 4 * tj + j) * C) + c2);
            if (RELU) o[i][j] = __builtin_elementwise_max(o[i][j], (f2)(0.0f));
        }
    }
    if (amax_out) {
        // upper bound of max |y| for the next layer's split-fp16 scale (pre-pool values bound the pooled ones): LDS maximum
        // per workgroup, then one global atomic per workgroup and only if it would raise the slot
        float m = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (live && 4 * ti + i < H && 4 * tj + j < W) m = fmaxf(m, fmaxf(fabsf(o[i][j].x), fabsf(o[i][j].y)));
        atomicMax(&wg_amax, __float_as_uint(m));
        __syncthreads();
        if (threadIdx.x == 0 && wg_amax > *(volatile unsigned *)amax_out) atomicMax(amax_out, wg_amax);
    }
    if (POOL) {
        const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f2 v = __builtin_elementwise_max(__builtin_elementwise_max(o[2 * i][2 * j], o[2 * i][2 * j + 1]),
                                                 __builtin_elementwise_max(o[2 * i + 1][2 * j], o[2 * i + 1][2 * j + 1]));
                if (live && 2 * ti + i < Ho && 2 * tj + j < Wo)
                    *((f2 *)(y + (((int64_t)b * Ho + 2 * ti + i) * Wo + 2 * tj + j) * C) + c2) = v;
            }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (live && 4 * ti + i < H && 4 * tj + j < W)
                    *((f2 *)(y + (((int64_t)b * H + 4 * ti + i) * W + 4 * tj + j) * C) + c2) = o[i][j];
    }
}

// The second half of the output transform for the Z form of the products (wino_zgemm_h2_kernel, wino_gemm.hip):
// Z [24][T][C], plane 4 i + q = sum_j M[6 i + j] A^T[q][j]  ->  Y[p][q] = sum_i A^T[p][i] Z_i[q], then exactly the epilogue of
// wino4_output_kernel (rescale, bias, ReLU, max |y|, MaxPool2d).  24 loads per tile and channel pair instead of 36.
template <bool RELU, bool POOL>
__global__ __launch_bounds__(256) void wino4_output_z_kernel(const float *__restrict__ Zp, const float *__restrict__ bias,
                                                             int B, int H, int W, int C, float *__restrict__ y,
                                                             const unsigned *__restrict__ amax, float inv_su,
                                                             unsigned *__restrict__ amax_out) {
    __shared__ unsigned wg_amax;
    if (amax_out) {
        if (threadIdx.x == 0) wg_amax = 0u;
        __syncthreads();
    }
    const int c2n = C >> 1;
    const int64_t gid0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int TH = (H + 3) >> 2, TW = (W + 3) >> 2;
    const int64_t T = (int64_t)B * TH * TW;
    const bool live = gid0 < T * c2n;
    const int64_t gid = live ? gid0 : 0;
    const int c2 = (int)(gid % c2n);
    const int64_t t = gid / c2n;
    const int64_t plane = T * C;
    const float *p = Zp + t * C + 2 * c2;
    const float inv = amax ? inv_su / wino_h3_scale(*amax) : 1.0f;
    const f2 bv = bias ? *((const f2 *)bias + c2) : (f2)(0.0f);
    const int tj = (int)(t % TW);
    const int ti = (int)((t / TW) % TH);
    const int b = (int)(t / ((int64_t)TW * TH));
    f2 o[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f2 m[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) m[i] = __builtin_nontemporal_load((const f2 *)(p + (int64_t)(4 * i + q) * plane));
        wino4_at(m[0], m[1], m[2], m[3], m[4], m[5], o[0][q], o[1][q], o[2][q], o[3][q]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[i][j] = amax ? o[i][j] * inv + bv : o[i][j] + bv;
            if (RELU) o[i][j] = __builtin_elementwise_max(o[i][j], (f2)(0.0f));
        }
    if (amax_out) {
        float m = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (live && 4 * ti + i < H && 4 * tj + j < W) m = fmaxf(m, fmaxf(fabsf(o[i][j].x), fabsf(o[i][j].y)));
        atomicMax(&wg_amax, __float_as_uint(m));
        __syncthreads();
        if (threadIdx.x == 0 && wg_amax > *(volatile unsigned *)amax_out) atomicMax(amax_out, wg_amax);
    }
    if (POOL) {
        const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f2 v = __builtin_elementwise_max(__builtin_elementwise_max(o[2 * i][2 * j], o[2 * i][2 * j + 1]),
                                                 __builtin_elementwise_max(o[2 * i + 1][2 * j], o[2 * i + 1][2 * j + 1]));
                if (live && 2 * ti + i < Ho && 2 * tj + j < Wo)
                    *((f2 *)(y + (((int64_t)b * Ho + 2 * ti + i) * Wo + 2 * tj + j) * C) + c2) = v;
            }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (live && 4 * ti + i < H && 4 * tj + j < W)
                    *((f2 *)(y + (((int64_t)b * H + 4 * ti + i) * W + 4 * tj + j) * C) + c2) = o[i][j];
    }
}

/* Output transform of the Z form (cslam_wino_zgemm_h2_dev): arguments as cslam_wino4_output_scaled_dev without the residual. */
CSLAM_API int cslam_wino4_output_z_dev(const float *d_Z, const float *d_bias, int B, int H, int W, int C, int relu, int pool,
                                       const unsigned *d_amax, float inv_su, unsigned *d_amax_out, float *d_y, void *stream) {
    PTR_DEVICE(d_Z);
    ARG_CHECK(d_Z && d_y, "NULL argument");
    ARG_CHECK(B >= 1 && H >= 1 && W >= 1, "empty map");
    ARG_CHECK(C >= 2 && (C % 2) == 0, "C must be even");
    const int64_t n = (int64_t)B * ((H + 3) / 4) * ((W + 3) / 4) * (C / 2);
    ARG_CHECK(ceil_div64(n, 256) < (1LL << 31), "too many tiles for one launch");
    dim3 grid((unsigned)ceil_div64(n, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (relu && pool) hipLaunchKernelGGL((wino4_output_z_kernel<true, true>), grid, block, 0, st, d_Z, d_bias, B, H, W, C, d_y, d_amax, inv_su, d_amax_out);
    else if (relu) hipLaunchKernelGGL((wino4_output_z_kernel<true, false>), grid, block, 0, st, d_Z, d_bias, B, H, W, C, d_y, d_amax, inv_su, d_amax_out);
    else if (pool) hipLaunchKernelGGL((wino4_output_z_kernel<false, true>), grid, block, 0, st, d_Z, d_bias, B, H, W, C, d_y, d_amax, inv_su, d_amax_out);
    else hipLaunchKernelGGL((wino4_output_z_kernel<false, false>), grid, block, 0, st, d_Z, d_bias, B, H, W, C, d_y, d_amax, inv_su, d_amax_out);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

// Measured and rejected: evaluating the first layer (3 -> 64 + ReLU) inside this layer's input transform for
// conv1_2, so that the 12.8 MB-per-frame activation between them never touches HBM.  The fused kernel (6x6 window
// of first-layer outputs recomputed per tile from an 8x8x3 LDS patch, 972 packed FMAs per thread, 256 VGPRs) ran
// 3.9 ms per 256 frames against 1.2 ms (conv3x3_c3_kernel) + 2.0 ms (this kernel's conv1_2 launch) separately.

CSLAM_API int cslam_wino4_input_dev(const float *d_x, int B, int H, int W, int C, float *d_V, void *stream) {
    PTR_DEVICE(d_x);
    ARG_CHECK(d_x && d_V, "NULL argument");
    ARG_CHECK(B >= 1 && H >= 1 && W >= 1, "empty map");
    ARG_CHECK(C >= 2 && (C % 2) == 0, "C must be even");
    const int64_t n = (int64_t)B * ((H + 3) / 4) * ((W + 3) / 4) * (C / 2);
    ARG_CHECK(ceil_div64(n, 256) < (1LL << 31), "too many tiles for one launch");
    hipLaunchKernelGGL(wino4_input_kernel, dim3((unsigned)round_up64(ceil_div64(n, 256), 8)), dim3(256), 0,
                       (hipStream_t)stream, d_x, B, H, W, C, d_V);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

static int wino4_output_launch(const float *d_M, const float *d_bias, const float *d_res, int B, int H, int W, int C,
                               int relu, int pool, float *d_y, const unsigned *d_amax, float inv_su, unsigned *d_amax_out,
                               void *stream) {
    ARG_CHECK(d_M && d_y, "NULL argument");
    ARG_CHECK(!(d_res && pool), "a residual input cannot be combined with pooling");
    ARG_CHECK(B >= 1 && H >= 1 && W >= 1, "empty map");
    ARG_CHECK(C >= 2 && (C % 2) == 0, "C must be even");
    const int64_t n = (int64_t)B * ((H + 3) / 4) * ((W + 3) / 4) * (C / 2);
    ARG_CHECK(ceil_div64(n, 256) < (1LL << 31), "too many tiles for one launch");
    dim3 grid((unsigned)ceil_div64(n, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (relu && pool) hipLaunchKernelGGL((wino4_output_kernel<true, true>), grid, block, 0, st, d_M, d_bias, d_res, B, H, W, C, d_y, d_amax, inv_su, d_amax_out);
    else if (relu) hipLaunchKernelGGL((wino4_output_kernel<true, false>), grid, block, 0, st, d_M, d_bias, d_res, B, H, W, C, d_y, d_amax, inv_su, d_amax_out);
    else if (pool) hipLaunchKernelGGL((wino4_output_kernel<false, true>), grid, block, 0, st, d_M, d_bias, d_res, B, H, W, C, d_y, d_amax, inv_su, d_amax_out);
    else hipLaunchKernelGGL((wino4_output_kernel<false, false>), grid, block, 0, st, d_M, d_bias, d_res, B, H, W, C, d_y, d_amax, inv_su, d_amax_out);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

CSLAM_API int cslam_wino4_output_dev(const float *d_M, const float *d_bias, const float *d_res, int B, int H, int W,
                                     int C, int relu, int pool, float *d_y, void *stream) {
    PTR_DEVICE(d_M);
    return wino4_output_launch(d_M, d_bias, d_res, B, H, W, C, relu, pool, d_y, nullptr, 1.0f, nullptr, stream);
}

// ---- split-fp16 form: max |x| -> slot, input transform into [36, T, 3 C] fp16 (hi | lo | hi), output transform with the
// exact power-of-two rescale 1 / (sV sU) ----
CSLAM_API int cslam_absmax_dev(const float *d_x, int64_t n, unsigned *d_slot, void *stream) {
    PTR_DEVICE(d_x);
    ARG_CHECK(d_x && d_slot, "NULL argument");
    ARG_CHECK(n >= 4 && (n % 4) == 0, "n must be a positive multiple of 4");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(d_slot, 0, sizeof(unsigned), st));
    const int64_t blocks = ceil_div64(n / 4, 256);
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, d_x, n / 4, d_slot);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

CSLAM_API int cslam_wino4_input_h3_dev(const float *d_x, int B, int H, int W, int C, const unsigned *d_amax, void *d_V3,
                                       void *stream) {
    PTR_DEVICE(d_x);
    ARG_CHECK(d_x && d_V3 && d_amax, "NULL argument");
    ARG_CHECK(B >= 1 && H >= 1 && W >= 1, "empty map");
    ARG_CHECK(C >= 2 && (C % 2) == 0, "C must be even");
    const int64_t n = (int64_t)B * ((H + 3) / 4) * ((W + 3) / 4) * (C / 2);
    ARG_CHECK(ceil_div64(n, 256) < (1LL << 31), "too many tiles for one launch");
    if ((C % 4) == 0) {
        const int64_t n4 = n / 2;
        hipLaunchKernelGGL(wino4_input_h3x4_kernel, dim3((unsigned)round_up64(ceil_div64(n4, 256), 8)), dim3(256), 0,
                           (hipStream_t)stream, d_x, B, H, W, C, d_amax, (__half *)d_V3);
    } else {
        hipLaunchKernelGGL(wino4_input_h3_kernel, dim3((unsigned)round_up64(ceil_div64(n, 256), 8)), dim3(256), 0,
                           (hipStream_t)stream, d_x, B, H, W, C, d_amax, (__half *)d_V3);
    }
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

CSLAM_API int cslam_wino4_output_scaled_dev(const float *d_M, const float *d_bias, const float *d_res, int B, int H, int W,
                                            int C, int relu, int pool, const unsigned *d_amax, float inv_su,
                                            unsigned *d_amax_out, float *d_y, void *stream) {
    PTR_DEVICE(d_M);
    ARG_CHECK(inv_su > 0.0f, "inv_su must be positive");
    return wino4_output_launch(d_M, d_bias, d_res, B, H, W, C, relu, pool, d_y, d_amax, inv_su, d_amax_out, stream);
}
