// gemm_nt.hip -- PCA projection of the NetVLAD descriptor on gfx950 matrix cores.
//
// Replaces the tail of NetVLAD.compute_embedding, cslam/vpr/netvlad.py:231-237:
//     reduced = pca.transform(embedding)          (sklearn: X @ components_.T - mean_ @ components_.T,
//                                                  / sqrt(explained_variance_) when whitening)
//     out     = sklearn.preprocessing.normalize(reduced)
// as a batched fp32 GEMM  Y[b, d] = sum_k X[b, k] * Comp[d, k]  (both operands K-major, "NT"):
// the same 128x128x32 exact-f32 MFMA tile as sim_topk_mfma.hip (global_load_lds double buffer,
// XOR-swizzled source, ds_read_b128 fragments), split along K so that small batches still fill
// 256 CUs; split-K partials are reduced in a fixed order by the epilogue kernel (deterministic),
// which also subtracts mean_proj, applies the whitening scale and L2-normalises each row.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define TM 128
#define TN 128
#define TK 32
#define STAGE_BYTES (2 * TM * TK * 4)

__device__ __forceinline__ void glds16(const float *g, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// A [M, K] (rows = output features), B [N, K] (rows = batch); part [S][N][M]
__global__ __launch_bounds__(256, 2) void gemm_nt_splitk_kernel(const float *__restrict__ A, int64_t lda, int M,
                                                                const float *__restrict__ B, int64_t ldb, int N, int K,
                                                                int mt, int nt, int kt_per_split,
                                                                float *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l31 = lane & 31;
    int item = blockIdx.x;
    const int tm = item % mt; item /= mt;
    const int tn = item % nt; item /= nt;
    const int sp = item;
    const int nkt = K / TK;
    const int kt0 = sp * kt_per_split;
    int kt1 = kt0 + kt_per_split;
    if (kt1 > nkt) kt1 = nkt;

    const float *gA[4], *gB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int pch = i * 256 + tid, r = pch >> 3, slot = pch & 7, c = slot ^ ((r >> 1) & 7);
        int64_t ra = (int64_t)tm * TM + r; if (ra > M - 1) ra = M - 1;
        int64_t rb = (int64_t)tn * TN + r; if (rb > N - 1) rb = N - 1;
        gA[i] = A + ra * lda + c * 4;
        gB[i] = B + rb * ldb + c * 4;
    }
    const int wave_chunk = wave * 1024;
    auto stage_load = [&](int stage, int kt) {
        char *sA = smem + stage * STAGE_BYTES, *sB = sA + TM * TK * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(gA[i] + kt * TK, sA + i * 4096 + wave_chunk);
            glds16(gB[i] + kt * TK, sB + i * 4096 + wave_chunk);
        }
    };
    const int swz = (lane >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) foff[j] = (((2 * j + h) ^ swz) << 4);
    const int arow0 = (wm * 64 + l31) * 128, brow0 = (wn * 64 + l31) * 128;

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    if (kt1 > kt0) {
        stage_load(0, kt0);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        int cur = 0;
        for (int kt = kt0; kt < kt1; ++kt) {
            if (kt + 1 < kt1) stage_load(cur ^ 1, kt + 1);
            const char *sA = smem + cur * STAGE_BYTES, *sB = sA + TM * TK * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 a0 = *(const f32x4 *)(sA + arow0 + foff[j]);
                f32x4 a1 = *(const f32x4 *)(sA + arow0 + 32 * 128 + foff[j]);
                f32x4 b0 = *(const f32x4 *)(sB + brow0 + foff[j]);
                f32x4 b1 = *(const f32x4 *)(sB + brow0 + 32 * 128 + foff[j]);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b1[t], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b0[t], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc[1][1], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            cur ^= 1;
        }
    }
    // acc[m][n][r] = Y[batch = tn*128 + wn*64 + n*32 + l31][feat = tm*128 + wm*64 + m*32 + (r&3) + 8*(r>>2) + 4h]
    // registers 4g..4g+3 are 4 consecutive features -> one 16-byte store
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int bcol = tn * TN + wn * 64 + n * 32 + l31;
        if (bcol >= N) continue;
        float *prow = part + ((size_t)sp * N + bcol) * M;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int f0 = tm * TM + wm * 64 + m * 32 + 8 * g + 4 * h;
                if (f0 + 3 < M) {
                    f32x4 v = {acc[m][n][4 * g], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]};
                    *(f32x4 *)(prow + f0) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (f0 + e < M) prow[f0 + e] = acc[m][n][4 * g + e];
                }
            }
    }
}

// out[b, d] = normalize_row((sum_s part[s][b][d] - mean_proj[d]) * inv_scale[d])
__global__ __launch_bounds__(256) void pca_epilogue_kernel(const float *__restrict__ part, int S, int N, int M,
                                                           const float *__restrict__ mean_proj,
                                                           const float *__restrict__ inv_scale,
                                                           float *__restrict__ out) {
    __shared__ float red[16];
    const int b = blockIdx.x;
    float ss = 0.0f;
    for (int d = threadIdx.x; d < M; d += blockDim.x) {
        float v = 0.0f;
        for (int s = 0; s < S; ++s) v += part[((size_t)s * N + b) * M + d];
        if (mean_proj) v -= mean_proj[d];
        if (inv_scale) v *= inv_scale[d];
        out[(size_t)b * M + d] = v;
        ss += v * v;
    }
    // block reduce
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    float t = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    float nrm = sqrtf(t);
    float den = nrm == 0.0f ? 1.0f : nrm;          // sklearn normalize: zero rows stay zero
    for (int d = threadIdx.x; d < M; d += blockDim.x) out[(size_t)b * M + d] /= den;
}

// the same epilogue for the pair products: their power-of-two input scale is recomputed from the max |x| slot
__global__ __launch_bounds__(256) void pca_epilogue_pairs_kernel(const float *__restrict__ part, int S, int N, int M,
                                                                 const float *__restrict__ mean_proj,
                                                                 const float *__restrict__ inv_scale, float *__restrict__ out,
                                                                 const unsigned *__restrict__ amax, float a_given, float inv_sw) {
    __shared__ float red[16];
    const float a = fminf(fmaxf(amax ? __uint_as_float(*amax) : a_given, 1e-30f), 1e30f);
    int e;
    (void)frexpf(16384.0f / a, &e);
    const float ps = inv_sw / ldexpf(1.0f, e - 1);
    const int b = blockIdx.x;
    float ss = 0.0f;
    typedef float pf4 __attribute__((ext_vector_type(4)));
    // four consecutive outputs per thread (M is a multiple of 128 here), the partials of a split 16 bytes at a time, four
    // splits in flight: the scalar loop was a chain of S dependent-latency loads per output (0.11 ms per 256 rows)
    for (int d = 4 * threadIdx.x; d < M; d += 4 * blockDim.x) {
        pf4 v = (pf4)(0.0f);
        int s = 0;
        for (; s + 4 <= S; s += 4) {
            const pf4 p0 = *(const pf4 *)(part + ((size_t)s * N + b) * M + d), p1 = *(const pf4 *)(part + ((size_t)(s + 1) * N + b) * M + d);
            const pf4 p2 = *(const pf4 *)(part + ((size_t)(s + 2) * N + b) * M + d), p3 = *(const pf4 *)(part + ((size_t)(s + 3) * N + b) * M + d);
            v = (((v + p0) + p1) + p2) + p3;
        }
        for (; s < S; ++s) v += *(const pf4 *)(part + ((size_t)s * N + b) * M + d);
        v *= ps;
        if (mean_proj) v -= *(const pf4 *)(mean_proj + d);
        if (inv_scale) v *= *(const pf4 *)(inv_scale + d);
        *(pf4 *)(out + (size_t)b * M + d) = v;
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    float t = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    const float nrm = sqrtf(t);
    const float den = nrm == 0.0f ? 1.0f : nrm;
    for (int d = 4 * threadIdx.x; d < M; d += 4 * blockDim.x) {
        pf4 v = *(pf4 *)(out + (size_t)b * M + d);
        v /= den;
        *(pf4 *)(out + (size_t)b * M + d) = v;
    }
}

// Small batches (the online path: one keyframe per call): the projection is a matrix-vector product bound by
// streaming the 512 MB component matrix once.  One wave per pair of output rows, 16-byte loads, 8 KiB per row
// in flight per wave (2048 waves -> 32 MB in flight), float32 accumulation per lane, xor-shuffle reduction in a
// fixed order (deterministic).  Measured 3.5x faster than the 128-row MFMA tile at B = 1.
template <int BB>
__global__ __launch_bounds__(256) void pca_gemv_kernel(const float *__restrict__ comp, int64_t ldc,
                                                       const float *__restrict__ x, int64_t ldx, int Din, int Dout,
                                                       float *__restrict__ raw) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int j0 = gw * 2;
    if (j0 >= Dout) return;
    const int j1 = j0 + 1 < Dout ? j0 + 1 : j0;
    const float *r0 = comp + (int64_t)j0 * ldc, *r1 = comp + (int64_t)j1 * ldc;
    float acc0[BB], acc1[BB];
#pragma unroll
    for (int b = 0; b < BB; ++b) { acc0[b] = 0.0f; acc1[b] = 0.0f; }
    constexpr int U = 8;
    int k = lane * 4;
    for (; k + (U - 1) * 256 < Din; k += U * 256) {
        f32x4 a0[U], a1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a0[u] = __builtin_nontemporal_load((const f32x4 *)(r0 + k + u * 256));
            a1[u] = __builtin_nontemporal_load((const f32x4 *)(r1 + k + u * 256));
        }
#pragma unroll
        for (int b = 0; b < BB; ++b)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f32x4 xv = *(const f32x4 *)(x + (int64_t)b * ldx + k + u * 256);
                acc0[b] += a0[u][0] * xv[0] + a0[u][1] * xv[1] + a0[u][2] * xv[2] + a0[u][3] * xv[3];
                acc1[b] += a1[u][0] * xv[0] + a1[u][1] * xv[1] + a1[u][2] * xv[2] + a1[u][3] * xv[3];
            }
    }
    for (; k < Din; k += 256) {
        const f32x4 a0 = *(const f32x4 *)(r0 + k), a1 = *(const f32x4 *)(r1 + k);
#pragma unroll
        for (int b = 0; b < BB; ++b) {
            const f32x4 xv = *(const f32x4 *)(x + (int64_t)b * ldx + k);
            acc0[b] += a0[0] * xv[0] + a0[1] * xv[1] + a0[2] * xv[2] + a0[3] * xv[3];
            acc1[b] += a1[0] * xv[0] + a1[1] * xv[1] + a1[2] * xv[2] + a1[3] * xv[3];
        }
    }
#pragma unroll
    for (int b = 0; b < BB; ++b) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            acc0[b] += __shfl_xor(acc0[b], off, 64);
            acc1[b] += __shfl_xor(acc1[b], off, 64);
        }
        if (lane == 0) {
            raw[(size_t)b * Dout + j0] = acc0[b];
            if (j1 != j0) raw[(size_t)b * Dout + j1] = acc1[b];
        }
    }
}

// Split-K / GEMV partial sums, operand pairs of the pair form: per (device, stream), grow-only (common.h).
static StreamScratch g_part_scratch;

CSLAM_API int cslam_pca_project_dev(const float *d_x, int64_t ldx, const float *d_comp, int64_t ldc,
                                    const float *d_mean_proj, const float *d_inv_scale, int B, int Din, int Dout,
                                    float *d_out, void *stream) {
    PTR_DEVICE(d_x);
    ARG_CHECK(d_x && d_comp && d_out, "NULL argument");
    ARG_CHECK(B >= 0 && Din >= TK && Dout >= 1 && Din % TK == 0, "Din must be a positive multiple of 32");
    ARG_CHECK(((uintptr_t)d_x % 16 == 0) && ((uintptr_t)d_comp % 16 == 0), "x / comp must be 16-byte aligned");
    ARG_CHECK(ldx >= Din && ldc >= Din && ldx % 4 == 0 && ldc % 4 == 0, "row pitches must be >= Din and multiples of 4 floats");
    if (B == 0) return CSLAM_OK;
    hipStream_t st = (hipStream_t)stream;
    int dev = 0; HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop; HIP_TRY(hipGetDeviceProperties(&prop, dev));
    const int mt = (int)ceil_div64(Dout, TM), nt = (int)ceil_div64(B, TN), nkt = Din / TK;
    int S = (int)ceil_div64(2 * prop.multiProcessorCount, (int64_t)mt * nt);
    if (S > nkt / 8) S = nkt / 8;                  // at least 8 K steps per split
    if (S < 1) S = 1;
    const int kps = (int)ceil_div64(nkt, S);
    S = (int)ceil_div64(nkt, kps);
    size_t need = (size_t)S * B * Dout * 4;
    SCRATCH_GET(g_part_c, char *, g_part_scratch, dev, stream, need, (size_t)64 << 20);
    float *g_part = (float *)g_part_c;
    if (B <= 4 && Din % 4 == 0) {
        dim3 grid((unsigned)ceil_div64(ceil_div64(Dout, 2), 4)), block(256);
        if (B == 1) hipLaunchKernelGGL(pca_gemv_kernel<1>, grid, block, 0, st, d_comp, ldc, d_x, ldx, Din, Dout, g_part);
        else if (B == 2) hipLaunchKernelGGL(pca_gemv_kernel<2>, grid, block, 0, st, d_comp, ldc, d_x, ldx, Din, Dout, g_part);
        else if (B == 3) hipLaunchKernelGGL(pca_gemv_kernel<3>, grid, block, 0, st, d_comp, ldc, d_x, ldx, Din, Dout, g_part);
        else hipLaunchKernelGGL(pca_gemv_kernel<4>, grid, block, 0, st, d_comp, ldc, d_x, ldx, Din, Dout, g_part);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(pca_epilogue_kernel, dim3(B), dim3(256), 0, st, g_part, 1, B, Dout, d_mean_proj, d_inv_scale,
                           d_out);
        HIP_TRY(hipGetLastError());
        return CSLAM_OK;
    }
    static DeviceOnce once;
    int once_dev;
    if (once.todo(&once_dev)) {
        HIP_TRY(hipFuncSetAttribute((const void *)gemm_nt_splitk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    2 * STAGE_BYTES));
        once.done(once_dev);
    }
    hipLaunchKernelGGL(gemm_nt_splitk_kernel, dim3(mt * nt * S), dim3(256), 2 * STAGE_BYTES, st, d_comp, ldc, Dout, d_x,
                       ldx, B, Din, mt, nt, kps, g_part);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(pca_epilogue_kernel, dim3(B), dim3(256), 0, st, g_part, S, B, Dout, d_mean_proj, d_inv_scale,
                       d_out);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

// ---- the projection on the fp16 matrix pipe (batches): x and the components as exact fp16 hi / lo pairs, three of the four
// partial products in fp32 accumulators -- the arithmetic of the trunk's Winograd GEMM (csrc/wino_gemm.hip), whose kernel runs
// it: K = Din is cut into S splits that take the place of the 36 frequencies, pca_epilogue_kernel sums them.  The f32-input
// MFMA form above holds 120 TFLOP/s (0.58 ms per 256 frames at 32768 -> 4096); this one is bound by the 512 MB of weights.
int cslam_pair_gemm_launch(const void *d_A2, const void *d_B2, int nxi, int64_t T, int K, int N, float *d_M, hipStream_t st);

// x [B][Din] (pitch ldx) -> A2 [S][B][Din / S / 32][hi 32 | lo 32] fp16 pairs of s x, s = the power of two with max |x| s <= 2^14
__global__ __launch_bounds__(256) void pca_pairs_rows_kernel(const float *__restrict__ x, int64_t ldx, int B, int Din, int S,
                                                             const unsigned *__restrict__ amax, float a_given,
                                                             unsigned short *__restrict__ A2) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const int c4n = Din >> 2;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)B * c4n) return;
    const int b = (int)(gid / c4n), c = 4 * (int)(gid % c4n);
    const float a = fminf(fmaxf(amax ? __uint_as_float(*amax) : a_given, 1e-30f), 1e30f);
    int e;
    (void)frexpf(16384.0f / a, &e);
    const float sc = ldexpf(1.0f, e - 1);
    const f4 v = *(const f4 *)(x + (int64_t)b * ldx + c) * sc;
    const int ks = Din / S, s = c / ks, cc = c - s * ks;
    unsigned short *o = A2 + (((int64_t)s * B + b) * (ks >> 5) + (cc >> 5)) * 64 + (cc & 31);
    unsigned short h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const _Float16 hi = (_Float16)v[i];
        const _Float16 lo = (_Float16)(v[i] - (float)hi);
        h[i] = __builtin_bit_cast(unsigned short, hi); l[i] = __builtin_bit_cast(unsigned short, lo);
    }
    u2 hv, lv;
    hv.x = h[0] | ((unsigned)h[1] << 16); hv.y = h[2] | ((unsigned)h[3] << 16);
    lv.x = l[0] | ((unsigned)l[1] << 16); lv.y = l[2] | ((unsigned)l[3] << 16);
    *(u2 *)o = hv;
    *(u2 *)(o + 32) = lv;
}

__global__ __launch_bounds__(256) void pca_absmax_kernel(const float *__restrict__ x, int64_t ldx, int B, int Din, unsigned *__restrict__ slot) {
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)B * Din; i += (int64_t)gridDim.x * 256)
        m = fmaxf(m, fabsf(x[(i / Din) * ldx + (i % Din)]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(slot, __float_as_uint(m));
}

// x_bound > 0: a known bound of max |x| (e.g. 1 for L2-normalised VLAD vectors) instead of a pass over x.
// out = normalize_row(((x - 0) W^T - mean_proj) * inv_scale) with W given as pairs: d_W2 [S][Dout][Din / S / 32][hi 32 | lo 32] of
// sW W (`pca_pair_weights`), inv_sw = 1 / sW.  Workspaces (A2, partials, the max |x| slot) live in the grow-only scratch.
CSLAM_API int cslam_pca_project_pairs_dev(const float *d_x, int64_t ldx, float x_bound, const void *d_W2, float inv_sw, int S,
                                          const float *d_mean_proj, const float *d_inv_scale, int B, int Din, int Dout,
                                          float *d_out, void *stream) {
    PTR_DEVICE(d_x);
    ARG_CHECK(d_x && d_W2 && d_out, "NULL argument");
    ARG_CHECK(B >= 1 && S >= 1 && Din >= 32 * S && (Din % (32 * S)) == 0, "Din must be a multiple of 32 S");
    ARG_CHECK(Dout >= 128 && (Dout % 128) == 0, "Dout must be a multiple of 128");
    ARG_CHECK(((uintptr_t)d_x % 16 == 0) && ldx >= Din && ldx % 4 == 0, "x must be 16-byte aligned with a pitch that is a multiple of 4 floats");
    ARG_CHECK(inv_sw > 0.0f, "inv_sw must be positive");
    hipStream_t st = (hipStream_t)stream;
    int dev = 0; HIP_TRY(hipGetDevice(&dev));
    const size_t a2_bytes = (size_t)B * Din * 4, part_bytes = (size_t)S * B * Dout * 4;
    const size_t need = a2_bytes + part_bytes + 256;
    SCRATCH_GET(g_part_c, char *, g_part_scratch, dev, stream, need, (size_t)64 << 20);
    float *g_part = (float *)g_part_c;
    unsigned *slot = (unsigned *)g_part;
    unsigned short *A2 = (unsigned short *)((char *)g_part + 256);
    float *part = (float *)((char *)g_part + 256 + a2_bytes);
    if (!(x_bound > 0.0f)) {                                       // no bound given: one pass over x for max |x|
        HIP_TRY(hipMemsetAsync(slot, 0, 4, st));
        hipLaunchKernelGGL(pca_absmax_kernel, dim3(1024), dim3(256), 0, st, d_x, ldx, B, Din, slot);
    } else {
        slot = nullptr;
    }
    hipLaunchKernelGGL(pca_pairs_rows_kernel, dim3((unsigned)ceil_div64((int64_t)B * (Din / 4), 256)), dim3(256), 0, st, d_x, ldx, B,
                       Din, S, slot, x_bound, A2);
    HIP_TRY(hipGetLastError());
    const int rc = cslam_pair_gemm_launch(A2, d_W2, S, B, Din / S, Dout, part, st);
    if (rc != CSLAM_OK) return rc;
    // the input scale is only known on the device: the epilogue undoes it from the same slot
    hipLaunchKernelGGL(pca_epilogue_pairs_kernel, dim3(B), dim3(256), 0, st, part, S, B, Dout, d_mean_proj, d_inv_scale, d_out, slot,
                       x_bound, inv_sw);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}
