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

// Split-K / GEMV partial sums.  Buffers only ever grow and superseded ones stay allocated, so a pointer captured in
// a hipGraph (the online path replays one) stays valid when a later, larger batch needs more room.
static float *g_part = nullptr;
static size_t g_part_bytes = 0;
static int g_part_dev = -1;

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
    if (need > g_part_bytes || g_part_dev != dev) {
        size_t want = need > 2 * g_part_bytes ? need : 2 * g_part_bytes;
        if (want < ((size_t)64 << 20)) want = (size_t)64 << 20;
        float *fresh = nullptr;
        HIP_TRY(hipMalloc((void **)&fresh, want));      // the previous buffer is left alive on purpose (see above)
        g_part = fresh; g_part_bytes = want; g_part_dev = dev;
    }
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
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void *)gemm_nt_splitk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    2 * STAGE_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_nt_splitk_kernel, dim3(mt * nt * S), dim3(256), 2 * STAGE_BYTES, st, d_comp, ldc, Dout, d_x,
                       ldx, B, Din, mt, nt, kps, g_part);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(pca_epilogue_kernel, dim3(B), dim3(256), 0, st, g_part, S, B, Dout, d_mean_proj, d_inv_scale,
                       d_out);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}
