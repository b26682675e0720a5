// wino_fused.hip -- a 3x3 / stride 1 / pad 1 convolution with 64 input and 64 or 128 output channels as ONE kernel:
// Winograd F(2x2, 3x3) input transform, the 16 per-frequency products on the fp32 MFMA pipe and the output
// transform + bias + ReLU (+ the 2x2 max-pool that follows) without V or M ever leaving the compute unit (gfx950).
//
// Why: at 64 -> 64 channels (VGG-16 conv1_2 on 224 x 224 maps, cslam/vpr/netvlad.py:163-171) the three-kernel form
// (transform, batched GEMM, transform: winograd.hip) is HBM-bound in every phase -- V and M are 2.25x (F(4x4)) or 4x
// (F(2x2)) the activation and are written and read once each: 33.7 GB per 256 frames, 6.2 ms.  With K = 64 the whole
// contraction of a tile block fits in registers, so here HBM sees the activation once (x1.4 halo, mostly L2) and the
// pooled output once (4.1 GB per 256 frames); the work is then bound by the MFMA pipe: 16 * 64 * 64 * 2 flop per
// 2x2-pixel tile = 4.2e11 flop per 256 frames = 2.7 ms at the 157 TFLOP/s fp32-MFMA peak.
//
// Two forms.  The default is the persistent producer / consumer kernel further down (wino2_fused_c64_pipe_kernel);
// the first one (one tile block per workgroup; launched by the measurement build only: CSLAM_WF_WAVES=4 or 8) documents the
// phases: one workgroup owns 8 x 4 tiles (16 x 8 output pixels), K is walked in four 16-channel quarters:
//   P1  the 18 x 10 pixel patch of the quarter -> LDS (zero border = the convolution's padding)
//   P2  every thread transforms one tile x channel pair: V = B^T d B -> LDS [16 xi][32 tiles][16 ch, groups rotated]
//   P3  wave w multiplies the 32 tiles by U_xi[:, 16w..16w+15]: v_mfma_f32_16x16x4_f32, A from LDS as one
//       ds_read_b128 per (xi, 16-tile block) (conflict-free: see WF_VS), B straight from a pre-permuted
//       copy of U in global memory (262 KB, L2-resident; one float4 per lane per xi), 16 xi x 2 tile blocks x 4
//       accumulator registers = 128 VGPRs.  The next quarter's patch is fetched under the MFMAs.
// Epilogue: every lane holds all 16 frequencies of its 8 (tile, channel) pairs: A^T M A, bias, ReLU, max of the 2x2
// block, 64-byte runs of NHWC output.  Same arithmetic as wino_input_kernel / rocBLAS / wino_output_kernel up to the
// order of the K summation (MFMA fma chain in 4 groups of 16 channels).
#include "common.h"

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define WF_TBW 8
#define WF_TBH 4
#define WF_NT (WF_TBW * WF_TBH)
#define WF_PW (2 * WF_TBW + 2)
#define WF_PH (2 * WF_TBH + 2)
#define WF_NPIX (WF_PW * WF_PH)
#define WF_PS 24      // patch pixel pitch (floats): the transform's ds_read_b64 of two neighbouring tiles hit disjoint banks
#define WF_VS 16      // V row pitch (floats): unpadded, the four 4-channel groups of tile row r rotated by (r & 15) >> 1 --
                      // conflict-free for the ds_read_b128 lane groups of gfx950 ({0-3,12-15,20-27}, ...) and for the
                      // transform's ds_write_b64 (pitch 20 unrotated cost 2x on both: SQ_LDS_BANK_CONFLICT = half of
                      // SQ_LDS_IDX_ACTIVE)
#define WF_VSW(tile, grp) (4 * (((((tile) & 15) >> 1) + (grp)) & 3))
#ifndef WF_ABL
#define WF_ABL 0      // timing experiments only (tools/exp_fused_ablation.sh): bit mask of phases to leave out
#endif

// NW = waves per workgroup (16 output channels each), COUT = output channels of the layer: NW = COUT / 16 (one
// workgroup makes all of them) or NW = 4 with COUT / 64 workgroups per tile block (blockIdx.x interleaves them, so the
// ones sharing an input patch run side by side and the patch comes from L2).
template <int NW, int COUT, bool RELU, bool POOL>
__global__ __launch_bounds__(64 * NW, 8 / NW) void wino2_fused_c64_kernel(const float *__restrict__ x,
                                                                          const float *__restrict__ Up,
                                                                          const float *__restrict__ bias, int H, int W,
                                                                          float *__restrict__ y) {
    constexpr int NT = 64 * NW;                  // threads
    constexpr int NG = COUT / (16 * NW);         // workgroups per tile block
    constexpr int NL = (4 * WF_NPIX + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float s_d[WF_NPIX * WF_PS];
    __shared__ __attribute__((aligned(16))) float s_v[16 * WF_NT * WF_VS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int bx = blockIdx.x / NG, by = blockIdx.y;
    const int wave_g = (blockIdx.x % NG) * NW + wave;           // 16-channel group of the layer's output
    const int64_t b = blockIdx.z;
    const int gx0 = bx * (2 * WF_TBW) - 1, gy0 = by * (2 * WF_TBH) - 1;
    const float *xb = x + b * (int64_t)H * W * 64;

    // patch loader geometry: element e = (pixel, float4 f of the 16-channel quarter), 720 elements, <= 3 per thread
    int p_off[NL];         // LDS offset (floats), -1 = no element
    int64_t p_src[NL];     // global offset (floats) without the quarter, -1 = zero border
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int e = tid + NT * i;
        const int pix = e >> 2, f = e & 3;
        const int pr = pix / WF_PW, pc = pix - pr * WF_PW;
        const int gy = gy0 + pr, gx = gx0 + pc;
        const bool in = (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W);
        p_off[i] = e < 4 * WF_NPIX ? pix * WF_PS + 4 * f : -1;
        p_src[i] = (e < 4 * WF_NPIX && in) ? ((int64_t)gy * W + gx) * 64 + 4 * f : -1;
    }
    f4 pre[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) pre[i] = p_src[i] >= 0 ? *(const f4 *)(xb + p_src[i]) : (f4)(0.0f);
#pragma unroll
    for (int i = 0; i < NL; ++i)
        if (p_off[i] >= 0) *(f4 *)(s_d + p_off[i]) = pre[i];

    f4 acc[16][2];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) { acc[xi][0] = (f4)(0.0f); acc[xi][1] = (f4)(0.0f); }

    // transform geometry: this thread's tile and channel pair
    const int t_tile = (tid >> 3) & 31, t_cp = tid & 7;        // (threads 256.. of an 8-wave workgroup sit P2 out)
    const int t_ty = t_tile >> 3, t_tx = t_tile & 7;
    const float *t_src = s_d + ((2 * t_ty) * WF_PW + 2 * t_tx) * WF_PS + 2 * t_cp;
    float *t_dst = s_v + t_tile * WF_VS + WF_VSW(t_tile, t_cp >> 1) + 2 * (t_cp & 1);
    const float *a_src = s_v + r16 * WF_VS + WF_VSW(r16, g);
    const f4 *up = (const f4 *)Up + wave_g * 64 + lane;        // + ((kq * 16 + xi) * (COUT / 16)) * 64

    __syncthreads();
    for (int kq = 0; kq < 4; ++kq) {
        // B fragments of the first half of the frequencies: in flight during the transform
        f4 bq[16];
#pragma unroll
        for (int xi = 0; xi < 8; ++xi) bq[xi] = up[(int64_t)((((WF_ABL & 16) ? 0 : kq) * 16 + xi) * (COUT / 16)) * 64];
        if (!(WF_ABL & 1) && (NW == 4 || tid < 256)) {   // P2: V = B^T d B for (tile, channel pair)
            f2 d[4][4], r[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) d[i][j] = *(const f2 *)(t_src + (i * WF_PW + j) * WF_PS);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r[0][j] = d[0][j] - d[2][j];
                r[1][j] = d[1][j] + d[2][j];
                r[2][j] = d[2][j] - d[1][j];
                r[3][j] = d[1][j] - d[3][j];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *(f2 *)(t_dst + (4 * i + 0) * WF_NT * WF_VS) = r[i][0] - r[i][2];
                *(f2 *)(t_dst + (4 * i + 1) * WF_NT * WF_VS) = r[i][1] + r[i][2];
                *(f2 *)(t_dst + (4 * i + 2) * WF_NT * WF_VS) = r[i][2] - r[i][1];
                *(f2 *)(t_dst + (4 * i + 3) * WF_NT * WF_VS) = r[i][1] - r[i][3];
            }
        }
        __syncthreads();                                        // V complete; the patch buffer is free
        if (kq < 3 && !(WF_ABL & 8)) {
#pragma unroll
            for (int i = 0; i < NL; ++i)
                pre[i] = p_src[i] >= 0 ? *(const f4 *)(xb + p_src[i] + (kq + 1) * 16) : (f4)(0.0f);
        }
#pragma unroll
        for (int xi = 8; xi < 16; ++xi) bq[xi] = up[(int64_t)((((WF_ABL & 16) ? 0 : kq) * 16 + xi) * (COUT / 16)) * 64];
        // P3: 16 frequencies x 2 blocks of 16 tiles x 4 K-steps
        // the A fragments of frequency xi + 1 are read while the MFMAs of xi run (an LDS read issued right before its
        // consumer costs its whole latency: 16 times per quarter, a third of the MFMA time)
        f4 a0 = *(const f4 *)(a_src), a1 = *(const f4 *)(a_src + 16 * WF_VS);
#pragma unroll
        for (int xi = 0; xi < ((WF_ABL & 2) ? 1 : 16); ++xi) {
            f4 n0 = a0, n1 = a1;
            if (xi < 15) {
                n0 = *(const f4 *)(a_src + ((xi + 1) * WF_NT) * WF_VS);
                n1 = *(const f4 *)(a_src + ((xi + 1) * WF_NT + 16) * WF_VS);
            }
            __builtin_amdgcn_sched_barrier(0);                  // (the scheduler would sink the reads back down)
            // the two tile blocks alternate so that no MFMA waits on the one issued just before it
            // (32-cycle issue, 40-cycle dependent latency)
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, bq[xi].x, acc[xi][0], 0, 0, 0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, bq[xi].x, acc[xi][1], 0, 0, 0);
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, bq[xi].y, acc[xi][0], 0, 0, 0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, bq[xi].y, acc[xi][1], 0, 0, 0);
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, bq[xi].z, acc[xi][0], 0, 0, 0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, bq[xi].z, acc[xi][1], 0, 0, 0);
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, bq[xi].w, acc[xi][0], 0, 0, 0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, bq[xi].w, acc[xi][1], 0, 0, 0);
            a0 = n0; a1 = n1;
        }
        if (kq < 3) {
#pragma unroll
            for (int i = 0; i < NL; ++i)
                if (p_off[i] >= 0) *(f4 *)(s_d + p_off[i]) = pre[i];
        }
        __syncthreads();                                        // next patch visible; V free
    }

    // epilogue: lane (r16, g) holds M_xi[tile = 16 mb + 4 g + v][channel 16 wave_g + r16] in acc[xi][mb][v]
    const int co = 16 * wave_g + r16;
    const float bv = bias ? bias[co] : 0.0f;
    const int Ho = POOL ? H >> 1 : H, Wo = POOL ? W >> 1 : W;
    float *yb = y + b * (int64_t)Ho * Wo * COUT + co;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int tile = 16 * mb + 4 * g + v;
            const int ty = tile >> 3, tx = tile & 7;
            float m[16];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) m[xi] = acc[xi][mb][v];
            float t0[4], t1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t0[j] = m[0 + j] + m[4 + j] + m[8 + j];
                t1[j] = m[4 + j] - m[8 + j] - m[12 + j];
            }
            float y00 = t0[0] + t0[1] + t0[2] + bv, y01 = t0[1] - t0[2] - t0[3] + bv;
            float y10 = t1[0] + t1[1] + t1[2] + bv, y11 = t1[1] - t1[2] - t1[3] + bv;
            if (RELU) { y00 = fmaxf(y00, 0.0f); y01 = fmaxf(y01, 0.0f); y10 = fmaxf(y10, 0.0f); y11 = fmaxf(y11, 0.0f); }
            if ((WF_ABL & 4) && y00 != 12345.678f) continue;
            if (POOL) {
                const int py = by * WF_TBH + ty, px = bx * WF_TBW + tx;
                if (py < Ho && px < Wo) yb[((int64_t)py * Wo + px) * COUT] = fmaxf(fmaxf(y00, y01), fmaxf(y10, y11));
            } else {
                const int oy = (by * WF_TBH + ty) * 2, ox = (bx * WF_TBW + tx) * 2;
                if (oy < H && ox < W) yb[((int64_t)oy * W + ox) * COUT] = y00;
                if (oy < H && ox + 1 < W) yb[((int64_t)oy * W + ox + 1) * COUT] = y01;
                if (oy + 1 < H && ox < W) yb[((int64_t)(oy + 1) * W + ox) * COUT] = y10;
                if (oy + 1 < H && ox + 1 < W) yb[((int64_t)(oy + 1) * W + ox + 1) * COUT] = y11;
            }
        }
    }
}

// ---- producer / consumer form ----------------------------------------------------------------------------------------
// Measured on the kernel above (tools/exp_fused_ablation.py): its two co-resident workgroups run their phases in
// lock-step, so the transform / load / barrier time ADDS to the MFMA time (4.4 ms = 2.9 MFMA + 1.5 other per 256
// frames of conv1_2) instead of hiding under it.  Here one persistent workgroup per compute unit has 8 waves with fixed
// roles: waves 0..3 only multiply (wave w = output channels 16 w.. of the workgroup's 64), waves 4..7 only prepare V --
// wave 4 + p owns tile row p of the 8 x 4 tile block: it loads the 4 x 18 pixel rows it needs into a private LDS strip
// (no cross-wave synchronisation), transforms and writes V of the NEXT 16-channel quarter into the other half of a
// double-buffered V while the MFMA waves consume the current one.  One barrier per quarter.  The quarter sequence runs
// on across the workgroup's tile blocks (block = blockIdx.x, + gridDim.x, ...), so the first patch of the next block is
// loaded and transformed under the last MFMAs of the current one; COUT = 128 is two virtual blocks per tile block.
#define WP_STRIP (4 * WF_PW)                       // pixels of a producer wave's strip
#define WP_NL ((4 * WP_STRIP + 63) / 64)           // float4 elements per lane

struct WpBlock { int img, by, bx; };
__device__ __forceinline__ WpBlock wp_decode(int vb, int gxb, int gyb, int NG) {
    WpBlock r;
    const int per_img = gxb * gyb * NG;
    r.img = vb / per_img;
    const int rem = vb - r.img * per_img;
    r.by = rem / (gxb * NG);
    r.bx = (rem - r.by * gxb * NG) / NG;
    return r;
}

template <int COUT, bool RELU, bool POOL>
__global__ __launch_bounds__(512, 1) void wino2_fused_c64_pipe_kernel(const float *__restrict__ x,
                                                                      const float *__restrict__ Up,
                                                                      const float *__restrict__ bias,
                                                                      const float *__restrict__ res, int H, int W,
                                                                      int gxb, int gyb, int nvb, float *__restrict__ y) {
    constexpr int NG = COUT / 64;
    __shared__ __attribute__((aligned(16))) float s_v[2][16 * WF_NT * WF_VS];
    __shared__ __attribute__((aligned(16))) float s_d[4][WP_STRIP * WF_PS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nmine = (nvb - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = 4 * nmine;

    if (wave >= 4) {
        // ---------------- producer: V of quarter q = j + 1 during iteration j ----------------
        const int pw = wave - 4;
        float *sd = s_d[pw];
        int p_off[WP_NL], p_pr[WP_NL], p_pc[WP_NL];
#pragma unroll
        for (int i = 0; i < WP_NL; ++i) {
            const int e = lane + 64 * i;
            const int pix = e >> 2, f = e & 3;
            p_pr[i] = pix / WF_PW;
            p_pc[i] = pix - p_pr[i] * WF_PW;
            p_off[i] = e < 4 * WP_STRIP ? pix * WF_PS + 4 * f : -1;
        }
        int64_t p_src[WP_NL];
        auto geometry = [&](int blk) {
            const WpBlock c = wp_decode((int)blockIdx.x + blk * (int)gridDim.x, gxb, gyb, NG);
            const int gx0 = c.bx * (2 * WF_TBW) - 1, gy0 = c.by * (2 * WF_TBH) - 1 + 2 * pw;
#pragma unroll
            for (int i = 0; i < WP_NL; ++i) {
                const int gy = gy0 + p_pr[i], gx = gx0 + p_pc[i];
                const bool in = (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W) & (p_off[i] >= 0);
                p_src[i] = in ? ((int64_t)c.img * H * W + (int64_t)gy * W + gx) * 64 + (p_off[i] % WF_PS) : -1;
            }
        };
        const int t_tx = lane >> 3, t_cp = lane & 7;
        const float *t_src = sd + (2 * t_tx) * WF_PS + 2 * t_cp;
        const int t_dst = (pw * WF_TBW + t_tx) * WF_VS + WF_VSW(pw * WF_TBW + t_tx, t_cp >> 1) + 2 * (t_cp & 1);
        f4 pre[WP_NL];
        geometry(0);
#pragma unroll
        for (int i = 0; i < WP_NL; ++i) pre[i] = p_src[i] >= 0 ? *(const f4 *)(x + p_src[i]) : (f4)(0.0f);
        for (int j = -1; j < total; ++j) {
            const int q = j + 1;
            if (q < total && !(WF_ABL & 32)) {
#pragma unroll
                for (int i = 0; i < WP_NL; ++i)
                    if (p_off[i] >= 0) *(f4 *)(sd + p_off[i]) = pre[i];
                const int qn = q + 1;
                if (qn < total) {
                    if ((qn & 3) == 0) geometry(qn >> 2);
#pragma unroll
                    for (int i = 0; i < WP_NL; ++i)
                        pre[i] = p_src[i] >= 0 ? *(const f4 *)(x + p_src[i] + (qn & 3) * 16) : (f4)(0.0f);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the strip is this wave's own
                float *dst = s_v[q & 1] + t_dst;
                f2 d[4][4], r[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) d[i][jj] = *(const f2 *)(t_src + (i * WF_PW + jj) * WF_PS);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    r[0][jj] = d[0][jj] - d[2][jj];
                    r[1][jj] = d[1][jj] + d[2][jj];
                    r[2][jj] = d[2][jj] - d[1][jj];
                    r[3][jj] = d[1][jj] - d[3][jj];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    *(f2 *)(dst + (4 * i + 0) * WF_NT * WF_VS) = r[i][0] - r[i][2];
                    *(f2 *)(dst + (4 * i + 1) * WF_NT * WF_VS) = r[i][1] + r[i][2];
                    *(f2 *)(dst + (4 * i + 2) * WF_NT * WF_VS) = r[i][2] - r[i][1];
                    *(f2 *)(dst + (4 * i + 3) * WF_NT * WF_VS) = r[i][1] - r[i][3];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // reads of the strip before its next fill
            }
            __syncthreads();
        }
    } else {
        // ---------------- consumer: the MFMAs of quarter j during iteration j ----------------
        const int r16 = lane & 15, g = lane >> 4;
        const int wave_g = ((int)blockIdx.x % NG) * 4 + wave;           // gridDim.x is a multiple of NG
        const int co = 16 * wave_g + r16;
        const float bv = bias ? bias[co] : 0.0f;
        const int Ho = POOL ? H >> 1 : H, Wo = POOL ? W >> 1 : W;
        const f4 *up = (const f4 *)Up + wave_g * 64 + lane;
        f4 bq[16];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) bq[xi] = up[(int64_t)(xi * (COUT / 16)) * 64];
        f4 acc[16][2];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) { acc[xi][0] = (f4)(0.0f); acc[xi][1] = (f4)(0.0f); }
        __syncthreads();                                                // j = -1
        for (int j = 0; j < total; ++j) {
            const int kn = (j + 1) & 3;
            const float *a_src = s_v[j & 1] + r16 * WF_VS + WF_VSW(r16, g);
            f4 a0 = *(const f4 *)(a_src), a1 = *(const f4 *)(a_src + 16 * WF_VS);
#pragma unroll
            for (int xi = 0; xi < ((WF_ABL & 64) ? 1 : 16); ++xi) {
                f4 n0 = a0, n1 = a1;
                if (xi < 15) {
                    n0 = *(const f4 *)(a_src + ((xi + 1) * WF_NT) * WF_VS);
                    n1 = *(const f4 *)(a_src + ((xi + 1) * WF_NT + 16) * WF_VS);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, bq[xi].x, acc[xi][0], 0, 0, 0);
                acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, bq[xi].x, acc[xi][1], 0, 0, 0);
                acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, bq[xi].y, acc[xi][0], 0, 0, 0);
                acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, bq[xi].y, acc[xi][1], 0, 0, 0);
                acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, bq[xi].z, acc[xi][0], 0, 0, 0);
                acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, bq[xi].z, acc[xi][1], 0, 0, 0);
                acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, bq[xi].w, acc[xi][0], 0, 0, 0);
                acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, bq[xi].w, acc[xi][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                bq[xi] = up[(int64_t)((kn * 16 + xi) * (COUT / 16)) * 64];   // the next quarter's weights, in place
                a0 = n0; a1 = n1;
            }
            if ((j & 3) == 3) {
                // output transform of the finished block: lane (r16, g) holds M_xi[tile 16 mb + 4 g + v][channel co]
                const WpBlock c = wp_decode((int)blockIdx.x + (j >> 2) * (int)gridDim.x, gxb, gyb, NG);
                float *yb = y + (int64_t)c.img * Ho * Wo * COUT + co;
                const float *rb = res ? res + (int64_t)c.img * H * W * COUT + co : nullptr;   // shortcut (no pooling)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int tile = 16 * mb + 4 * g + v;
                        const int ty = tile >> 3, tx = tile & 7;
                        float m[16];
#pragma unroll
                        for (int xi = 0; xi < 16; ++xi) m[xi] = acc[xi][mb][v];
                        float t0[4], t1[4];
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            t0[jj] = m[0 + jj] + m[4 + jj] + m[8 + jj];
                            t1[jj] = m[4 + jj] - m[8 + jj] - m[12 + jj];
                        }
                        float y00 = t0[0] + t0[1] + t0[2] + bv, y01 = t0[1] - t0[2] - t0[3] + bv;
                        float y10 = t1[0] + t1[1] + t1[2] + bv, y11 = t1[1] - t1[2] - t1[3] + bv;
                        if (!POOL && rb) {
                            const int oy = (c.by * WF_TBH + ty) * 2, ox = (c.bx * WF_TBW + tx) * 2;
                            if (oy < H && ox < W) y00 += rb[((int64_t)oy * W + ox) * COUT];
                            if (oy < H && ox + 1 < W) y01 += rb[((int64_t)oy * W + ox + 1) * COUT];
                            if (oy + 1 < H && ox < W) y10 += rb[((int64_t)(oy + 1) * W + ox) * COUT];
                            if (oy + 1 < H && ox + 1 < W) y11 += rb[((int64_t)(oy + 1) * W + ox + 1) * COUT];
                        }
                        if (RELU) {
                            y00 = fmaxf(y00, 0.0f); y01 = fmaxf(y01, 0.0f);
                            y10 = fmaxf(y10, 0.0f); y11 = fmaxf(y11, 0.0f);
                        }
                        if (POOL) {
                            const int py = c.by * WF_TBH + ty, px = c.bx * WF_TBW + tx;
                            if (py < Ho && px < Wo)
                                yb[((int64_t)py * Wo + px) * COUT] = fmaxf(fmaxf(y00, y01), fmaxf(y10, y11));
                        } else {
                            const int oy = (c.by * WF_TBH + ty) * 2, ox = (c.bx * WF_TBW + tx) * 2;
                            if (oy < H && ox < W) yb[((int64_t)oy * W + ox) * COUT] = y00;
                            if (oy < H && ox + 1 < W) yb[((int64_t)oy * W + ox + 1) * COUT] = y01;
                            if (oy + 1 < H && ox < W) yb[((int64_t)(oy + 1) * W + ox) * COUT] = y10;
                            if (oy + 1 < H && ox + 1 < W) yb[((int64_t)(oy + 1) * W + ox + 1) * COUT] = y11;
                        }
                    }
                }
#pragma unroll
                for (int xi = 0; xi < 16; ++xi) { acc[xi][0] = (f4)(0.0f); acc[xi][1] = (f4)(0.0f); }
            }
            __syncthreads();
        }
    }
}

template <int COUT>
static int launch_fused_pipe(const float *d_x, const float *d_Up, const float *d_bias, const float *d_res, int B, int H,
                             int W, int relu, int pool, float *d_y, hipStream_t st) {
    const int n_cu = cslam_cu_count();
    ARG_CHECK(n_cu > 0, "no HIP device");
    const int gxb = (int)ceil_div64(W, 2 * WF_TBW), gyb = (int)ceil_div64(H, 2 * WF_TBH);
    const int64_t nvb = (int64_t)B * gxb * gyb * (COUT / 64);
    ARG_CHECK(nvb < (1ll << 30), "too many tile blocks for one launch");
    int grid_n = (int)(nvb < n_cu ? nvb : n_cu);
    grid_n -= grid_n % (COUT / 64);
    dim3 grid((unsigned)grid_n), block(512);
    if (relu && pool) hipLaunchKernelGGL((wino2_fused_c64_pipe_kernel<COUT, true, true>), grid, block, 0, st, d_x, d_Up, d_bias, d_res, H, W, gxb, gyb, (int)nvb, d_y);
    else if (relu) hipLaunchKernelGGL((wino2_fused_c64_pipe_kernel<COUT, true, false>), grid, block, 0, st, d_x, d_Up, d_bias, d_res, H, W, gxb, gyb, (int)nvb, d_y);
    else if (pool) hipLaunchKernelGGL((wino2_fused_c64_pipe_kernel<COUT, false, true>), grid, block, 0, st, d_x, d_Up, d_bias, d_res, H, W, gxb, gyb, (int)nvb, d_y);
    else hipLaunchKernelGGL((wino2_fused_c64_pipe_kernel<COUT, false, false>), grid, block, 0, st, d_x, d_Up, d_bias, d_res, H, W, gxb, gyb, (int)nvb, d_y);
    return CSLAM_OK;
}

// ---- F(4x4, 3x3) producer / consumer form ----------------------------------------------------------------------------
// The same persistent role split with 6 x 6 input tiles and 4 x 4 output tiles: 36 frequencies, 1.78x fewer MFMAs per
// output pixel than F(2x2) (144 per 256 pixels and 16 input channels against 128 per 128).  A workgroup block is 4 x 4
// tiles (16 x 16 output pixels, one 16-row MFMA operand); the 36 x 4 accumulator registers per lane leave no room for a
// quarter's 36 B fragments, so those stream through a 3-pair ring (refilled in place three pairs = 768 MFMA cycles ahead
// of their use, across quarter boundaries), the A fragments one pair ahead; the two frequencies of a pair alternate so
// that no MFMA waits on its predecessor.  Transform constants as in winograd.hip (wino4_bt / wino4_at), so the
// arithmetic equals the three-kernel F(4x4) form up to the order of the K summation.
#define W4_T 4                                     // tiles per block side
#define W4_PW (4 * W4_T + 2)                       // 18 patch pixels per row
#define W4_PS 20                                   // patch pixel pitch (floats): ds_read_b32 of two tiles conflict-free
#define W4_STRIP (6 * W4_PW)                       // pixels of a producer wave's strip (the 6 rows of its tile row)
#define W4_NL ((4 * W4_STRIP + 63) / 64)
#define W4_NT (W4_T * W4_T)
#ifndef W4_BR
#define W4_BR 6                                    // B-fragment pairs in flight (divides 18)
#endif

__device__ __forceinline__ void w4_bt(float &d0, float &d1, float &d2, float &d3, float &d4, float &d5) {
    const float r0 = 4.0f * d0 - 5.0f * d2 + d4;
    const float r1 = -4.0f * (d1 + d2) + d3 + d4;
    const float r2 = 4.0f * (d1 - d2) - d3 + d4;
    const float r3 = 2.0f * (d3 - d1) - d2 + d4;
    const float r4 = 2.0f * (d1 - d3) - d2 + d4;
    const float r5 = 4.0f * d1 - 5.0f * d3 + d5;
    d0 = r0; d1 = r1; d2 = r2; d3 = r3; d4 = r4; d5 = r5;
}
__device__ __forceinline__ void w4_at(float m0, float m1, float m2, float m3, float m4, float m5, float &s0, float &s1,
                                      float &s2, float &s3) {
    const float a = m1 + m2, b = m1 - m2, c = m3 + m4, e = m3 - m4;
    s0 = m0 + a + c;
    s1 = b + 2.0f * e;
    s2 = a + 4.0f * c;
    s3 = b + 8.0f * e + m5;
}

template <int COUT, bool RELU, bool POOL>
__global__ __launch_bounds__(512, 1) void wino4_fused_c64_pipe_kernel(const float *__restrict__ x,
                                                                      const float *__restrict__ Up,
                                                                      const float *__restrict__ bias,
                                                                      const float *__restrict__ res, int H, int W,
                                                                      int gxb, int gyb, int nvb, float *__restrict__ y) {
    constexpr int NG = COUT / 64;
    __shared__ __attribute__((aligned(16))) float s_v[2][36 * W4_NT * WF_VS];
    __shared__ __attribute__((aligned(16))) float s_d[4][W4_STRIP * W4_PS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nmine = (nvb - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = 4 * nmine;

    if (wave >= 4) {
        // ---------------- producer: V of quarter q = j + 1 during iteration j ----------------
        const int pw = wave - 4;
        float *sd = s_d[pw];
        int p_off[W4_NL], p_pr[W4_NL], p_pc[W4_NL];
#pragma unroll
        for (int i = 0; i < W4_NL; ++i) {
            const int e = lane + 64 * i;
            const int pix = e >> 2, f = e & 3;
            p_pr[i] = pix / W4_PW;
            p_pc[i] = pix - p_pr[i] * W4_PW;
            p_off[i] = e < 4 * W4_STRIP ? pix * W4_PS + 4 * f : -1;
        }
        int64_t p_src[W4_NL];
        auto geometry = [&](int blk) {
            const WpBlock c = wp_decode((int)blockIdx.x + blk * (int)gridDim.x, gxb, gyb, NG);
            const int gx0 = c.bx * (4 * W4_T) - 1, gy0 = c.by * (4 * W4_T) - 1 + 4 * pw;
#pragma unroll
            for (int i = 0; i < W4_NL; ++i) {
                const int gy = gy0 + p_pr[i], gx = gx0 + p_pc[i];
                const bool in = (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W) & (p_off[i] >= 0);
                p_src[i] = in ? ((int64_t)c.img * H * W + (int64_t)gy * W + gx) * 64 + (p_off[i] % W4_PS) : -1;
            }
        };
        const int t_tx = lane >> 4, t_c = lane & 15;
        const int t_tile = pw * W4_T + t_tx;
        const float *t_src = sd + (4 * t_tx) * W4_PS + t_c;
        const int t_dst = t_tile * WF_VS + WF_VSW(t_tile, t_c >> 2) + (t_c & 3);
        // the patch of quarter q + 2 is requested while quarter q is transformed: one quarter (~2 us) ahead did not
        // cover the latency of these 64-byte-per-pixel loads under load (0.5 ms of 3.0 on conv1_2)
        f4 pre[W4_NL], pre2[W4_NL];
        geometry(0);
#pragma unroll
        for (int i = 0; i < W4_NL; ++i) pre[i] = p_src[i] >= 0 ? *(const f4 *)(x + p_src[i]) : (f4)(0.0f);
#pragma unroll
        for (int i = 0; i < W4_NL; ++i) pre2[i] = p_src[i] >= 0 ? *(const f4 *)(x + p_src[i] + 16) : (f4)(0.0f);
        for (int j = -1; j < total; ++j) {
            const int q = j + 1;
            if (q < total && !(WF_ABL & 32)) {
#pragma unroll
                for (int i = 0; i < W4_NL; ++i)
                    if (p_off[i] >= 0) *(f4 *)(sd + p_off[i]) = pre[i];
#pragma unroll
                for (int i = 0; i < W4_NL; ++i) pre[i] = pre2[i];
                const int qn = q + 2;
                if (qn < total && !(WF_ABL & 2048)) {
                    if ((qn & 3) == 0) geometry(qn >> 2);
#pragma unroll
                    for (int i = 0; i < W4_NL; ++i)
                        pre2[i] = p_src[i] >= 0 ? *(const f4 *)(x + p_src[i] + (qn & 3) * 16) : (f4)(0.0f);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the strip is this wave's own
                float *dst = s_v[q & 1] + t_dst;
                float d[6][6];
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int jj = 0; jj < 6; ++jj) d[i][jj] = t_src[(i * W4_PW + jj) * W4_PS];
                if (!(WF_ABL & 1024)) {
#pragma unroll
                    for (int jj = 0; jj < 6; ++jj) w4_bt(d[0][jj], d[1][jj], d[2][jj], d[3][jj], d[4][jj], d[5][jj]);
#pragma unroll
                    for (int i = 0; i < 6; ++i) w4_bt(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5]);
                }
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int jj = 0; jj < 6; ++jj) dst[(6 * i + jj) * W4_NT * WF_VS] = d[i][jj];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // reads of the strip before its next fill
            }
            __syncthreads();
        }
    } else {
        // ---------------- consumer: the MFMAs of quarter j during iteration j ----------------
        const int r16 = lane & 15, g = lane >> 4;
        // (s_setprio on either role and the A prefetch distance, 1 or 2 pairs, make no difference beyond the +-2 % run-to-run
        // noise: profiles/r01_exp_fused_ablation.log)
        const int wave_g = ((int)blockIdx.x % NG) * 4 + wave;           // gridDim.x is a multiple of NG
        const int co = 16 * wave_g + r16;
        const float bv = bias ? bias[co] : 0.0f;
        const int Ho = POOL ? H >> 1 : H, Wo = POOL ? W >> 1 : W;
        const f4 *up = (const f4 *)Up + wave_g * 64 + lane;             // + ((kq * 36 + xi) * (COUT / 16)) * 64
        f4 bq[W4_BR][2];
#pragma unroll
        for (int p = 0; p < W4_BR; ++p) {
            bq[p][0] = up[(int64_t)((2 * p) * (COUT / 16)) * 64];
            bq[p][1] = up[(int64_t)((2 * p + 1) * (COUT / 16)) * 64];
        }
        f4 acc[36];
#pragma unroll
        for (int xi = 0; xi < 36; ++xi) acc[xi] = (f4)(0.0f);
        __syncthreads();                                                // j = -1
        for (int j = 0; j < total; ++j) {
            const int kq = j & 3, kn = (j + 1) & 3;
            const float *a_src = s_v[j & 1] + r16 * WF_VS + WF_VSW(r16, g);
            // A fragments two pairs (512 MFMA cycles) ahead of their use, ring of 3
            f4 ar[3][2];
            ar[0][0] = *(const f4 *)(a_src); ar[0][1] = *(const f4 *)(a_src + W4_NT * WF_VS);
            ar[1][0] = *(const f4 *)(a_src + 2 * W4_NT * WF_VS); ar[1][1] = *(const f4 *)(a_src + 3 * W4_NT * WF_VS);
#pragma unroll
            for (int p = 0; p < ((WF_ABL & 64) ? 1 : 18); ++p) {
                if (p < 16) {
                    ar[(p + 2) % 3][0] = *(const f4 *)(a_src + ((2 * p + 4) * W4_NT) * WF_VS);
                    ar[(p + 2) % 3][1] = *(const f4 *)(a_src + ((2 * p + 5) * W4_NT) * WF_VS);
                }
                __builtin_amdgcn_sched_barrier(0);
                const f4 a0 = ar[p % 3][0], a1 = ar[p % 3][1];
                const f4 b0 = bq[p % W4_BR][0], b1 = bq[p % W4_BR][1];
                acc[2 * p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc[2 * p], 0, 0, 0);
                acc[2 * p + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc[2 * p + 1], 0, 0, 0);
                acc[2 * p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc[2 * p], 0, 0, 0);
                acc[2 * p + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc[2 * p + 1], 0, 0, 0);
                acc[2 * p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc[2 * p], 0, 0, 0);
                acc[2 * p + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc[2 * p + 1], 0, 0, 0);
                acc[2 * p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc[2 * p], 0, 0, 0);
                acc[2 * p + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc[2 * p + 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                {   // pair p + W4_BR (of this quarter or the next one) into the slot just used
                    const int pn = p + W4_BR < 18 ? p + W4_BR : p + W4_BR - 18;
                    const int kk = p + W4_BR < 18 ? kq : kn;
                    bq[p % W4_BR][0] = up[(int64_t)((kk * 36 + 2 * pn) * (COUT / 16)) * 64];
                    bq[p % W4_BR][1] = up[(int64_t)((kk * 36 + 2 * pn + 1) * (COUT / 16)) * 64];
                }
            }
            if (kq == 3) {
                // output transform of the finished block: lane (r16, g) holds M_xi[tile (g, v)][channel co] in acc[xi][v]
                const WpBlock c = wp_decode((int)blockIdx.x + (j >> 2) * (int)gridDim.x, gxb, gyb, NG);
                float *yb = y + (int64_t)c.img * Ho * Wo * COUT + co;
                const float *rb = res ? res + (int64_t)c.img * H * W * COUT + co : nullptr;
                if (WF_ABL & 128) {      // timing only: the accumulators stay live through one sum and one store
                    f4 t = acc[0];
#pragma unroll
                    for (int xi = 1; xi < 36; ++xi) t += acc[xi];
                    if (c.by * 16 + 4 * g < Ho && c.bx * 16 < Wo)
                        yb[((int64_t)(c.by * 16 + 4 * g) * Wo + c.bx * 16) * COUT] = t.x + t.y + t.z + t.w;
                }
#pragma unroll
                for (int v = 0; v < ((WF_ABL & 128) ? 0 : 4); ++v) {
                    float s[4][6], o[4][4];
#pragma unroll
                    for (int jj = 0; jj < 6; ++jj)
                        w4_at(acc[jj][v], acc[6 + jj][v], acc[12 + jj][v], acc[18 + jj][v], acc[24 + jj][v],
                              acc[30 + jj][v], s[0][jj], s[1][jj], s[2][jj], s[3][jj]);
                    const int oy0 = (c.by * W4_T + g) * 4, ox0 = (c.bx * W4_T + v) * 4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        w4_at(s[i][0], s[i][1], s[i][2], s[i][3], s[i][4], s[i][5], o[i][0], o[i][1], o[i][2], o[i][3]);
                        if (!POOL) {
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {
                                o[i][jj] += bv;
                                if (rb && oy0 + i < H && ox0 + jj < W)
                                    o[i][jj] += rb[((int64_t)(oy0 + i) * W + ox0 + jj) * COUT];
                                if (RELU) o[i][jj] = fmaxf(o[i][jj], 0.0f);
                            }
                        }
                    }
                    if (POOL) {
                        // bias and ReLU after the 2 x 2 maximum: rounding is monotone, so max(a + b, c + b) == max(a, c) + b
                        // bit for bit, and ReLU commutes with max -- 4 instead of 16 of each per tile
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int jj = 0; jj < 2; ++jj) {
                                const int py = (oy0 >> 1) + i, px = (ox0 >> 1) + jj;
                                float m = fmaxf(fmaxf(o[2 * i][2 * jj], o[2 * i][2 * jj + 1]),
                                                fmaxf(o[2 * i + 1][2 * jj], o[2 * i + 1][2 * jj + 1])) + bv;
                                if (RELU) m = fmaxf(m, 0.0f);
                                if (py < Ho && px < Wo) yb[((int64_t)py * Wo + px) * COUT] = m;
                            }
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj)
                                if (oy0 + i < H && ox0 + jj < W) yb[((int64_t)(oy0 + i) * W + ox0 + jj) * COUT] = o[i][jj];
                    }
                }
#pragma unroll
                for (int xi = 0; xi < 36; ++xi) acc[xi] = (f4)(0.0f);
            }
            __syncthreads();
        }
    }
}

template <int COUT>
static int launch_fused4_pipe(const float *d_x, const float *d_Up, const float *d_bias, const float *d_res, int B, int H,
                              int W, int relu, int pool, float *d_y, hipStream_t st) {
    const int n_cu = cslam_cu_count();
    ARG_CHECK(n_cu > 0, "no HIP device");
    const int gxb = (int)ceil_div64(W, 4 * W4_T), gyb = (int)ceil_div64(H, 4 * W4_T);
    const int64_t nvb = (int64_t)B * gxb * gyb * (COUT / 64);
    ARG_CHECK(nvb < (1ll << 30), "too many tile blocks for one launch");
    int grid_n = (int)(nvb < n_cu ? nvb : n_cu);
    grid_n -= grid_n % (COUT / 64);
    dim3 grid((unsigned)grid_n), block(512);
    if (relu && pool) hipLaunchKernelGGL((wino4_fused_c64_pipe_kernel<COUT, true, true>), grid, block, 0, st, d_x, d_Up, d_bias, d_res, H, W, gxb, gyb, (int)nvb, d_y);
    else if (relu) hipLaunchKernelGGL((wino4_fused_c64_pipe_kernel<COUT, true, false>), grid, block, 0, st, d_x, d_Up, d_bias, d_res, H, W, gxb, gyb, (int)nvb, d_y);
    else if (pool) hipLaunchKernelGGL((wino4_fused_c64_pipe_kernel<COUT, false, true>), grid, block, 0, st, d_x, d_Up, d_bias, d_res, H, W, gxb, gyb, (int)nvb, d_y);
    else hipLaunchKernelGGL((wino4_fused_c64_pipe_kernel<COUT, false, false>), grid, block, 0, st, d_x, d_Up, d_bias, d_res, H, W, gxb, gyb, (int)nvb, d_y);
    return CSLAM_OK;
}

CSLAM_API int cslam_wino4_fused_c64_dev(const float *d_x, const float *d_Up, const float *d_bias,
                                        const float *d_residual, int B, int H, int W, int Cout, int relu, int pool,
                                        float *d_y, void *stream) {
    PTR_DEVICE(d_x);
    ARG_CHECK(d_x && d_Up && d_y, "NULL argument");
    ARG_CHECK(B >= 1 && H >= 1 && W >= 1, "empty map");
    ARG_CHECK(Cout == 64 || Cout == 128, "Cout must be 64 or 128");
    ARG_CHECK(!pool || ((H % 2) == 0 && (W % 2) == 0), "pooling needs even H and W");
    ARG_CHECK(!(pool && d_residual), "a shortcut cannot be added to a pooled output");
    hipStream_t st = (hipStream_t)stream;
    const int rc = Cout == 64 ? launch_fused4_pipe<64>(d_x, d_Up, d_bias, d_residual, B, H, W, relu, pool, d_y, st)
                              : launch_fused4_pipe<128>(d_x, d_Up, d_bias, d_residual, B, H, W, relu, pool, d_y, st);
    if (rc != CSLAM_OK) return rc;
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

template <int NW, int COUT>
static void launch_fused_c64(const float *d_x, const float *d_Up, const float *d_bias, int B, int H, int W, int relu,
                             int pool, float *d_y, hipStream_t st) {
    const int64_t gx = ceil_div64(W, 2 * WF_TBW) * (COUT / (16 * NW)), gy = ceil_div64(H, 2 * WF_TBH);
    dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)B), block(64 * NW);
    if (relu && pool) hipLaunchKernelGGL((wino2_fused_c64_kernel<NW, COUT, true, true>), grid, block, 0, st, d_x, d_Up, d_bias, H, W, d_y);
    else if (relu) hipLaunchKernelGGL((wino2_fused_c64_kernel<NW, COUT, true, false>), grid, block, 0, st, d_x, d_Up, d_bias, H, W, d_y);
    else if (pool) hipLaunchKernelGGL((wino2_fused_c64_kernel<NW, COUT, false, true>), grid, block, 0, st, d_x, d_Up, d_bias, H, W, d_y);
    else hipLaunchKernelGGL((wino2_fused_c64_kernel<NW, COUT, false, false>), grid, block, 0, st, d_x, d_Up, d_bias, H, W, d_y);
}

CSLAM_API int cslam_wino2_fused_c64_dev(const float *d_x, const float *d_Up, const float *d_bias,
                                        const float *d_residual, int B, int H, int W, int Cout, int relu, int pool,
                                        float *d_y, void *stream) {
    PTR_DEVICE(d_x);
    ARG_CHECK(d_x && d_Up && d_y, "NULL argument");
    ARG_CHECK(B >= 1 && H >= 1 && W >= 1, "empty map");
    ARG_CHECK(Cout == 64 || Cout == 128, "Cout must be 64 or 128");
    ARG_CHECK(!pool || ((H % 2) == 0 && (W % 2) == 0), "pooling needs even H and W");
    ARG_CHECK(!(pool && d_residual), "a shortcut cannot be added to a pooled output");
    ARG_CHECK(ceil_div64(H, 2 * WF_TBH) <= 65535 && B <= 65535, "map too tall / batch too large for one launch");
    hipStream_t st = (hipStream_t)stream;
    // the persistent producer / consumer kernel; the first form (one tile block per workgroup) is compiled into the measurement build
    // only (CSLAM_WF_WAVES=4 or 8 there)
    int wide = 0;
#ifdef CSLAM_ABLATIONS
    if (const char *e = getenv("CSLAM_WF_WAVES")) wide = atoi(e);
#endif
    if (wide == 0 || d_residual) {
        const int rc = Cout == 64 ? launch_fused_pipe<64>(d_x, d_Up, d_bias, d_residual, B, H, W, relu, pool, d_y, st)
                                  : launch_fused_pipe<128>(d_x, d_Up, d_bias, d_residual, B, H, W, relu, pool, d_y, st);
        if (rc != CSLAM_OK) return rc;
    }
#ifdef CSLAM_ABLATIONS
    else if (Cout == 64) launch_fused_c64<4, 64>(d_x, d_Up, d_bias, B, H, W, relu, pool, d_y, st);
    else if (wide == 8) launch_fused_c64<8, 128>(d_x, d_Up, d_bias, B, H, W, relu, pool, d_y, st);
    else launch_fused_c64<4, 128>(d_x, d_Up, d_bias, B, H, W, relu, pool, d_y, st);
#endif
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

CSLAM_API int cslam_wino2_fused64_dev(const float *d_x, const float *d_Up, const float *d_bias, int B, int H, int W,
                                      int relu, int pool, float *d_y, void *stream) {
    PTR_DEVICE(d_x);
    return cslam_wino2_fused_c64_dev(d_x, d_Up, d_bias, nullptr, B, H, W, 64, relu, pool, d_y, stream);
}
