// wino_fused_h.hip -- the one-kernel F(4x4, 3x3) Winograd convolution of the 64-input-channel layers (VGG-16 conv1_2,
// conv2_1: cslam/vpr/netvlad.py:163-171,227) on the fp16 matrix pipe with fp32-grade results (gfx950).
//
// wino_fused.hip runs the same convolution on the f32-input MFMA (1/16 of the fp16 rate), where it is bound by the
// matrix pipe at half of its peak.  Here the Winograd-domain operands are exact fp16 pairs, as in wino_gemm.hip:
//   V = B^T (sV x) B  ->  (vh, vl) with vh + vl = V to 22 bits, packed as ONE dword per value [vh | vl << 16]
//   U = sU G g G^T    ->  (uh, ul), packed the same way (vpr/winograd.py `fused64_pair_weights`)
// A lane's 8-half MFMA operand is then (vh, vl) of 4 consecutive channels, and with the B operand built in registers as
// (uh, uh) resp. (ul, ul) of the same channels -- one v_perm_b32 per dword --
//   acc += A . dup(uh)      = sum (vh + vl) uh
//   acc += A . dup(ul)      = sum (vh + vl) ul        (v_mfma_f32_16x16x32_f16, fp32 accumulate)
// is the full product (the 2^-22 term vl ul included) with TWO matrix instructions of 16 cycles per frequency and
// 16-channel quarter, against four f32 instructions of 32 cycles: the matrix phase shrinks 4x and the kernel becomes
// bound by what it moves (activation in, pooled activation out, the packed U stream from L2).
//
// Structure: all 8 waves of a persistent workgroup are alike (no producer / consumer roles: with the matrix phase this
// short the registers that a weight ring deep enough to cover L2 latency would need are better spent on a second wave per
// SIMD).  One iteration = NB blocks of 4 x 4 tiles side by side (16 x 16 output pixels each); wave w owns output
// channels 16 (w % (COUT/16)) .. + 15 of block w / (COUT/16): COUT = 64 -> NB = 2, COUT = 128 -> NB = 1; the two waves
// that share a weight fragment (COUT = 64) find it in the compute unit's L1.  Per 16-channel quarter:
//   (a) the patch rows prefetched into registers -> LDS, barrier;
//   (b) thread (tile, channel): V = B^T d B from 36 ds_read_b32, split, 36 ds_write_b32, barrier;
//   (c) the next quarter's patch is requested, then 36 x 2 MFMAs per wave with A = one ds_read_b128 per frequency and B
//       from the packed, L2-resident U (ring of WH_BR pairs), barrier.
// After the fourth quarter every lane holds all 36 frequencies of its 4 (tile, channel) pairs: A^T M A, exact rescale by
// 1 / (sV sU), bias, (shortcut), ReLU, 2x2 max, NHWC stores, max |y| for the next layer's scale.
#include <stdlib.h>
#include "common.h"

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define WH_PS 16                       // patch pixel pitch in floats: the LDS-DMA image is lane-linear, so no padding (the
                                       // transform's ds_read_b32 of two tiles of a 32-lane group then collide 2-way)
#define WH_VS 16                       // V row = 16 dwords (16 channels x [hi | lo]), 16-byte groups rotated by (tile & 15) >> 1
#define WH_VSW(tile, grp) (4 * (((((tile) & 15) >> 1) + (grp)) & 3))
#define WH_BR_DEFAULT 2                // weight fragments in flight, in pairs of frequencies (template parameter WH_BR; a ring of 3
                                       // or 4 pairs spills more than its depth gains: 2.39 / 2.47 / 2.63 ms at 2 / 3 / 4, round 2)

__device__ __forceinline__ void wh_glds16(const float *g, float *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ float wh_scale(unsigned amax_bits) {          // = wino_h3_scale (winograd.hip)
    const float a = fminf(fmaxf(__uint_as_float(amax_bits), 1e-30f), 1e30f);
    int e;
    (void)frexpf(327.68f / a, &e);
    return ldexpf(1.0f, e - 1);
}
__device__ __forceinline__ void wh_bt(float &d0, float &d1, float &d2, float &d3, float &d4, float &d5) {
    const float r0 = 4.0f * d0 - 5.0f * d2 + d4;
    const float r1 = -4.0f * (d1 + d2) + d3 + d4;
    const float r2 = 4.0f * (d1 - d2) - d3 + d4;
    const float r3 = 2.0f * (d3 - d1) - d2 + d4;
    const float r4 = 2.0f * (d1 - d3) - d2 + d4;
    const float r5 = 4.0f * d1 - 5.0f * d3 + d5;
    d0 = r0; d1 = r1; d2 = r2; d3 = r3; d4 = r4; d5 = r5;
}
__device__ __forceinline__ void wh_at(float m0, float m1, float m2, float m3, float m4, float m5, float &s0, float &s1,
                                      float &s2, float &s3) {
    const float a = m1 + m2, b = m1 - m2, c = m3 + m4, e = m3 - m4;
    s0 = m0 + a + c;
    s1 = b + 2.0f * e;
    s2 = a + 4.0f * c;
    s3 = b + 8.0f * e + m5;
}
__device__ __forceinline__ unsigned wh_pack(float v) {                    // [fp16(v) | fp16(v - fp16(v)) << 16]
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    return (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
}

// DBG & 32 (CSLAM_WFH_PROF=1, results stay correct): waves 0 and 4 of workgroup 0 add up the shader cycles (s_memtime) they
// spend in each phase of a quarter and leave them in the buffer `cslam_debug_wfh_prof_dev` points the kernel at:
// [wave 0 | wave 4][transform, barrier 1, matrix loop, prefetch + stem work, output transform, barrier 2, quarters]
__device__ unsigned long long *wfh_prof = nullptr;
// DBG: timing-only ablations (wrong results), CSLAM_WFH_DBG: 1 = every weight fragment from ONE address (L1 hits: no L2
// latency), 2 = no input transform, 4 = no MFMAs, 8 = no patch loads after the first, 16 = no output transform / stores
// STEM (COUT = 64 only): the 64-channel input of this convolution is itself the 3 -> 64 channel first convolution of the
// trunk (+ bias + ReLU), computed HERE from the planar 3-channel image instead of being written to HBM by one kernel and read
// back by this one (VGG-16 conv1_1 -> conv1_2: 3.3 GB each way per 256 frames, and the patch bursts every wave's weight loads
// queued behind).  Per iteration the 20 x 36 x 3 image patch of the block is loaded once, split into fp16 pairs and kept in
// LDS; per 16-channel quarter the 18 x 34 patch of first-layer outputs is 39 row tiles of v_mfma_f32_16x16x32_f16 over the
// 27 taps (3 products: xh wh + xl wh + xh wl, fp32-grade like everything else here), written to the single patch buffer
// behind the quarter's main MFMAs.  x = the planar image [B][3][H][W]; st_w1 = `stem_pair_weights` (scaled by the power of
// two 1 / st_inv_sw); amax_in holds max |image|: the input scale s1 is the power of two with max |image| s1 <= 2^14, and the
// scale of THIS layer's V comes from the rigorous bound max_co(|b1[co]| + max |image| * st_sumw[co]), st_sumw[co] = the sum
// of |first-layer weights| of channel co (its true maximum is only known after the kernel has run).
#define ST_IW 36
#define ST_IPL (20 * ST_IW)
#define ST_IMG (3 * ST_IPL)
#define ST_GRP 2                       // row tiles of the first layer a wave works on at once
#define ST_PAD 768                     // zero dwords behind the image: the unused K slots of lane group 3 read there
template <int COUT, bool RELU, bool POOL, int DBG, int WH_BR, bool RES = false, bool STEM = false>
__global__ __launch_bounds__(512, 2) void wino4_fused_c64_h_kernel(
    const float *__restrict__ x, const unsigned *__restrict__ Uh, const float *__restrict__ bias,
    const float *__restrict__ res, int H, int W, int gxs, int gyb, int nsb, const unsigned *__restrict__ amax_in,
    float inv_su, unsigned *__restrict__ amax_out, const float *__restrict__ zero16, float *__restrict__ y,
    const unsigned *__restrict__ st_w1, const float *__restrict__ st_b1, const float *__restrict__ st_sumw, float st_inv_sw) {
    static_assert(!STEM || (COUT == 64 && !RES), "the stem form is the 3 -> 64 -> 64 channel pair only");
    constexpr int NG = COUT / 16;                  // 16-channel output groups: 4 or 8
    constexpr int NB = 8 / NG;                     // blocks per iteration: 2 or 1
    constexpr int NT = 16 * NB;                    // tiles per iteration
    constexpr int PWX = 16 * NB + 2;               // patch width in pixels
    constexpr int NPIX = 18 * PWX;
    constexpr int NL = (4 * NPIX + 511) / 512;     // 16-byte patch elements per thread and quarter (5 | 3)
    constexpr int PBUF = NL * 512 * 4;             // floats per patch buffer (whole LDS-DMA rounds)
    extern __shared__ __attribute__((aligned(16))) char wh_smem[];
    constexpr int NPB = STEM ? 1 : 2;                               // patch buffers
    float *s_p = (float *)wh_smem;                                  // [NPB][PBUF]
    unsigned *s_v = (unsigned *)(wh_smem + NPB * PBUF * 4);         // [36][NT][16]
    unsigned *s_img = s_v + 36 * NT * WH_VS;                        // STEM: [20][36][3] packed [hi | lo] image patch + ST_PAD zeros
    float *s_raw = (float *)(s_img + ST_IMG + ST_PAD);              // STEM: [NL * 512] the same patch as it arrives (planar order)
    unsigned *s_w1 = (unsigned *)(s_raw + NL * 512);                // STEM: first-layer weights [4 quarters][hi | lo][64 lanes][4] + bias [64]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float sc_, st_s1 = 1.0f, st_inv1 = 1.0f;
    if constexpr (STEM) {
        const float a0 = fminf(fmaxf(__uint_as_float(*amax_in), 1e-30f), 1e30f);
        int e;
        (void)frexpf(16384.0f / a0, &e);
        st_s1 = ldexpf(1.0f, e - 1);
        st_inv1 = st_inv_sw / st_s1;
        float bound = 0.0f;
        for (int c = 0; c < 64; ++c) bound = fmaxf(bound, fabsf(st_b1 ? st_b1[c] : 0.0f) + a0 * st_sumw[c]);
        sc_ = wh_scale(__float_as_uint(bound));
    } else {
        sc_ = wh_scale(*amax_in);
    }
    const float sc = sc_;
    const float inv = inv_su / sc;

    // ---- patch loader: element e = i * 512 + tid = (pixel e >> 2, float4 e & 3 of the 16-channel quarter), LDS-DMA into the
    // lane-linear image [pixel][16 floats]; pixels outside the map (the convolution's zero padding, ragged blocks) and the
    // round-up elements read 16 zero bytes
    int p_src[NL];                                 // float offset into x without the quarter, -1 = zeros
    int c_img = 0, c_by = 0, c_sx = 0;             // the iteration the loader geometry points at
    auto geometry = [&](int sb) {
        const int per_img = gxs * gyb;
        c_img = sb / per_img;
        const int rem = sb - c_img * per_img;
        c_by = rem / gxs;
        c_sx = rem - c_by * gxs;
        const int gx0 = c_sx * (16 * NB) - 1, gy0 = c_by * 16 - 1;
        if constexpr (STEM) {
            // element e = i * 512 + tid of the planar 3 x 20 x 36 image patch whose origin is one pixel up / left of the patch
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int e = tid + 512 * i;
                const int ci = (e >= ST_IPL) + (e >= 2 * ST_IPL);
                const int rem = e - ci * ST_IPL;
                const int r = (rem * 1821) >> 16, c = rem - ST_IW * r;             // rem / 36 for rem < 2200
                const int gy = gy0 - 1 + r, gx = gx0 - 1 + c;
                const bool in = (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W) & (e < ST_IMG);
                p_src[i] = in ? ((c_img * 3 + ci) * H + gy) * W + gx : -1;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = tid + 512 * i;
            const int pix = e >> 2, f = e & 3;
            const int pr = pix / PWX, pc = pix - pr * PWX;
            const int gy = gy0 + pr, gx = gx0 + pc;
            const bool in = (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W) & (pix < NPIX);
            p_src[i] = in ? ((c_img * H + gy) * W + gx) * 64 + 4 * f : -1;
        }
    };
    auto fetch = [&](int buf, int kq) {
#pragma unroll
        for (int i = 0; i < NL; ++i)
            wh_glds16(p_src[i] >= 0 ? x + p_src[i] + kq * 16 : zero16, s_p + buf * PBUF + i * (512 * 4) + wave * 256);
    };
    // ---- STEM: image patch in (one dword per lane by LDS-DMA: no registers held across the phases), split, first layer
    auto img_request = [&]() {
#pragma unroll
        for (int i = 0; i < NL; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p_src[i] >= 0 ? x + p_src[i] : zero16),
                                             (__attribute__((address_space(3))) void *)(s_raw + i * 512 + wave * 64), 4, 0, 0);
    };
    auto img_split = [&]() {                       // raw planar patch -> [row][col][channel] packed fp16 pairs of st_s1 * x
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = tid + 512 * i;
            if (e < ST_IMG) {
                const int ci = (e >= ST_IPL) + (e >= 2 * ST_IPL);
                const int rem = e - ci * ST_IPL;
                s_img[3 * rem + ci] = wh_pack(s_raw[e] * st_s1);
            }
        }
    };
    // K slots of the first layer's MFMA (lane group gq = lane >> 4, slot j): gq < 3: tap (ky = gq, kx = j / 3, ci = j % 3), the
    // 8 dwords are CONSECUTIVE in the [row][col][channel] image; gq = 3, j < 3: the ninth tap (kx = 2, ci = 2) of row ky = j
    // (stride one image row), j >= 3: zero weights, the reads land in the zero padding behind the image
    const int st_gq = lane >> 4;
    const int st_lane_off = st_gq < 3 ? st_gq * (3 * ST_IW) : 8;
    const int st_lane_stride = st_gq < 3 ? 1 : 3 * ST_IW;
    auto produce = [&](int kq, int b_by, int b_sx) {               // patch quarter kq of block (b_by, b_sx) -> s_p
        const int gx0 = b_sx * (16 * NB) - 1, gy0 = b_by * 16 - 1;
        // weights and bias from LDS (8 KB, staged once): as global loads they queued behind the next quarter's weight fragments
        const u4 w1h = ((const u4 *)s_w1)[(kq * 2 + 0) * 64 + lane], w1l = ((const u4 *)s_w1)[(kq * 2 + 1) * 64 + lane];
        const float b1 = __uint_as_float(s_w1[2048 + 16 * kq + (lane & 15)]);
        // blocks whose 18 x 34 patch lies inside the map (61 % of a 224 x 224 frame's) skip the per-pixel tests
        const bool whole = (gy0 >= 0) & (gy0 + 18 <= H) & (gx0 >= 0) & (gx0 + PWX <= W);
        constexpr int NMT = (NPIX + 15) / 16;                       // 39 row tiles of 16 pixels, wave w takes w, w + 8, ...
        constexpr int PER = (NMT + 7) / 8;
        // a wave's row tiles TWO at a time: the image reads of both are in flight before the first MFMA and their three-MFMA
        // chains interleave (one tile at a time the phase is a chain of LDS and MFMA latencies; all five at once spill)
        const h8 Bh = __builtin_bit_cast(h8, w1h), Bl = __builtin_bit_cast(h8, w1l);
#pragma unroll 1
        for (int i0 = 0; i0 < PER; i0 += ST_GRP) {
            unsigned pk[ST_GRP][8];
#pragma unroll
            for (int i = 0; i < ST_GRP; ++i) {
                const int mt = wave + 8 * (i0 + i);
                int p = 16 * mt + (lane & 15);
                p = p < NPIX ? p : NPIX - 1;
                const int pr = (p * 1928) >> 16, pc = p - PWX * pr;                // p / 34 for p < 700
                const unsigned *ib = s_img + (pr * ST_IW + pc) * 3 + st_lane_off;
#pragma unroll
                for (int j = 0; j < 8; ++j) pk[i][j] = ib[j * st_lane_stride];
            }
            f4 c[ST_GRP];
#pragma unroll
            for (int i = 0; i < ST_GRP; ++i) {
                u4 ah, al;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    ah[d] = __builtin_amdgcn_perm(pk[i][2 * d + 1], pk[i][2 * d], 0x05040100u); // (hi, hi) of slots 2d, 2d + 1
                    al[d] = __builtin_amdgcn_perm(pk[i][2 * d + 1], pk[i][2 * d], 0x07060302u); // (lo, lo)
                }
                const h8 Ah = __builtin_bit_cast(h8, ah), Al = __builtin_bit_cast(h8, al);
                c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, Bh, (f4)(0.0f), 0, 0, 0);
                c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bl, c[i], 0, 0, 0);
                c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bh, c[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < ST_GRP; ++i) {
                const int mt = wave + 8 * (i0 + i);
                if (mt < NMT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int p2 = 16 * mt + 4 * st_gq + r;                    // lane holds rows 4 gq + r of column lane & 15
                        const float v = fmaxf(c[i][r] * st_inv1 + b1, 0.0f);       // exact power-of-two rescale, bias, ReLU
                        bool in = true;
                        if (!whole) {
                            const int pr2 = (p2 * 1928) >> 16, pc2 = p2 - PWX * pr2;
                            const int gy = gy0 + pr2, gx = gx0 + pc2;
                            in = (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W);
                        }
                        s_p[p2 * WH_PS + (lane & 15)] = in ? v : 0.0f;             // outside the map: the second layer's zero padding
                    }
                }
            }
        }
    };
    // ---- transform geometry: thread = (tile, channel of the quarter); COUT = 128: threads 256.. sit the transform out
    // (measured and rejected: the stem producer's row tiles interleaved into the matrix loop -- 110 spilled registers, 5.7 ms
    // against 3.2, profiles/r02_v26_stem_interleave_rejected.log; for NB = 1, where half the threads sit the transform out: two threads per (tile, channel), three
    // frequency rows each -- and the output stores staged through LDS as whole pixels: 1.50 / 1.49 ms against 1.43,
    // profiles/r02_v21_fused_h_conv2_1_ab_rejected.log)
    const int t_tile = tid >> 4, t_c = tid & 15;
    const bool t_on = t_tile < NT;
    const int t_bl = (t_tile >> 4) & (NB - 1), t_tt = t_tile & 15;
    const int t_src = ((4 * (t_tt >> 2)) * PWX + 16 * t_bl + 4 * (t_tt & 3)) * WH_PS + t_c;
    const int t_dst = t_tile * WH_VS + WH_VSW(t_tile, t_c >> 2) + (t_c & 3);
    // ---- MFMA geometry
    const int r16 = lane & 15, g = lane >> 4;
    const int m_bl = NB == 2 ? wave >> 2 : 0, m_grp = NB == 2 ? wave & 3 : wave;
    const int co = 16 * m_grp + r16;
    const float bv = bias ? bias[co] : 0.0f;
    const int Ho = POOL ? H >> 1 : H, Wo = POOL ? W >> 1 : W;
    const u4 *up = (const u4 *)Uh + m_grp * 64 + lane;              // + ((kq * 36 + xi) * NG) * 64
    const unsigned *a_src = s_v + (16 * m_bl + r16) * WH_VS + WH_VSW(r16, g);

    f4 acc[36];
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) acc[xi] = (f4)(0.0f);
    float my_amax = 0.0f;
    unsigned long long *prof = nullptr;
    unsigned long long tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tl = 0;
    if constexpr ((DBG & 32) != 0) {
        if (blockIdx.x == 0 && (tid & 255) == 0) prof = wfh_prof;
    }
#define WH_TICK(i) do { if constexpr ((DBG & 32) != 0) { if (prof) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tl; tl = t_; } } } while (0)

    const int n_mine = (nsb - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total_q = 4 * n_mine;                                 // quarters this workgroup walks
    // The patch of quarter Q + 2 is requested at the END of the MFMAs of quarter Q (its buffer, that of quarter Q, was
    // consumed by Q's transform), BEHIND the first weight fragments of quarter Q + 1: a wave's loads return in order, so
    // anything issued after an HBM-latency LDS-DMA waits for it -- with the patch requested at the top of a quarter every
    // weight fragment of that quarter queued behind it (0.4 ms of 2.6 on conv1_2, profiles/r02_v5_fused_h_ablation.log).
    int l_q = 0;                                                    // next quarter to request
    int n_img = 0, n_by = 0, n_sx = 0;                              // block the loader has most recently moved to
    auto request_next = [&]() {
        if (l_q < total_q && !((DBG & 8) && l_q >= 2)) {
            if ((l_q & 3) == 0) {
                geometry((int)blockIdx.x + (l_q >> 2) * (int)gridDim.x);
                n_img = c_img; n_by = c_by; n_sx = c_sx;
            }
            fetch(l_q & 1, l_q & 3);
        }
        ++l_q;
    };
    if constexpr (STEM) {
        for (int i = tid; i < ST_PAD; i += 512) s_img[ST_IMG + i] = 0u;
        for (int i = tid; i < 2048; i += 512) s_w1[i] = st_w1[i];
        if (tid < 64) s_w1[2048 + tid] = st_b1 ? __float_as_uint(st_b1[tid]) : 0u;
        if (n_mine > 0) {
            geometry((int)blockIdx.x);
            n_img = c_img; n_by = c_by; n_sx = c_sx;
            img_request();
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            img_split();
            __syncthreads();
            produce(0, n_by, n_sx);
        }
    } else if (n_mine > 0) {
        request_next();
        request_next();
    }
    int o_img = n_img, o_by = n_by, o_sx = n_sx;                    // block of iteration 0
    u4 bq[WH_BR][2];
#pragma unroll
    for (int p = 0; p < WH_BR; ++p) {
        bq[p][0] = up[(int64_t)((2 * p) * NG) * 64];
        bq[p][1] = up[(int64_t)((2 * p + 1) * NG) * 64];
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int it = 0; it < n_mine; ++it) {
#pragma unroll 1
        for (int kq = 0; kq < 4; ++kq) {
            const int pb = STEM ? 0 : kq & 1;                       // 4 quarters per iteration: the buffer parity is kq's
            if constexpr ((DBG & 32) != 0) { if (prof) { tl = __builtin_amdgcn_s_memtime(); tacc[6] += 1; } }
            // (b) V = B^T (sV d) B, split into fp16 pairs
            if (t_on && !(DBG & 2)) {
                const float *src = s_p + pb * PBUF + t_src;
                float d[6][6];
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) d[i][j] = src[(i * PWX + j) * WH_PS] * sc;
#pragma unroll
                for (int j = 0; j < 6; ++j) wh_bt(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);
#pragma unroll
                for (int i = 0; i < 6; ++i) wh_bt(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5]);
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) s_v[(6 * i + j) * NT * WH_VS + t_dst] = wh_pack(d[i][j]);
            }
            // V complete: LDS stores drained, then a RAW barrier -- __syncthreads() would also wait for the LDS-DMA of the
            // next patch (a pending LDS write), which is meant to stay in flight across this barrier
            WH_TICK(0);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0) only
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            WH_TICK(1);
            // (c) 36 frequencies x 2 MFMAs, weight fragments WH_BR pairs ahead (the first WH_BR pairs were requested at the
            // end of the previous quarter)
            // A fragments (LDS) two pairs ahead of their use: issued right before its consumer a ds_read costs its whole
            // latency 18 times per quarter (the matrix work of a pair is only 64 cycles)
            u4 ar[3][2];
            ar[0][0] = *(const u4 *)(a_src); ar[0][1] = *(const u4 *)(a_src + NT * WH_VS);
            ar[1][0] = *(const u4 *)(a_src + 2 * NT * WH_VS); ar[1][1] = *(const u4 *)(a_src + 3 * NT * WH_VS);
#pragma unroll
            for (int p = 0; p < ((DBG & 4) ? 1 : 18); ++p) {
                if (p < 16) {
                    ar[(p + 2) % 3][0] = *(const u4 *)(a_src + (2 * p + 4) * NT * WH_VS);
                    ar[(p + 2) % 3][1] = *(const u4 *)(a_src + (2 * p + 5) * NT * WH_VS);
                }
                const u4 a0 = ar[p % 3][0], a1 = ar[p % 3][1];
                const u4 b0 = bq[p % WH_BR][0], b1 = bq[p % WH_BR][1];
                const h8 A0 = __builtin_bit_cast(h8, a0), A1 = __builtin_bit_cast(h8, a1);
                u4 t0, t1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    t0[e] = __builtin_amdgcn_perm(b0[e], b0[e], 0x01000100u);      // (uh, uh)
                    t1[e] = __builtin_amdgcn_perm(b1[e], b1[e], 0x01000100u);
                }
                acc[2 * p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0, __builtin_bit_cast(h8, t0), acc[2 * p], 0, 0, 0);
                acc[2 * p + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1, __builtin_bit_cast(h8, t1), acc[2 * p + 1], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    t0[e] = __builtin_amdgcn_perm(b0[e], b0[e], 0x03020302u);      // (ul, ul)
                    t1[e] = __builtin_amdgcn_perm(b1[e], b1[e], 0x03020302u);
                }
                acc[2 * p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0, __builtin_bit_cast(h8, t0), acc[2 * p], 0, 0, 0);
                acc[2 * p + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1, __builtin_bit_cast(h8, t1), acc[2 * p + 1], 0, 0, 0);
                if (p + WH_BR < 18) {
                    bq[p % WH_BR][0] = up[(DBG & 1) ? 0 : (int64_t)((kq * 36 + 2 * (p + WH_BR)) * NG) * 64];
                    bq[p % WH_BR][1] = up[(DBG & 1) ? 0 : (int64_t)((kq * 36 + 2 * (p + WH_BR) + 1) * NG) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);                  // keep the rings as deep as written: no hoisting of later loads
            }
            WH_TICK(2);
            // first weight fragments of the next quarter, THEN the patch of the quarter after it (see request_next)
            {
                const int nkq = (kq + 1) & 3;
#pragma unroll
                for (int p = 0; p < WH_BR; ++p) {                  // every ring slot is free here: pair p of the next quarter -> slot p
                    bq[p][0] = up[(DBG & 1) ? 0 : (int64_t)((nkq * 36 + 2 * p) * NG) * 64];
                    bq[p][1] = up[(DBG & 1) ? 0 : (int64_t)((nkq * 36 + 2 * p + 1) * NG) * 64];
                }
                if constexpr (!STEM) request_next();                // quarter 4 it + kq + 2 (at kq = 2 the loader moves to the next block)
            }
            if constexpr (STEM) {
                // The patch of the NEXT quarter is computed here, behind this quarter's MFMAs (its buffer was consumed by this
                // quarter's transform, two barriers ago).  The image patch of the next block is requested behind the third
                // quarter's MFMAs (one LDS-DMA round per iteration instead of four per-quarter bursts), lands under the fourth
                // quarter and is split into pairs before the next block's first patch quarter needs it.
                const bool more = it + 1 < n_mine;
                if (kq < 3) produce(kq + 1, o_by, o_sx);
                if (kq == 2 && more) {                              // AFTER produce(): its weight loads must not queue behind the DMA
                    geometry((int)blockIdx.x + (it + 1) * (int)gridDim.x);
                    n_img = c_img; n_by = c_by; n_sx = c_sx;
                    img_request();
                }
                if (kq == 3 && more) {
                    __builtin_amdgcn_s_waitcnt(0x0F70 | 2 * WH_BR);     // vmcnt: all but the weight fragments just requested
                    __syncthreads();                                // every wave's share of the raw patch has landed; the last
                    img_split();                                    // reader of s_img (quarter 3's patch) is two barriers back
                    __syncthreads();
                    produce(0, n_by, n_sx);
                }
            }
            WH_TICK(3);
            if (kq == 3 && (DBG & 16)) {
                f4 t = acc[0];
#pragma unroll
                for (int xi = 1; xi < 36; ++xi) t += acc[xi];
                if (t.x + t.y + t.z + t.w == 1.2345e-30f) y[0] = t.x;       // keeps the accumulators live
#pragma unroll
                for (int xi = 0; xi < 36; ++xi) acc[xi] = (f4)(0.0f);
            }
            if (kq == 3 && !(DBG & 16)) {
                // ---- output transform: lane (r16, g) holds M_xi[tile (g, v)][channel co] in acc[xi][v].  32-bit element
                // offsets inside the image (H W COUT < 2^31), one wave-uniform test for blocks that lie inside the map
                float *yb = y + (int64_t)o_img * Ho * Wo * COUT + co;
                const float *rb = (RES && !POOL) ? res + (int64_t)o_img * H * W * COUT + co : nullptr;
                const int bx = o_sx * NB + m_bl;
                const bool inside = (o_by * 16 + 16 <= H) & (bx * 16 + 16 <= W);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    __builtin_amdgcn_sched_barrier(0);              // one tile at a time: 40 live values beside the accumulators
                    float s[4][6], o[4][4];
#pragma unroll
                    for (int jj = 0; jj < 6; ++jj)
                        wh_at(acc[jj][v], acc[6 + jj][v], acc[12 + jj][v], acc[18 + jj][v], acc[24 + jj][v],
                              acc[30 + jj][v], s[0][jj], s[1][jj], s[2][jj], s[3][jj]);
                    const int oy0 = (o_by * 4 + g) * 4, ox0 = (bx * 4 + v) * 4;
                    const int e00 = (oy0 * W + ox0) * COUT;         // element offset of the tile's first pixel (unpooled map)
                    if (POOL) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            wh_at(s[i][0], s[i][1], s[i][2], s[i][3], s[i][4], s[i][5], o[i][0], o[i][1], o[i][2], o[i][3]);
                        // 2 x 2 maximum FIRST, then rescale, bias, ReLU on the 4 survivors instead of all 16: the rescale is a
                        // multiplication by a positive power of two (exact, monotone) and x -> x + b rounds monotonically, so
                        // max(a s + b, c s + b) == max(a, c) s + b bit for bit, and ReLU commutes with max.  The maximum of
                        // the pooled values is exactly max |next layer's input|.
                        const int p00 = ((oy0 >> 1) * Wo + (ox0 >> 1)) * COUT;
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int jj = 0; jj < 2; ++jj) {
                                float m = fmaxf(fmaxf(o[2 * i][2 * jj], o[2 * i][2 * jj + 1]),
                                                fmaxf(o[2 * i + 1][2 * jj], o[2 * i + 1][2 * jj + 1]));
                                m = m * inv + bv;
                                if (RELU) m = fmaxf(m, 0.0f);
                                if (inside || ((oy0 >> 1) + i < Ho && (ox0 >> 1) + jj < Wo)) {
                                    my_amax = fmaxf(my_amax, fabsf(m));
                                    yb[p00 + (i * Wo + jj) * COUT] = m;
                                }
                            }
                    } else {
                        // one output row at a time (4 live values beside s[][])
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            __builtin_amdgcn_sched_barrier(0);
                            wh_at(s[i][0], s[i][1], s[i][2], s[i][3], s[i][4], s[i][5], o[i][0], o[i][1], o[i][2], o[i][3]);
                            const int er = e00 + i * W * COUT;
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {
                                const bool in = inside || (oy0 + i < H && ox0 + jj < W);
                                float val = o[i][jj] * inv + bv;                  // exact power-of-two rescale, then bias
                                if (RES && in) val += rb[er + jj * COUT];
                                if (RELU) val = fmaxf(val, 0.0f);
                                if (in) {
                                    my_amax = fmaxf(my_amax, fabsf(val));
                                    yb[er + jj * COUT] = val;
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int xi = 0; xi < 36; ++xi) acc[xi] = (f4)(0.0f);
            }
            WH_TICK(4);
            // The patch of the NEXT quarter was requested one quarter ago, ahead of every weight fragment this quarter used:
            // loads return in order, so it has landed in this wave; the barrier makes that true for all waves.  (Still
            // outstanding, deliberately: the weight fragments and the patch requested just above.)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();                           // V is free again, the next patch visible to everyone
            asm volatile("" ::: "memory");
            WH_TICK(5);
        }
        o_img = n_img; o_by = n_by; o_sx = n_sx;
    }
    if constexpr ((DBG & 32) != 0) {
        if (prof)
            for (int i = 0; i < 7; ++i) prof[(tid >> 8) * 8 + i] = tacc[i];
    }
#undef WH_TICK
    if (amax_out) {
        // max |y| of everything this workgroup wrote: wave maximum, LDS maximum, one global atomic per workgroup and only if
        // it would raise the slot
        unsigned *s_amax = s_v;                                     // V is no longer needed
        if (tid == 0) *s_amax = 0u;
        __syncthreads();
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) my_amax = fmaxf(my_amax, __shfl_xor(my_amax, o, 64));
        if (lane == 0) atomicMax(s_amax, __float_as_uint(my_amax));
        __syncthreads();
        if (tid == 0 && *s_amax > *(volatile unsigned *)amax_out) atomicMax(amax_out, *s_amax);
    }
}

template <int COUT>
static int launch_fused_h(const float *d_x, const unsigned *d_Uh, const float *d_bias, const float *d_res, int B, int H, int W,
                          int relu, int pool, const unsigned *d_amax, float inv_su, unsigned *d_amax_out, float *d_y,
                          hipStream_t st) {
    const int n_cu = cslam_cu_count();
    ARG_CHECK(n_cu > 0, "no HIP device");
    constexpr int NB = 8 / (COUT / 16);
    const int gxs = (int)ceil_div64(W, 16 * NB), gyb = (int)ceil_div64(H, 16);
    const int64_t nsb = (int64_t)B * gxs * gyb;
    ARG_CHECK(nsb < (1ll << 30), "too many tile blocks for one launch");
    ARG_CHECK((int64_t)B * H * W * 64 < (1ll << 31), "activation too large for the 32-bit patch offsets");
    const int grid_n = (int)(nsb < n_cu ? nsb : n_cu);
    dim3 grid((unsigned)grid_n), block(512);
    constexpr int NPIX = 18 * (16 * NB + 2), NL = (4 * NPIX + 511) / 512;
    constexpr int lds = 2 * NL * 512 * 16 + 36 * 16 * NB * 64;
    static float *zero16 = nullptr;                                 // 16 zero bytes per device the kernel has run on
    static int zero_dev = -1;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (!zero16 || zero_dev != dev) {
        HIP_TRY(hipMalloc((void **)&zero16, 256));                  // never freed: a captured graph may point at it
        HIP_TRY(hipMemset(zero16, 0, 256));
        zero_dev = dev;
    }
#define WH_LAUNCH_DB(R, P, D, BR) do { \
        HIP_TRY(hipFuncSetAttribute((const void *)wino4_fused_c64_h_kernel<COUT, R, P, D, BR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        hipLaunchKernelGGL((wino4_fused_c64_h_kernel<COUT, R, P, D, BR>), grid, block, lds, st, d_x, d_Uh, d_bias, d_res, \
                           H, W, gxs, gyb, (int)nsb, d_amax, inv_su, d_amax_out, zero16, d_y, (const unsigned *)nullptr, (const float *)nullptr, (const float *)nullptr, 0.0f); } while (0)
#define WH_LAUNCH_D(R, P, D) WH_LAUNCH_DB(R, P, D, WH_BR_DEFAULT)
#define WH_LAUNCH(R, P) WH_LAUNCH_DB(R, P, 0, WH_BR_DEFAULT)
#ifdef CSLAM_ABLATIONS
    const char *dbg_env = getenv("CSLAM_WFH_DBG");                  // timing-only ablations (wrong results), relu + pool form only: measurement build
    const int dbg = dbg_env ? atoi(dbg_env) : 0;
    if (dbg && relu) {
        if (pool) { switch (dbg) { case 1: WH_LAUNCH_D(true, true, 1); break; case 2: WH_LAUNCH_D(true, true, 2); break; case 4: WH_LAUNCH_D(true, true, 4); break;
                                   case 5: WH_LAUNCH_D(true, true, 5); break; case 8: WH_LAUNCH_D(true, true, 8); break; case 16: WH_LAUNCH_D(true, true, 16); break;
                                   case 6: WH_LAUNCH_D(true, true, 6); break; default: WH_LAUNCH_D(true, true, 31); } }
        else { switch (dbg) { case 1: WH_LAUNCH_D(true, false, 1); break; case 2: WH_LAUNCH_D(true, false, 2); break; case 4: WH_LAUNCH_D(true, false, 4); break;
                              case 5: WH_LAUNCH_D(true, false, 5); break; case 8: WH_LAUNCH_D(true, false, 8); break; case 16: WH_LAUNCH_D(true, false, 16); break;
                              case 6: WH_LAUNCH_D(true, false, 6); break; default: WH_LAUNCH_D(true, false, 31); } }
        return CSLAM_OK;
    }
#endif
    if (d_res) {                                                    // shortcut add (never pooled): its own instantiation
        const void *fr = relu ? (const void *)wino4_fused_c64_h_kernel<COUT, true, false, 0, WH_BR_DEFAULT, true>
                              : (const void *)wino4_fused_c64_h_kernel<COUT, false, false, 0, WH_BR_DEFAULT, true>;
        HIP_TRY(hipFuncSetAttribute(fr, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if (relu) hipLaunchKernelGGL((wino4_fused_c64_h_kernel<COUT, true, false, 0, WH_BR_DEFAULT, true>), grid, block, lds, st, d_x, d_Uh,
                                     d_bias, d_res, H, W, gxs, gyb, (int)nsb, d_amax, inv_su, d_amax_out, zero16, d_y, (const unsigned *)nullptr, (const float *)nullptr, (const float *)nullptr, 0.0f);
        else hipLaunchKernelGGL((wino4_fused_c64_h_kernel<COUT, false, false, 0, WH_BR_DEFAULT, true>), grid, block, lds, st, d_x, d_Uh,
                                d_bias, d_res, H, W, gxs, gyb, (int)nsb, d_amax, inv_su, d_amax_out, zero16, d_y, (const unsigned *)nullptr, (const float *)nullptr, (const float *)nullptr, 0.0f);
        return CSLAM_OK;
    }
#ifdef CSLAM_ABLATIONS
    const char *prof_env = getenv("CSLAM_WFH_PROF");
    if (prof_env && atoi(prof_env) && relu && !d_res) {             // per-phase cycle counts of workgroup 0 (conv1_2's / conv2_1's forms)
        if (COUT == 64 && pool) { WH_LAUNCH_DB(true, true, 32, WH_BR_DEFAULT); return CSLAM_OK; }
        if (COUT == 128 && !pool) { WH_LAUNCH_DB(true, false, 32, WH_BR_DEFAULT); return CSLAM_OK; }
    }
#endif
    if (relu && pool) WH_LAUNCH(true, true);
    else if (relu) WH_LAUNCH(true, false);
    else if (pool) WH_LAUNCH(false, true);
    else WH_LAUNCH(false, false);
#undef WH_LAUNCH
#undef WH_LAUNCH_D
#undef WH_LAUNCH_DB
    return CSLAM_OK;
}

CSLAM_API int cslam_wino4_fused_c64_h_dev(const float *d_x, const void *d_Uh, const float *d_bias, const float *d_residual,
                                          int B, int H, int W, int Cout, int relu, int pool, const unsigned *d_amax,
                                          float inv_su, unsigned *d_amax_out, float *d_y, void *stream) {
    PTR_DEVICE(d_x);
    ARG_CHECK(d_x && d_Uh && d_y && d_amax, "NULL argument");
    ARG_CHECK(B >= 1 && H >= 1 && W >= 1, "empty map");
    ARG_CHECK(Cout == 64 || Cout == 128, "Cout must be 64 or 128");
    ARG_CHECK(!pool || ((H % 2) == 0 && (W % 2) == 0), "pooling needs even H and W");
    ARG_CHECK(!(pool && d_residual), "a shortcut cannot be added to a pooled output");
    ARG_CHECK(inv_su > 0.0f, "inv_su must be positive");
    hipStream_t st = (hipStream_t)stream;
    const int rc = Cout == 64
        ? launch_fused_h<64>(d_x, (const unsigned *)d_Uh, d_bias, d_residual, B, H, W, relu, pool, d_amax, inv_su, d_amax_out, d_y, st)
        : launch_fused_h<128>(d_x, (const unsigned *)d_Uh, d_bias, d_residual, B, H, W, relu, pool, d_amax, inv_su, d_amax_out, d_y, st);
    if (rc != CSLAM_OK) return rc;
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

// The stem form: conv 3 -> 64 + bias + ReLU -> conv 64 -> 64 + bias + ReLU (+ MaxPool2d) in one launch.
static int launch_stem_h(const float *d_x0, const unsigned *d_w1, const float *d_b1, const float *d_sumw, float inv_sw, const unsigned *d_Uh,
                         const float *d_bias, int B, int H, int W, int pool, const unsigned *d_amax, float inv_su,
                         unsigned *d_amax_out, float *d_y, hipStream_t st) {
    int dev = 0, n_cu = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    const int gxs = (int)ceil_div64(W, 32), gyb = (int)ceil_div64(H, 16);
    const int64_t nsb = (int64_t)B * gxs * gyb;
    ARG_CHECK(nsb < (1ll << 30), "too many tile blocks for one launch");
    ARG_CHECK((int64_t)B * H * W * 3 < (1ll << 31), "image batch too large for the 32-bit patch offsets");
    const int grid_n = (int)(nsb < n_cu ? nsb : n_cu);
    constexpr int NPIX = 18 * 34, NL = (4 * NPIX + 511) / 512;
    constexpr int lds = NL * 512 * 16 + 36 * 32 * 64 + (ST_IMG + ST_PAD) * 4 + NL * 512 * 4 + (2048 + 64) * 4;
    static float *zero16[16] = {nullptr};                           // 16 zero bytes per device
    ARG_CHECK(dev < 16, "device index");
    if (!zero16[dev]) {
        HIP_TRY(hipMalloc((void **)&zero16[dev], 256));             // never freed: a captured graph may point at it
        HIP_TRY(hipMemset(zero16[dev], 0, 256));
    }
    dim3 grid((unsigned)grid_n), block(512);
#define ST_LAUNCH(P, D) do { \
        HIP_TRY(hipFuncSetAttribute((const void *)wino4_fused_c64_h_kernel<64, true, P, D, WH_BR_DEFAULT, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        hipLaunchKernelGGL((wino4_fused_c64_h_kernel<64, true, P, D, WH_BR_DEFAULT, false, true>), grid, block, lds, st, d_x0, d_Uh, d_bias, \
                           (const float *)nullptr, H, W, gxs, gyb, (int)nsb, d_amax, inv_su, d_amax_out, zero16[dev], d_y, d_w1, d_b1, d_sumw, inv_sw); } while (0)
#ifdef CSLAM_ABLATIONS
    const char *prof_env = getenv("CSLAM_WFH_PROF");
    if (prof_env && atoi(prof_env) && pool) { ST_LAUNCH(true, 32); return CSLAM_OK; }
#endif
    if (pool) ST_LAUNCH(true, 0);
    else ST_LAUNCH(false, 0);
#undef ST_LAUNCH
    return CSLAM_OK;
}

CSLAM_API int cslam_wino4_stem_c64_h_dev(const float *d_x0, const void *d_w1, const float *d_b1, const float *d_sumw, float inv_sw,
                                         const void *d_Uh, const float *d_bias, int B, int H, int W, int pool,
                                         const unsigned *d_amax_x0, float inv_su, unsigned *d_amax_out, float *d_y, void *stream) {
    PTR_DEVICE(d_x0);
    ARG_CHECK(d_x0 && d_w1 && d_sumw && d_Uh && d_y && d_amax_x0, "NULL argument");
    ARG_CHECK(B >= 1 && H >= 1 && W >= 1, "empty map");
    ARG_CHECK(!pool || ((H % 2) == 0 && (W % 2) == 0), "pooling needs even H and W");
    ARG_CHECK(inv_sw > 0.0f && inv_su > 0.0f, "scales must be positive");
    const int rc = launch_stem_h(d_x0, (const unsigned *)d_w1, d_b1, d_sumw, inv_sw, (const unsigned *)d_Uh, d_bias, B, H, W, pool,
                                 d_amax_x0, inv_su, d_amax_out, d_y, (hipStream_t)stream);
    if (rc != CSLAM_OK) return rc;
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

/* diagnostics: where CSLAM_WFH_PROF=1 launches leave their per-phase cycle counts (16 x 8 bytes, device memory; NULL = off) */
CSLAM_API int cslam_debug_wfh_prof_dev(void *d_buf16) {
    unsigned long long *p = (unsigned long long *)d_buf16;
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(wfh_prof), &p, sizeof(p)));
    return CSLAM_OK;
}
