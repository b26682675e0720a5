// conv_stem_direct_h.hip -- VGG-16 conv1_1 (3 -> 64, ReLU) + conv1_2 (64 -> 64, ReLU, MaxPool2d) as ONE direct (non-Winograd)
// kernel on the fp16 matrix pipe with fp32-grade results (gfx950): cslam/vpr/netvlad.py:163-171,227.
//
// Why (round 4).  The one-kernel F(4x4) form of this pair (wino_fused_h.hip, STEM) does a quarter of the multiplications but is
// bound by the delivery of its Winograd-domain weights -- 36 matrices of 64 x 64, each used by the 32 tiles of an iteration only --
// through the compute unit's vector path (matrix pipe busy 0.15, 25 spilled registers, 2.14 x its algorithmic bytes from L2).  The
// direct form has NINE 64 x 64 matrices, and as exact fp16 pairs they are 147 KB: they fit the REGISTER FILES of a compute unit.
//
// Work decomposition: the K dimension (64 input channels x 9 taps) is split over the four waves of a workgroup BY INPUT CHANNEL:
// wave w owns channels 16 w .. 16 w + 15 of the 64-channel map between the two layers, all nine taps, all 64 output channels:
//   * its share of conv1_2's weights -- [9 taps][2 x 32 output channels][hi | lo] MFMA A fragments = 144 registers -- is loaded
//     ONCE per kernel and never moves again: no weight stream, no weight ring in LDS, no stage barriers;
//   * it computes ITS 16 channels of conv1_1 (+ bias + ReLU) on the block's patch itself (v_mfma_f32_16x16x32_f16 over the 27
//     taps of the 3-channel image, as wino_fused_h.hip's stem), splits them into exact fp16 pairs and keeps them in a patch of
//     its own in LDS: the patch is wave-private, so producing and consuming it needs no workgroup barrier either;
//   * nine shifted reads of that patch (an immediate offset per tap) feed 4 pixel tiles x 2 channel tiles x 3 products of
//     v_mfma_f32_32x32x16_f16: 8 fragment reads per 24 MFMAs (the eight-wave 128-channel direct kernel: 8 per 12);
//   * the four partial sums of a block (128 pixels x 64 channels each) meet in LDS once per block: wave w sends three pixel
//     tiles and finishes the fourth (bias, ReLU, 2 x 2 max, NHWC stores, max |y|).  Two barriers per block.
// One wave per SIMD, 512 registers: 144 weights + 128 accumulators + fragments.  Block = 8 x 16 output pixels, persistent
// workgroups.  HBM: the 3-channel image in (x 1.4 halo, from L2), the pooled activation out; nothing else.
//
// Arithmetic (as conv_direct_h.hip / wino_fused_h.hip): image scaled by the power of two s1 (max |image| s1 <= 2^14), first-layer
// weights by 1 / inv_sw1: acc1 = xh wh + xl wh + xh wl, a1 = relu(acc1 inv_sw1 / s1 + b1) (exact rescale).  a1 is scaled by the
// power of two s_x derived from the RIGOROUS bound max_co(|b1[co]| + max |image| sum |w1[co]|) (its true maximum is only known
// after the kernel has run), split into hi + lo; w2 split offline (`stem_direct_pair_weights`): acc2 = wh xh + wh xl + wl xh,
// y = pool(relu(acc2 inv_sw2 / s_x + b2)).
#include <stdlib.h>
#include <type_traits>
#include <hip/hip_fp16.h>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define SD_PP 80                       // bytes per patch pixel of a wave: [hi 16 halfs | lo 16 halfs | 16 pad]: 5 sixteen-byte slots, coprime with 16
#define SD_RP 1536                     // bytes per patch row (18 pixels = 1440, to a multiple of 256: the two rows of a ds_read_b128 lane
                                       // group then fall on 16 distinct slots for every tap)
#define SD_PATCHB (10 * SD_RP)         // 15 360 per wave
#define SD_IW 20
#define SD_IMG (12 * SD_IW * 3)        // dwords of the packed [hi | lo] image patch [12][20][3]
#define SD_XTRA 9216                   // per wave: what of its 24 KB of partial sums does not fit its (then dead) patch
#define SD_NPIX 180                    // 10 x 18 patch pixels

struct StemDirectArgs {
    const float *x0; const unsigned *w1; const float *b1; const float *sumw; float inv_sw1;
    const f16x8 *w2; const float *bias; float inv_sw2;
    int B, H, W, gxb, gyb, nblk;
    const unsigned *amax_in; unsigned *amax_out; float *y;
};

#ifdef CSLAM_ABLATIONS
__device__ unsigned long long *sd_prof = nullptr;             // measurement build: [main loop (+ the next block's first layer), exchange writes + barrier, image + exchange reads, epilogue, blocks] ticks of wave 0 / workgroup 0
extern "C" __attribute__((visibility("default"))) int cslam_debug_sd_prof_dev(void *d_buf) {
    unsigned long long *q = (unsigned long long *)d_buf;
    return hipMemcpyToSymbol(HIP_SYMBOL(sd_prof), &q, sizeof(q)) == hipSuccess ? 0 : -2;
}
#define SD_PROF 1
#else
#define SD_PROF 0
#endif

__device__ __forceinline__ __amdgpu_buffer_rsrc_t sd_rsrc(const char *base, int64_t bytes) {
    const uint64_t a = (uint64_t)base;
    const uint64_t u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)a);
    const int n = __builtin_amdgcn_readfirstlane((int)(bytes > 0x7ffffff0 ? 0x7ffffff0 : bytes));
    return __builtin_amdgcn_make_buffer_rsrc((void *)u, 0, n, 0x00020000);
}
__device__ __forceinline__ unsigned sd_pack(float v) {                    // [fp16(v) | fp16(v - fp16(v)) << 16]
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    return (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
}

// max over lanes l, l ^ 1, l ^ 16, l ^ 17 (the four pixels of a 2 x 2 pooling window) without going through LDS: a DPP quad
// permutation and gfx950's v_permlane16_swap (rows 1 / 3 of the first operand <-> rows 0 / 2 of the second: of two copies of v one ends
// up holding the even rows twice, the other the odd rows).  As `__shfl_xor` each step was a ds_bpermute_b32 with a full LDS round
// trip behind it: 64 of them per block, 6 000 of a block's 18 000 cycles with one wave per SIMD.
__device__ __forceinline__ float sd_pool4(float v) {
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1, 0, 3, 2]
    asm("v_max_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(a));   // (fmaxf would first canonicalise both operands: two more instructions per maximum)
    // (as inline assembly: hipcc 7.2 folds the two results of __builtin_amdgcn_permlane16_swap(v, v) into ONE value and drops the maximum
    // that follows; the s_nop covers the VALU-write -> permlane-read hazard the compiler cannot see in here)
    float b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v), "+v"(b));
    asm("v_max_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(b));
    return v;
}

template <bool POOL, int PATCHES>
__global__ __launch_bounds__(256, 1) void conv_stem_direct_h_kernel(StemDirectArgs p) {
    extern __shared__ __attribute__((aligned(16))) char sd_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int gq = lane >> 4, l15 = lane & 15;
    // LDS: [PATCHES][4 waves] patches | [4 waves] exchange overflow | image patch
    char *const s_xtra_all = sd_smem + PATCHES * 4 * SD_PATCHB;
    unsigned *const s_img = (unsigned *)(s_xtra_all + 4 * SD_XTRA);
    float *const s_bias = (float *)(s_img + SD_IMG);          // the second layer's 64 bias values (read per block: 32 registers otherwise)
    if (tid < 64) s_bias[tid] = p.bias ? p.bias[tid] : 0.0f;   // visible behind the first block's barrier

    // ---- scales
    const float a0 = fminf(fmaxf(__uint_as_float(*p.amax_in), 1e-30f), 1e30f);
    int e_;
    (void)frexpf(16384.0f / a0, &e_);
    const float s1 = ldexpf(1.0f, e_ - 1);
    const float inv1 = p.inv_sw1 / s1;
    float bound = 0.0f;
    for (int c = 0; c < 64; ++c) bound = fmaxf(bound, fabsf(p.b1 ? p.b1[c] : 0.0f) + a0 * p.sumw[c]);
    bound = fminf(fmaxf(bound, 1e-30f), 1e30f);
    (void)frexpf(32752.0f / bound, &e_);
    const float sx = ldexpf(1.0f, e_ - 1);
    const float inv2 = p.inv_sw2 / sx;

    const int n_mine = (p.nblk - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (n_mine <= 0) return;

    // ---- this wave's operands, register-resident for the whole kernel
    f16x8 wr[9][2][2];                                         // [tap][32-channel tile][hi | lo]: A fragments of conv1_2
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) wr[tap][n][hl] = p.w2[(((wave * 9 + tap) * 2 + n) * 2 + hl) * 64 + lane];
    const u32x4 w1h = ((const u32x4 *)p.w1)[(wave * 2 + 0) * 64 + lane], w1l = ((const u32x4 *)p.w1)[(wave * 2 + 1) * 64 + lane];
    const f16x8 W1h = __builtin_bit_cast(f16x8, w1h), W1l = __builtin_bit_cast(f16x8, w1l);
    float b1v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) b1v[r] = p.b1 ? p.b1[16 * wave + 4 * gq + r] : 0.0f;

    // ---- image patch: element e = j * 256 + tid of the planar [3][12][20] patch whose origin is two pixels up / left of the block
    int e_dst[3], e_rc[3];                                     // destination dword in s_img (-1: none); (plane << 16 | row << 8 | column)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int e = j * 256 + tid;
        const int ci = (e >= 240) + (e >= 480);
        const int rem = e - 240 * ci;
        const int r = (rem * 3277) >> 16, c = rem - 20 * r;     // rem / 20 for rem < 240
        e_dst[j] = e < SD_IMG ? (r * SD_IW + c) * 3 + ci : -1;
        e_rc[j] = (ci << 16) | (r << 8) | c;
    }
    struct Blk { int img, by, bx; };
    auto decode_blk = [&](int bi) {
        int blk = (int)blockIdx.x + bi * (int)gridDim.x;
        blk = blk < p.nblk ? blk : p.nblk - 1;
        const int per_img = p.gxb * p.gyb;
        Blk b;
        b.img = blk / per_img;
        const int rem = blk - b.img * per_img;
        b.by = rem / p.gxb; b.bx = rem - b.by * p.gxb;
        return b;
    };
    float raw[3];
    auto img_load = [&](const Blk &b) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ci = e_rc[j] >> 16, r = (e_rc[j] >> 8) & 255, c = e_rc[j] & 255;
            const int gy = b.by * 8 - 2 + r, gx = b.bx * 16 - 2 + c;
            const bool in = (e_dst[j] >= 0) & (gy >= 0) & (gy < p.H) & (gx >= 0) & (gx < p.W);
            raw[j] = in ? p.x0[((int64_t)(b.img * 3 + ci) * p.H + gy) * p.W + gx] : 0.0f;
        }
    };
    auto img_store = [&]() {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (e_dst[j] >= 0) s_img[e_dst[j]] = sd_pack(raw[j] * s1);
    };

    // ---- first layer: K slots of the 16x16x32 MFMA (lane group gq, slot j): gq < 3: tap (ky = gq, kx = j / 3, ci = j % 3), eight
    // CONSECUTIVE dwords of the [row][col][channel] image; gq = 3, j < 3: the ninth tap (kx = 2, ci = 2) of row ky = j (stride one
    // image row); j >= 3: zero weights (`stem_pair_weights`), the slot re-reads j = 2 (any finite value)
    int st_joff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) st_joff[j] = gq < 3 ? gq * (3 * SD_IW) + j : 8 + (j < 3 ? j : 2) * (3 * SD_IW);
    // a1 s_x = relu(acc1 (inv1 s_x) + b1 s_x): the power of two s_x goes through the rounding of the sum unchanged
    const float k1 = inv1 * sx;
    float b1s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) b1s[r] = b1v[r] * sx;
    // The first layer of a block runs in SIX steps of two 16-pixel tiles (12 tiles = 192 >= 180 patch pixels), each in two halves: the
    // image reads of step g (`conv1_read`: 16 ds_read_b32) and its arithmetic (`conv1_finish`: 6 MFMAs, ~100 VALU instructions, 4
    // ds_write_b64).  In the steady state the steps of block i + 1 are interleaved with the taps of block i's second layer: with ONE
    // wave per SIMD nobody else fills the matrix pipe while this wave does vector work (as separate phases: first layer 5.2k, main
    // loop 7.1k, exchange + epilogue 5.6k cycles per block).
    unsigned pk[2][8];
    auto conv1_read = [&](int g) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int pp = 16 * (2 * g + i) + l15;
            pp = pp < SD_NPIX ? pp : SD_NPIX - 1;
            const int pr = (pp * 3641) >> 16, pc = pp - 18 * pr;                  // pp / 18 for pp < 4096
            const unsigned *ib = s_img + (pr * SD_IW + pc) * 3;
#pragma unroll
            for (int j = 0; j < 8; ++j) pk[i][j] = ib[st_joff[j]];
        }
    };
    auto conv1_finish = [&](int g, const Blk &b, char *patch) {
        const int gy0 = b.by * 8 - 1, gx0 = b.bx * 16 - 1;
        f32x4 c[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u32x4 ah, al;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                ah[d] = __builtin_amdgcn_perm(pk[i][2 * d + 1], pk[i][2 * d], 0x05040100u);  // (hi, hi) of slots 2d, 2d + 1
                al[d] = __builtin_amdgcn_perm(pk[i][2 * d + 1], pk[i][2 * d], 0x07060302u);  // (lo, lo)
            }
            const f16x8 Ph = __builtin_bit_cast(f16x8, ah), Pl = __builtin_bit_cast(f16x8, al);
            // weights as the A operand: D[m = channel][n = pixel], a lane holds channels 4 gq .. + 3 of pixel l15
            c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W1l, Ph, (f32x4)(0.0f), 0, 0, 0);
            c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W1h, Pl, c[i], 0, 0, 0);
            c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W1h, Ph, c[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pp = 16 * (2 * g + i) + l15;
            const int pr = (pp * 3641) >> 16, pc = pp - 18 * pr;
            const int gy = gy0 + pr, gx = gx0 + pc;
            const bool in = (gy >= 0) & (gy < p.H) & (gx >= 0) & (gx < p.W);      // outside the map: the second layer's zero padding
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = in ? fmaxf(c[i][r] * k1 + b1s[r], 0.0f) : 0.0f;
            const __half2 h01 = __floats2half2_rn(v[0], v[1]), h23 = __floats2half2_rn(v[2], v[3]);
            const float2 f01 = __half22float2(h01), f23 = __half22float2(h23);
            const __half2 l01 = __floats2half2_rn(v[0] - f01.x, v[1] - f01.y), l23 = __floats2half2_rn(v[2] - f23.x, v[3] - f23.y);
            if (pp < SD_NPIX) {
                char *d = patch + pr * SD_RP + pc * SD_PP + 8 * gq;
                *(uint2 *)d = make_uint2(*(const unsigned *)&h01, *(const unsigned *)&h23);
                *(uint2 *)(d + 32) = make_uint2(*(const unsigned *)&l01, *(const unsigned *)&l23);
            }
        }
    };

    // ---- second layer.  acc[a] = pixel tile t = (a + wave) & 3 of the block (rows 2 t, 2 t + 1): acc[0] is the tile this wave
    // finishes, acc[1..3] go to waves (wave + a) & 3 -- the rotation sits in the fragment addresses, every register index is a constant
    f32x16 acc[4][2];
    int pa_off[4];                                             // lane's patch offset for acc index a at tap (0, 0)
#pragma unroll
    for (int a = 0; a < 4; ++a) pa_off[a] = (2 * ((a + wave) & 3) + (l31 >> 4)) * SD_RP + (l31 & 15) * SD_PP + 16 * h;
    // fragments: ONE register set (32 registers, not 64: with 144 weight and 128 accumulator registers a second set spilled).  Per tap
    // the products run lo first: G1 = wh xl, then G2 = wl xh, G3 = wh xh.  The NEXT tap's lo fragments are requested behind G1 (they
    // have G2 + G3 = 512 matrix cycles to land), this tap's hi fragments at its top, under G1 (256 cycles).
    f16x8 fbh[4], fbl[4];
    auto read_hi = [&](auto tap_tag, const char *patch) {
        constexpr int TAP = decltype(tap_tag)::value;
        constexpr int OFF = (TAP / 3) * SD_RP + (TAP % 3) * SD_PP;
#pragma unroll
        for (int a = 0; a < 4; ++a) fbh[a] = *(const f16x8 *)(patch + pa_off[a] + OFF);
    };
    auto read_lo = [&](auto tap_tag, const char *patch) {
        constexpr int TAP = decltype(tap_tag)::value;
        constexpr int OFF = (TAP / 3) * SD_RP + (TAP % 3) * SD_PP + 32;
#pragma unroll
        for (int a = 0; a < 4; ++a) fbl[a] = *(const f16x8 *)(patch + pa_off[a] + OFF);
    };
    // One tap = two scheduling regions.  STEP < 6: step STEP of the NEXT block's first layer rides along -- its image reads under G1,
    // its arithmetic under G2 + G3, one MFMA of this tap between every handful of its vector instructions.
    auto tap_body = [&](auto tap_tag, auto next_tag, auto step_tag, const char *patch, const Blk &nblk, char *npatch) {
        constexpr int TAP = decltype(tap_tag)::value;
        constexpr int NEXT = decltype(next_tag)::value;
        constexpr int STEP = decltype(step_tag)::value;
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        read_hi(tap_tag, patch);
        if constexpr (STEP < 6) conv1_read(STEP);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[a][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[TAP][n][0], fbl[a], TAP == 0 ? z : acc[a][n], 0, 0, 0);
        if constexpr (STEP < 6) {
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);          // the hi fragments first
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // VALU (addresses)
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // DS read
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NEXT < 9) read_lo(next_tag, patch);
        if constexpr (STEP < 6) conv1_finish(STEP, nblk, npatch);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[a][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[TAP][n][1], fbh[a], acc[a][n], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[a][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[TAP][n][0], fbh[a], acc[a][n], 0, 0, 0);
        if constexpr (STEP < 6) {
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);          // the next tap's lo fragments first
#pragma unroll
            for (int i = 0; i < 22; ++i) {                              // 16 of this tap + 6 of the first layer
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // DS write
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#define SD_TAP(T) std::integral_constant<int, T>{}

    // ---- exchange: chunk c = ((a - 1) * 2 + n) * 4 + q of a wave's 24 (1 KB each, lane-linear float4): the first 15 in its own
    // patch (dead by then), the rest in its overflow area
    auto xchg_ptr = [&](int w, int c, const int cur) -> char * {
        return c < 15 ? sd_smem + (cur * 4 + w) * SD_PATCHB + c * 1024 + lane * 16
                      : s_xtra_all + w * SD_XTRA + (c - 15) * 1024 + lane * 16;
    };

    float my_amax = 0.0f;
    const int Ho = POOL ? p.H >> 1 : p.H, Wo = POOL ? p.W >> 1 : p.W;

    [[maybe_unused]] unsigned long long t_main = 0, t_xw = 0, t_xr = 0, t_epi = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    Blk cb = decode_blk(0);
    img_load(cb);
    img_store();
    __syncthreads();
    {
        char *const patch0 = sd_smem + wave * SD_PATCHB;
#pragma unroll
        for (int g = 0; g < 6; ++g) { conv1_read(g); conv1_finish(g, cb, patch0); __builtin_amdgcn_sched_barrier(0); }
    }
    Blk nb = decode_blk(1);
    img_load(nb);
    __syncthreads();                                           // everybody is done with block 0's image
    img_store();                                               // block 1's
    Blk nnb = decode_blk(2);
    img_load(nnb);                                             // block 2's: in flight until the first exchange
    for (int bi = 0; bi < n_mine; ++bi) {
        const int cur = bi & 1;
        char *const patch = sd_smem + (cur * 4 + wave) * SD_PATCHB;
        char *const npatch = sd_smem + ((cur ^ 1) * 4 + wave) * SD_PATCHB;
        __syncthreads();                                       // the next block's image is whole; everybody has read the last block's partial sums (they sit in `npatch`)
        if (SD_PROF) t1 = __builtin_amdgcn_s_memtime();
        read_lo(SD_TAP(0), patch);
#define SD_NONE std::integral_constant<int, 9>{}
        tap_body(SD_TAP(0), SD_TAP(1), SD_TAP(0), patch, nb, npatch); tap_body(SD_TAP(1), SD_TAP(2), SD_TAP(1), patch, nb, npatch);
        tap_body(SD_TAP(2), SD_TAP(3), SD_TAP(2), patch, nb, npatch); tap_body(SD_TAP(3), SD_TAP(4), SD_TAP(3), patch, nb, npatch);
        tap_body(SD_TAP(4), SD_TAP(5), SD_TAP(4), patch, nb, npatch); tap_body(SD_TAP(5), SD_TAP(6), SD_TAP(5), patch, nb, npatch);
        tap_body(SD_TAP(6), SD_TAP(7), SD_NONE, patch, nb, npatch); tap_body(SD_TAP(7), SD_TAP(8), SD_NONE, patch, nb, npatch);
        tap_body(SD_TAP(8), SD_TAP(9), SD_NONE, patch, nb, npatch);
        if (SD_PROF) { t2 = __builtin_amdgcn_s_memtime(); t_main += t2 - t1; }
        // ---- partial sums of the three pixel tiles other waves finish
#pragma unroll
        for (int a = 1; a < 4; ++a)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(float4 *)xchg_ptr(wave, ((a - 1) * 2 + n) * 4 + q, cur) =
                        make_float4(acc[a][n][4 * q], acc[a][n][4 * q + 1], acc[a][n][4 * q + 2], acc[a][n][4 * q + 3]);
        __syncthreads();
        if (SD_PROF) { t3 = __builtin_amdgcn_s_memtime(); t_xw += t3 - t2; }
        // the image of block bi + 2 (everybody is through block bi + 1's first layer), consumed HERE, in front of this block's stores:
        // loads and stores share vmcnt and return out of order, so a load consumed behind a store waits for the store (vmcnt(0)) -- at the
        // top of the next iteration that was an HBM write round trip per block
        img_store();
        const Blk n3b = decode_blk(bi + 3);
        img_load(n3b);
        // (eight reads in flight, then their 32 additions: left to itself hipcc issued ONE read at a time, each behind a full
        // lgkmcnt(0) -- 32 LDS round trips per block with nobody else on the SIMD to fill them)
#pragma unroll
        for (int a = 1; a < 4; ++a) {
            const int src = (wave - a) & 3;                    // whose acc[a] is my pixel tile
            float4 t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = *(const float4 *)xchg_ptr(src, (a - 1) * 8 + i, cur);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int n = i >> 2, q = i & 3;
                acc[0][n][4 * q] += t[i].x; acc[0][n][4 * q + 1] += t[i].y; acc[0][n][4 * q + 2] += t[i].z; acc[0][n][4 * q + 3] += t[i].w;
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 64, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (SD_PROF) { t4 = __builtin_amdgcn_s_memtime(); t_xr += t4 - t3; }
        // ---- epilogue of pixel tile `wave`: lane = pixel (row l31 >> 4, column l31 & 15), 16-byte runs of four channels
        {
            const int oy = cb.by * 8 + 2 * wave + (l31 >> 4), ox = cb.bx * 16 + (l31 & 15);
            // buffer stores, the image's output map = the buffer: a lane that does not store (three of a pooling window's four, pixels
            // beyond a ragged edge) gets an offset out of its range and is dropped by the hardware -- no branch around the stores, so the
            // compiler can COUNT them in its waits (behind a store under a branch every later wait became vmcnt(0): an HBM write round trip)
            bool store;
            int yoff;                                                                           // byte offset of channel 4 h of the lane's pixel
            if (POOL) {
                store = ((l31 & 17) == 0) && (oy >> 1) < Ho && (ox >> 1) < Wo;                   // the window's top-left lane
                yoff = (((oy >> 1) * Wo + (ox >> 1)) * 64 + 4 * h) * 4;
            } else {
                store = oy < p.H && ox < p.W;
                yoff = ((oy * p.W + ox) * 64 + 4 * h) * 4;
            }
            yoff = store ? yoff : 0x7fffffff;
            const __amdgpu_buffer_rsrc_t rsY = sd_rsrc((const char *)(p.y + (int64_t)cb.img * Ho * Wo * 64), (int64_t)Ho * Wo * 256);
            float4 bvs[2][4];                                  // the lane's channels 32 n + 8 q + 4 h .. + 3 (read together: see above)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q) bvs[n][q] = *(const float4 *)(s_bias + 32 * n + 8 * q + 4 * h);
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v = make_float4(acc[0][n][4 * q], acc[0][n][4 * q + 1], acc[0][n][4 * q + 2], acc[0][n][4 * q + 3]);
                    if (POOL) {
                        // 2 x 2 maximum first (lanes l ^ 1, l ^ 16), then the exact rescale, bias and ReLU on the survivor: all monotone
                        v.x = sd_pool4(v.x); v.y = sd_pool4(v.y); v.z = sd_pool4(v.z); v.w = sd_pool4(v.w);
                    }
                    const float4 bv = bvs[n][q];
                    v.x = fmaxf(v.x * inv2 + bv.x, 0.0f); v.y = fmaxf(v.y * inv2 + bv.y, 0.0f);
                    v.z = fmaxf(v.z * inv2 + bv.z, 0.0f); v.w = fmaxf(v.w * inv2 + bv.w, 0.0f);
                    my_amax = fmaxf(my_amax, store ? fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)) : 0.0f);
                    u32x4 bits;
                    bits.x = __float_as_uint(v.x); bits.y = __float_as_uint(v.y); bits.z = __float_as_uint(v.z); bits.w = __float_as_uint(v.w);
                    __builtin_amdgcn_raw_buffer_store_b128(bits, rsY, yoff, (32 * n + 8 * q) * 4, 0);
                }
        }
        if (SD_PROF) t_epi += __builtin_amdgcn_s_memtime() - t4;
        cb = nb;
        nb = nnb;
        nnb = n3b;
    }
#ifdef CSLAM_ABLATIONS
    if (sd_prof && blockIdx.x == 0 && tid == 0) { sd_prof[0] = t_main; sd_prof[1] = t_xw; sd_prof[2] = t_xr; sd_prof[3] = t_epi; sd_prof[4] = (unsigned long long)n_mine; }
#endif

    if (p.amax_out) {
        __syncthreads();
        unsigned *s_amax = (unsigned *)sd_smem;
        if (tid == 0) *s_amax = 0u;
        __syncthreads();
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) my_amax = fmaxf(my_amax, __shfl_xor(my_amax, o, 64));
        if (lane == 0) atomicMax(s_amax, __float_as_uint(my_amax));
        __syncthreads();
        if (tid == 0 && *s_amax > *(volatile unsigned *)p.amax_out) atomicMax(p.amax_out, *s_amax);
    }
}

/* y = [pool](relu(conv3x3(relu(conv3x3(x0, w1) + b1), w2) + bias)): x0 planar [B][3][H][W] float32, y NHWC [B][H'][W'][64].
 * d_w1 / inv_sw1 / d_sumw = `stem_pair_weights` (vpr/winograd.py); d_w2r / inv_sw2 = `stem_direct_pair_weights`: the second layer's
 * weights as exact fp16 pairs in MFMA-fragment order [4 channel quarters][9 taps][2][hi | lo][64 lanes][8]; d_amax_x0: 4-byte slot
 * holding the bits of (a bound of) max |x0|; d_amax_out (or NULL): zeroed slot that receives max |y|. */
CSLAM_API int cslam_conv_stem_direct_h_dev(const float *d_x0, const void *d_w1, const float *d_b1, const float *d_sumw, float inv_sw1,
                                           const void *d_w2r, const float *d_bias, float inv_sw2, int B, int H, int W, int pool,
                                           const unsigned *d_amax_x0, unsigned *d_amax_out, float *d_y, void *stream) {
    PTR_DEVICE(d_x0);
    ARG_CHECK(d_x0 && d_w1 && d_sumw && d_w2r && d_y && d_amax_x0, "NULL argument");
    ARG_CHECK(B >= 1 && H >= 1 && W >= 1, "empty map");
    ARG_CHECK(!pool || ((H % 2) == 0 && (W % 2) == 0), "pooling needs even H and W");
    ARG_CHECK(inv_sw1 > 0.0f && inv_sw2 > 0.0f, "scales must be positive");
    ARG_CHECK((int64_t)H * W * 256 < 0x7ffffff0ll, "one image's output map must stay below 2 GiB (32-bit buffer offsets)");
    StemDirectArgs a;
    a.x0 = d_x0; a.w1 = (const unsigned *)d_w1; a.b1 = d_b1; a.sumw = d_sumw; a.inv_sw1 = inv_sw1;
    a.w2 = (const f16x8 *)d_w2r; a.bias = d_bias; a.inv_sw2 = inv_sw2;
    a.B = B; a.H = H; a.W = W;
    a.gxb = (int)ceil_div64(W, 16); a.gyb = (int)ceil_div64(H, 8);
    const int64_t nblk = (int64_t)B * a.gxb * a.gyb;
    ARG_CHECK(nblk < (1ll << 30), "too many blocks for one launch");
    a.nblk = (int)nblk;
    a.amax_in = d_amax_x0; a.amax_out = d_amax_out; a.y = d_y;
    const int n_cu = cslam_cu_count();
    ARG_CHECK(n_cu > 0, "no HIP device");
    const int grid = (int)(nblk < n_cu ? nblk : n_cu);
    hipStream_t st = (hipStream_t)stream;
    constexpr int lds = 2 * 4 * SD_PATCHB + 4 * SD_XTRA + SD_IMG * 4 + 256;
#define SD_LAUNCH(P) do { \
        static DeviceOnce once; int once_dev; \
        if (once.todo(&once_dev)) { \
            HIP_TRY(hipFuncSetAttribute((const void *)conv_stem_direct_h_kernel<P, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
            once.done(once_dev); } \
        hipLaunchKernelGGL((conv_stem_direct_h_kernel<P, 2>), dim3(grid), dim3(256), lds, st, a); } while (0)
    if (pool) SD_LAUNCH(true);
    else SD_LAUNCH(false);
#undef SD_LAUNCH
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}
