// conv_stem_direct_h.hip -- VGG-16 conv1_1 (3 -> 64, ReLU) + conv1_2 (64 -> 64, ReLU, MaxPool2d) as ONE direct (non-Winograd)
// kernel on the fp16 matrix pipe with fp32-grade results (gfx950): cslam/vpr/netvlad.py:163-171,227.
//
// Why (round 4).  The one-kernel F(4x4) form of this pair (wino_fused_h.hip, STEM) does a quarter of the multiplications but is
// bound by the delivery of its Winograd-domain weights -- 36 matrices of 64 x 64, each used by the 32 tiles of an iteration only --
// through the compute unit's vector path (matrix pipe busy 0.15, 25 spilled registers, 2.14 x its algorithmic bytes from L2).  The
// direct form has NINE 64 x 64 matrices, and as exact fp16 pairs they are 147 KB: they fit the REGISTER FILES of a compute unit.
//
// Work decomposition: four waves per workgroup, one per SIMD, 512 registers each.  Wave w owns OUTPUT channels 16 w .. 16 w + 15
// of the second layer for all 128 pixels of an 8 x 16 block:
//   * its share of conv1_2's weights -- [9 taps][2 x 32 input channels][hi | lo] A fragments of v_mfma_f32_16x16x32_f16 = 144
//     registers -- is loaded ONCE per kernel and never moves again: no weight stream, no weight ring in LDS, no stage barriers;
//   * the 10 x 18-pixel patch of the 64-channel map between the layers lives in LDS as exact fp16 pairs, shared by the four waves,
//     double buffered; wave w computes channels 16 w .. + 15 of it (conv1_1 + bias + ReLU on the 16x16x32 MFMA over the 27 taps of
//     the 3-channel image, as wino_fused_h.hip's stem) for block i + 1 WHILE it multiplies block i -- with one wave per SIMD nobody
//     else fills the matrix pipe, so the first layer's vector work is interleaved into the second layer's MFMA stream by hand;
//   * a B fragment = one patch row (16 pixels) x 32 channels at a column shift dx: it serves the THREE taps (dy = 0, 1, 2) that
//     read this patch row for three different output rows -- 2 fragment reads per up to 9 MFMAs;
//   * no partial sums: a wave finishes its own 16 channels (2 x 2 max = one in-lane maximum of two row accumulators and one DPP
//     quad permutation; bias, ReLU, NHWC stores, max |y|).  ONE barrier per block.
// (First form of this file, measured and replaced: the K dimension split over the four waves by input channel, each wave with a
// private 16-channel patch and 32x32x16 MFMAs -- four partial sums per block met in LDS: 96 KB of ds_write_b128 + 24 reads + 96
// additions + v_permlane16_swap pooling per block, 5 of a block's 15 thousand cycles: 2.93 ms per 256 frames.)
// HBM: the 3-channel image in (x 1.4 halo, from L2), the pooled activation out; nothing else.
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
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define SD_PP 320                      // bytes per patch pixel: [hi 64 halfs | lo 64 halfs | 64 pad] = 20 sixteen-byte slots, and the 16-byte chunk c
                                       // (8 channels) of a half sits at chunk c ^ SD_SWZ(pixel column).  Found by search over pitch and
                                       // swizzle (profiles/r04_v55_*): every ds_read_b128 lane group of a fragment read (8 pixels of K group
                                       // g, the other 8 of K group g + 1) falls on 16 distinct slots for every column shift, K step and half,
                                       // and the first layer's ds_write_b64 (16 consecutive pixels, one 8-byte quarter-chunk each) collide
                                       // 2.5-way on average -- 288 bytes without swizzle: the same reads, 3.75-way writes, 1.1k cycles per block
#define SD_SWZ(pc) (((pc) >> 1) & 3)
#define SD_RP (18 * SD_PP)             // 5 760 bytes per patch row
#define SD_PATCHB (10 * SD_RP)         // 57 600 per buffer
#define SD_IW 20
#define SD_IMG (12 * SD_IW * 3)        // dwords of the packed [hi | lo] image patch [12][20][3]
#define SD_T9 (20 * 17)                // ... and of its third channel once more, column-major [20][17] (rows 12 .. 16: zeros): the ninth tap
#define SD_IMGB (SD_IMG + SD_T9)       // dwords per image buffer
#define SD_NPIX 180                    // 10 x 18 patch pixels
#define SD_LDS (2 * SD_PATCHB + 2 * SD_IMGB * 4)

struct StemDirectArgs {
    const float *x0; const unsigned *w1; const float *b1; const float *sumw; float inv_sw1;
    const f16x8 *w2; const float *bias; float inv_sw2;
    int B, H, W, gxb, gyb, nblk;
    const unsigned *amax_in; unsigned *amax_out; float *y;
};

#ifdef CSLAM_ABLATIONS
__device__ unsigned long long *sd_prof = nullptr;             // measurement build: [six columns (+ the next block's first layer), epilogue, barrier, blocks] ticks of wave 0 / workgroup 0
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
// v - (float)half HI of the packed pair h: one v_fma_mix_f32 (the fp16 operand is read straight out of the packed register; written as
// fmaf((float)half, -1, v) hipcc 7.2 converts the half back to float and subtracts: three instructions per value)
template <int HI>
__device__ __forceinline__ float sd_sub_half(float v, __half2 h) {
    float d;
    const unsigned hb = *(const unsigned *)&h;
    if (HI) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hb), "v"(v));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hb), "v"(v));
    return d;
}
__device__ __forceinline__ float sd_max(float a, float b) {   // (fmaxf first canonicalises both operands: two more instructions per maximum)
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// max(v, v of lane ^ 1): a DPP quad permutation (as `__shfl_xor` it is a ds_bpermute_b32 with an LDS round trip behind it)
__device__ __forceinline__ float sd_max_xor1(float v) {
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1, 0, 3, 2]
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(a));   // (fmaxf would first canonicalise both operands: two more instructions per maximum)
    return r;
}

// DBG (builds with -DCSLAM_ABLATIONS only; WRONG results, timing): 1 = no first layer inside the loop, 2 = no fragment reads after a block's
// first two, 4 = no epilogue stores, 8 = first layer without its image reads, 16 = ... without its patch stores
template <bool POOL, int DBG = 0>
__global__ __launch_bounds__(256, 1) void conv_stem_direct_h_kernel(StemDirectArgs p) {
    extern __shared__ __attribute__((aligned(16))) char sd_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gq = lane >> 4, l15 = lane & 15;
    // LDS: [2] patches | [2] image patches
    unsigned *const s_img_all = (unsigned *)(sd_smem + 2 * SD_PATCHB);

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

    // Blocks to workgroups, XCD-aware: workgroup w runs on XCD w % 8 (round-robin dispatch), and every XCD has an L2 of its own.  Each XCD
    // takes one CONTIGUOUS eighth of the blocks, its workgroups walk it side by side -- the blocks whose image patches overlap (left /
    // right neighbours, and the row below 14 blocks later) are then read through ONE L2 at about the same time.  (Block b -> workgroup
    // b % grid: every halo line was fetched by two or three XCDs, 723 MB of L2 misses for a 154 MB image input.)
    const bool by_xcd = (gridDim.x & 7) == 0;
    const int wg_xcd = by_xcd ? (int)blockIdx.x & 7 : 0, wg_j = by_xcd ? (int)blockIdx.x >> 3 : (int)blockIdx.x;
    const int wg_per = by_xcd ? (int)gridDim.x >> 3 : (int)gridDim.x;
    const int per_xcd = by_xcd ? (p.nblk + 7) >> 3 : p.nblk;
    const int blk_beg = wg_xcd * per_xcd;
    const int blk_cnt = min(per_xcd, p.nblk - blk_beg);        // blocks of this XCD's range (may be <= 0 for the last ones of a tiny launch)
    const int n_mine = blk_cnt > wg_j ? (blk_cnt - wg_j + wg_per - 1) / wg_per : 0;
    if (n_mine <= 0) return;

    // ---- this wave's operands, register-resident for the whole kernel
    f16x8 wr[9][2][2];                                         // [tap][32-channel K step][hi | lo]: A fragments of conv1_2 (16 output channels x 32)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) {
                wr[tap][ks][hl] = p.w2[(((wave * 9 + tap) * 2 + ks) * 2 + hl) * 64 + lane];
                // into the accumulation half of the register file: they are MFMA operands and nothing else; left to itself hipcc keeps them
                // in the 256 architectural registers and moves every accumulator the vector unit touches through v_accvgpr_read instead
                asm volatile("" : "+a"(wr[tap][ks][hl]));
            }
    const u32x4 w1h = ((const u32x4 *)p.w1)[(wave * 2 + 0) * 64 + lane], w1l = ((const u32x4 *)p.w1)[(wave * 2 + 1) * 64 + lane];
    const f16x8 W1h = __builtin_bit_cast(f16x8, w1h), W1l = __builtin_bit_cast(f16x8, w1l);
    // a1 s_x = relu(acc1 (inv1 s_x) + b1 s_x): the power of two s_x goes through the rounding of the sum unchanged
    const float k1 = inv1 * sx;
    float b1s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) b1s[r] = (p.b1 ? p.b1[16 * wave + 4 * gq + r] : 0.0f) * sx;
    float4 bv2;                                                // the second layer's bias of the lane's channels 16 wave + 4 gq .. + 3
    bv2 = p.bias ? *(const float4 *)(p.bias + 16 * wave + 4 * gq) : make_float4(0.f, 0.f, 0.f, 0.f);

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
        int blk = wg_j + bi * wg_per;
        blk = blk_beg + (blk < blk_cnt ? blk : blk_cnt - 1);
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
    auto img_store = [&](unsigned *s_img) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (e_dst[j] >= 0) {
                const unsigned v = sd_pack(raw[j] * s1);
                s_img[e_dst[j]] = v;
                if ((e_rc[j] >> 16) == 2) s_img[SD_IMG + (e_rc[j] & 255) * 17 + ((e_rc[j] >> 8) & 255)] = v;
            }
    };

    // ---- first layer: K slots of the 16x16x32 MFMA (lane group gq, slot j): gq < 3: tap (ky = gq, kx = j / 3, ci = j % 3), eight
    // CONSECUTIVE dwords of the [row][col][channel] image; gq = 3, j < 3: the ninth tap (kx = 2, ci = 2) of row ky = j; j >= 3: zero
    // weights (`stem_pair_weights`) on any finite value.  The third channel is kept a second time, column-major, so that group 3 reads
    // eight consecutive dwords as well (column pc + 2, rows pr .. pr + 7): every lane's eight slots are ONE base address + immediates
    // (with a per-slot stride that depended on the lane group every slot cost an address addition: 16 of a tile's 40 instructions).
    // Per tile t (pixels 16 t .. 16 t + 15 of the 10 x 18 patch; the last four of tile 11 do not exist and repeat pixel 179): the
    // lane's source offset in an image buffer and its destination offset in a patch, both in bytes, precomputed
    int c1_src[12], c1_dst[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) {
        int pp = 16 * t + l15;
        pp = pp < SD_NPIX ? pp : SD_NPIX - 1;
        const int pr = (pp * 3641) >> 16, pc = pp - 18 * pr;                      // pp / 18 for pp < 4096
        c1_src[t] = (gq < 3 ? ((pr + gq) * SD_IW + pc) * 3 : SD_IMG + (pc + 2) * 17 + pr) * 4;
        c1_dst[t] = pr * SD_RP + pc * SD_PP + (((2 * wave + (gq >> 1)) ^ SD_SWZ(pc)) << 4) + (gq & 1) * 8;
    }
    for (int i = tid; i < 2 * 20 * 5; i += 256)                // rows 12 .. 16 of the column-major copies: never written again
        s_img_all[(i / 100) * SD_IMGB + SD_IMG + ((i % 100) / 5) * 17 + 12 + (i % 5)] = 0u;
    // The first layer of a block runs in SIX steps of two 16-pixel tiles (12 tiles = 192 >= 180 patch pixels): the image reads of step g
    // (`conv1_read`: 16 ds_read_b32) and its arithmetic in six pieces (`conv1_piece`: 6 MFMAs, ~100 VALU instructions, 4 ds_write_b64 in
    // all).  In the steady state step g of block i + 1 rides in column g of block i's second layer.
    unsigned pk[2][8];
    auto conv1_read = [&](int g, const unsigned *s_img) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned *ib = (const unsigned *)((const char *)s_img + c1_src[2 * g + i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) pk[i][j] = ib[j];
        }
    };
    // the arithmetic of a step in SIX pieces (tile i = piece & 1 ...): 0 / 1 = the tile's matrix products, 2 / 3 = bias, ReLU, zero
    // padding, 4 / 5 = the split into fp16 pairs and the two LDS stores
    f32x4 c1[2];
    float c1v[2][4];
    auto conv1_piece = [&](int g, int piece, const Blk &b, char *patch) {
        const int i = piece & 1;
        if (piece < 2) {
            u32x4 ah, al;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                ah[d] = __builtin_amdgcn_perm(pk[i][2 * d + 1], pk[i][2 * d], 0x05040100u);  // (hi, hi) of slots 2d, 2d + 1
                al[d] = __builtin_amdgcn_perm(pk[i][2 * d + 1], pk[i][2 * d], 0x07060302u);  // (lo, lo)
            }
            const f16x8 Ph = __builtin_bit_cast(f16x8, ah), Pl = __builtin_bit_cast(f16x8, al);
            // weights as the A operand: D[m = channel][n = pixel], a lane holds channels 4 gq .. + 3 of pixel l15
            c1[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W1l, Ph, (f32x4)(0.0f), 0, 0, 0);
            c1[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W1h, Pl, c1[i], 0, 0, 0);
            c1[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W1h, Ph, c1[i], 0, 0, 0);
            asm volatile("" : "+v"(c1[i]));                   // an architectural register: the vector unit reads it next
            return;
        }
        if (piece < 4) {
            // (no test for pixels outside the map here: `border_fixup` zeroes them, once per block and only in blocks that touch the border)
#pragma unroll
            for (int r = 0; r < 4; ++r) c1v[i][r] = fmaxf(c1[i][r] * k1 + b1s[r], 0.0f);
            return;
        }
        const __half2 h01 = __floats2half2_rn(c1v[i][0], c1v[i][1]), h23 = __floats2half2_rn(c1v[i][2], c1v[i][3]);
        const float d0 = sd_sub_half<0>(c1v[i][0], h01), d1 = sd_sub_half<1>(c1v[i][1], h01);
        const float d2 = sd_sub_half<0>(c1v[i][2], h23), d3 = sd_sub_half<1>(c1v[i][3], h23);
        const __half2 l01 = __floats2half2_rn(d0, d1), l23 = __floats2half2_rn(d2, d3);
        if ((DBG & 16) == 0 && (2 * g + i < 11 || l15 < 4)) {  // (g is a constant where this is called: tile 11 alone has pixels that do not exist)
            char *d = patch + c1_dst[2 * g + i];
            *(uint2 *)d = make_uint2(*(const unsigned *)&h01, *(const unsigned *)&h23);
            *(uint2 *)(d + 128) = make_uint2(*(const unsigned *)&l01, *(const unsigned *)&l23);
        }
    };
    // the second layer's zero padding: patch pixels outside the map are zeroed AFTER the first layer has written them (this wave's 16
    // channels; same wave, so LDS order does it), in the blocks that touch the border only -- 20 % of a 224 x 224 frame's -- instead
    // of a test per pixel and value in every block
    auto border_fixup = [&](const Blk &b, char *patch) {
        const int gy0 = b.by * 8 - 1, gx0 = b.bx * 16 - 1;
        if ((gy0 >= 0) & (gy0 + 10 <= p.H) & (gx0 >= 0) & (gx0 + 18 <= p.W)) return;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int pp = lane + 64 * k;
            const int pr = (pp * 3641) >> 16, pc = pp - 18 * pr;
            const int gy = gy0 + pr, gx = gx0 + pc;
            if (pp < SD_NPIX && !((gy >= 0) & (gy < p.H) & (gx >= 0) & (gx < p.W))) {
                char *d = patch + pr * SD_RP + pc * SD_PP;
                const int c0 = ((2 * wave) ^ SD_SWZ(pc)) << 4, c1_ = ((2 * wave + 1) ^ SD_SWZ(pc)) << 4;
                *(uint4 *)(d + c0) = make_uint4(0u, 0u, 0u, 0u); *(uint4 *)(d + c1_) = make_uint4(0u, 0u, 0u, 0u);
                *(uint4 *)(d + 128 + c0) = make_uint4(0u, 0u, 0u, 0u); *(uint4 *)(d + 128 + c1_) = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    };

    // ---- epilogue of a finished block, in FOUR pieces (two output rows = one pooled row each) that ride in the NEXT block's first two
    // columns: the accumulators are copied out (`eacc`) when a block's columns are through and the next block starts at once -- with one
    // wave per SIMD the matrix pipe would otherwise idle through the epilogue's ~150 vector instructions and stores (1.2k of a block's
    // 12.6k cycles).  Lane = (pixel column l15, channels 16 wave + 4 gq .. + 3).  Buffer stores, the image's output map = the buffer: a
    // lane that does not store (the odd column of a pooling window, pixels beyond a ragged edge, the pieces in front of the first block)
    // gets an offset out of its range and is dropped by the hardware -- no branch around the stores, so the compiler can count them
    float my_amax = 0.0f;
    const int Ho = POOL ? p.H >> 1 : p.H, Wo = POOL ? p.W >> 1 : p.W;
    f32x4 eacc[8];
    Blk ecb = {0, 0, 0};
    bool e_ok = false;
    auto epi_piece = [&](int k) {
        const __amdgpu_buffer_rsrc_t rsY = sd_rsrc((const char *)(p.y + (int64_t)ecb.img * Ho * Wo * 64), (int64_t)Ho * Wo * 256);
        const int ox = ecb.bx * 16 + l15;
        const int ch_off = (16 * wave + 4 * gq) * 4;
        if (POOL) {
            const int py = ecb.by * 4 + k;
            float4 v;
            // 2 x 2 maximum first (the two rows sit in one lane, the two columns in lanes l, l ^ 1), then the exact rescale, bias and ReLU
            // on the survivor: all monotone
            v.x = sd_max_xor1(sd_max(eacc[2 * k][0], eacc[2 * k + 1][0])); v.y = sd_max_xor1(sd_max(eacc[2 * k][1], eacc[2 * k + 1][1]));
            v.z = sd_max_xor1(sd_max(eacc[2 * k][2], eacc[2 * k + 1][2])); v.w = sd_max_xor1(sd_max(eacc[2 * k][3], eacc[2 * k + 1][3]));
            v.x = fmaxf(v.x * inv2 + bv2.x, 0.0f); v.y = fmaxf(v.y * inv2 + bv2.y, 0.0f);
            v.z = fmaxf(v.z * inv2 + bv2.z, 0.0f); v.w = fmaxf(v.w * inv2 + bv2.w, 0.0f);
            const bool store = e_ok & ((l15 & 1) == 0) & ((ox >> 1) < Wo) & (py < Ho);          // (bitwise: `&&` became branches on exec)
            float m = sd_max(sd_max(v.x, v.y), sd_max(v.z, v.w));
            int off = (py * Wo + (ox >> 1)) * 256 + ch_off;
            asm volatile("" : "+v"(m), "+v"(off));             // computed by every lane: inside a select hipcc turns them into a branch, and a branch ends the region
            my_amax = sd_max(my_amax, store ? m : 0.0f);
            u32x4 bits;
            bits.x = __float_as_uint(v.x); bits.y = __float_as_uint(v.y); bits.z = __float_as_uint(v.z); bits.w = __float_as_uint(v.w);
            if (!(DBG & 4)) __builtin_amdgcn_raw_buffer_store_b128(bits, rsY, store ? off : 0x7fffffff, 0, 0);
        } else {
#pragma unroll
            for (int r = 2 * k; r < 2 * k + 2; ++r) {
                const int oy = ecb.by * 8 + r;
                float4 v;
                v.x = fmaxf(eacc[r][0] * inv2 + bv2.x, 0.0f); v.y = fmaxf(eacc[r][1] * inv2 + bv2.y, 0.0f);
                v.z = fmaxf(eacc[r][2] * inv2 + bv2.z, 0.0f); v.w = fmaxf(eacc[r][3] * inv2 + bv2.w, 0.0f);
                const bool store = e_ok & (ox < p.W) & (oy < p.H);
                float m = sd_max(sd_max(v.x, v.y), sd_max(v.z, v.w));
                int off = (oy * p.W + ox) * 256 + ch_off;
                asm volatile("" : "+v"(m), "+v"(off));
                my_amax = sd_max(my_amax, store ? m : 0.0f);
                u32x4 bits;
                bits.x = __float_as_uint(v.x); bits.y = __float_as_uint(v.y); bits.z = __float_as_uint(v.z); bits.w = __float_as_uint(v.w);
                __builtin_amdgcn_raw_buffer_store_b128(bits, rsY, store ? off : 0x7fffffff, 0, 0);
            }
        }
    };

    // ---- second layer.  acc[r]: output row r of the block, lane (l15, gq) = pixel column l15, channels 16 wave + 4 gq .. + 3.
    // Column (dx, ks): for the ten patch rows R the fragment (row R, columns l15 + dx, channels 32 ks + 8 gq .. + 7; hi and lo) is read
    // once and multiplied into the up to three output rows r = R - dy with the weights of tap (dy, dx).
    // One scheduling region per patch row (`rstep`): the reads of the fragment TWO rows ahead (a ring of three register pairs, running
    // on into the next column), the row's 3 - 9 MFMAs, and one piece of the next block's first layer -- with ONE wave per SIMD nothing
    // else covers an LDS round trip or fills the matrix pipe while vector work runs.  (Left to sched_group_barrier pipelines over a
    // whole column hipcc put every read right in front of its first MFMA: 13.6k cycles per block for 7.5k of MFMA issue.)
    f32x4 acc[8];
    int fr_off[3];                                             // the lane's fragment offset at column shift dx: pixel l15 + dx, chunk gq ^ swizzle
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) fr_off[dx] = (l15 + dx) * SD_PP + ((gq ^ SD_SWZ(l15 + dx)) << 4);
    f16x8 fh[3], fl[3];
    auto frag_read = [&](auto col_tag, auto r_tag, const char *patch) {
        constexpr int COL = decltype(col_tag)::value, R = decltype(r_tag)::value;
        constexpr int DX = COL >> 1, KS = COL & 1, SLOT = (R + COL) % 3;
        fh[SLOT] = *(const f16x8 *)(patch + fr_off[DX] + R * SD_RP + KS * 64);
        fl[SLOT] = *(const f16x8 *)(patch + fr_off[DX] + R * SD_RP + KS * 64 + 128);
    };
    auto rstep = [&](auto col_tag, auto r_tag, const char *patch, const unsigned *s_img, const Blk &nblk, char *npatch) {
        constexpr int COL = decltype(col_tag)::value, R = decltype(r_tag)::value;
        constexpr int DX = COL >> 1, KS = COL & 1, SLOT = (R + COL) % 3;
        // fragments two rows ahead: rows 8 and 9 request rows 0 and 1 of the next column
        if constexpr (DBG & 2) { }
        else if constexpr (R + 2 < 10) frag_read(col_tag, std::integral_constant<int, (R + 2) % 10>{}, patch);
        else if constexpr (COL < 5) frag_read(std::integral_constant<int, (COL + 1) % 6>{}, std::integral_constant<int, (R + 2) % 10>{}, patch);
        if constexpr (R == 0 && !(DBG & 1) && !(DBG & 8)) conv1_read(COL, s_img);
#pragma unroll
        for (int prod = 0; prod < 3; ++prod)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int r = R - dy;
                if (r < 0 || r > 7) continue;
                const f16x8 a = prod == 2 ? wr[3 * dy + DX][KS][1] : wr[3 * dy + DX][KS][0];
                const f16x8 b = prod == 1 ? fl[SLOT] : fh[SLOT];
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, (COL == 0 && prod == 0 && dy == 0) ? (f32x4)(0.0f) : acc[r], 0, 0, 0);
            }
        if constexpr (R >= 2 && R <= 7 && !(DBG & 1)) conv1_piece(COL, R - 2, nblk, npatch);
        if constexpr (COL < 2 && (R == 1 || R == 8)) epi_piece(2 * COL + (R == 8 ? 1 : 0));    // the previous block's epilogue
        // inside the region: the reads first, then a few vector instructions behind every MFMA (hipcc otherwise runs a piece's ~25
        // vector instructions in one block with the matrix pipe idle)
        __builtin_amdgcn_sched_group_barrier(0x100, R == 0 ? 18 : 2, 0);
        constexpr int NM = 3 * ((R < 2 ? R + 1 : 3) - (R > 7 ? R - 7 : 0)) + ((R == 2 || R == 3) ? 3 : 0);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, (COL < 2 && (R == 1 || R == 8)) ? 6 : 3, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // The last patch row of a column multiplies into ONE accumulator (output row 7), the first row of the next column into one as well
    // (row 0): as regions of their own each was a chain of three dependent MFMAs.  Together, alternating, no MFMA waits for its predecessor.
    auto edge = [&](auto col_tag, const char *patch, const unsigned *s_img) {
        constexpr int COL = decltype(col_tag)::value;           // row 9 of column COL + row 0 of column COL + 1
        constexpr int DX = COL >> 1, KS = COL & 1, SLOT = (9 + COL) % 3;
        constexpr int DX2 = (COL + 1) >> 1, KS2 = (COL + 1) & 1, SLOT2 = (COL + 1) % 3;
        if constexpr (!(DBG & 2)) frag_read(std::integral_constant<int, COL + 1>{}, std::integral_constant<int, 1>{}, patch);
        if constexpr (!(DBG & 1) && !(DBG & 8)) conv1_read(COL + 1, s_img);
#pragma unroll
        for (int prod = 0; prod < 3; ++prod) {
            acc[7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(prod == 2 ? wr[6 + DX][KS][1] : wr[6 + DX][KS][0], prod == 1 ? fl[SLOT] : fh[SLOT], acc[7], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(prod == 2 ? wr[DX2][KS2][1] : wr[DX2][KS2][0], prod == 1 ? fl[SLOT2] : fh[SLOT2], acc[0], 0, 0, 0);
        }
        // (row 2 of the next column lands in the ring slot row 9 was just multiplied from: behind its MFMAs)
        if constexpr (!(DBG & 2)) frag_read(std::integral_constant<int, COL + 1>{}, std::integral_constant<int, 2>{}, patch);
        __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto column = [&](auto col_tag, const char *patch, const unsigned *s_img, const Blk &nblk, char *npatch) {
        constexpr int COL_ = decltype(col_tag)::value;
        if constexpr (COL_ == 0) rstep(col_tag, std::integral_constant<int, 0>{}, patch, s_img, nblk, npatch);      // (the other columns' row 0: `edge`)
        rstep(col_tag, std::integral_constant<int, 1>{}, patch, s_img, nblk, npatch);
        rstep(col_tag, std::integral_constant<int, 2>{}, patch, s_img, nblk, npatch); rstep(col_tag, std::integral_constant<int, 3>{}, patch, s_img, nblk, npatch);
        rstep(col_tag, std::integral_constant<int, 4>{}, patch, s_img, nblk, npatch); rstep(col_tag, std::integral_constant<int, 5>{}, patch, s_img, nblk, npatch);
        rstep(col_tag, std::integral_constant<int, 6>{}, patch, s_img, nblk, npatch); rstep(col_tag, std::integral_constant<int, 7>{}, patch, s_img, nblk, npatch);
        rstep(col_tag, std::integral_constant<int, 8>{}, patch, s_img, nblk, npatch);
        if constexpr (COL_ == 5) rstep(col_tag, std::integral_constant<int, 9>{}, patch, s_img, nblk, npatch);
        else edge(col_tag, patch, s_img);
    };
#define SD_C(T) std::integral_constant<int, T>{}

    [[maybe_unused]] unsigned long long t_main = 0, t_epi = 0, t_bar = 0, t1 = 0, t2 = 0, t3 = 0;
    // ---- prologue: block 0's first layer, block 1's image
    Blk cb = decode_blk(0);
    img_load(cb);
    img_store(s_img_all);
    Blk nb = decode_blk(1);
    img_load(nb);
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 6; ++g) {
        conv1_read(g, s_img_all);
#pragma unroll
        for (int piece = 0; piece < 6; ++piece) conv1_piece(g, piece, cb, sd_smem);
        __builtin_amdgcn_sched_barrier(0);
    }
    border_fixup(cb, sd_smem);
    img_store(s_img_all + SD_IMGB);
    __syncthreads();
    for (int bi = 0; bi < n_mine; ++bi) {
        const int cur = bi & 1;
        const char *const patch = sd_smem + cur * SD_PATCHB;
        char *const npatch = sd_smem + (cur ^ 1) * SD_PATCHB;
        const unsigned *const s_img_next = s_img_all + (cur ^ 1) * SD_IMGB;    // block bi + 1's image
        const Blk nnb = decode_blk(bi + 2);
        img_load(nnb);                                         // in flight until this block's columns are through
        if (SD_PROF) t1 = __builtin_amdgcn_s_memtime();
        frag_read(SD_C(0), SD_C(0), patch);
        frag_read(SD_C(0), SD_C(1), patch);
        column(SD_C(0), patch, s_img_next, nb, npatch); column(SD_C(1), patch, s_img_next, nb, npatch); column(SD_C(2), patch, s_img_next, nb, npatch);
        column(SD_C(3), patch, s_img_next, nb, npatch); column(SD_C(4), patch, s_img_next, nb, npatch); column(SD_C(5), patch, s_img_next, nb, npatch);
        if (!(DBG & 1)) border_fixup(nb, npatch);
        if (SD_PROF) { t2 = __builtin_amdgcn_s_memtime(); t_main += t2 - t1; }
        // the image of block bi + 2 into the buffer block bi's came from (read during iteration bi - 1), consumed HERE, in front of this
        // block's stores: loads and stores share vmcnt and return out of order, so a load consumed behind a store waits for the store
        img_store(s_img_all + cur * SD_IMGB);
        // the finished block's accumulators out of the way: its epilogue rides in the next block's first two columns
#pragma unroll
        for (int r = 0; r < 8; ++r) eacc[r] = acc[r];
        ecb = cb;
        e_ok = true;
        if (SD_PROF) { t3 = __builtin_amdgcn_s_memtime(); t_epi += t3 - t2; }
        // ONE barrier per block: the next block's patch and the image after it are whole, everybody is through this block's patch
        __syncthreads();
        if (SD_PROF) t_bar += __builtin_amdgcn_s_memtime() - t3;
        cb = nb;
        nb = nnb;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) epi_piece(k);                  // the last block's
#ifdef CSLAM_ABLATIONS
    if (sd_prof && blockIdx.x == 0 && tid == 0) { sd_prof[0] = t_main; sd_prof[1] = t_epi; sd_prof[2] = t_bar; sd_prof[3] = 0; sd_prof[4] = (unsigned long long)n_mine; }
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
 * weights as exact fp16 pairs in MFMA-fragment order [4 output-channel quarters][9 taps][2 K steps][hi | lo][64 lanes][8]; d_amax_x0: 4-byte slot
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
    constexpr int lds = SD_LDS;
#define SD_LAUNCH(P) do { \
        static DeviceOnce once; int once_dev; \
        if (once.todo(&once_dev)) { \
            HIP_TRY(hipFuncSetAttribute((const void *)conv_stem_direct_h_kernel<P>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
            once.done(once_dev); } \
        hipLaunchKernelGGL((conv_stem_direct_h_kernel<P>), dim3(grid), dim3(256), lds, st, a); } while (0)
#ifdef CSLAM_ABLATIONS
    if (const char *e = getenv("CSLAM_SD_DBG")) {              // timing-only ablations (wrong results): measurement build
        const int d = atoi(e);
#define SD_LAUNCH_D(D) do { HIP_TRY(hipFuncSetAttribute((const void *)conv_stem_direct_h_kernel<true, D>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        hipLaunchKernelGGL((conv_stem_direct_h_kernel<true, D>), dim3(grid), dim3(256), lds, st, a); HIP_TRY(hipGetLastError()); return CSLAM_OK; } while (0)
        if (d == 1) SD_LAUNCH_D(1);
        if (d == 2) SD_LAUNCH_D(2);
        if (d == 3) SD_LAUNCH_D(3);
        if (d == 4) SD_LAUNCH_D(4);
        if (d == 7) SD_LAUNCH_D(7);
        if (d == 8) SD_LAUNCH_D(8);
        if (d == 16) SD_LAUNCH_D(16);
        if (d == 24) SD_LAUNCH_D(24);
#undef SD_LAUNCH_D
    }
#endif
    if (pool) SD_LAUNCH(true);
    else SD_LAUNCH(false);
#undef SD_LAUNCH
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}
