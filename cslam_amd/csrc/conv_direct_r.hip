// conv_direct_r.hip -- 3x3 / stride 1 / pad 1 convolution 64 -> 128 channels (VGG-16 conv2_1 on 112 x 112 maps: cslam/vpr/netvlad.py:
// 163-171,227) as ONE direct kernel whose weights never leave the register files (gfx950), fp32-grade on the fp16 matrix pipe.
//
// The form of conv_stem_direct_h.hip with the activation read from HBM instead of computed from the image: 64 x 128 x 9 weights as exact
// fp16 pairs are 295 KB -- 288 registers per lane for each of the four one-per-SIMD waves of a workgroup.  Wave w owns OUTPUT channels
// 32 w .. 32 w + 31 (two 16-row A tiles of v_mfma_f32_16x16x32_f16) for all 128 pixels of an 8 x 16 block:
//   * weights [9 taps][2 K steps of 32 channels][2 channel tiles][hi | lo] loaded once per kernel: no weight stream through L2 / LDS, no
//     weight ring, no stage barriers (conv_direct_h.hip streams 16 KB per (tap, slab) stage and synchronises after each);
//   * the 10 x 18-pixel x 64-channel patch of the input in LDS as exact fp16 pairs (conv_stem_direct_h.hip's layout: 320-byte pixels,
//     swizzled 16-byte chunks), shared by the four waves, double buffered: block i + 1's patch is loaded (float32, HBM / L2 -> registers)
//     during block i's first columns and split into the idle buffer during its later ones, a few vector instructions behind every MFMA;
//   * a B fragment = one patch row (16 pixels) x 32 channels at a column shift dx serves the three taps dy = 0, 1, 2 and both channel
//     tiles: 2 fragment reads per up to 18 MFMAs (conv_direct_h.hip: 8 per 12);
//   * no partial sums, ONE barrier per block.
// Arithmetic as conv_direct_h.hip: x scaled by the power of two s_x from the producer's max |x| slot and split into hi + lo when the patch
// is staged; w split offline (`direct_r_pair_weights`); acc = wh xh + wh xl + wl xh; y = [pool](relu(acc / (s_x s_w) + bias)).
#include <stdlib.h>
#include <type_traits>
#include <hip/hip_fp16.h>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float dr_f32x2 __attribute__((ext_vector_type(2)));

#define DR_PP 320                      // bytes per patch pixel: [hi 64 halfs | lo 64 halfs | 64 pad]; chunk c of a half at c ^ DR_SWZ(column)
#define DR_SWZ(pc) (((pc) >> 1) & 3)   // (conv_stem_direct_h.hip: conflict-free fragment reads for every column shift)
#define DR_RP (18 * DR_PP)
#define DR_PATCHB (10 * DR_RP)         // 57 600 per buffer
#define DR_NPIX 180
#define DR_NEL 12                      // float4 elements per thread and patch: 180 pixels x 16 = 2880 <= 12 x 256
#define DR_LDS (2 * DR_PATCHB + 4096)   // + a sink for the elements of the last staging round that do not exist (no branch around a store)

struct ConvDirectRArgs {
    const float *x; const f16x8 *w2; const float *bias; float *y;
    int B, H, W, gxb, gyb, nblk;
    int relu;
    const unsigned *amax_in; float inv_sw; unsigned *amax_out;
    float wl1, bmax; unsigned *bound_out;          // OUTP: y leaves in pair format (conv_igemm.hip), scaled for the bound max|x| wl1 + bmax
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t dr_rsrc(const char *base, int64_t bytes) {
    const uint64_t a = (uint64_t)base;
    const uint64_t u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)a);
    const int n = __builtin_amdgcn_readfirstlane((int)(bytes > 0x7ffffff0 ? 0x7ffffff0 : bytes));
    return __builtin_amdgcn_make_buffer_rsrc((void *)u, 0, n, 0x00020000);
}
#ifdef CSLAM_ABLATIONS
__device__ unsigned long long *dr_prof = nullptr;             // measurement build: [six columns (+ staging), epilogue, barrier, blocks] ticks of wave 0 / workgroup 0
extern "C" __attribute__((visibility("default"))) int cslam_debug_dr_prof_dev(void *d_buf) {
    unsigned long long *q = (unsigned long long *)d_buf;
    return hipMemcpyToSymbol(HIP_SYMBOL(dr_prof), &q, sizeof(q)) == hipSuccess ? 0 : -2;
}
#define DR_PROF 1
#else
#define DR_PROF 0
#endif
template <int HI>
__device__ __forceinline__ float dr_sub_half(float v, __half2 h) {       // v - (float)half HI of h: one v_fma_mix_f32
    float d;
    const unsigned hb = *(const unsigned *)&h;
    if (HI) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hb), "v"(v));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hb), "v"(v));
    return d;
}
__device__ __forceinline__ float dr_max(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float dr_max_xor1(float v) {                   // max(v, v of lane ^ 1): DPP quad permutation
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    return dr_max(v, a);
}

// DBG (builds with -DCSLAM_ABLATIONS only; WRONG results, timing): 1 = no patch staging inside the loop, 4 = no stores
// OUTP (no pooling): y is written in the PAIR FORMAT of conv_igemm.hip -- [pixel][32-channel block][hi 32 | lo 32] fp16 of s y, s the
// power of two that brings the bound max|x| wl1 + bmax (stored to bound_out) into [2^13, 2^14) -- so that the next layer (conv2_2,
// conv_direct_h.hip) stages its patch without converting anything (0.34 of its 2.5 ms were the split's vector instructions)
typedef _Float16 dr_f16x2 __attribute__((ext_vector_type(2)));
template <int HI>
__device__ __forceinline__ float dr_fma_half(unsigned hb, float s, float v) { return __builtin_fmaf((float)__builtin_bit_cast(dr_f16x2, hb)[HI], s, v); }
__device__ __forceinline__ unsigned dr_pk(float a, float b) { const __half2 h = __floats2half2_rn(a, b); return *(const unsigned *)&h; }
template <bool POOL, bool RELU, int DBG = 0, bool OUTP = false>
__global__ __launch_bounds__(256, 1) void conv3x3_direct_r_kernel(ConvDirectRArgs p) {
    static_assert(!(POOL && OUTP), "the pair-format output is written un-pooled");
    extern __shared__ __attribute__((aligned(16))) char dr_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gq = lane >> 4, l15 = lane & 15;

    // power-of-two input scale: max |x| s_x <= 2^15 - 16
    const float amax = fminf(fmaxf(__uint_as_float(*p.amax_in), 1e-30f), 1e30f);
    int e_;
    (void)frexpf(32752.0f / amax, &e_);
    const float sx = ldexpf(1.0f, e_ - 1);
    const float inv = p.inv_sw / sx;
    float s_out = 1.0f, neg1 = -1.0f;
    asm volatile("" : "+v"(neg1));                             // (a run-time -1: conv_direct_p.hip)
    if (OUTP) {
        const float bound = (amax * p.wl1 + p.bmax) * 1.001f;  // >= max |y| whatever the rounding of the products
        int eo;
        (void)frexpf(fminf(fmaxf(bound, 1e-30f), 1e30f), &eo);
        s_out = ldexpf(1.0f, 14 - eo);                         // conv_igemm.hip::ci_scale
        if (blockIdx.x == 0 && threadIdx.x == 0) *p.bound_out = __float_as_uint(bound);
    }

    // blocks to workgroups by XCD (contiguous eighths: the halo neighbours share goes through one L2)
    const bool by_xcd = (gridDim.x & 7) == 0;
    const int wg_xcd = by_xcd ? (int)blockIdx.x & 7 : 0, wg_j = by_xcd ? (int)blockIdx.x >> 3 : (int)blockIdx.x;
    const int wg_per = by_xcd ? (int)gridDim.x >> 3 : (int)gridDim.x;
    const int per_xcd = by_xcd ? (p.nblk + 7) >> 3 : p.nblk;
    const int blk_beg = wg_xcd * per_xcd;
    const int blk_cnt = min(per_xcd, p.nblk - blk_beg);
    const int n_mine = blk_cnt > wg_j ? (blk_cnt - wg_j + wg_per - 1) / wg_per : 0;
    if (n_mine <= 0) return;

    // ---- this wave's weights: [tap][K step][channel tile][hi | lo], 288 registers; 256 of them pinned in the accumulation half of the
    // register file (MFMA operands and nothing else)
    f16x8 wr[9][2][2][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
                    wr[tap][ks][mt][hl] = p.w2[((((wave * 9 + tap) * 2 + ks) * 2 + mt) * 2 + hl) * 64 + lane];
                    if (tap < 6) asm volatile("" : "+a"(wr[tap][ks][mt][hl]));    // 192 + the 64 accumulators = the 256 AGPRs
                }
    float4 bv[2];                                              // bias of the lane's channels 32 wave + 16 mt + 4 gq .. + 3
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
        bv[mt] = p.bias ? *(const float4 *)(p.bias + 32 * wave + 16 * mt + 4 * gq) : make_float4(0.f, 0.f, 0.f, 0.f);

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

    // ---- patch staging: element e = i * 256 + tid = (pixel e >> 4, float4 e & 15 of its 64 channels)
    int st_dst[DR_NEL];                                        // destination in a patch buffer (-1: none); the source offset is recomputed from
    unsigned st_pc = 0;                                        // the thread id at every load (12 more registers spilled); packed columns (5 bits each)
#pragma unroll
    for (int i = 0; i < DR_NEL; ++i) {
        const int e = i * 256 + tid, px = e >> 4, f4 = e & 15;
        const int pr = (px * 3641) >> 16, pc = px - 18 * pr;   // px / 18 for px < 4096
        const bool valid = px < DR_NPIX;
        st_dst[i] = valid ? pr * DR_RP + pc * DR_PP + (((f4 >> 1) ^ DR_SWZ(pc)) << 4) + (f4 & 1) * 8 : -1;   // (-1: see patch_split)
        if (i < 6) st_pc |= (unsigned)(valid ? pc : 0) << (5 * i);
    }
    unsigned st_pc2 = 0;
#pragma unroll
    for (int i = 6; i < DR_NEL; ++i) {
        const int e = i * 256 + tid, px = e >> 4;
        const int pr = (px * 3641) >> 16, pc = px - 18 * pr;
        st_pc2 |= (unsigned)(px < DR_NPIX ? pc : 0) << (5 * (i - 6));
    }
    const int img_bytes = p.H * p.W * 256;
    u32x4 stg[DR_NEL];
    // the image is the buffer: rows above and below it are out of its range and read as zero; columns left and right of it would read the
    // neighbouring row's pixels and are zeroed when the registers are split.  Every lane issues every load (no branch): counted waits.
    auto patch_load = [&](const Blk &b, int i) {
        const __amdgpu_buffer_rsrc_t rsX = dr_rsrc((const char *)p.x + (int64_t)b.img * img_bytes, img_bytes);
        const int blk_off = ((b.by * 8 - 1) * p.W + (b.bx * 16 - 1)) * 256;
        int t = tid;
        asm volatile("" : "+v"(t));                            // opaque: the twelve offsets are loop invariants otherwise, hoisted and spilled
        const int e = i * 256 + t, px = e >> 4;
        const int pr = (px * 3641) >> 16, pc = px - 18 * pr;
        int off = ((pr * p.W + pc) * 64 + (e & 15) * 4) * 4 + blk_off;
        asm volatile("" : "+v"(off));                          // computed by every lane: written as one select hipcc turns the "expensive" arm
        stg[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, px < DR_NPIX ? off : 0x7fffffff, 0, 0);     // into a branch, and a branch ends the region
    };
    auto patch_split = [&](const Blk &b, int i, char *patch) {
        // (vector instructions are what this staging costs -- three slots per MFMA and wave: the column test is one bit-field extract, one
        // add and one unsigned compare, the scale two packed multiplies; conv3x3_direct_r2_kernel below, where it was measured)
        const unsigned pc = __builtin_amdgcn_ubfe(i < 6 ? st_pc : st_pc2, 5 * (i < 6 ? i : i - 6), 5);
        const float s = (unsigned)(b.bx * 16 - 1) + pc < (unsigned)p.W ? sx : 0.0f;         // (column - 1 wraps to a large number)
        dr_f32x2 w01 = dr_f32x2{__uint_as_float(stg[i].x), __uint_as_float(stg[i].y)} * dr_f32x2{s, s};
        dr_f32x2 w23 = dr_f32x2{__uint_as_float(stg[i].z), __uint_as_float(stg[i].w)} * dr_f32x2{s, s};
        asm("" : "+v"(w01), "+v"(w23));                        // (used AS pairs: hipcc keeps the two v_pk_mul_f32)
        const float w0 = w01[0], w1 = w01[1], w2 = w23[0], w3 = w23[1];
        const __half2 h01 = __floats2half2_rn(w0, w1), h23 = __floats2half2_rn(w2, w3);
        const float d0 = dr_sub_half<0>(w0, h01), d1 = dr_sub_half<1>(w1, h01), d2 = dr_sub_half<0>(w2, h23), d3 = dr_sub_half<1>(w3, h23);
        const __half2 l01 = __floats2half2_rn(d0, d1), l23 = __floats2half2_rn(d2, d3);
        // (branch-free: a region must stay ONE basic block for its instruction order to be set; what does not exist goes to the sink)
        char *d = st_dst[i] >= 0 ? patch + st_dst[i] : dr_smem + 2 * DR_PATCHB + (tid & 63) * 16;
        *(uint2 *)d = make_uint2(*(const unsigned *)&h01, *(const unsigned *)&h23);
        *(uint2 *)(d + 128) = make_uint2(*(const unsigned *)&l01, *(const unsigned *)&l23);
    };

    // ---- the products.  acc[r][mt]: output row r, lane (l15, gq) = pixel column l15, channels 32 wave + 16 mt + 4 gq .. + 3
    f32x4 acc[8][2];
    int fr_off[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) fr_off[dx] = (l15 + dx) * DR_PP + ((gq ^ DR_SWZ(l15 + dx)) << 4);
    f16x8 fh[3], fl[3];
    auto frag_read = [&](auto col_tag, auto r_tag, const char *patch) {
        constexpr int COL = decltype(col_tag)::value, R = decltype(r_tag)::value;
        constexpr int DX = COL >> 1, KS = COL & 1, SLOT = (R + COL) % 3;
        fh[SLOT] = *(const f16x8 *)(patch + fr_off[DX] + R * DR_RP + KS * 64);
        fl[SLOT] = *(const f16x8 *)(patch + fr_off[DX] + R * DR_RP + KS * 64 + 128);
    };
    // ---- epilogue, a row (pair) at a time: output row r has its last contribution in the LAST column's patch row r + 2, so its bias /
    // ReLU / stores ride in that column's next region, under the MFMAs of the rows still open -- only the last row (pair) is left
    // behind the block's last MFMA.  (As one phase behind the columns: 3.5k of a block's 21k cycles, the matrix pipe idle.)  Buffer stores,
    // the image's output map = the buffer (a lane that does not store gets an offset out of its range: no branch).
    float my_amax = 0.0f;
    const int Ho = POOL ? p.H >> 1 : p.H, Wo = POOL ? p.W >> 1 : p.W;
    Blk eb = {0, 0, 0};                                        // the block being multiplied
    auto epi_rows = [&](int k) {                               // POOL: pooled row k (output rows 2 k, 2 k + 1); else output row k
        const __amdgpu_buffer_rsrc_t rsY = dr_rsrc((const char *)(p.y + (int64_t)eb.img * Ho * Wo * 128), (int64_t)Ho * Wo * 512);
        const int ox = eb.bx * 16 + l15;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int ch_off = (32 * wave + 16 * mt + 4 * gq) * 4;
            float4 v;
            bool store;
            int off;
            if (POOL) {
                const int py = eb.by * 4 + k;
                v.x = dr_max_xor1(dr_max(acc[2 * k][mt][0], acc[2 * k + 1][mt][0])); v.y = dr_max_xor1(dr_max(acc[2 * k][mt][1], acc[2 * k + 1][mt][1]));
                v.z = dr_max_xor1(dr_max(acc[2 * k][mt][2], acc[2 * k + 1][mt][2])); v.w = dr_max_xor1(dr_max(acc[2 * k][mt][3], acc[2 * k + 1][mt][3]));
                store = ((l15 & 1) == 0) & ((ox >> 1) < Wo) & (py < Ho);
                off = (py * Wo + (ox >> 1)) * 512 + ch_off;
            } else {
                const int oy = eb.by * 8 + k;
                v.x = acc[k][mt][0]; v.y = acc[k][mt][1]; v.z = acc[k][mt][2]; v.w = acc[k][mt][3];
                store = (ox < p.W) & (oy < p.H);
                off = (oy * p.W + ox) * 512 + ch_off;
            }
            v.x = v.x * inv + bv[mt].x; v.y = v.y * inv + bv[mt].y; v.z = v.z * inv + bv[mt].z; v.w = v.w * inv + bv[mt].w;
            if (RELU) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
            float m = dr_max(dr_max(fabsf(v.x), fabsf(v.y)), dr_max(fabsf(v.z), fabsf(v.w)));
            asm volatile("" : "+v"(m), "+v"(off));             // (both computed by every lane: no branch around them)
            my_amax = dr_max(my_amax, store ? m : 0.0f);
            if (OUTP) {
                // the lane's four channels 32 wave + 16 mt + 4 gq .. + 3 of the pixel: block `wave`, hi run at its halfs 16 mt + 4 gq, lo run 64 bytes on
                const float u0 = v.x * s_out, u1 = v.y * s_out, u2 = v.z * s_out, u3 = v.w * s_out;
                const unsigned h01 = dr_pk(u0, u1), h23 = dr_pk(u2, u3);
                const unsigned l01 = dr_pk(dr_fma_half<0>(h01, neg1, u0), dr_fma_half<1>(h01, neg1, u1));
                const unsigned l23 = dr_pk(dr_fma_half<0>(h23, neg1, u2), dr_fma_half<1>(h23, neg1, u3));
                const int offp = store ? off - ch_off + wave * 128 + (16 * mt + 4 * gq) * 2 : 0x7fffffff;
                typedef unsigned dr_u32x2 __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64((dr_u32x2){h01, h23}, rsY, offp, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64((dr_u32x2){l01, l23}, rsY, offp, 64, 0);
            } else {
            u32x4 bits;
            bits.x = __float_as_uint(v.x); bits.y = __float_as_uint(v.y); bits.z = __float_as_uint(v.z); bits.w = __float_as_uint(v.w);
            if (!(DBG & 4)) __builtin_amdgcn_raw_buffer_store_b128(bits, rsY, store ? off : 0x7fffffff, 0, 0);
            else asm volatile("" :: "v"(bits));
            }
        }
    };

    // one scheduling region per patch row: the fragment two rows ahead, the row's 6 - 18 MFMAs, and a slice of the next block's staging
    // (columns 0 / 1: its loads, columns 2 .. 5: the split of three elements each)
    auto rstep = [&](auto col_tag, auto r_tag, const char *patch, const Blk &nblk, char *npatch) {
        constexpr int COL = decltype(col_tag)::value, R = decltype(r_tag)::value;
        constexpr int DX = COL >> 1, KS = COL & 1, SLOT = (R + COL) % 3;
        if constexpr (R + 2 < 10) frag_read(col_tag, std::integral_constant<int, (R + 2) % 10>{}, patch);
        else if constexpr (COL < 5) frag_read(std::integral_constant<int, (COL + 1) % 6>{}, std::integral_constant<int, (R + 2) % 10>{}, patch);
        if constexpr (COL == 0 && R >= 1 && R <= 6 && !(DBG & 1)) { patch_load(nblk, 2 * (R - 1)); patch_load(nblk, 2 * (R - 1) + 1); }
#pragma unroll
        for (int prod = 0; prod < 3; ++prod)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int r = R - dy;
                    if (r < 0 || r > 7) continue;
                    const f16x8 a = prod == 2 ? wr[3 * dy + DX][KS][mt][1] : wr[3 * dy + DX][KS][mt][0];
                    const f16x8 b = prod == 1 ? fl[SLOT] : fh[SLOT];
                    acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, (COL == 0 && prod == 0 && dy == 0) ? (f32x4)(0.0f) : acc[r][mt], 0, 0, 0);
                }
        if constexpr (COL >= 3 && (R == 1 || R == 3 || R == 5 || R == 7) && !(DBG & 1)) patch_split(nblk, 4 * (COL - 3) + (R - 1) / 2, npatch);
        if constexpr (COL == 5 && !POOL && R >= 3) epi_rows(R - 3);                                 // output row R - 3 was finished by the last region
        if constexpr (COL == 5 && POOL && (R == 4 || R == 6 || R == 8)) epi_rows((R - 4) / 2);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        constexpr int NM = 6 * ((R < 2 ? R + 1 : 3) - (R > 7 ? R - 7 : 0));
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, COL == 5 ? 4 : 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto column = [&](auto col_tag, const char *patch, const Blk &nblk, char *npatch) {
        rstep(col_tag, std::integral_constant<int, 0>{}, patch, nblk, npatch); rstep(col_tag, std::integral_constant<int, 1>{}, patch, nblk, npatch);
        rstep(col_tag, std::integral_constant<int, 2>{}, patch, nblk, npatch); rstep(col_tag, std::integral_constant<int, 3>{}, patch, nblk, npatch);
        rstep(col_tag, std::integral_constant<int, 4>{}, patch, nblk, npatch); rstep(col_tag, std::integral_constant<int, 5>{}, patch, nblk, npatch);
        rstep(col_tag, std::integral_constant<int, 6>{}, patch, nblk, npatch); rstep(col_tag, std::integral_constant<int, 7>{}, patch, nblk, npatch);
        rstep(col_tag, std::integral_constant<int, 8>{}, patch, nblk, npatch); rstep(col_tag, std::integral_constant<int, 9>{}, patch, nblk, npatch);
    };
#define DR_C(T) std::integral_constant<int, T>{}

    [[maybe_unused]] unsigned long long t_main = 0, t_epi = 0, t_bar = 0, t1 = 0, t2 = 0, t3 = 0;
    // ---- prologue: block 0's patch
    Blk cb = decode_blk(0);
#pragma unroll
    for (int i = 0; i < DR_NEL; ++i) patch_load(cb, i);
#pragma unroll
    for (int i = 0; i < DR_NEL; ++i) patch_split(cb, i, dr_smem);
    __syncthreads();
    for (int bi = 0; bi < n_mine; ++bi) {
        const int cur = bi & 1;
        const char *const patch = dr_smem + cur * DR_PATCHB;
        char *const npatch = dr_smem + (cur ^ 1) * DR_PATCHB;
        const Blk nb = decode_blk(bi + 1);
        eb = cb;
        if (DR_PROF) t1 = __builtin_amdgcn_s_memtime();
        frag_read(DR_C(0), DR_C(0), patch);
        frag_read(DR_C(0), DR_C(1), patch);
        column(DR_C(0), patch, nb, npatch); column(DR_C(1), patch, nb, npatch); column(DR_C(2), patch, nb, npatch);
        column(DR_C(3), patch, nb, npatch); column(DR_C(4), patch, nb, npatch); column(DR_C(5), patch, nb, npatch);
        if (DR_PROF) { t2 = __builtin_amdgcn_s_memtime(); t_main += t2 - t1; }
        epi_rows(POOL ? 3 : 7);                                // the last row (pair): nothing left to hide it under
        if (DR_PROF) { t3 = __builtin_amdgcn_s_memtime(); t_epi += t3 - t2; }
        // ONE barrier per block: the next block's patch is whole, everybody is through this block's
        __syncthreads();
        if (DR_PROF) t_bar += __builtin_amdgcn_s_memtime() - t3;
        cb = nb;
    }
#ifdef CSLAM_ABLATIONS
    if (dr_prof && blockIdx.x == 0 && tid == 0) { dr_prof[0] = t_main; dr_prof[1] = t_epi; dr_prof[2] = t_bar; dr_prof[3] = (unsigned long long)n_mine; }
#endif

    if (p.amax_out) {
        unsigned *s_amax = (unsigned *)dr_smem;
        if (tid == 0) *s_amax = 0u;
        __syncthreads();
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) my_amax = fmaxf(my_amax, __shfl_xor(my_amax, o, 64));
        if (lane == 0) atomicMax(s_amax, __float_as_uint(my_amax));
        __syncthreads();
        if (tid == 0 && *s_amax > *(volatile unsigned *)p.amax_out) atomicMax(p.amax_out, *s_amax);
    }
}

// ---- 128 -> 128 channels (VGG-16 conv2_2, + MaxPool2d) on the same register-resident form (round 6) ----------------------------------
// 128 x 128 x 9 weights as fp16 pairs are 590 KB: more than a compute unit's 512 KB of registers, which is why conv_direct_h.hip streams
// them through an LDS ring -- 16 KB per (tap, 32-channel slab) and 256 pixels, 14.8 GB of L2 -> LDS traffic per 256 frames, and with it the
// kernel sits 15-25 % above what its flop and its HBM bytes cost at the board's power cap.  HALF the output channels
// are 295 KB and fit: the two workgroups 2 j, 2 j + 1 of an XCD walk the SAME blocks, one per output-channel half, so the input leaves HBM
// once (the partner's reads hit the XCD's L2) and no weight ever moves again.  Wave w owns output channels 64 half + 16 w .. + 15 (ONE 16-row
// A tile) for all 128 pixels of an 8 x 16 block and all 128 input channels: a block is two passes over the 64-channel patch layout of
// the kernel above (slab 0, slab 1; the accumulators stay), each pass stages the next one's patch -- (same block, slab 1), then (next
// block, slab 0) -- inside its MFMA stream exactly as above; 2 fragment reads per up to 9 MFMAs; the epilogue rides in slab 1's last column.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define DR2_LDS (2 * (DR_PATCHB + 1024))     // two patch buffers, each with the sink of the elements that do not exist behind it
struct ConvDirectR2Args {
    const float *x; const f16x8 *w2; const float *bias; float *y;
    int B, H, W, gxb, gyb, nblk;
    const unsigned *amax_in; float inv_sw; unsigned *amax_out;
};
template <bool POOL, bool RELU, int DBG = 0>
__global__ __launch_bounds__(256, 1) void conv3x3_direct_r2_kernel(ConvDirectR2Args p) {
    extern __shared__ __attribute__((aligned(16))) char dr_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gq = lane >> 4, l15 = lane & 15;

    const float amax = fminf(fmaxf(__uint_as_float(*p.amax_in), 1e-30f), 1e30f);
    int e_;
    (void)frexpf(32752.0f / amax, &e_);
    const float sx = ldexpf(1.0f, e_ - 1);
    const float inv = p.inv_sw / sx;

    // blocks to workgroup PAIRS by XCD (contiguous eighths); the launch has an even number of workgroups per XCD
    const bool by_xcd = (gridDim.x & 15) == 0;
    const int wg_xcd = by_xcd ? (int)blockIdx.x & 7 : 0, wg_raw = by_xcd ? (int)blockIdx.x >> 3 : (int)blockIdx.x;
    const int half = wg_raw & 1, wg_j = wg_raw >> 1;
    const int wg_per = (by_xcd ? (int)gridDim.x >> 3 : (int)gridDim.x) >> 1;
    const int per_xcd = by_xcd ? (p.nblk + 7) >> 3 : p.nblk;
    const int blk_beg = wg_xcd * per_xcd;
    const int blk_cnt = min(per_xcd, p.nblk - blk_beg);
    const int n_mine = blk_cnt > wg_j ? (blk_cnt - wg_j + wg_per - 1) / wg_per : 0;
    if (n_mine <= 0) return;

    // ---- this wave's weights: [tap][K step][slab][hi | lo], 288 registers; taps 0 .. 6 (224) + the 32 accumulators = the 256 AGPRs
    f16x8 wr[9][2][2][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
                    wr[tap][ks][sl][hl] = p.w2[((((((half * 4 + wave) * 9 + tap) * 2 + ks) * 2 + sl) * 2 + hl) * 64) + lane];
                    if (tap < 7) asm volatile("" : "+a"(wr[tap][ks][sl][hl]));
                }
    const float4 bv = p.bias ? *(const float4 *)(p.bias + 64 * half + 16 * wave + 4 * gq) : make_float4(0.f, 0.f, 0.f, 0.f);

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

    // ---- patch staging (the kernel above, the pixel pitch of the source 128 channels): element e = i * 256 + tid = (pixel e >> 4,
    // float4 e & 15 of the slab's 64 channels)
    // (this kernel has registers to spare and a vector instruction budget that is short -- three slots per MFMA and wave, and half the
    // MFMAs per patch of the kernel above: the per-element source offsets live in registers (an add and a select per request instead of
    // ten instructions), the elements that do not exist go to a sink at the SAME offset behind either patch buffer (one add per
    // destination), the scale is two packed multiplies, the column test one bit-field extract, one add and one unsigned compare)
    constexpr int R2_BUF = DR_PATCHB + 1024;                   // a patch buffer + its sink
    int st_dst[DR_NEL], st_src[DR_NEL];
    unsigned st_pc = 0, st_pc2 = 0;
#pragma unroll
    for (int i = 0; i < DR_NEL; ++i) {
        const int e = i * 256 + tid, px = e >> 4, f4 = e & 15;
        const int pr = (px * 3641) >> 16, pc = px - 18 * pr;
        const bool valid = px < DR_NPIX;
        st_dst[i] = valid ? pr * DR_RP + pc * DR_PP + (((f4 >> 1) ^ DR_SWZ(pc)) << 4) + (f4 & 1) * 8 : DR_PATCHB + (tid & 63) * 16;
        st_src[i] = valid ? ((pr * p.W + pc) * 128 + f4 * 4) * 4 : -1;
        if (i < 6) st_pc |= (unsigned)(valid ? pc : 0) << (5 * i);
        else st_pc2 |= (unsigned)(valid ? pc : 0) << (5 * (i - 6));
    }
    const int img_bytes = p.H * p.W * 512;
    u32x4 stg[DR_NEL];
    auto patch_load = [&](const Blk &b, int slab, int i) {
        const __amdgpu_buffer_rsrc_t rsX = dr_rsrc((const char *)p.x + (int64_t)b.img * img_bytes, img_bytes);
        const int blk_off = ((b.by * 8 - 1) * p.W + (b.bx * 16 - 1)) * 512 + slab * 256;
        int off = st_src[i] + blk_off;
        asm volatile("" : "+v"(off));                          // (computed by every lane: one select, no branch)
        stg[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, st_src[i] >= 0 ? off : 0x7fffffff, 0, 0);
    };
    auto patch_split = [&](const Blk &b, int i, char *patch) {
        const unsigned pc = __builtin_amdgcn_ubfe(i < 6 ? st_pc : st_pc2, 5 * (i < 6 ? i : i - 6), 5);
        const float s = (unsigned)(b.bx * 16 - 1) + pc < (unsigned)p.W ? sx : 0.0f;         // (column - 1 wraps to a large number)
        f32x2 w01 = f32x2{__uint_as_float(stg[i].x), __uint_as_float(stg[i].y)} * f32x2{s, s};
        f32x2 w23 = f32x2{__uint_as_float(stg[i].z), __uint_as_float(stg[i].w)} * f32x2{s, s};
        asm("" : "+v"(w01), "+v"(w23));                        // (used AS pairs: hipcc keeps the two v_pk_mul_f32)
        const __half2 h01 = __floats2half2_rn(w01[0], w01[1]), h23 = __floats2half2_rn(w23[0], w23[1]);
        const float d0 = dr_sub_half<0>(w01[0], h01), d1 = dr_sub_half<1>(w01[1], h01), d2 = dr_sub_half<0>(w23[0], h23), d3 = dr_sub_half<1>(w23[1], h23);
        const __half2 l01 = __floats2half2_rn(d0, d1), l23 = __floats2half2_rn(d2, d3);
        char *d = patch + st_dst[i];
        *(uint2 *)d = make_uint2(*(const unsigned *)&h01, *(const unsigned *)&h23);
        *(uint2 *)(d + 128) = make_uint2(*(const unsigned *)&l01, *(const unsigned *)&l23);
    };

    // ---- the products.  acc[r]: output row r, lane (l15, gq) = pixel column l15, channels 64 half + 16 wave + 4 gq .. + 3
    f32x4 acc[8];
    int fr_off[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) fr_off[dx] = (l15 + dx) * DR_PP + ((gq ^ DR_SWZ(l15 + dx)) << 4);
    f16x8 fh[3], fl[3];
    auto frag_read = [&](auto col_tag, auto r_tag, const char *patch) {
        constexpr int COL = decltype(col_tag)::value, R = decltype(r_tag)::value;
        constexpr int DX = COL >> 1, KS = COL & 1, SLOT = (R + COL) % 3;
        fh[SLOT] = *(const f16x8 *)(patch + fr_off[DX] + R * DR_RP + KS * 64);
        fl[SLOT] = *(const f16x8 *)(patch + fr_off[DX] + R * DR_RP + KS * 64 + 128);
    };
    float my_amax = 0.0f;
    const int Ho = POOL ? p.H >> 1 : p.H, Wo = POOL ? p.W >> 1 : p.W;
    Blk eb = {0, 0, 0};
    auto epi_rows = [&](int k) {                               // POOL: pooled row k (output rows 2 k, 2 k + 1); else output row k
        const __amdgpu_buffer_rsrc_t rsY = dr_rsrc((const char *)(p.y + (int64_t)eb.img * Ho * Wo * 128), (int64_t)Ho * Wo * 512);
        const int ox = eb.bx * 16 + l15;
        const int ch_off = (64 * half + 16 * wave + 4 * gq) * 4;
        float4 v;
        bool store;
        int off;
        if (POOL) {
            const int py = eb.by * 4 + k;
            v.x = dr_max_xor1(dr_max(acc[2 * k][0], acc[2 * k + 1][0])); v.y = dr_max_xor1(dr_max(acc[2 * k][1], acc[2 * k + 1][1]));
            v.z = dr_max_xor1(dr_max(acc[2 * k][2], acc[2 * k + 1][2])); v.w = dr_max_xor1(dr_max(acc[2 * k][3], acc[2 * k + 1][3]));
            store = ((l15 & 1) == 0) & ((ox >> 1) < Wo) & (py < Ho);
            off = (py * Wo + (ox >> 1)) * 512 + ch_off;
        } else {
            const int oy = eb.by * 8 + k;
            v.x = acc[k][0]; v.y = acc[k][1]; v.z = acc[k][2]; v.w = acc[k][3];
            store = (ox < p.W) & (oy < p.H);
            off = (oy * p.W + ox) * 512 + ch_off;
        }
        v.x = v.x * inv + bv.x; v.y = v.y * inv + bv.y; v.z = v.z * inv + bv.z; v.w = v.w * inv + bv.w;
        if (RELU) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
        float m = dr_max(dr_max(fabsf(v.x), fabsf(v.y)), dr_max(fabsf(v.z), fabsf(v.w)));
        asm volatile("" : "+v"(m), "+v"(off));
        my_amax = dr_max(my_amax, store ? m : 0.0f);
        u32x4 bits;
        bits.x = __float_as_uint(v.x); bits.y = __float_as_uint(v.y); bits.z = __float_as_uint(v.z); bits.w = __float_as_uint(v.w);
        if (!(DBG & 4)) __builtin_amdgcn_raw_buffer_store_b128(bits, rsY, store ? off : 0x7fffffff, 0, 0);
        else asm volatile("" :: "v"(bits));
    };

    // one scheduling region per patch row, as above with half the MFMAs (one channel tile): the fragment two rows ahead, the row's
    // 3 - 9 MFMAs, a slice of the next pass's staging
    auto rstep = [&](auto col_tag, auto r_tag, auto slab_tag, const char *patch, const Blk &nblk, int nslab, char *npatch) {
        constexpr int COL = decltype(col_tag)::value, R = decltype(r_tag)::value, SL = decltype(slab_tag)::value;
        constexpr int DX = COL >> 1, KS = COL & 1, SLOT = (R + COL) % 3;
        if constexpr (R + 2 < 10) frag_read(col_tag, std::integral_constant<int, (R + 2) % 10>{}, patch);
        else if constexpr (COL < 5) frag_read(std::integral_constant<int, (COL + 1) % 6>{}, std::integral_constant<int, (R + 2) % 10>{}, patch);
        if constexpr (COL == 0 && R >= 1 && R <= 6 && !(DBG & 1) && !(DBG & 8)) { patch_load(nblk, nslab, 2 * (R - 1)); patch_load(nblk, nslab, 2 * (R - 1) + 1); }
#pragma unroll
        for (int prod = 0; prod < 3; ++prod)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int r = R - dy;
                if (r < 0 || r > 7) continue;
                const f16x8 a = prod == 2 ? wr[3 * dy + DX][KS][SL][1] : wr[3 * dy + DX][KS][SL][0];
                const f16x8 b = prod == 1 ? fl[SLOT] : fh[SLOT];
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, (SL == 0 && COL == 0 && prod == 0 && dy == 0) ? (f32x4)(0.0f) : acc[r], 0, 0, 0);
            }
        // (the split waits for its load: three columns = 216 MFMAs behind the requests -- with two, as in the kernel above whose columns
        // are twice as long, the first splits stalled on HBM latency: 2.03-2.09 ms against 1.98-2.00; splits in the last two columns only: no better)
        if constexpr (COL >= 3 && (R == 1 || R == 3 || R == 5 || R == 7) && !(DBG & 1) && !(DBG & 16)) patch_split(nblk, 4 * (COL - 3) + (R - 1) / 2, npatch);
        if constexpr (SL == 1 && COL == 5 && !POOL && R >= 3) epi_rows(R - 3);
        if constexpr (SL == 1 && COL == 5 && POOL && (R == 4 || R == 6 || R == 8)) epi_rows((R - 4) / 2);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        constexpr int NM = 3 * ((R < 2 ? R + 1 : 3) - (R > 7 ? R - 7 : 0));
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, (SL == 1 && COL == 5) ? 8 : 4, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto column = [&](auto col_tag, auto slab_tag, const char *patch, const Blk &nblk, int nslab, char *npatch) {
        rstep(col_tag, std::integral_constant<int, 0>{}, slab_tag, patch, nblk, nslab, npatch); rstep(col_tag, std::integral_constant<int, 1>{}, slab_tag, patch, nblk, nslab, npatch);
        rstep(col_tag, std::integral_constant<int, 2>{}, slab_tag, patch, nblk, nslab, npatch); rstep(col_tag, std::integral_constant<int, 3>{}, slab_tag, patch, nblk, nslab, npatch);
        rstep(col_tag, std::integral_constant<int, 4>{}, slab_tag, patch, nblk, nslab, npatch); rstep(col_tag, std::integral_constant<int, 5>{}, slab_tag, patch, nblk, nslab, npatch);
        rstep(col_tag, std::integral_constant<int, 6>{}, slab_tag, patch, nblk, nslab, npatch); rstep(col_tag, std::integral_constant<int, 7>{}, slab_tag, patch, nblk, nslab, npatch);
        rstep(col_tag, std::integral_constant<int, 8>{}, slab_tag, patch, nblk, nslab, npatch); rstep(col_tag, std::integral_constant<int, 9>{}, slab_tag, patch, nblk, nslab, npatch);
    };
    auto pass = [&](auto slab_tag, const char *patch, const Blk &nblk, int nslab, char *npatch) {
        frag_read(DR_C(0), DR_C(0), patch);
        frag_read(DR_C(0), DR_C(1), patch);
        column(DR_C(0), slab_tag, patch, nblk, nslab, npatch); column(DR_C(1), slab_tag, patch, nblk, nslab, npatch);
        column(DR_C(2), slab_tag, patch, nblk, nslab, npatch); column(DR_C(3), slab_tag, patch, nblk, nslab, npatch);
        column(DR_C(4), slab_tag, patch, nblk, nslab, npatch); column(DR_C(5), slab_tag, patch, nblk, nslab, npatch);
    };

    // ---- prologue: (block 0, slab 0) into buffer 0
    Blk cb = decode_blk(0);
#pragma unroll
    for (int i = 0; i < DR_NEL; ++i) patch_load(cb, 0, i);
#pragma unroll
    for (int i = 0; i < DR_NEL; ++i) patch_split(cb, i, dr_smem);
    __syncthreads();
    char *const buf0 = dr_smem, *const buf1 = dr_smem + R2_BUF;
    for (int bi = 0; bi < n_mine; ++bi) {
        const Blk nb = decode_blk(bi + 1);
        eb = cb;
        pass(DR_C(0), buf0, cb, 1, buf1);                      // slab 0 of the block; its slab 1 is staged into the other buffer
        __syncthreads();
        pass(DR_C(1), buf1, nb, 0, buf0);                      // slab 1; the next block's slab 0 is staged
        epi_rows(POOL ? 3 : 7);
        __syncthreads();
        cb = nb;
    }
    if (p.amax_out) {
        unsigned *s_amax = (unsigned *)dr_smem;
        if (tid == 0) *s_amax = 0u;
        __syncthreads();
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) my_amax = fmaxf(my_amax, __shfl_xor(my_amax, o, 64));
        if (lane == 0) atomicMax(s_amax, __float_as_uint(my_amax));
        __syncthreads();
        if (tid == 0 && *s_amax > *(volatile unsigned *)p.amax_out) atomicMax(p.amax_out, *s_amax);
    }
}

/* y = [pool](relu(conv3x3(x, w) + bias)), Cin = Cout = 128; x, y NHWC float32.  d_w2r2 = `direct_r2_pair_weights` (vpr/winograd.py):
 * [2 output-channel halves][4 quarters of a half][9 taps][2 K steps][2 input-channel slabs][hi | lo][64 lanes][8] halfs of s_w w, inv_sw =
 * 1 / s_w; d_amax = 4-byte slot holding (a bound of) max |x|; d_amax_out (or NULL): zeroed slot that receives max |y|.
 * VGG-16 conv2_2: cslam/vpr/netvlad.py:163-171,227. */
CSLAM_API int cslam_conv3x3_direct_r2_dev(const float *d_x, const void *d_w2r2, const float *d_bias, int B, int H, int W, int Cin,
                                          int Cout, int relu, int pool, const unsigned *d_amax, float inv_sw,
                                          unsigned *d_amax_out, float *d_y, void *stream) {
    PTR_DEVICE(d_x);
    ARG_CHECK(d_x && d_w2r2 && d_y && d_amax, "NULL argument");
    ARG_CHECK(B >= 1 && H >= 1 && W >= 1, "empty map");
    ARG_CHECK(Cin == 128 && Cout == 128, "Cin and Cout must be 128");
    ARG_CHECK(!pool || ((H % 2) == 0 && (W % 2) == 0), "pooling needs even H and W");
    ARG_CHECK(inv_sw > 0.0f, "inv_sw must be positive");
    ARG_CHECK((int64_t)H * W * 512 < 0x7ffffff0ll, "one image's maps must stay below 2 GiB (32-bit buffer offsets)");
    ConvDirectR2Args a;
    a.x = d_x; a.w2 = (const f16x8 *)d_w2r2; a.bias = d_bias; a.y = d_y;
    a.B = B; a.H = H; a.W = W;
    a.gxb = (int)ceil_div64(W, 16); a.gyb = (int)ceil_div64(H, 8);
    const int64_t nblk = (int64_t)B * a.gxb * a.gyb;
    ARG_CHECK(nblk < (1ll << 30), "too many blocks for one launch");
    a.nblk = (int)nblk;
    a.amax_in = d_amax; a.inv_sw = inv_sw; a.amax_out = d_amax_out;
    const int n_cu = cslam_cu_count();
    ARG_CHECK(n_cu >= 2, "no HIP device");
    // workgroup pairs: (block stream, output-channel half); whole pairs only, XCD-aligned when the device allows it
    int grid = n_cu % 16 == 0 ? n_cu : (n_cu & ~1);
    if (2 * nblk < grid) grid = (int)(2 * nblk);
    hipStream_t st = (hipStream_t)stream;
#define DR2_LAUNCH(P, R) do { \
        static DeviceOnce once; int once_dev; \
        if (once.todo(&once_dev)) { \
            HIP_TRY(hipFuncSetAttribute((const void *)conv3x3_direct_r2_kernel<P, R>, hipFuncAttributeMaxDynamicSharedMemorySize, DR2_LDS)); \
            once.done(once_dev); } \
        hipLaunchKernelGGL((conv3x3_direct_r2_kernel<P, R>), dim3(grid), dim3(256), DR2_LDS, st, a); } while (0)
#ifdef CSLAM_ABLATIONS
    if (const char *e = getenv("CSLAM_DR_DBG")) {              // timing-only ablations (wrong results): 1 = no staging inside the loop, 4 = no stores, 8 = no requests, 16 = no splits
        const int d = atoi(e);
#define DR2_LAUNCH_D(D) do { HIP_TRY(hipFuncSetAttribute((const void *)conv3x3_direct_r2_kernel<true, true, D>, hipFuncAttributeMaxDynamicSharedMemorySize, DR2_LDS)); \
        hipLaunchKernelGGL((conv3x3_direct_r2_kernel<true, true, D>), dim3(grid), dim3(256), DR2_LDS, st, a); HIP_TRY(hipGetLastError()); return CSLAM_OK; } while (0)
        if (d == 1) DR2_LAUNCH_D(1);
        if (d == 4) DR2_LAUNCH_D(4);
        if (d == 5) DR2_LAUNCH_D(5);
        if (d == 8) DR2_LAUNCH_D(8);                           // no requests inside the loop (the splits work on stale registers)
        if (d == 16) DR2_LAUNCH_D(16);                         // requests, no splits
#undef DR2_LAUNCH_D
    }
#endif
    if (pool && relu) DR2_LAUNCH(true, true);
    else if (pool) DR2_LAUNCH(true, false);
    else if (relu) DR2_LAUNCH(false, true);
    else DR2_LAUNCH(false, false);
#undef DR2_LAUNCH
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

/* y = [pool](relu(conv3x3(x, w) + bias)), Cin = 64, Cout = 128; x, y NHWC float32.  d_w2r = `direct_r_pair_weights` (vpr/winograd.py):
 * [4 output-channel quarters][9 taps][2 K steps][2 channel tiles][hi | lo][64 lanes][8] halfs of s_w w, inv_sw = 1 / s_w; d_amax = 4-byte
 * slot holding (a bound of) max |x|; d_amax_out (or NULL): zeroed slot that receives max |y|. */
static int conv_direct_r_launch(const float *d_x, const void *d_w2r, const float *d_bias, int B, int H, int W, int Cin,
                                int Cout, int relu, int pool, const unsigned *d_amax, float inv_sw,
                                unsigned *d_amax_out, float *d_y, void *stream, int out_pairs, float wl1, float bmax, unsigned *d_bound_out) {
    PTR_DEVICE(d_x);
    ARG_CHECK(!out_pairs || (!pool && relu && d_bound_out && wl1 >= 0.0f && bmax >= 0.0f), "pair-format output: ReLU, no pooling, wl1, bmax and the bound slot");
    ARG_CHECK(d_x && d_w2r && d_y && d_amax, "NULL argument");
    ARG_CHECK(B >= 1 && H >= 1 && W >= 1, "empty map");
    ARG_CHECK(Cin == 64 && Cout == 128, "Cin must be 64 and Cout 128");
    ARG_CHECK(!pool || ((H % 2) == 0 && (W % 2) == 0), "pooling needs even H and W");
    ARG_CHECK(inv_sw > 0.0f, "inv_sw must be positive");
    ARG_CHECK((int64_t)H * W * 512 < 0x7ffffff0ll, "one image's maps must stay below 2 GiB (32-bit buffer offsets)");
    ConvDirectRArgs a;
    a.x = d_x; a.w2 = (const f16x8 *)d_w2r; a.bias = d_bias; a.y = d_y;
    a.B = B; a.H = H; a.W = W;
    a.gxb = (int)ceil_div64(W, 16); a.gyb = (int)ceil_div64(H, 8);
    const int64_t nblk = (int64_t)B * a.gxb * a.gyb;
    ARG_CHECK(nblk < (1ll << 30), "too many blocks for one launch");
    a.nblk = (int)nblk;
    a.relu = relu; a.amax_in = d_amax; a.inv_sw = inv_sw; a.amax_out = d_amax_out;
    a.wl1 = wl1; a.bmax = bmax; a.bound_out = d_bound_out;
    const int n_cu = cslam_cu_count();
    ARG_CHECK(n_cu > 0, "no HIP device");
    const int grid = (int)(nblk < n_cu ? nblk : n_cu);
    hipStream_t st = (hipStream_t)stream;
    if (out_pairs) {
        static DeviceOnce once_p; int once_dev_p;
        if (once_p.todo(&once_dev_p)) {
            HIP_TRY(hipFuncSetAttribute((const void *)conv3x3_direct_r_kernel<false, true, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DR_LDS));
            once_p.done(once_dev_p); }
        hipLaunchKernelGGL((conv3x3_direct_r_kernel<false, true, 0, true>), dim3(grid), dim3(256), DR_LDS, st, a);
        HIP_TRY(hipGetLastError());
        return CSLAM_OK;
    }
#define DR_LAUNCH(P, R) do { \
        static DeviceOnce once; int once_dev; \
        if (once.todo(&once_dev)) { \
            HIP_TRY(hipFuncSetAttribute((const void *)conv3x3_direct_r_kernel<P, R>, hipFuncAttributeMaxDynamicSharedMemorySize, DR_LDS)); \
            once.done(once_dev); } \
        hipLaunchKernelGGL((conv3x3_direct_r_kernel<P, R>), dim3(grid), dim3(256), DR_LDS, st, a); } while (0)
#ifdef CSLAM_ABLATIONS
    if (const char *e = getenv("CSLAM_DR_DBG")) {
        const int d = atoi(e);
#define DR_LAUNCH_D(D) do { HIP_TRY(hipFuncSetAttribute((const void *)conv3x3_direct_r_kernel<false, true, D>, hipFuncAttributeMaxDynamicSharedMemorySize, DR_LDS)); \
        hipLaunchKernelGGL((conv3x3_direct_r_kernel<false, true, D>), dim3(grid), dim3(256), DR_LDS, st, a); HIP_TRY(hipGetLastError()); return CSLAM_OK; } while (0)
        if (d == 1) DR_LAUNCH_D(1);
        if (d == 4) DR_LAUNCH_D(4);
        if (d == 5) DR_LAUNCH_D(5);
#undef DR_LAUNCH_D
    }
#endif
    if (pool && relu) DR_LAUNCH(true, true);
    else if (pool) DR_LAUNCH(true, false);
    else if (relu) DR_LAUNCH(false, true);
    else DR_LAUNCH(false, false);
#undef DR_LAUNCH
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}
CSLAM_API int cslam_conv3x3_direct_r_dev(const float *d_x, const void *d_w2r, const float *d_bias, int B, int H, int W, int Cin,
                                         int Cout, int relu, int pool, const unsigned *d_amax, float inv_sw,
                                         unsigned *d_amax_out, float *d_y, void *stream) {
    return conv_direct_r_launch(d_x, d_w2r, d_bias, B, H, W, Cin, Cout, relu, pool, d_amax, inv_sw, d_amax_out, d_y, stream, 0, 0.0f, 0.0f, nullptr);
}
/* The same convolution (64 -> 128 channels, + bias + ReLU, no pooling) with y written in the PAIR FORMAT of cslam_conv_igemm_h2p_dev:
 * [B,H,W,4 blocks][hi 32 | lo 32] fp16 of s y, s the power of two of the bound *d_amax wl1 + bmax, which goes to d_bound_out (wl1 = max_co
 * sum |w[co]|, bmax = max |bias|); d_amax_out (optional, zeroed) receives the measured max |y|.  VGG-16 conv2_1 in front of a conv2_2
 * that stages its patch from pairs (cslam_conv3x3_direct_hp_dev): cslam/vpr/netvlad.py:163-171,227. */
CSLAM_API int cslam_conv3x3_direct_r_pairs_dev(const float *d_x, const void *d_w2r, const float *d_bias, int B, int H, int W, int Cin,
                                               int Cout, const unsigned *d_amax, float inv_sw, float wl1, float bmax,
                                               unsigned *d_amax_out, unsigned *d_bound_out, void *d_y, void *stream) {
    return conv_direct_r_launch(d_x, d_w2r, d_bias, B, H, W, Cin, Cout, 1, 0, d_amax, inv_sw, d_amax_out, (float *)d_y, stream, 1, wl1, bmax, d_bound_out);
}
