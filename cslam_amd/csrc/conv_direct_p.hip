// conv_direct_p.hip -- 3x3 / stride 1 / pad 1 convolution 64 -> 64 channels BETWEEN PAIR-FORMAT MAPS as one direct kernel whose weights
// never leave the register files (gfx950): the four stride-1 convolutions of ResNet-18/34's layer1 (cslam/vpr/cosplace_utils/network.py:
// 38-68, the reference's default extractor) -- a quarter of CosPlace's extract pass, and the layer the implicit GEMM (conv_igemm.hip)
// runs worst: its 64-column tile gives every activation block half the MFMAs a 128-column tile gets.
//
// The decomposition of conv_stem_direct_h.hip's second layer (four waves per workgroup, one per SIMD; wave w owns OUTPUT channels
// 16 w .. 16 w + 15 for all 128 pixels of a block; its 144 registers of weight fragments are loaded once per kernel; a B fragment = one
// patch row x 32 channels at a column shift dx serves the three taps dy = 0, 1, 2: 2 fragment reads per up to 9 MFMAs; no partial sums,
// ONE barrier per block), with what the pair format (conv_igemm.hip: [pixel][32-channel block][hi 32 | lo 32] fp16 of s x) adds:
//   * the input patch comes by LDS-DMA (buffer_load_dwordx4 ... lds): no register, no conversion, no VALU work per value (the float32
//     form XF32, for the pooled stem output that opens the chain, stages and splits it like conv_direct_r.hip).  A 256-byte pixel is exactly one LDS bank row; chunk s of pixel column c sits in
//     slot s ^ (2 c & 15): a fragment read (8 pixels of K group g, 8 of g ^ 1, consecutive columns) falls on 16 distinct slots for
//     every column shift.  The DMA writes 64 consecutive 16-byte slots per instruction and every lane chooses its SOURCE chunk, so the
//     swizzle costs nothing.
//   * a block is TWO HALF-BLOCKS of 8 x 8 pixels, consecutive in the order (image, block row, column of eight): for maps whose width
//     is a multiple of 16 that is the 8 x 16 block of the other direct kernels, for ResNet's 56 x 56 maps the seventh half of a block
//     row pairs with the first of the next (8 x 16 blocks would leave an eighth of the MFMAs on pixels that do not exist).  A half's
//     patch is 10 x 10 pixels = 25 DMA instructions; lanes 0-7 / 8-15 of a fragment read the two halves.
//   * the epilogue writes pairs (two 8-byte runs per lane and output row) scaled for the bound max|x| wl1 + bmax (+ max|shortcut|) as
//     conv_igemm.hip does, reads the shortcut in either format, and rides in the NEXT block's columns 2-4; the next block's DMA requests
//     ride in columns 0-1, this block's shortcut loads in column 0; everything vector-memory is drained by ONE s_waitcnt vmcnt(0) at the
//     block's end, four columns after the last request.
// Arithmetic as conv_igemm.hip: acc = wh xh + wh xl + wl xh in fp32, y = act(acc / (s_x s_w) + bias (+ shortcut)).
// Measured (profiles/r06_*): 0.58 / 0.63 ms per 1000 frames of 56 x 56 without / with a shortcut (the implicit GEMM: 0.80 / 0.88), fabric
// traffic 1.001 x algorithmic, SQ_LDS_BANK_CONFLICT 0; back to back it holds the board at 1399 W of its 1400 W cap: its time is its energy
// (0.63 pJ per issued flop + the bytes) over the cap, which is why the eight-wave form (NRW = 4) and tighter schedules change nothing.
#include <stdlib.h>
#include <type_traits>
#include <hip/hip_fp16.h>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define DP_HALFB (100 * 256)               // a half-block's 10 x 10-pixel patch
#define DP_NDMA(NW) ((50 + (NW) - 1) / (NW))   // DMA instructions per wave and block: 4 x 13 = 52, 8 x 7 = 56 >= 50 (the last ones: a sink behind the patch)
#define DP_PATCHB(NW) ((NW) * DP_NDMA(NW) * 1024)   // 53 248 / 57 344 bytes per buffer
#define DP_RP (10 * 256)                    // bytes per patch row of a half
#define DP_LDS(NW) (2 * DP_PATCHB(NW))
#define DP_OOB 0x7fffffff

struct ConvDirectPArgs {
    const char *x; const unsigned *xbound; const unsigned *amax_in;
    const f16x8 *w2; const float *bias; float inv_sw;
    const char *res; const unsigned *res_bound;
    char *y; float wl1, bmax; unsigned *bound_out; unsigned *amax_out;
    int B, H, W, gxh, gyb, nhalf, nblk;
    int relu;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t dp_rsrc(const char *base, int64_t bytes) {
    const uint64_t a = (uint64_t)base;
    const uint64_t u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)a);
    const int n = __builtin_amdgcn_readfirstlane((int)(bytes > 0x7ffffff0ll ? 0x7ffffff0ll : bytes));
    return __builtin_amdgcn_make_buffer_rsrc((void *)u, 0, n, 0x00020000);
}
// (a __device__ function, not the builtin inside the kernel's lambda: conv_igemm.hip)
__device__ __forceinline__ void dp_blds16(__amdgpu_buffer_rsrc_t rs, int voff, char *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)lds_wave_base, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ float dp_scale(unsigned amax_bits) {            // conv_igemm.hip::ci_scale
    const float a = fminf(fmaxf(__uint_as_float(amax_bits), 1e-30f), 1e30f);
    int e;
    (void)frexpf(a, &e);
    return ldexpf(1.0f, 14 - e);
}
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// (float)half HI of hb * s + v: hipcc selects ONE v_fma_mix_f32 for this (as inline assembly the scheduler cannot place it: a
// sched_group_barrier pipeline leaves every asm statement of a region behind the region's last MFMA)
template <int HI>
__device__ __forceinline__ float dp_fma_half(unsigned hb, float s, float v) {
    return __builtin_fmaf((float)__builtin_bit_cast(f16x2, hb)[HI], s, v);
}
__device__ __forceinline__ unsigned dp_pk(float a, float b) {              // (fp16 rn(a), fp16 rn(b)) in one dword
    const __half2 h = __floats2half2_rn(a, b);
    return *(const unsigned *)&h;
}

// RES: shortcut 0 none, 1 float32 NHWC, 2 pair format; OUTP: y in pair format (else float32 NHWC)
// DBG (builds with -DCSLAM_ABLATIONS only; WRONG results, timing): 1 no epilogue, 2 no patch requests inside the loop, 4 no shortcut loads,
// 8 no fragment reads after a block's first two, 16 no stores (the epilogue's arithmetic stays)
// NRW: output rows of a block per wave.  8: four waves, one per SIMD (512 registers each); 4: EIGHT waves, two per SIMD -- wave w = channel
// quarter w & 3 of the block's rows 4 (w >> 2) .. + 3 (six patch rows: a fragment serves two taps on average instead of 2.4), 256
// registers each: the two waves of a SIMD cover one another's vector-memory instructions (45 per block: each holds its wave's issue for
// tens of cycles, and with one wave per SIMD the matrix pipe idles meanwhile -- the ablations of profiles/r06_b_dp_ablations.log)
// XF32: x is a float32 NHWC map (the pooled stem output that opens ResNet's pair-format chain), scaled by the power of two of its max |x|
// slot and split into pairs while the patch is staged through registers (conv_direct_r.hip's staging: 13 16-byte loads per thread in a
// block's first column, their split in columns 1 .. 3, into this kernel's LDS image)
template <int RES, bool OUTP, int NRW, int DBG = 0, bool XF32 = false>
__global__ __launch_bounds__(2048 / NRW, 1) void conv3x3_direct_p_kernel(ConvDirectPArgs p) {
    constexpr int NW = 32 / NRW, NPR = NRW + 2, NDMA = DP_NDMA(NW), PATCHB = DP_PATCHB(NW);
    extern __shared__ __attribute__((aligned(16))) char dp_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave & 3, rh = wave >> 2;                   // channel quarter, row group (0 with four waves)
    const int gq = lane >> 4, l15 = lane & 15;
    const int lh = l15 >> 3, l7 = l15 & 7;                     // the lane's half-block and pixel column inside it

    const float sc = dp_scale(XF32 ? *p.amax_in : *p.xbound);
    const float inv = p.inv_sw / sc;
    float s_out = 1.0f, inv_sres = 0.0f;
    if (OUTP) {
        const float xmax = __uint_as_float(*p.amax_in);
        const float rmax = RES ? __uint_as_float(*p.res_bound) : 0.0f;
        const float bound = (xmax * p.wl1 + p.bmax + rmax) * 1.001f;       // >= max |y| whatever the rounding of the products
        s_out = dp_scale(__float_as_uint(bound));
        if (blockIdx.x == 0 && tid == 0) *p.bound_out = __float_as_uint(bound);
    }
    if (RES == 2) inv_sres = 1.0f / dp_scale(*p.res_bound);
    const float floor_ = p.relu ? 0.0f : -INFINITY;
    float neg1 = -1.0f;                                        // (a run-time value: with the literal hipcc folds fma(h, -1, u) into a conversion and a subtraction)
    asm volatile("" : "+v"(neg1));

    // blocks to workgroups by XCD (contiguous eighths: the halo neighbours share goes through one L2)
    const bool by_xcd = (gridDim.x & 7) == 0;
    const int wg_xcd = by_xcd ? (int)blockIdx.x & 7 : 0, wg_j = by_xcd ? (int)blockIdx.x >> 3 : (int)blockIdx.x;
    const int wg_per = by_xcd ? (int)gridDim.x >> 3 : (int)gridDim.x;
    const int per_xcd = by_xcd ? (p.nblk + 7) >> 3 : p.nblk;
    const int blk_beg = wg_xcd * per_xcd;
    const int blk_cnt = min(per_xcd, p.nblk - blk_beg);
    const int n_mine = blk_cnt > wg_j ? (blk_cnt - wg_j + wg_per - 1) / wg_per : 0;
    if (n_mine <= 0) return;

    // ---- this wave's weights: [tap][32-channel K step][hi | lo], register-resident for the whole kernel (conv_stem_direct_h.hip)
    f16x8 wr[9][2][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) {
                wr[tap][ks][hl] = p.w2[(((wq * 9 + tap) * 2 + ks) * 2 + hl) * 64 + lane];
                // pinned in the accumulation half of the register file (conv_stem_direct_h.hip).  With two waves per SIMD hipcc splits a wave's
                // 256 registers 128 + 128: the accumulators and the first 28 fragments fill the accumulation half, the last 8 stay architectural
                if (NRW == 8 || tap < 7) asm volatile("" : "+a"(wr[tap][ks][hl]));
                else asm volatile("" : "+v"(wr[tap][ks][hl]));
            }
    const float4 bv = p.bias ? *(const float4 *)(p.bias + 16 * wq + 4 * gq) : make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- blocks.  Half-block hb -> (image, block row, column of eight); a block = halves 2 k, 2 k + 1.  All of it wave-uniform.
    const int img_px = p.H * p.W, imgB = img_px * 256;
    struct Blk { int by8[2], bx8[2], org[2]; int64_t ioff; int range; };   // org: byte offset of a half's first pixel from image img0 (both formats: 256 B
                                                                                 // per pixel); ioff / range: image img0's offset in a tensor, bytes of it and the next
    // A workgroup's blocks are wg_per apart: the (image, block row, column) of its next block's first half is the current one plus a
    // constant step with two carries -- no division in the loop (as divisions the decode was ~130 scalar instructions in front of a
    // block's first MFMA)
    const int per_img = p.gxh * p.gyb;
    int st_hb = 2 * (blk_beg + wg_j), st_img = st_hb / per_img, st_by = (st_hb - st_img * per_img) / p.gxh, st_bx = st_hb - st_img * per_img - st_by * p.gxh;
    const int step_hb = 2 * wg_per, step_img = step_hb / per_img, step_by = (step_hb - step_img * per_img) / p.gxh,
              step_bx = step_hb - step_img * per_img - step_by * p.gxh;
    // (integer masks, no bool -> int conversions: hipcc 7.2 moves a uniform value that passes through a zero-extended condition into the
    // vector registers, and the buffer descriptors built from it then cost v_readfirstlane chains in front of a block's first MFMAs)
    auto make_blk = [&](int live_mask) {                       // the block at the state; live_mask: -1 it exists, 0 it does not
        Blk b;
        int bx1 = st_bx + 1;
        const int z0 = (bx1 - p.gxh) >> 31;                    // -1: same block row; 0: the row is through
        bx1 &= z0;
        int by1 = st_by + 1 + z0;
        const int z1 = (by1 - p.gyb) >> 31;                    // -1: same image; 0: the image is through
        by1 &= z1;
        const int dimg = 1 + z1;
        b.ioff = (int64_t)st_img * imgB;                       // buffers begin at image img0 and cover it and the next one (a block's second
        b.range = imgB + (imgB & ~((p.B - st_img - 2) >> 31)); // half may be the next image's first); x, y and the shortcut share the geometry
        const int ok0 = live_mask & ((st_hb - p.nhalf) >> 31), ok1 = live_mask & ((st_hb + 1 - p.nhalf) >> 31);
        b.by8[0] = ((st_by * 8) & ok0) | ((1 << 24) & ~ok0);    // a half that does not exist: every row is outside the map
        b.by8[1] = ((by1 * 8) & ok1) | ((1 << 24) & ~ok1);
        b.bx8[0] = st_bx * 8; b.bx8[1] = bx1 * 8;
        b.org[0] = (st_by * 8 * p.W + st_bx * 8) * 256;
        b.org[1] = (dimg * img_px + by1 * 8 * p.W + bx1 * 8) * 256;
        return b;
    };
    auto advance_blk = [&]() {
        st_hb += step_hb;
        int t = st_bx + step_bx - p.gxh, n = t >> 31;
        st_bx = t + (n & p.gxh);
        t = st_by + step_by + 1 + n - p.gyb; n = t >> 31;
        st_by = t + (n & p.gyb);
        st_img += step_img + 1 + n;
    };
    auto rsrc_of = [&](const char *base, const Blk &b) { return dp_rsrc(base + b.ioff, b.range); };

    // ---- patch requests: instruction i of this wave = DMA piece j = 4 i + wave: half j / 25, its pixels 4 (j % 25) .. + 3 (patch row
    // major, 10 per row); lane -> pixel (lane >> 4) of the four, physical slot lane & 15 = logical 16-byte chunk ^ (2 c & 15)
    int dm_rel[NDMA], dm_r[NDMA], dm_c[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int j = NW * i + wave;
        const int pp = 4 * (j % 25) + (lane >> 4);
        const int r = pp / 10, c = pp - 10 * r;
        const int s = (lane & 15) ^ ((2 * c) & 15);
        dm_r[i] = r - 1; dm_c[i] = c - 1;
        dm_rel[i] = ((r - 1) * p.W + (c - 1)) * 256 + s * 16;
    }
    auto dma_issue = [&](auto i_tag, const Blk &b, __amdgpu_buffer_rsrc_t rsX, char *npatch) {
        constexpr int I = decltype(i_tag)::value;
        const int j = NW * I + wave;                           // wave-uniform; the half is a compile-time fact except for the instruction that holds piece 25
        int by8, bx8, org;
        if constexpr (NW * I + NW - 1 < 25) { by8 = b.by8[0]; bx8 = b.bx8[0]; org = b.org[0]; }
        else if constexpr (NW * I >= 25) { by8 = b.by8[1]; bx8 = b.bx8[1]; org = b.org[1]; }
        else { const int m = (24 - j) >> 31; by8 = (b.by8[1] & m) | (b.by8[0] & ~m); bx8 = (b.bx8[1] & m) | (b.bx8[0] & ~m); org = (b.org[1] & m) | (b.org[0] & ~m); }
        if constexpr (NW * I + NW - 1 >= 50) by8 |= ((49 - j) >> 31) & (1 << 24);          // pieces 50 ..: the sink
        int rel = dm_rel[I];
        asm volatile("" : "+v"(rel));                          // (pins the request's arithmetic to its region: hipcc otherwise gathers all thirteen in front of the first)
        const int off = org + rel;
        const bool in = ((unsigned)(by8 + dm_r[I]) < (unsigned)p.H) & ((unsigned)(bx8 + dm_c[I]) < (unsigned)p.W);
        dp_blds16(rsX, in ? off : DP_OOB, npatch + j * 1024);
    };

    // ---- XF32: the patch through registers.  Element e = i * 256 + tid = (patch pixel e >> 4 of the 2 x 100, channels 4 (e & 15) .. + 3)
    constexpr int NST = XF32 ? 13 : 1;
    int sg_rel[NST], sg_rc[NST], sg_dst[NST];
    u32x4 stg[NST];
    if (XF32) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int e = i * 256 + tid, pp = e >> 4, f4 = e & 15;
            const int h = pp >= 100 ? 1 : 0, q = pp - 100 * h;
            const int r = q / 10, c = q - 10 * r;
            const int s_hi = (f4 >> 3) * 8 + ((f4 & 7) >> 1);                 // logical chunk of the hi halves: K step, group of 8 channels
            sg_rel[i] = ((r - 1) * p.W + (c - 1)) * 256 + f4 * 16;
            sg_rc[i] = pp < 200 ? ((r - 1) & 0xffff) | (((c - 1) & 0x3fff) << 16) | (h << 30) : (1 << 14);    // (row 16384: outside every map)
            sg_dst[i] = pp < 200 ? h * DP_HALFB + q * 256 + ((s_hi ^ ((2 * c) & 15)) << 4) + (f4 & 1) * 8 : 2 * DP_HALFB + (tid & 63) * 16;   // (no element: the sink)
        }
    }
    auto stage_load = [&](auto i_tag, const Blk &b, __amdgpu_buffer_rsrc_t rsX) {
        constexpr int I = decltype(i_tag)::value;
        int rc = sg_rc[I];
        asm volatile("" : "+v"(rc));                           // (pinned to its region, as the requests)
        const int m = -(rc >> 30);                             // -1: second half
        const int by8 = (b.by8[1] & m) | (b.by8[0] & ~m), bx8 = (b.bx8[1] & m) | (b.bx8[0] & ~m), org = (b.org[1] & m) | (b.org[0] & ~m);
        const int r = (rc << 16) >> 16, c = (rc << 2) >> 18;
        const bool in = ((unsigned)(by8 + r) < (unsigned)p.H) & ((unsigned)(bx8 + c) < (unsigned)p.W);
        stg[I] = __builtin_amdgcn_raw_buffer_load_b128(rsX, in ? org + sg_rel[I] : DP_OOB, 0, 0);
    };
    auto stage_split = [&](auto i_tag, char *npatch) {
        constexpr int I = decltype(i_tag)::value;
        u32x4 v = stg[I];
        asm volatile("" : "+v"(v));
        const float w0 = __uint_as_float(v.x) * sc, w1 = __uint_as_float(v.y) * sc, w2 = __uint_as_float(v.z) * sc, w3 = __uint_as_float(v.w) * sc;
        const unsigned h01 = dp_pk(w0, w1), h23 = dp_pk(w2, w3);
        const unsigned l01 = dp_pk(dp_fma_half<0>(h01, neg1, w0), dp_fma_half<1>(h01, neg1, w1));
        const unsigned l23 = dp_pk(dp_fma_half<0>(h23, neg1, w2), dp_fma_half<1>(h23, neg1, w3));
        char *d = npatch + sg_dst[I];
        *(u32x2 *)d = (u32x2){h01, h23};
        *(u32x2 *)(npatch + (sg_dst[I] ^ 64)) = (u32x2){l01, l23};      // (the lo halves: logical chunk + 4 = physical chunk ^ 4)
    };

    // ---- the lane's output pixel of a block: half lh, column l7; offset of the block's row 0 from image img0, and how many of the 8 rows exist
    struct OutPx { int pix, rows; };
    auto out_px = [&](const Blk &b) {
        OutPx o;
        const int by8 = lh ? b.by8[1] : b.by8[0], bx8 = lh ? b.bx8[1] : b.bx8[0], org = lh ? b.org[1] : b.org[0];
        o.pix = org + l7 * 256;
        const int left = p.H - by8;                            // (negative for a half that does not exist)
        o.rows = (bx8 + l7 < p.W) ? (left < 8 ? left : 8) : 0;
        return o;
    };
    const int rowB = p.W * 256;
    // byte offsets of the lane's four channels 16 wave + 4 gq .. + 3 inside a pixel: pair format [block wave >> 1][hi | lo][32], float32
    const int ch_pair = (wq >> 1) * 128 + (16 * (wq & 1) + 4 * gq) * 2;
    const int ch_f32 = (16 * wq + 4 * gq) * 4;
    const int row0 = rh * NRW;                                 // this wave's first output row of the block = its first patch row

    // ---- shortcut of the block being multiplied: loaded in its first column, used by its epilogue one block later
    u32x4 rnew[NRW], rres[NRW];
    auto res_load = [&](int r, const OutPx &o, __amdgpu_buffer_rsrc_t rsR) {
        if (RES == 0) return;
        int off = o.pix + (row0 + r) * rowB + (RES == 2 ? ch_pair : ch_f32);
        off = row0 + r < o.rows ? off : DP_OOB;
        if (RES == 2) {
            const u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(rsR, off, 0, 0), c = __builtin_amdgcn_raw_buffer_load_b64(rsR, off, 64, 0);
            rnew[r] = (u32x4){a.x, a.y, c.x, c.y};
        } else {
            rnew[r] = __builtin_amdgcn_raw_buffer_load_b128(rsR, off, 0, 0);
        }
    };

    // ---- epilogue of the finished block (accumulators copied to eacc), one output row in two pieces
    float my_amax = 0.0f;
    f32x4 eacc[NRW];
    OutPx eo = {0, 0};
    float ev[4];
    auto epi_a = [&](int r) {                                  // rescale, bias, shortcut, activation, max |y|
        asm volatile("" : "+v"(eacc[r]));                    // (pins the piece to its region: pure arithmetic on values that are ready at the top of
                                                               // the block is otherwise gathered there, in front of the block's first MFMAs)
        ev[0] = __builtin_fmaf(eacc[r][0], inv, bv.x); ev[1] = __builtin_fmaf(eacc[r][1], inv, bv.y);
        ev[2] = __builtin_fmaf(eacc[r][2], inv, bv.z); ev[3] = __builtin_fmaf(eacc[r][3], inv, bv.w);
        if (RES == 2) {
            ev[0] = dp_fma_half<0>(rres[r].x, inv_sres, ev[0]); ev[1] = dp_fma_half<1>(rres[r].x, inv_sres, ev[1]);
            ev[2] = dp_fma_half<0>(rres[r].y, inv_sres, ev[2]); ev[3] = dp_fma_half<1>(rres[r].y, inv_sres, ev[3]);
            ev[0] = dp_fma_half<0>(rres[r].z, inv_sres, ev[0]); ev[1] = dp_fma_half<1>(rres[r].z, inv_sres, ev[1]);
            ev[2] = dp_fma_half<0>(rres[r].w, inv_sres, ev[2]); ev[3] = dp_fma_half<1>(rres[r].w, inv_sres, ev[3]);
        } else if (RES == 1) {
            ev[0] += __uint_as_float(rres[r].x); ev[1] += __uint_as_float(rres[r].y); ev[2] += __uint_as_float(rres[r].z); ev[3] += __uint_as_float(rres[r].w);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) ev[e] = __builtin_fmaxf(ev[e], floor_);
        const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(ev[0]), __builtin_fabsf(ev[1])), __builtin_fmaxf(__builtin_fabsf(ev[2]), __builtin_fabsf(ev[3])));
        my_amax = __builtin_fmaxf(my_amax, row0 + r < eo.rows ? m : 0.0f);
        asm volatile("" : "+v"(my_amax));                    // (pinned to its region: left alone hipcc gathers the eight rows' maxima into one chain in front of a later MFMA)
    };
    auto epi_b = [&](int r, __amdgpu_buffer_rsrc_t rsY) {       // the stores
        asm volatile("" : "+v"(ev[0]), "+v"(ev[1]), "+v"(ev[2]), "+v"(ev[3]));
        int off = eo.pix + (row0 + r) * rowB + (OUTP ? ch_pair : ch_f32);
        off = row0 + r < eo.rows ? off : DP_OOB;
        if (OUTP) {
            const float u0 = ev[0] * s_out, u1 = ev[1] * s_out, u2 = ev[2] * s_out, u3 = ev[3] * s_out;
            const unsigned h01 = dp_pk(u0, u1), h23 = dp_pk(u2, u3);
            const unsigned l01 = dp_pk(dp_fma_half<0>(h01, neg1, u0), dp_fma_half<1>(h01, neg1, u1));
            const unsigned l23 = dp_pk(dp_fma_half<0>(h23, neg1, u2), dp_fma_half<1>(h23, neg1, u3));
            if (!(DBG & 16)) {
                __builtin_amdgcn_raw_buffer_store_b64((u32x2){h01, h23}, rsY, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64((u32x2){l01, l23}, rsY, off, 64, 0);
            } else asm volatile("" :: "v"(h01), "v"(h23), "v"(l01), "v"(l23), "v"(off));
        } else {
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(ev[0]), __float_as_uint(ev[1]), __float_as_uint(ev[2]), __float_as_uint(ev[3])}, rsY, off, 0, 0);
        }
    };

    // ---- the products.  acc[r]: output row r of both halves, lane (l15, gq) = pixel (half lh, column l7), channels 16 wave + 4 gq .. + 3
    f32x4 acc[NRW];
    int fro[3][2][2];                                          // fragment offset at column shift dx, K step, hi | lo: pixel column l7 + dx of half lh
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) {
                const int c = l7 + dx;
                fro[dx][ks][hl] = lh * DP_HALFB + row0 * DP_RP + c * 256 + (((ks * 8 + hl * 4 + gq) ^ ((2 * c) & 15)) << 4);
            }
    f16x8 fh[3], fl[3];
    auto frag_read = [&](auto col_tag, auto r_tag, const char *patch) {
        constexpr int COL = decltype(col_tag)::value, R = decltype(r_tag)::value;
        constexpr int DX = COL >> 1, KS = COL & 1, SLOT = (COL * NPR + R) % 3;
        fh[SLOT] = *(const f16x8 *)(patch + fro[DX][KS][0] + R * DP_RP);
        fl[SLOT] = *(const f16x8 *)(patch + fro[DX][KS][1] + R * DP_RP);
    };
    // what rides in the region of (column, patch row) -- nothing in the matrix loop waits for it.  Four waves (NRW = 8):
    //   column 0, rows 1..8: the shortcut loads of this block's output rows 0..7
    //   columns 0 / 1, rows 2..7 / 2..8: the 13 patch requests of the next block
    //   columns 2 / 3 / 4, rows 2..7: the previous block's epilogue, output row 3 (COL - 2) + (R - 2) / 2 .. (16 pieces, rows 0..7)
    //   column 5, rows 2..3: the block after next is decoded (scalar unit)
    // eight waves (NRW = 4): shortcut loads column 0 rows 1..4, requests column 0 rows 2..4 and column 1 rows 1..4, the epilogue's
    // 8 pieces in columns 2 / 3 / 4 rows 1..3
    Blk cb, nb, nnb, eb;
    OutPx co;
    __amdgpu_buffer_rsrc_t rsXn, rsRc, rsYe;
    int more2 = 0;                                        // a block after next exists
    auto extras = [&](auto col_tag, auto r_tag, char *npatch) {
        constexpr int COL = decltype(col_tag)::value, R = decltype(r_tag)::value;
        if constexpr (COL == 0 && R == 0) { co = out_px(cb); rsRc = rsrc_of(RES ? p.res : p.y, cb); rsXn = rsrc_of(p.x, nb); rsYe = rsrc_of(p.y, eb); }
        if constexpr (COL == 0 && R >= 1 && R <= NRW && !(DBG & 4)) res_load(R - 1, co, rsRc);
        if constexpr (NRW == 8 && XF32) {
            // loads: column 0 rows 1..7 (two each: 13 + one repeat), splits: columns 1 / 2 rows 2..8 (13 of the 14 slots)
            if constexpr (COL == 0 && R >= 1 && R <= 7) { stage_load(std::integral_constant<int, 2 * (R - 1)>{}, nb, rsXn); if constexpr (R < 7) stage_load(std::integral_constant<int, 2 * (R - 1) + 1>{}, nb, rsXn); }
            if constexpr ((COL == 1 || COL == 2) && R >= 2 && R <= 8) { constexpr int I = 7 * (COL - 1) + (R - 2); if constexpr (I < 13) stage_split(std::integral_constant<int, I>{}, npatch); }
            if constexpr (COL >= 3 && COL <= 5 && R >= 2 && R <= 7 && !(DBG & 1)) {
                constexpr int piece = 6 * (COL - 3) + (R - 2);
                if constexpr (piece < 16) { if constexpr ((piece & 1) == 0) epi_a(piece >> 1); else epi_b(piece >> 1, rsYe); }
            }
        } else if constexpr (NRW == 8) {
            if constexpr (COL == 0 && R >= 2 && R <= 7 && !(DBG & 2)) dma_issue(std::integral_constant<int, R - 2>{}, nb, rsXn, npatch);
            if constexpr (COL == 1 && R >= 2 && R <= 8 && !(DBG & 2)) dma_issue(std::integral_constant<int, R + 4>{}, nb, rsXn, npatch);
            if constexpr (COL >= 2 && COL <= 4 && R >= 2 && R <= 7 && !(DBG & 1)) {
                constexpr int piece = 6 * (COL - 2) + (R - 2);
                if constexpr (piece < 16) { if constexpr ((piece & 1) == 0) epi_a(piece >> 1); else epi_b(piece >> 1, rsYe); }
            }
        } else {
            if constexpr (COL == 0 && R >= 2 && R <= 4 && !(DBG & 2)) dma_issue(std::integral_constant<int, R - 2>{}, nb, rsXn, npatch);
            if constexpr (COL == 1 && R >= 1 && R <= 4 && !(DBG & 2)) dma_issue(std::integral_constant<int, R + 2>{}, nb, rsXn, npatch);
            if constexpr (COL >= 2 && COL <= 4 && R >= 1 && R <= 3 && !(DBG & 1)) {
                constexpr int piece = 3 * (COL - 2) + (R - 1);
                if constexpr (piece < 8) { if constexpr ((piece & 1) == 0) epi_a(piece >> 1); else epi_b(piece >> 1, rsYe); }
            }
        }
        if constexpr (COL == 5 && R == 2) { asm volatile("" : "+s"(st_hb)); advance_blk(); }
        if constexpr (COL == 5 && R == 3) { asm volatile("" : "+s"(st_hb)); nnb = make_blk(more2); }
    };
    auto rstep = [&](auto col_tag, auto r_tag, const char *patch, char *npatch) {
        constexpr int COL = decltype(col_tag)::value, R = decltype(r_tag)::value;
        constexpr int DX = COL >> 1, KS = COL & 1, SLOT = (COL * NPR + R) % 3;
        if constexpr (DBG & 8) { }
        else if constexpr (R + 2 < NPR) frag_read(col_tag, std::integral_constant<int, (R + 2) % NPR>{}, patch);
        else if constexpr (COL < 5) frag_read(std::integral_constant<int, (COL + 1) % 6>{}, std::integral_constant<int, (R + 2) % NPR>{}, patch);
#pragma unroll
        for (int prod = 0; prod < 3; ++prod)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int r = R - dy;
                if (r < 0 || r > NRW - 1) continue;
                const f16x8 a = prod == 2 ? wr[3 * dy + DX][KS][1] : wr[3 * dy + DX][KS][0];
                const f16x8 b = prod == 1 ? fl[SLOT] : fh[SLOT];
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, (COL == 0 && prod == 0 && dy == 0) ? (f32x4)(0.0f) : acc[r], 0, 0, 0);
            }
        extras(col_tag, r_tag, npatch);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        constexpr int NM = 3 * ((R < 2 ? R + 1 : 3) - (R > NRW - 1 ? R - (NRW - 1) : 0));
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // the last patch row of a column and the first of the next each feed ONE accumulator: merged and alternated (conv_stem_direct_h.hip)
    auto edge = [&](auto col_tag, const char *patch) {
        constexpr int COL = decltype(col_tag)::value;
        constexpr int DX = COL >> 1, KS = COL & 1, SLOT = (COL * NPR + NPR - 1) % 3;
        constexpr int DX2 = (COL + 1) >> 1, KS2 = (COL + 1) & 1, SLOT2 = ((COL + 1) * NPR) % 3;
        if constexpr (!(DBG & 8)) frag_read(std::integral_constant<int, COL + 1>{}, std::integral_constant<int, 1>{}, patch);
#pragma unroll
        for (int prod = 0; prod < 3; ++prod) {
            acc[NRW - 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(prod == 2 ? wr[6 + DX][KS][1] : wr[6 + DX][KS][0], prod == 1 ? fl[SLOT] : fh[SLOT], acc[NRW - 1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(prod == 2 ? wr[DX2][KS2][1] : wr[DX2][KS2][0], prod == 1 ? fl[SLOT2] : fh[SLOT2], acc[0], 0, 0, 0);
        }
        if constexpr (!(DBG & 8)) frag_read(std::integral_constant<int, COL + 1>{}, std::integral_constant<int, 2>{}, patch);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto column = [&](auto col_tag, const char *patch, char *npatch) {
        constexpr int COL_ = decltype(col_tag)::value;
        if constexpr (COL_ == 0) rstep(col_tag, std::integral_constant<int, 0>{}, patch, npatch);
        rstep(col_tag, std::integral_constant<int, 1>{}, patch, npatch);
        rstep(col_tag, std::integral_constant<int, 2>{}, patch, npatch); rstep(col_tag, std::integral_constant<int, 3>{}, patch, npatch);
        rstep(col_tag, std::integral_constant<int, 4>{}, patch, npatch);
        if constexpr (NRW == 8) {
            rstep(col_tag, std::integral_constant<int, 5>{}, patch, npatch);
            rstep(col_tag, std::integral_constant<int, 6>{}, patch, npatch); rstep(col_tag, std::integral_constant<int, 7>{}, patch, npatch);
            rstep(col_tag, std::integral_constant<int, 8>{}, patch, npatch);
        }
        if constexpr (COL_ == 5) rstep(col_tag, std::integral_constant<int, NPR - 1>{}, patch, npatch);
        else edge(col_tag, patch);
    };
#define DP_C(T) std::integral_constant<int, T>{}

    // ---- prologue: block 0's patch
    cb = make_blk(-1);
    advance_blk();
    nb = make_blk((1 - n_mine) >> 31);
    eb = cb;
    rsXn = rsrc_of(p.x, cb);
    if constexpr (XF32) {
        stage_load(DP_C(0), cb, rsXn); stage_load(DP_C(1), cb, rsXn); stage_load(DP_C(2), cb, rsXn); stage_load(DP_C(3), cb, rsXn);
        stage_load(DP_C(4), cb, rsXn); stage_load(DP_C(5), cb, rsXn); stage_load(DP_C(6), cb, rsXn); stage_load(DP_C(7), cb, rsXn);
        stage_load(DP_C(8), cb, rsXn); stage_load(DP_C(9), cb, rsXn); stage_load(DP_C(10), cb, rsXn); stage_load(DP_C(11), cb, rsXn);
        stage_load(DP_C(12), cb, rsXn);
        stage_split(DP_C(0), dp_smem); stage_split(DP_C(1), dp_smem); stage_split(DP_C(2), dp_smem); stage_split(DP_C(3), dp_smem);
        stage_split(DP_C(4), dp_smem); stage_split(DP_C(5), dp_smem); stage_split(DP_C(6), dp_smem); stage_split(DP_C(7), dp_smem);
        stage_split(DP_C(8), dp_smem); stage_split(DP_C(9), dp_smem); stage_split(DP_C(10), dp_smem); stage_split(DP_C(11), dp_smem);
        stage_split(DP_C(12), dp_smem);
    } else {
    dma_issue(DP_C(0), cb, rsXn, dp_smem); dma_issue(DP_C(1), cb, rsXn, dp_smem); dma_issue(DP_C(2), cb, rsXn, dp_smem);
    dma_issue(DP_C(3), cb, rsXn, dp_smem); dma_issue(DP_C(4), cb, rsXn, dp_smem); dma_issue(DP_C(5), cb, rsXn, dp_smem);
    dma_issue(DP_C(6), cb, rsXn, dp_smem);
    if constexpr (NDMA == 13) {
        dma_issue(DP_C(7), cb, rsXn, dp_smem); dma_issue(DP_C(8), cb, rsXn, dp_smem);
        dma_issue(DP_C(9), cb, rsXn, dp_smem); dma_issue(DP_C(10), cb, rsXn, dp_smem); dma_issue(DP_C(11), cb, rsXn, dp_smem);
        dma_issue(DP_C(12), cb, rsXn, dp_smem);
    }
    }
#pragma unroll
    for (int r = 0; r < NRW; ++r) { rres[r] = (u32x4)(0u); rnew[r] = (u32x4)(0u); eacc[r] = (f32x4)(0.0f); acc[r] = (f32x4)(0.0f); }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_s_barrier();
    int bi = 0;
    do {
        const int cur = bi & 1;
        more2 = (bi + 2 - n_mine) >> 31;
        const char *const patch = dp_smem + cur * PATCHB;
        char *const npatch = dp_smem + (cur ^ 1) * PATCHB;
        frag_read(DP_C(0), DP_C(0), patch);
        frag_read(DP_C(0), DP_C(1), patch);
        column(DP_C(0), patch, npatch); column(DP_C(1), patch, npatch); column(DP_C(2), patch, npatch);
        column(DP_C(3), patch, npatch); column(DP_C(4), patch, npatch); column(DP_C(5), patch, npatch);
        // everything vector-memory of this block -- the next patch, the shortcut rows, the previous block's stores -- is through
        __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
        for (int r = 0; r < NRW; ++r) { eacc[r] = acc[r]; if (RES) rres[r] = rnew[r]; }
        eo = co;
        eb = cb;
        __builtin_amdgcn_s_barrier();
        cb = nb;
        nb = nnb;
    } while (++bi < n_mine);
    rsYe = rsrc_of(p.y, eb);
#pragma unroll
    for (int r = 0; r < NRW; ++r) { epi_a(r); epi_b(r, rsYe); }    // the last block's

    if (p.amax_out) {
        __syncthreads();
        unsigned *s_amax = (unsigned *)dp_smem;
        if (tid == 0) *s_amax = 0u;
        __syncthreads();
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) my_amax = fmaxf(my_amax, __shfl_xor(my_amax, o, 64));
        if (lane == 0) atomicMax(s_amax, __float_as_uint(my_amax));
        __syncthreads();
        if (tid == 0 && *s_amax > *(volatile unsigned *)p.amax_out) atomicMax(p.amax_out, *s_amax);
    }
}

/* y = act(conv3x3(x, w) + bias (+ res)), stride 1, pad 1, Cin = Cout = 64, x a PAIR-FORMAT map [B,H,W,64] (conv_igemm.hip: 128 bytes
 * [hi 32 | lo 32] fp16 per pixel and 32-channel block, scaled by the power of two of its bound slot d_xbound) or (x_pairs = 0, no
 * shortcut) a float32 NHWC map scaled by the power of two of *d_amax_in and split while its patch is staged.  d_w2r / inv_sw =
 * `stem_direct_pair_weights(weight)`: [4 output-channel quarters][9 taps][2 K steps][hi | lo][64 lanes][8] halfs of s_w w.  d_res
 * (optional shortcut, y's shape): pair format with its bound slot (res_pairs) or float32 NHWC with d_res_bound = a bound of max |res|.
 * out_pairs: y leaves in pair format scaled for the bound max|x| wl1 + bmax (+ *d_res_bound), which goes to d_bound_out (d_amax_in =
 * the measured max |x|; wl1 = max_co sum |w[co]|, bmax = max |bias|); else y is float32 NHWC.  d_amax_out (optional, zeroed):
 * receives max |y|.  Same contract as cslam_conv_igemm_h2p_dev with x_pairs = 1, KH = KW = 3, stride = pad = 1. */
CSLAM_API int cslam_conv3x3_direct_p_dev(const void *d_x, int x_pairs, const unsigned *d_xbound, const void *d_w2r, const float *d_bias,
                                         const void *d_res, int res_pairs, const unsigned *d_res_bound, int B, int H, int W, int Cin,
                                         int Cout, int relu, const unsigned *d_amax_in, float inv_sw, float wl1, float bmax,
                                         unsigned *d_amax_out, int out_pairs, unsigned *d_bound_out, void *d_y, void *stream) {
    PTR_DEVICE(d_x);
    ARG_CHECK(d_x && (d_xbound || !x_pairs) && d_w2r && d_y && d_amax_in, "NULL argument");
    ARG_CHECK(x_pairs || (!d_res && H < 16384), "a float32 input map goes without a shortcut");
    ARG_CHECK(B >= 1 && H >= 1 && W >= 1, "empty map");
    ARG_CHECK(Cin == 64 && Cout == 64, "Cin and Cout must be 64");
    ARG_CHECK(inv_sw > 0.0f, "inv_sw must be positive");
    ARG_CHECK(!out_pairs || (d_bound_out && wl1 >= 0.0f && bmax >= 0.0f), "pair-format output needs wl1, bmax and the bound slot");
    ARG_CHECK(!d_res || d_res_bound, "the shortcut's bound slot is missing");
    ARG_CHECK((int64_t)H * W * 512 + (int64_t)11 * W * 256 < 0x7ffffff0ll, "two images' maps must stay below 2 GiB (32-bit buffer offsets)");
    ConvDirectPArgs a;
    a.x = (const char *)d_x; a.xbound = d_xbound; a.amax_in = d_amax_in;
    a.w2 = (const f16x8 *)d_w2r; a.bias = d_bias; a.inv_sw = inv_sw;
    a.res = (const char *)d_res; a.res_bound = d_res_bound;
    a.y = (char *)d_y; a.wl1 = wl1; a.bmax = bmax; a.bound_out = d_bound_out; a.amax_out = d_amax_out;
    a.B = B; a.H = H; a.W = W;
    a.gxh = (int)ceil_div64(W, 8); a.gyb = (int)ceil_div64(H, 8);
    const int64_t nhalf = (int64_t)B * a.gxh * a.gyb;
    ARG_CHECK(nhalf < (1ll << 30), "too many blocks for one launch");
    a.nhalf = (int)nhalf; a.nblk = (int)((nhalf + 1) / 2);
    a.relu = relu;
    const int n_cu = cslam_cu_count();
    ARG_CHECK(n_cu > 0, "no HIP device");
    const int grid = a.nblk < n_cu ? a.nblk : n_cu;
    hipStream_t st = (hipStream_t)stream;
    // four waves, one per SIMD.  The eight-wave form (NRW = 4, two per SIMD: measurement build, CSLAM_DP_NRW=4) does not fit its 128 + 128
    // registers (84 - 236 bytes of scratch: 1.05 ms), and without the epilogue, where it does fit, it is no faster (0.475 against 0.479 ms:
    // profiles/r06_c_dp_ablations.log) -- what separates this kernel from its matrix-only time is not issue slots a second wave could fill
    int nrw = 8;
#ifdef CSLAM_ABLATIONS
    if (const char *e = getenv("CSLAM_DP_NRW")) nrw = atoi(e) == 4 ? 4 : 8;
#define DP_NRW4(X) X
#else
#define DP_NRW4(X) do { } while (0)
#endif
#define DP_LAUNCH_N(R_, O_, N_) do { \
        static DeviceOnce once; int once_dev; \
        if (once.todo(&once_dev)) { \
            HIP_TRY(hipFuncSetAttribute((const void *)conv3x3_direct_p_kernel<R_, O_, N_>, hipFuncAttributeMaxDynamicSharedMemorySize, DP_LDS(32 / N_))); \
            once.done(once_dev); } \
        hipLaunchKernelGGL((conv3x3_direct_p_kernel<R_, O_, N_>), dim3(grid), dim3(2048 / N_), DP_LDS(32 / N_), st, a); } while (0)
#define DP_LAUNCH(R_, O_) do { if (nrw == 8) DP_LAUNCH_N(R_, O_, 8); else DP_NRW4(DP_LAUNCH_N(R_, O_, 4)); } while (0)
    const int rm = !d_res ? 0 : (res_pairs ? 2 : 1);
#ifdef CSLAM_ABLATIONS
    if (const char *e = getenv("CSLAM_DP_DBG")) {              // timing-only ablations (wrong results): measurement build, pair-format output
        const int d = atoi(e);
#define DP_LAUNCH_D2(R_, N_, D) do { HIP_TRY(hipFuncSetAttribute((const void *)conv3x3_direct_p_kernel<R_, true, N_, D>, hipFuncAttributeMaxDynamicSharedMemorySize, DP_LDS(32 / N_))); \
            hipLaunchKernelGGL((conv3x3_direct_p_kernel<R_, true, N_, D>), dim3(grid), dim3(2048 / N_), DP_LDS(32 / N_), st, a); } while (0)
#define DP_LAUNCH_D(D) do { if (rm == 2) { if (nrw == 8) DP_LAUNCH_D2(2, 8, D); else DP_LAUNCH_D2(2, 4, D); } \
                            else { if (nrw == 8) DP_LAUNCH_D2(0, 8, D); else DP_LAUNCH_D2(0, 4, D); } \
        HIP_TRY(hipGetLastError()); return CSLAM_OK; } while (0)
        if (d == 1) DP_LAUNCH_D(1);
        if (d == 2) DP_LAUNCH_D(2);
        if (d == 4) DP_LAUNCH_D(4);
        if (d == 7) DP_LAUNCH_D(7);
        if (d == 16) DP_LAUNCH_D(16);
        if (d == 23) DP_LAUNCH_D(23);
#undef DP_LAUNCH_D
#undef DP_LAUNCH_D2
    }
#endif
    if (!x_pairs) {
#define DP_LAUNCH_X(O_) do { \
        static DeviceOnce once; int once_dev; \
        if (once.todo(&once_dev)) { \
            HIP_TRY(hipFuncSetAttribute((const void *)conv3x3_direct_p_kernel<0, O_, 8, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DP_LDS(4))); \
            once.done(once_dev); } \
        hipLaunchKernelGGL((conv3x3_direct_p_kernel<0, O_, 8, 0, true>), dim3(grid), dim3(256), DP_LDS(4), st, a); } while (0)
        if (out_pairs) DP_LAUNCH_X(true); else DP_LAUNCH_X(false);
#undef DP_LAUNCH_X
        HIP_TRY(hipGetLastError());
        return CSLAM_OK;
    }
    if (out_pairs) { if (rm == 0) DP_LAUNCH(0, true); else if (rm == 1) DP_LAUNCH(1, true); else DP_LAUNCH(2, true); }
    else { if (rm == 0) DP_LAUNCH(0, false); else if (rm == 1) DP_LAUNCH(1, false); else DP_LAUNCH(2, false); }
#undef DP_LAUNCH
#undef DP_LAUNCH_N
#undef DP_NRW4
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}
