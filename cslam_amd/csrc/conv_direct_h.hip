// conv_direct_h.hip -- 3x3 / stride 1 / pad 1 convolution as ONE kernel on the fp16 matrix pipe with fp32-grade results (gfx950):
// the direct (non-Winograd) form for the layers whose F(4x4) pipeline is bound by what it MOVES, VGG-16 conv2_1 (64 -> 128
// channels on 112 x 112 maps) and conv2_2 (128 -> 128, + MaxPool2d): cslam/vpr/netvlad.py:163-171,227.
//
// Why a direct form beside a Winograd trunk (round 4).  conv2_2 through transform / 36 products / transform moves 17 GB per 256
// frames (x in, V out, V in, Z out, Z in, y out: 10 x the activation) and takes 2.8 ms at 0.75-0.9 of the copy rate; conv2_1 as
// the one-kernel F(4x4) convolution streams 1.18 MB of weight fragments per 16 tiles through the compute unit's 64 B/clk vector
// path (matrix pipe busy 0.15, 80 bytes of scratch per lane).  The direct form does 4 x the multiplications of F(4x4) -- but as
// an implicit GEMM whose operands barely move: per 16 x 16-pixel block the input patch is read ONCE from HBM / L2 (x 1.27 halo)
// into LDS and the 3 x 3 taps are nine SHIFTED READS of that one patch (an immediate offset per tap), the weights of a (tap,
// 32-channel slab) are 16 KB for 1536 matrix-pipe cycles, the output leaves once.  HBM: activation in + out, nothing else.
//
// Arithmetic: both operands as exact fp16 pairs, three products per fp32-grade product (v_mfma_f32_32x32x16_f16, fp32
// accumulate), as wino_gemm.hip / sim_topk_pair.hip:
//     x (float32, NHWC) is scaled by the power of two s_x from the previous layer's max |x| slot and split into hi + lo WHEN THE
//     PATCH IS STAGED (registers -> LDS; nothing in HBM changes format); w is split offline (vpr/winograd.py
//     `direct_pair_weights`: [9 taps][Cout][Cin/32][hi 32 | lo 32], s_w a power of two); acc = xh wh + xl wh + xh wl, the result
//     is acc / (s_x s_w) (exact), + bias, ReLU, 2 x 2 max, max |y| for the next layer's scale.
//
// Work decomposition: persistent workgroups of 8 waves walk 16 x 16-pixel output blocks (all 128 output channels of a block:
// 256 x 128 accumulators = 64 registers per lane).  Waves 4 (pixels) x 2 (channels): a wave owns 4 rows x 16 pixels x 64
// channels = 2 x 2 MFMA tiles; an MFMA tile is 2 output rows x 16 pixels -- one row of 2 x 2 pooling windows, the four pixels of
// a window in lanes l, l ^ 1, l ^ 16, l ^ 17 (two shuffles per value for the fused MaxPool2d).
// LDS: the 18 x 18-pixel patch of one 32-channel slab as [hi 32 | lo 32] halfs, pixel pitch 144 B and row pitch 2816 B (every
// ds_read_b128 lane group -- 2 x 8 pixels of two rows -- then falls on 16 distinct 16-byte slots for every tap), double
// buffered (2 x 50.7 KB); weight stages of 128 channels x 128 B, XOR-swizzled, LDS-DMA, ring of two (2 x 16 KB).
// K loop: stage = (slab, tap); the loop is flattened over the blocks of a workgroup, so the patch of the next slab / block is
// loaded (float32 -> registers) during the first taps of the current slab and split into the idle patch buffer during the later ones.
#include <stdlib.h>
#include <type_traits>
#include <hip/hip_fp16.h>
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CD_PIXB 144                    // bytes per patch pixel: 64 hi + 64 lo + 16 pad
#define CD_ROWB 2816                   // bytes per patch row (18 pixels = 2592, padded to a multiple of 256)
#define CD_PATCHB (18 * CD_ROWB)       // 50 688
#define CD_WROWB 128                   // bytes per weight row of a stage: [hi 32 | lo 32]
#define CD_WSTAGE (128 * CD_WROWB)     // 16 KB: 128 output channels
#define CD_NW 3                        // weight stages in the ring: requests run two stages ahead
#define CD_LDS (2 * CD_PATCHB + CD_NW * CD_WSTAGE + 512 + 256)  // + the 128 bias values + a sink for the patch elements that do not exist

struct ConvDirectArgs {
    const float *x; const char *w2; const float *bias; float *y;
    int B, H, W, Cin, nslab;           // nslab = Cin / 32
    int gxb, gyb, nblk;                // blocks per row / column of a frame, blocks in all
    int relu, pool;
    const unsigned *amax_in; float inv_sw; unsigned *amax_out;
    int stagger_cycles;                // start-up offset per phase (workgroup >> 3 & 3), 0 = none
};

// LDS-DMA in the MUBUF encoding (`buffer_load_dwordx4 ... lds`; sim_topk_pair.hip has the why: hipcc counts LDS reads again -- behind
// a pending `global_load_lds` every lgkmcnt(N) became lgkmcnt(0) --, the stage's offset travels in an SGPR, the lane's in one register)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cd_rsrc(const char *base, int64_t bytes) {
    const uint64_t a = (uint64_t)base;
    const uint64_t u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)a);
    const int n = __builtin_amdgcn_readfirstlane((int)(bytes > 0x7fffffff ? 0x7fffffff : bytes));
    return __builtin_amdgcn_make_buffer_rsrc((void *)u, 0, n, 0x00020000);
}
__device__ __forceinline__ void cd_blds16(__amdgpu_buffer_rsrc_t rs, int voff, int soff, char *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)lds_wave_base, 16, voff, soff, 0, 0);
}

// DBG (builds with -DCSLAM_ABLATIONS only; WRONG results): 1 = no weight requests after the prologue, 2 = no patch staging after the
// prologue, 3 = both; + 4 = no stage barrier / wait; + 8 = fragments read once per workgroup only; 32 = patch loaded but not split into
// LDS; 64 = patch split into LDS but not loaded
#ifdef CSLAM_ABLATIONS
__device__ unsigned long long *cd_prof = nullptr;             // measurement build: [stage loops, epilogues, blocks] ticks of wave 0 / workgroup 0
extern "C" __attribute__((visibility("default"))) int cslam_debug_cd_prof_dev(void *d_buf) {
    unsigned long long *q = (unsigned long long *)d_buf;
    return hipMemcpyToSymbol(HIP_SYMBOL(cd_prof), &q, sizeof(q)) == hipSuccess ? 0 : -2;
}
#define CD_PROF 1
#else
#define CD_PROF 0
#endif
// max over lanes l, l ^ 1, l ^ 16, l ^ 17 (the four pixels of a 2 x 2 pooling window) without going through LDS: a DPP quad
// permutation and gfx950's v_permlane16_swap (of two copies of v one ends up holding the even 16-lane rows twice, the other the odd
// rows); as `__shfl_xor` each step was a ds_bpermute_b32 with an LDS round trip behind it (conv_stem_direct_h.hip has the numbers)
__device__ __forceinline__ float cd_pool4(float v) {
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1, 0, 3, 2]
    asm("v_max_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(a));   // (fmaxf would first canonicalise both operands: two more instructions per maximum)
    // (as inline assembly: hipcc 7.2 folds the two results of __builtin_amdgcn_permlane16_swap(v, v) into ONE value and drops the maximum
    // that follows; the s_nop covers the VALU-write -> permlane-read hazard the compiler cannot see in here)
    float b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v), "+v"(b));
    asm("v_max_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(b));
    return v;
}

// v - (float)half HI of the packed pair h: one v_fma_mix_f32, the fp16 operand read in place (conv_stem_direct_h.hip's sd_sub_half)
template <int HI>
__device__ __forceinline__ float cd_sub_half(float v, __half2 h) {
    float d;
    const unsigned hb = *(const unsigned *)&h;
    if (HI) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hb), "v"(v));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hb), "v"(v));
    return d;
}

// XP: x is a PAIR-FORMAT map (conv_igemm.hip: [pixel][32-channel block][hi 32 | lo 32] fp16 of s x; amax_in = its BOUND slot, which fixes
// s).  A pixel's slab is the same 128 bytes at the same address as in a float32 map, so the loads do not change; an element is one of
// its eight 16-byte chunks and goes to LDS as it is -- the split (two packed conversions, four v_fma_mix, two more conversions per
// element: 0.34 of conv2_2's 2.5 ms) is gone.
template <bool RELU, bool POOL, int DBG = 0, bool PIXA = false, bool XP = false>
__global__ __launch_bounds__(512, 2) void conv3x3_direct_h_kernel(ConvDirectArgs p) {
    extern __shared__ __attribute__((aligned(16))) char cd_smem[];
    char *s_patch = cd_smem;                                   // [2][CD_PATCHB]
    char *s_w = cd_smem + 2 * CD_PATCHB;                       // [CD_NW][CD_WSTAGE]
    // the bias in LDS: read from global memory in the epilogue, every load behind the block's first stores made the compiler wait
    // for ALL of them (vmcnt(0): loads and stores share the counter and return out of order) -- 16 store round trips per block
    float *s_bias = (float *)(cd_smem + 2 * CD_PATCHB + CD_NW * CD_WSTAGE);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                   // 4 x 2 waves: rows 4 wm .. 4 wm + 3, channels 64 wn .. + 63
    const int h = lane >> 5, l31 = lane & 31;
    if (tid < 128) s_bias[tid] = p.bias ? p.bias[tid] : 0.0f;   // visible behind the prologue's barrier

    // power-of-two input scale: max |x| s_x <= 2^15 - 16 (fp16 holds 65504; the products stay far inside fp32)
    const float amax = fminf(fmaxf(__uint_as_float(*p.amax_in), 1e-30f), 1e30f);
    int e_;
    (void)frexpf(XP ? amax : 32752.0f / amax, &e_);
    const float sx = XP ? ldexpf(1.0f, 14 - e_) : ldexpf(1.0f, e_ - 1);        // XP: the producer's scale (conv_igemm.hip::ci_scale of the bound)
    const float inv = p.inv_sw / sx;

    // Blocks to workgroups, XCD-aware (as conv_stem_direct_h.hip): workgroup w runs on XCD w % 8, every XCD has its own L2; each XCD
    // takes one contiguous eighth of the blocks and its workgroups walk it side by side, so that the halo a block shares with its
    // neighbours is read through ONE L2 at about the same time (block b -> workgroup b % grid: the halo came through two or three)
    const bool by_xcd = (gridDim.x & 7) == 0;
    const int wg_xcd = by_xcd ? (int)blockIdx.x & 7 : 0, wg_j = by_xcd ? (int)blockIdx.x >> 3 : (int)blockIdx.x;
    const int wg_per = by_xcd ? (int)gridDim.x >> 3 : (int)gridDim.x;
    const int per_xcd = by_xcd ? (p.nblk + 7) >> 3 : p.nblk;
    const int blk_beg = wg_xcd * per_xcd;
    const int blk_cnt = min(per_xcd, p.nblk - blk_beg);
    const int n_mine = blk_cnt > wg_j ? (blk_cnt - wg_j + wg_per - 1) / wg_per : 0;
    if (n_mine <= 0) return;
    const int nslabs_total = n_mine * p.nslab;

    // ---- patch staging: element e = i * 512 + tid = (pixel e >> 3, float4 e & 7 of the 32-channel slab); 324 pixels x 8 = 2592
    constexpr int NPL = 6;                                     // ceil(2592 / 512)
    // (the per-element geometry is recomputed from the thread id where it is used -- four places per slab, a handful of integer
    // instructions each -- behind an opaque copy of the id: kept in registers across the whole K loop, hoisted there by the
    // compiler, its 12 values pushed the kernel over 256 registers and the weight pointers into scratch)
    auto elem = [&](int i, int &dst, int &pr, int &pc) {
        int t = tid;
        asm volatile("" : "+v"(t));
        const int e = i * 512 + t;
        const int px = e >> 3, f4 = e & 7;
        pr = (px * 3641) >> 16;                                // px / 18 for px < 4096
        pc = px - pr * 18;
        dst = e < 2592 ? pr * CD_ROWB + pc * CD_PIXB + f4 * (XP ? 16 : 8) : -1;
    };
    float4 stg[NPL / 2];                                       // the patch goes through the registers in two halves (12 registers, not 24)
    unsigned inmask = 0;                                       // bit half * 3 + j: element j of the half lies inside the image
    // what a patch LOAD needs of the geometry stays in seven registers: the element's byte offset from the patch's first pixel and its
    // column (5 bits each), so that a load is one add and sits at the very top of its stage -- the stage waits give a request until
    // the end of the NEXT stage to land, and these come from HBM
    int inv_off[NPL];
    unsigned pcpack = 0, validpack = 0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int e = i * 512 + tid, px = e >> 3;
        const int pr = (px * 3641) >> 16, pc = px - pr * 18;
        const bool valid = e < 2592;
        inv_off[i] = valid ? (pr * p.W + pc) * p.Cin * 4 + (e & 7) * 16 : 0;
        pcpack |= (unsigned)(valid ? pc : 0) << (5 * i);
        validpack |= (valid ? 1u : 0u) << i;
    }
    char *const s_sink = cd_smem + 2 * CD_PATCHB + CD_NW * CD_WSTAGE + 512 + (tid & 15) * (XP ? 16 : 8);
    // block coordinates are decoded ONCE per block on the scalar unit (three run-time divisions: ~90 dependent scalar instructions,
    // which sat at the top of two stages per slab): `cb` = the block being multiplied, `nb` = the next one (the last block again
    // behind the workgroup's last)
    struct Blk { int by, bx; const char *xb; };               // xb = the block's image
    auto decode_blk = [&](int bi) {
        const int blk = blk_beg + wg_j + bi * wg_per;
        const int per_img = p.gxb * p.gyb;
        const int img = blk / per_img, rem = blk - img * per_img;
        Blk b;
        b.by = rem / p.gxb; b.bx = rem - b.by * p.gxb;
        b.xb = (const char *)p.x + (int64_t)img * p.H * p.W * p.Cin * 4;
        return b;
    };
    Blk cb = decode_blk(0), nb = decode_blk(n_mine > 1 ? 1 : 0);
    const int img_bytes = p.H * p.W * p.Cin * 4;
    const int lane_ch = (tid & 7) * 16;                        // the lane's float4 inside a pixel's 32-channel slab
    // float32 patch of (block b, slab sl), half `half`, into registers: buffer loads (the lane's offset is one register, the slab's
    // an SGPR)
    auto patch_load = [&](const Blk &b, int sl, int half) {
        // the image is the buffer: rows above and below it are out of its range and read as zero; the columns left and right of it
        // would read the neighbouring row's pixels and are masked when the registers are split into LDS (a select right here would
        // wait for the load).  EVERY lane of EVERY wave issues every load: the stage barriers count the outstanding requests
        // (vmcnt(NV)), and a load under `if (inside)` is skipped by waves whose lanes are all outside (s_cbranch_execz) -- such a
        // wave then waited for fewer of its OLDER requests than it had to, and with the weights slow to arrive (another stream
        // thrashing the L2) multiplied a ring slot that had not landed (round 4's first form)
        const __amdgpu_buffer_rsrc_t rsX = cd_rsrc(b.xb, img_bytes);
        const int blk_off = ((b.by * 16 - 1) * p.W + (b.bx * 16 - 1)) * p.Cin * 4;
        const int gx0 = b.bx * 16 - 1;
        unsigned m = inmask & ~(7u << (3 * half));
#pragma unroll
        for (int j = 0; j < NPL / 2; ++j) {
            const int i = half * (NPL / 2) + j;
            const int gx = gx0 + (int)((pcpack >> (5 * i)) & 31u);
            const bool in = ((validpack >> i) & 1u) & (gx >= 0) & (gx < p.W);
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsX, inv_off[i] + blk_off, sl * 128, 0);
            stg[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            m |= (in ? 1u : 0u) << (3 * half + j);
        }
        inmask = m;
    };
    // registers -> exact fp16 pairs of s_x x -> LDS, elements [j0, j1) of the half.  Branch-free (an element that does not exist --
    // the last 480 of 6 x 512 -- goes to a sink): the stage must stay ONE basic block for its instruction order to be set
    auto patch_store = [&](int buf, int half, int j0, int j1) {
        char *dst = s_patch + buf * CD_PATCHB;
#pragma unroll
        for (int j = 0; j < NPL / 2; ++j) {
            if (j < j0 || j >= j1) continue;
            int pd, pr, pc;
            elem(half * (NPL / 2) + j, pd, pr, pc);
            const bool in = (inmask >> (3 * half + j)) & 1u;
            if (XP) {
                char *dx = pd >= 0 ? dst + pd : s_sink;
                *(uint4 *)dx = in ? make_uint4(__float_as_uint(stg[j].x), __float_as_uint(stg[j].y), __float_as_uint(stg[j].z), __float_as_uint(stg[j].w))
                                  : make_uint4(0u, 0u, 0u, 0u);
                continue;
            }
            const float w0 = in ? stg[j].x * sx : 0.0f, w1 = in ? stg[j].y * sx : 0.0f, w2 = in ? stg[j].z * sx : 0.0f, w3 = in ? stg[j].w * sx : 0.0f;
            const __half2 h01 = __floats2half2_rn(w0, w1), h23 = __floats2half2_rn(w2, w3);
            // w - (float)hi straight out of the packed register (v_fma_mix_f32; through __half22float2 hipcc rounds every value a second
            // time with a scalar v_cvt_f16_f32, converts that back and subtracts: three instructions per value instead of one)
            const __half2 l01 = __floats2half2_rn(cd_sub_half<0>(w0, h01), cd_sub_half<1>(w1, h01));
            const __half2 l23 = __floats2half2_rn(cd_sub_half<0>(w2, h23), cd_sub_half<1>(w3, h23));
            char *d = pd >= 0 ? dst + pd : s_sink;
            *(uint2 *)d = make_uint2(*(const unsigned *)&h01, *(const unsigned *)&h23);
            *(uint2 *)(d + 64) = make_uint2(*(const unsigned *)&l01, *(const unsigned *)&l23);
        }
    };

    // ---- weight loader: chunk pch = i * 512 + tid -> row pch >> 3 (output channel), physical 16-byte slot pch & 7 holding
    // logical chunk slot ^ ((row >> 1) & 7) of the row's 128-byte block (wino_gemm.hip's image: conflict-free ds_read_b128)
    const int tap_stride = 128 * p.nslab * CD_WROWB;
    const __amdgpu_buffer_rsrc_t rsW = cd_rsrc(p.w2, (int64_t)9 * tap_stride);
    int voffW[2];                                              // the lane's two chunks inside a (tap, slab)'s 128 rows
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pch = i * 512 + tid;
        const int r = pch >> 3, c = (pch & 7) ^ ((r >> 1) & 7);
        voffW[i] = r * p.nslab * CD_WROWB + (c << 4);
    }
    int woff = 0;                                              // running scalar offset: the weights of the stage after the current one

    // ---- fragment addressing
    // A (patch): lane (l31, h) = pixel (row l31 >> 4, column l31 & 15) of an MFMA tile of 2 rows x 16 pixels; tile t of the wave =
    // rows 4 wm + 2 t, + 1; K chunk 2 s + h (hi) | 4 + 2 s + h (lo) for K step s.  Everything else is an immediate offset.
    const int a_base = (4 * wm + (l31 >> 4)) * CD_ROWB + (l31 & 15) * CD_PIXB + h * 16;
    // B (weights): row = output channel 64 wn + 32 n + l31, swizzled chunk
    const int swz = (lane >> 1) & 7;
    int foff[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int lo = 0; lo < 2; ++lo) foff[s][lo] = ((4 * lo + 2 * s + h) ^ swz) << 4;
    const int b_base = (wn * 64 + l31) * CD_WROWB;

    // acc[t][n]: MFMA tile (pixel tile t, channel tile n) with the WEIGHTS as the A operand: D[m = channel][col = pixel], so a lane
    // holds, for ONE pixel (column l31), the 16 channels (r & 3) + 8 (r >> 2) + 4 h of the tile -- four runs of four consecutive
    // channels: 16-byte stores (with the pixels as A a lane held 16 pixels of one channel: sixteen 4-byte stores per tile)
    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][n][r] = 0.0f;
    float my_amax = 0.0f;
    const int Ho = POOL ? p.H >> 1 : p.H, Wo = POOL ? p.W >> 1 : p.W;

    // fragments: two register sets (K step s of a stage in set s): a K step is multiplied while the next one is read, and the
    // stage's second K step is multiplied BEHIND the stage barrier, under the first reads of the next stage (the rotation of
    // sim_topk_pair.hip, which also says why the reads sit behind the first MFMAs of a region: hipcc cannot count LDS reads
    // while an LDS-DMA is pending)
    f16x8 fa[2][2][2], fb[2][2][2];                            // [set][hi | lo][tile]
    auto read_frags = [&](int u, const char *sA, const char *sB, int s) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            fa[u][0][t] = *(const f16x8 *)(sA + t * 2 * CD_ROWB + s * 32);
            fa[u][1][t] = *(const f16x8 *)(sA + t * 2 * CD_ROWB + 64 + s * 32);
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            fb[u][0][n] = *(const f16x8 *)(sB + n * 32 * CD_WROWB + foff[s][0]);
            fb[u][1][n] = *(const f16x8 *)(sB + n * 32 * CD_WROWB + foff[s][1]);
        }
    };
    auto multiply = [&](int u) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[t][n] = PIXA ? __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[u][0][t], fb[u][0][n], acc[t][n], 0, 0, 0)
                                                         : __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[u][0][n], fa[u][0][t], acc[t][n], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[t][n] = PIXA ? __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[u][1][t], fb[u][0][n], acc[t][n], 0, 0, 0)
                                                         : __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[u][0][n], fa[u][1][t], acc[t][n], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[t][n] = PIXA ? __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[u][0][t], fb[u][1][n], acc[t][n], 0, 0, 0)
                                                         : __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[u][1][n], fa[u][0][t], acc[t][n], 0, 0, 0);
    };

    // MFMAs 3 k .. 3 k + 2 of multiply(u)'s twelve, in its order (the sums do not change)
    auto multiply3 = [&](int u, int k) {
#pragma unroll
        for (int idx = 3 * k; idx < 3 * k + 3; ++idx) {
            const int prod = idx >> 2, t = (idx >> 1) & 1, n = idx & 1;
            const f16x8 a = prod == 2 ? fb[u][1][n] : fb[u][0][n];
            const f16x8 b = prod == 1 ? fa[u][1][t] : fa[u][0][t];
            acc[t][n] = PIXA ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[t][n], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t][n], 0, 0, 0);
        }
    };

    // ---- prologue: patch of slab 0, weights of stage 0
#pragma unroll
    for (int b = 0; b < 2; ++b) {                              // stages 0 and 1: (tap 0 | 1, slab 0); the offset moves on to (tap 2, slab 0)
#pragma unroll
        for (int i = 0; i < 2; ++i) cd_blds16(rsW, voffW[i], woff, s_w + b * CD_WSTAGE + wave * 1024 + i * (512 * 16));
        woff += tap_stride;
    }
    patch_load(cb, 0, 0); patch_store(0, 0, 0, NPL / 2);
    patch_load(cb, 0, 1); patch_store(0, 1, 0, NPL / 2);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();

    int g = 0, gslab = 0, wb = 0;                              // global stage and slab counters, ring slot of the current stage's weights
    // One stage = one tap of one slab.  TAP is a compile-time constant (the nine taps of a slab are unrolled): the tap's shift of the
    // patch is an immediate offset of the fragment reads and nothing per stage is left to the scalar unit but the buffer parities --
    // with `tap = st % 9`, `dy = tap / 3` at run time every stage opened with ~40 dependent scalar instructions that BOTH waves of a
    // SIMD executed at the same time, right behind the barrier, with the matrix pipe idle.
    auto stage_body = [&](auto tap_tag, auto first_tag, int sl, int sl_next) {
        constexpr int TAP = decltype(tap_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;     // first stage of a block: nothing is carried in
        constexpr int DY = TAP / 3, DX = TAP - 3 * DY;
        // next slab's patch in two halves: loaded (float32 -> registers) at taps 0 / 4, split into the idle buffer at taps 3 / 7
        // (branch-free: behind the workgroup's very last slab its own patch is staged once more, into the idle buffer).  Their ~200 /
        // ~120 VALU instructions per wave are INTERLEAVED with the stage's MFMAs (sched_group_barrier pipelines below): at the top of
        // the stage, where program order puts them, both waves of a SIMD executed them at the same time right behind the barrier with
        // the matrix pipe idle -- 20 % of the kernel (profiles/r04_v24_direct_conv_ablations.log: 2.84 ms, without patch staging 2.25).
        // Schedule over a slab's nine taps: 0 load half 0 | 2, 3 split its three elements into LDS | 4 load half 1 | 6, 7, 8 split
        // (a load has landed two stages later: the stage waits leave only the newest requests in flight).
        constexpr bool PL = (TAP == 0 || TAP == 4) && !(DBG & 2) && !(DBG & 64);
        constexpr int HALF = TAP >= 4 ? 1 : 0;
        constexpr int ST1 = (DBG & (2 | 32)) ? -1 : TAP == 2 ? 0 : TAP == 3 ? 1 : TAP == 6 ? 0 : TAP == 7 ? 1 : TAP == 8 ? 2 : -1;   // element split under K step 1
        constexpr int ST2 = (DBG & (2 | 32)) ? -1 : TAP == 3 ? 2 : -1;                                                         // ... under K step 0
        constexpr bool PS = ST1 >= 0;
        if ((DBG & 32) && TAP == 8) { asm volatile("" :: "v"(stg[0].x), "v"(stg[1].y), "v"(stg[2].z), "v"(stg[0].w), "v"(stg[1].x), "v"(stg[2].y)); }
        // the slab whose patch is staged during this one: the next slab of this block, or slab 0 of the next block
        const bool same_blk = sl + 1 < p.nslab;
        Blk tb;
        tb.by = same_blk ? cb.by : nb.by; tb.bx = same_blk ? cb.bx : nb.bx; tb.xb = same_blk ? cb.xb : nb.xb;
        const char *sA = s_patch + (gslab & 1) * CD_PATCHB + a_base + DY * CD_ROWB + DX * CD_PIXB;
        const char *sB = s_w + wb * CD_WSTAGE + b_base;
        auto request_weights = [&]() {
            // the weights of stage g + 2 into the buffer stage g - 1 was read from (every wave finished that before the last barrier)
            // (the very last stage of the workgroup requests the weights of a stage nobody will run: harmless)
            const int wb2 = wb == 0 ? 2 : wb - 1;
            char *d = s_w + wb2 * CD_WSTAGE + wave * 1024;
#pragma unroll
            for (int i = 0; i < 2; ++i) cd_blds16(rsW, voffW[i], woff, d + i * (512 * 16));
            // a RUNNING offset (scalar): written as (TAP + 1) * tap_stride + ... the nine taps' addresses are loop invariants that the
            // compiler keeps in registers across the K loop (as 64-bit per-lane pointers, round 4's first form: 36 VGPRs, the kernel
            // spilled).  Behind tap 6's request -- tap 8 of this slab -- comes tap 0 of the next slab: a relative step.
            woff += TAP == 6 ? (sl_next - sl) * CD_WROWB - 8 * tap_stride : tap_stride;
        };
        // ---- behind the barrier: this stage's first reads, the next stage's weights, the previous stage's second K step
        if (!(DBG & 8) || g == 0) read_frags(0, sA, sB, 0);
        // (a stage that writes the patch requests its weights BEHIND its last LDS write: while an LDS-DMA is pending hipcc puts
        // vmcnt(0) in front of every LDS write that may alias its destination -- behind this stage's own requests that would be
        // their full latency)
        // a stage that loads issues its three requests FIRST
        if (PL) patch_load(tb, sl_next, HALF);
        if (!(DBG & 1) && !PS) request_weights();
        if constexpr (!FIRST) {
            if (PS) patch_store((gslab + 1) & 1, HALF, ST1, ST1 + 1);
            multiply(1);
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);      // DS read
            if (PS) {
                // the groups are upper bounds, filled from the stage's end: what does not fit is left at its top
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);              // VALU
                    __builtin_amdgcn_sched_group_barrier(0x004, 3, 0);              // SALU
                }
                __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);                  // DS write (hi, lo)
            } else {
                if (PL) {
                    __builtin_amdgcn_sched_group_barrier(0x004, 16, 0);     // the patch loads and what they need
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);      // MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // VMEM read (LDS-DMA)
                if (PL) __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (PL) __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- K step 0 over the reads of K step 1
        if (!(DBG & 8) || g == 0) read_frags(1, sA, sB, 1);
        if (ST2 >= 0) patch_store((gslab + 1) & 1, HALF, ST2, ST2 + 1);
        multiply(0);
        if (!(DBG & 1) && PS) request_weights();
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        if (ST2 >= 0) {
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
                __builtin_amdgcn_sched_group_barrier(0x004, 4, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        } else if (PS) {
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 10, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // my requests for stage g + 1 (issued a stage ago) have landed -- the two of this stage, and the patch loads issued with
        // them at taps 0 / 4, stay in flight: a counted vmcnt --, my reads of stage g and my patch stores are done: raw barrier
        if (!(DBG & 4)) {
            constexpr int NV = (DBG & 1 ? 0 : 2) + (PL ? NPL / 2 : 0);
            __builtin_amdgcn_s_waitcnt(NV | 0x70);             // vmcnt(NV) lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        ++g;
        wb = wb == 2 ? 0 : wb + 1;
        if (TAP == 8) ++gslab;
    };
#define CD_TAP(T, F) stage_body(std::integral_constant<int, T>{}, F, sl, sl_next)

    // Every workgroup has the same work per block, so all 256 compute units reach their epilogue at the same moment: 33 MB of
    // stores in one burst (the epilogue measured 11-15k cycles per block, HBM-write-bound, the matrix pipes idle) and nothing
    // written in between.  A start-up offset of a quarter of a block per workgroup (by blockIdx / 8 & 3: neighbours on one XCD
    // differ) spreads the bursts over the block period.
    if (p.stagger_cycles > 0) {
        const unsigned long long until = __builtin_amdgcn_s_memtime() + (unsigned long long)(((int)blockIdx.x >> 3) & 3) * p.stagger_cycles;
        while (__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(32);
    }
    unsigned long long t_loop = 0, t_epi = 0, t0 = 0, t1 = 0;
    for (int bi = 0; bi < n_mine; ++bi) {
        if (CD_PROF) t0 = __builtin_amdgcn_s_memtime();
        for (int sl = 0; sl < p.nslab; ++sl) {
            const int sl_next = sl + 1 < p.nslab ? sl + 1 : 0;
            if (sl == 0) CD_TAP(0, std::true_type{}); else CD_TAP(0, std::false_type{});
            CD_TAP(1, std::false_type{}); CD_TAP(2, std::false_type{}); CD_TAP(3, std::false_type{}); CD_TAP(4, std::false_type{});
            CD_TAP(5, std::false_type{}); CD_TAP(6, std::false_type{}); CD_TAP(7, std::false_type{}); CD_TAP(8, std::false_type{});
        }
        multiply(1);                                           // the block's last K step, then its outputs
        if (CD_PROF) { t1 = __builtin_amdgcn_s_memtime(); t_loop += t1 - t0; }
        // ---- block epilogue
        const int blk = blk_beg + wg_j + bi * wg_per;
        const int img = blk / (p.gxb * p.gyb);
        const int by = cb.by, bx = cb.bx;
        cb = nb;
        nb = decode_blk(bi + 2 < n_mine ? bi + 2 : n_mine - 1);
        if constexpr (PIXA) {
            // pixels as the A operand: lane (l31, h) holds, for tile (t, n), channel 64 wn + 32 n + l31 of the 16 pixels
            // m = (r & 3) + 8 (r >> 2) + 4 h of the tile (row m >> 4, column m & 15): a store instruction writes 128 contiguous bytes
            // of two pixels -- whole lines; the four pixels of a pooling window sit in one lane
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int co = wn * 64 + n * 32 + l31;
                const float bv = s_bias[co];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int oy0 = by * 16 + 4 * wm + 2 * t;
                    if (POOL) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int r0 = 2 * (j & 1) + 4 * (j >> 1);            // columns (r0 & 3) + 8 (r0 >> 2) + 4 h, + 1; rows: r0, r0 + 8
                            float m = fmaxf(fmaxf(acc[t][n][r0], acc[t][n][r0 + 1]), fmaxf(acc[t][n][r0 + 8], acc[t][n][r0 + 9]));
                            m = m * inv + bv;                                     // max first: exact rescale and + b are monotone
                            if (RELU) m = fmaxf(m, 0.0f);
                            const int col = (r0 & 3) + 8 * ((r0 >> 2) & 1) + 4 * h;
                            const int py = oy0 >> 1, pxo = (bx * 16 + col) >> 1;
                            if (py < Ho && pxo < Wo) {
                                my_amax = fmaxf(my_amax, fabsf(m));
                                p.y[(((int64_t)img * Ho + py) * Wo + pxo) * 128 + co] = m;
                            }
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int m_ = (r & 3) + 8 * (r >> 2) + 4 * h;
                            const int oy = oy0 + (m_ >> 4), ox = bx * 16 + (m_ & 15);
                            float v = acc[t][n][r] * inv + bv;
                            if (RELU) v = fmaxf(v, 0.0f);
                            if (oy < p.H && ox < p.W) {
                                my_amax = fmaxf(my_amax, fabsf(v));
                                p.y[(((int64_t)img * p.H + oy) * p.W + ox) * 128 + co] = v;
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][n][r] = 0.0f;
                }
            }
        } else {
            // (the lane's geometry is recomputed HERE from an opaque copy of the thread id: values that live across the K loop only
            // to be used in the epilogue were spilled, and every reload from scratch in front of a store made the compiler wait for
            // ALL earlier stores -- 16 HBM round trips per block, 11-13k cycles of a 55k-cycle block)
            int ln = tid;
            asm volatile("" : "+v"(ln));
            const int el = ln & 31, eh = (ln >> 5) & 1;
            // the lane's 32 bias values, read BEFORE the first store: behind a pending LDS-DMA the compiler puts `vmcnt(0)` in front of
            // an LDS read that may alias its destination -- in between the stores that is an HBM round trip per read
            float4 bvs[2][4];
            {
                const float *bb = s_bias + wn * 64 + 4 * eh;
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int q = 0; q < 4; ++q) bvs[n][q] = *(const float4 *)(bb + n * 32 + 8 * q);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int oy = by * 16 + 4 * wm + 2 * t + (el >> 4), ox = bx * 16 + (el & 15);      // this lane's pixel
                bool store;
                float *yb;                                                                          // channel 64 wn + 4 h of the pixel
                if (POOL) {
                    store = ((el & 17) == 0) && (oy >> 1) < Ho && (ox >> 1) < Wo;                   // the window's top-left lane
                    yb = p.y + (((int64_t)img * Ho + (oy >> 1)) * Wo + (ox >> 1)) * 128 + wn * 64 + 4 * eh;
                } else {
                    store = oy < p.H && ox < p.W;
                    yb = p.y + (((int64_t)img * p.H + oy) * p.W + ox) * 128 + wn * 64 + 4 * eh;
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {              // channels 64 wn + 32 n + 8 q + 4 h .. + 3
                        float4 v = make_float4(acc[t][n][4 * q], acc[t][n][4 * q + 1], acc[t][n][4 * q + 2], acc[t][n][4 * q + 3]);
                        if (POOL) {
                            // 2 x 2 maximum FIRST (lanes l ^ 1: the column neighbour, l ^ 16: the row below), then the exact
                            // power-of-two rescale, bias and ReLU on the survivor: all monotone, so the result is the same bit for bit
                            v.x = cd_pool4(v.x); v.y = cd_pool4(v.y); v.z = cd_pool4(v.z); v.w = cd_pool4(v.w);
                        }
                        const float4 bv = bvs[n][q];
                        v.x = v.x * inv + bv.x; v.y = v.y * inv + bv.y; v.z = v.z * inv + bv.z; v.w = v.w * inv + bv.w;
                        if (RELU) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
                        if (store) {
                            my_amax = fmaxf(my_amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                            *(float4 *)(yb + n * 32 + 8 * q) = v;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][n][r] = 0.0f;
                }
            }
        }
        if (CD_PROF) t_epi += __builtin_amdgcn_s_memtime() - t1;
    }
#ifdef CSLAM_ABLATIONS
    if (cd_prof && blockIdx.x == 0 && tid == 0) { cd_prof[0] = t_loop; cd_prof[1] = t_epi; cd_prof[2] = (unsigned long long)n_mine; }
#endif

    if (p.amax_out) {
        unsigned *s_amax = (unsigned *)cd_smem;
        if (tid == 0) *s_amax = 0u;
        __syncthreads();
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) my_amax = fmaxf(my_amax, __shfl_xor(my_amax, o, 64));
        if (lane == 0) atomicMax(s_amax, __float_as_uint(my_amax));
        __syncthreads();
        if (tid == 0 && *s_amax > *(volatile unsigned *)p.amax_out) atomicMax(p.amax_out, *s_amax);
    }
}

/* 3 x 3 / stride 1 / pad 1 convolution, Cout = 128, Cin a multiple of 32: y = [pool](relu(conv(x, w) + bias)); x, y NHWC float32.
 * d_w2 = `direct_pair_weights` (vpr/winograd.py): [9][128][Cin/32][hi 32 | lo 32] halfs of s_w w, inv_sw = 1 / s_w; d_amax = 4-byte
 * slot holding (a bound of) max |x|; d_amax_out (or NULL): zeroed slot that receives max |y|. */
static int conv_direct_h_launch(const float *d_x, const void *d_w2, const float *d_bias, int B, int H, int W, int Cin,
                                int Cout, int relu, int pool, const unsigned *d_amax, float inv_sw,
                                unsigned *d_amax_out, float *d_y, void *stream, int x_pairs) {
    PTR_DEVICE(d_x);
    ARG_CHECK(d_x && d_w2 && d_y && d_amax, "NULL argument");
    ARG_CHECK(B >= 1 && H >= 1 && W >= 1, "empty map");
    ARG_CHECK(Cout == 128, "Cout must be 128");
    ARG_CHECK(Cin >= 32 && (Cin % 32) == 0, "Cin must be a multiple of 32");
    ARG_CHECK(!pool || ((H % 2) == 0 && (W % 2) == 0), "pooling needs even H and W");
    ARG_CHECK(inv_sw > 0.0f, "inv_sw must be positive");
    ARG_CHECK((int64_t)H * W * Cin * 4 < (1ll << 31), "one image must stay below 2 GiB (32-bit buffer offsets)");
    ConvDirectArgs a;
    a.x = d_x; a.w2 = (const char *)d_w2; a.bias = d_bias; a.y = d_y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.nslab = Cin / 32;
    a.gxb = (int)ceil_div64(W, 16); a.gyb = (int)ceil_div64(H, 16);
    const int64_t nblk = (int64_t)B * a.gxb * a.gyb;
    ARG_CHECK(nblk < (1ll << 30), "too many blocks for one launch");
    a.nblk = (int)nblk;
    const int n_cu = cslam_cu_count();
    ARG_CHECK(n_cu > 0, "no HIP device");
    a.relu = relu; a.pool = pool; a.amax_in = d_amax; a.inv_sw = inv_sw; a.amax_out = d_amax_out;
    // a quarter of a block's time per phase (~ 1.5k matrix-pipe cycles per stage + epilogue), only when every workgroup has several blocks
    a.stagger_cycles = nblk >= 4 * (int64_t)n_cu ? (a.nslab * 9 * 2000 + 6000) / 4 : 0;
#ifdef CSLAM_ABLATIONS
    if (const char *se = getenv("CSLAM_CD_STAGGER")) a.stagger_cycles = atoi(se);      // measurement build only
#endif
    const int grid = (int)(nblk < n_cu ? nblk : n_cu);
    hipStream_t st = (hipStream_t)stream;
#define CD_LAUNCH(R, P) do { \
        static DeviceOnce once; int once_dev; \
        if (once.todo(&once_dev)) { \
            HIP_TRY(hipFuncSetAttribute((const void *)conv3x3_direct_h_kernel<R, P>, hipFuncAttributeMaxDynamicSharedMemorySize, CD_LDS)); \
            once.done(once_dev); } \
        hipLaunchKernelGGL((conv3x3_direct_h_kernel<R, P>), dim3(grid), dim3(512), CD_LDS, st, a); } while (0)
#ifdef CSLAM_ABLATIONS
    if (const char *e = getenv("CSLAM_CD_DBG")) {              // timing-only ablations (wrong results): measurement build
        const int d = atoi(e);
#define CD_LAUNCH_D1(D, P) do { HIP_TRY(hipFuncSetAttribute((const void *)conv3x3_direct_h_kernel<true, P, D>, hipFuncAttributeMaxDynamicSharedMemorySize, CD_LDS)); \
        hipLaunchKernelGGL((conv3x3_direct_h_kernel<true, P, D>), dim3(grid), dim3(512), CD_LDS, st, a); HIP_TRY(hipGetLastError()); return CSLAM_OK; } while (0)
#define CD_LAUNCH_D(D) do { if (pool) CD_LAUNCH_D1(D, true); else CD_LAUNCH_D1(D, false); } while (0)
        if (d == 1) CD_LAUNCH_D(1);
        if (d == 2) CD_LAUNCH_D(2);
        if (d == 3) CD_LAUNCH_D(3);
        if (d == 7) CD_LAUNCH_D(7);
        if (d == 11) CD_LAUNCH_D(11);
        if (d == 15) CD_LAUNCH_D(15);
        if (d == 32) CD_LAUNCH_D(32);
        if (d == 64) CD_LAUNCH_D(64);
        if (d == 16 && !pool) { HIP_TRY(hipFuncSetAttribute((const void *)conv3x3_direct_h_kernel<true, false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CD_LDS));
                       hipLaunchKernelGGL((conv3x3_direct_h_kernel<true, false, 0, true>), dim3(grid), dim3(512), CD_LDS, st, a); HIP_TRY(hipGetLastError()); return CSLAM_OK; }
#undef CD_LAUNCH_D1
#undef CD_LAUNCH_D
    }
#endif
    if (x_pairs) {
#define CD_LAUNCH_P(R, P) do { \
        static DeviceOnce once; int once_dev; \
        if (once.todo(&once_dev)) { \
            HIP_TRY(hipFuncSetAttribute((const void *)conv3x3_direct_h_kernel<R, P, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CD_LDS)); \
            once.done(once_dev); } \
        hipLaunchKernelGGL((conv3x3_direct_h_kernel<R, P, 0, false, true>), dim3(grid), dim3(512), CD_LDS, st, a); } while (0)
        if (relu && pool) CD_LAUNCH_P(true, true);
        else if (relu) CD_LAUNCH_P(true, false);
        else if (pool) CD_LAUNCH_P(false, true);
        else CD_LAUNCH_P(false, false);
#undef CD_LAUNCH_P
        HIP_TRY(hipGetLastError());
        return CSLAM_OK;
    }
    if (relu && pool) CD_LAUNCH(true, true);
    else if (relu) CD_LAUNCH(true, false);
    else if (pool) CD_LAUNCH(false, true);
    else CD_LAUNCH(false, false);
#undef CD_LAUNCH
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}
CSLAM_API int cslam_conv3x3_direct_h_dev(const float *d_x, const void *d_w2, const float *d_bias, int B, int H, int W, int Cin,
                                         int Cout, int relu, int pool, const unsigned *d_amax, float inv_sw,
                                         unsigned *d_amax_out, float *d_y, void *stream) {
    return conv_direct_h_launch(d_x, d_w2, d_bias, B, H, W, Cin, Cout, relu, pool, d_amax, inv_sw, d_amax_out, d_y, stream, 0);
}
/* The same convolution reading a PAIR-FORMAT map (cslam_conv_igemm_h2p_dev's format, as cslam_conv3x3_direct_r_pairs_dev writes it):
 * d_x [B,H,W,Cin/32][hi 32 | lo 32] fp16 of s x, s the power of two of the 4-byte BOUND slot d_xbound; y float32 NHWC as above.  The
 * patch is staged without conversion (VGG-16 conv2_2 behind a conv2_1 that writes pairs: cslam/vpr/netvlad.py:163-171,227). */
CSLAM_API int cslam_conv3x3_direct_hp_dev(const void *d_x, const unsigned *d_xbound, const void *d_w2, const float *d_bias, int B, int H,
                                          int W, int Cin, int Cout, int relu, int pool, float inv_sw, unsigned *d_amax_out, float *d_y,
                                          void *stream) {
    return conv_direct_h_launch((const float *)d_x, d_w2, d_bias, B, H, W, Cin, Cout, relu, pool, d_xbound, inv_sw, d_amax_out, d_y, stream, 1);
}
