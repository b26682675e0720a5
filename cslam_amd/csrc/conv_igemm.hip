// conv_igemm.hip -- every convolution of the ResNet trunks as ONE implicit-GEMM kernel on exact fp16 pairs (gfx950).
//
// For CosPlace's default backbone (cslam/vpr/cosplace_utils/network.py:38-68: torchvision ResNet-18 without avgpool / fc; the
// reference's DEFAULT extractor, global_descriptor_loop_closure_detection.py:56-60): the 7x7 / stride-2 stem (with its MaxPool2d fused),
// the 3x3 layers of either stride and the 1x1 / stride-2 shortcuts.  Rounds 1-4 ran the stride-1 3x3 layers through the fp32 Winograd
// pipeline with library products and the rest through torch (MIOpen / CK, f32-input matrix pipe: 13 % of the trunk's multiply-adds, 60 %
// of its kernel time, profiles/r05_v20_c2_kernel_split.log).  BatchNorm is folded into weight and bias on the host (vpr/winograd.py::fold_bn).
//
//     y[b, ho, wo, co] = act( sum_{kh, kw, ci} x[b, ho s - p + kh, wo s - p + kw, ci] w[co, ci, kh, kw] + bias[co] (+ res[b, ho, wo, co]) )
//
// as a GEMM  Y [P = B Ho Wo pixels, Cout] = A [P, K] W^T [K, Cout],  K = KH KW Cin walked in blocks of 32 channels of one tap
// (kh, kw); the A block of a tap is never materialised: every K stage gathers it from the NHWC activation (one 128-byte run per
// pixel; pixels of the zero padding read as zero: an offset past the buffer descriptor, the hardware fills in the zeros).
// Arithmetic: the pair scheme of csrc/wino_gemm.hip.  x times a power of two s splits exactly into fp16 hi + lo; the weights are split
// offline (vpr/winograd.py::igemm_pair_weights); acc += xh wh; acc += xl wh; acc += xh wl on v_mfma_f32_32x32x16_f16 with fp32
// accumulation (the dropped xl wl is 2^-22 of the product): an fp32-grade convolution at a third of the fp16 matrix rate.
// LDS stage = the [hi 32 | lo 32] row images of the pair GEMM (128 bytes per row and K block, 16-byte chunks XOR-swizzled by the row,
// fragment reads conflict-free), double buffered; the weight tile comes by LDS-DMA from rows laid out the same way.  The activation tile:
//   AM 0  float32 NHWC map: 16-byte buffer loads two K blocks ahead of the MFMAs (two register sets), split one block ahead -- scale, 2
//         packed conversions, 4 v_fma_mix differences, 2 packed conversions per 16 bytes -> ds_write_b128 -- behind the MFMAs of the
//         current block; s from the 4-byte max |x| slot the producing layer left;
//   AM 1  the 3-channel 7x7 stem.  A tap's channels are 12 bytes, so a K block is a whole kernel ROW: the 21 values (kw, c) of row kh are
//         consecutive floats of the NHWC image; 7 blocks of 32 slots (21 used, the weights of the others zero), sixteen 4-byte buffer
//         loads per thread and block.  (K packed densely over (kh, kw, c) -- 147 values in 5 blocks, the tap decoded per element --
//         measured slower: 3.25 against 2.70 ms per 1000 frames.)  With `pool`: 8 x 16-pixel tiles and MaxPool2d(3, 2, 1) in the epilogue
//         (ResNet's 7x7 / 2 / 64-channel geometry: the persistent patch-form kernel further down instead);
//   AM 2  PAIR-FORMAT map, written by a previous layer's epilogue as [pixel][32-channel block][hi 32 | lo 32] -- the row image itself:
//         plain LDS-DMA like the weights, no register and no VALU work per K block.  Its scale is fixed before the producer has run: the
//         power of two for the bound max|x| max_co sum|w| + max|b| (+ max|shortcut|), carried in a 4-byte slot beside the measured maximum.
// Workgroup = TM pixels x TN output channels on four waves: 128 x 128 and 128 x 64 (2 x 2 waves), 256 x 64 (4 x 1; AM 2), two or three
// per CU.  Epilogue: rescale, bias, shortcut (either format), ReLU, store (either format) in 128-byte runs through per-tile buffer
// descriptors (no branch: rows beyond the last pixel are out of range), max |y| into the slot the next layer reads.
#include "common.h"
#include <type_traits>
#include <hip/hip_fp16.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float cf4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned cu4 __attribute__((ext_vector_type(4)));

#define CI_ROWB 128
#define CI_TM 128
#define CI_OOB 0x40000000              // stem: offset sentinel (row and column sentinels add up to 2^31: still past a descriptor of < 2^30 bytes)

struct ConvIgemmArgs {
    const float *x; const char *w2; const float *bias; const float *res; float *y;
    int B, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
    int P;                 // output pixels B Ho Wo
    int nk;                // K blocks: KH KW Cin / 32 (STEM: KH)
    int ncb;               // Cin / 32
    int relu;
    int n_mt, ntb, nx, mt_per_x;   // pixel tiles, channel tiles, XCDs the 1-D grid is laid out for, pixel tiles per XCD (workgroup -> tile: see the kernel)
    int pool;              // stem only: MaxPool2d(3, 2, 1) of the ReLU output fused (8 x 16-pixel tiles, y = the POOLED map, zeroed by the host)
    const unsigned *amax_in; float inv_sw; unsigned *amax_out;
    // pair-format activations (AM = 2 reads them, out_pairs writes them): [pixel][channel block of 32][hi 32 | lo 32] halfs of s x, s the
    // power of two ci_scale() derives from the 4-byte BOUND slot that travels with the tensor
    const unsigned *xbound;            // AM = 2: the bound x's pairs were scaled by
    int out_pairs;                     // y leaves as pairs scaled by ci_scale(bound_out)
    float wl1, bmax;                   // max over output channels of sum |w|, max |bias|: bound_out = max|x| wl1 + bmax (+ max|res|)
    unsigned *bound_out;               // receives that bound (written by workgroup (0, 0))
    int res_pairs;                     // res is in pair format, scaled by ci_scale(*res_bound)
    const unsigned *res_bound;         // bound slot of res (pairs) / a bound of max |res| (float32 res, when out_pairs)
    int dbg;                           // measurement build (-DCSLAM_ABLATIONS, CSLAM_CI_DBG; WRONG results, timing) of the AM = 2 loop: 1 no
                                       // activation requests after the prologue, 2 no weight requests, 4 no barrier, 8 no wait for the
                                       // requests, 16 no products, 32 no fragment reads either, 64 no epilogue, 128 activation requests for the taps with kw = 0 only
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ci_rsrc(const char *base, int64_t bytes) {
    const uint64_t a = (uint64_t)base;
    const uint64_t u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)a);
    const int n = __builtin_amdgcn_readfirstlane((int)(bytes > 0x7fffffff ? 0x7fffffff : bytes));
    return __builtin_amdgcn_make_buffer_rsrc((void *)u, 0, n, 0x00020000);
}
__device__ __forceinline__ cf4 ci_bload16(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    const cu4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return (cf4){__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}
// (a __device__ function, not the builtin inside the kernel's lambda: with the LDS-DMA builtin called from a lambda hipcc 7.2's HOST
// pass emits no launch stub for the kernel and says nothing -- the library then fails to load with an undefined symbol)
__device__ __forceinline__ void ci_blds16(__amdgpu_buffer_rsrc_t rs, int voff, int soff, char *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)lds_wave_base, 16, voff, soff, 0, 0);
}
// the power of two that brings max |x| into [2^13, 2^14): the hi halves keep 11 bits, the lo halves stay normal fp16 numbers for
// every value within 2^-10 of the maximum (smaller ones lose nothing that matters: their absolute error is 2^-25 of the scaled maximum)
__device__ __forceinline__ float ci_scale(unsigned amax_bits) {
    const float a = fminf(fmaxf(__uint_as_float(amax_bits), 1e-30f), 1e30f);
    int e;
    (void)frexpf(a, &e);
    return ldexpf(1.0f, 14 - e);
}
template <int HI>
__host__ __device__ __forceinline__ float ci_sub_half(float v, __half2 h) {       // v - (float)half HI of h: one v_fma_mix_f32
#if defined(__HIP_DEVICE_COMPILE__)
    float d;
    const unsigned hb = *(const unsigned *)&h;
    if (HI) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hb), "v"(v));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hb), "v"(v));
    return d;
#else
    return v - (HI ? __high2float(h) : __low2float(h));
#endif
}

// AM: how the activation tile reaches LDS.  0: float32 NHWC, split while it is staged; 1: the 3-channel stem (float32); 2: pair-format
// NHWC written by a previous layer's epilogue -- the tile is then plain LDS-DMA like the weights, no register, no VALU work per K block
// Tile: TM pixels x TN output channels on four waves, WM = TM / 64 along the pixels and WN = 4 / WM along the channels: wave tile 64 x
// TN / WN.  128 x 128 and 128 x 64 (2 x 2 waves); 256 x 64 (4 x 1, pair format only): layers with 64 output channels -- a 64 x 32 wave
// tile reads one LDS fragment per MFMA, a 64 x 64 one two per three, and the LDS pipe is what bounds these kernels.
// KWS (AM = 2, 3x3 / stride 1 / pad 1, 128-pixel tiles): the three taps of a kernel row share ONE activation block.  In the flat pixel
// order the tiles use, tap kw of output pixel p is input pixel p + (kh - 1) W + (kw - 1): the 130 rows "pixels first - 1 .. first + 128
// at kernel row kh" are requested once per (slab, kh) and the taps read them at row offsets 0, 1, 2; what is wrong there -- the pixels
// at a row's ends, whose neighbour in the flat order belongs to another row -- is masked in the A fragments (32 v_cndmask per step for
// kw = 0 and 2).  A third of the activation requests (L2 -> LDS bytes -32 %).
// KWS with 256-pixel tiles (AM = 2, 128 output channels: the stride-1 layers of ResNet's layer2 .. layer4): four waves along the pixels,
// wave tile 64 x 128 = 8 accumulator tiles, ONE workgroup per compute unit with the whole register file.  A (slab, kernel row) block
// of 258 rows and three weight blocks feed 3 x 48 MFMAs per wave: 0.66 of the 128-pixel form's L2 -> LDS bytes per product, half its
// barriers per product (those bytes, not the matrix pipe, are what the board's power budget was being spent on: DESIGN.md section 8).
template <int TM, int TN, int AM, bool KWS = false>
__global__ __launch_bounds__(256, (KWS && TM == 256) ? 1 : 2) void conv_igemm_h2_kernel(ConvIgemmArgs p) {
    constexpr bool STEM = AM == 1, PAIRS = AM == 2;
    static_assert(!KWS || ((AM == 2 || AM == 0) && TM == 128) || (AM == 2 && TM == 256), "the shared-row form: 128-pixel tiles of the general forms, 256-pixel tiles of the pair format");
    constexpr int KA_ROWS = TM + 8;                    // KWS: TM + 2 rows of a (slab, kernel row) block, padded to whole 1 KB DMA pieces
    constexpr int KA_BYTES = KA_ROWS * CI_ROWB;        // 17 408 / 33 792: two of them, then the two weight stages
    constexpr int KA_NP = TM / 32 + 1;                 // its DMA pieces of 32 rows (the last: 8 rows, one wave's 1 KB)
    constexpr int NST = 2;                             // LDS stages
    constexpr int WM = TM / 64, WN = 4 / WM, TNW = TN / WN;
    constexpr int MT = 2, NT = TNW / 32;               // wave tile 64 x TNW in 32 x 32 MFMA tiles
    constexpr int NLA = TM * 8 / 256;                  // AM = 2: 16-byte activation chunks per thread and stage
    static_assert(TM == 128 || (TM == 256 && PAIRS), "the register-staged forms are written for 128-pixel tiles");
    static_assert(!KWS || AM == 2 || TM == 128, "float32 input with shared rows: 128-pixel tiles");
    constexpr int OPA = TM * CI_ROWB, OPB = TN * CI_ROWB;
    constexpr int STAGE = OPA + OPB;
    constexpr int NLB = TN * 8 / 256;                  // 16-byte weight chunks per thread and stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5, l31 = lane & 31;
    // Workgroup b runs on XCD b % nx.  Inside an XCD the channel tiles of ONE pixel tile follow each other (they read the same activation
    // tile at about the same time: one L2 miss, ntb - 1 hits); an XCD owns a contiguous range of pixel tiles (neighbours share the rows
    // above and below a tile: their halo is an L2 hit too).
    const int bx = blockIdx.x % p.nx, bg = blockIdx.x / p.nx;
    const int nt = bg % p.ntb, mt = bx * p.mt_per_x + bg / p.ntb;
    if (bg / p.ntb >= p.mt_per_x || mt >= p.n_mt) return;

    // ---- this thread's share of the activation tile: pixel row tid >> 1, channels 16 (tid & 1) .. + 15 of every K block
    const int ar = tid >> 1, hf = tid & 1;
    // 1-D tiles: CI_TM consecutive pixels of the [B][Ho][Wo] order; pooled stem: 8 x 16-pixel tiles of one image (tile row tr, column tc)
    const bool tile2d = STEM && p.pool;
    int tr = 0, tc = 0, tb = 0;
    if (tile2d) {
        const int tcols = p.Wo >> 4, tpi = (p.Ho >> 3) * tcols;
        tb = mt / tpi;
        const int t = mt - tb * tpi;
        tr = t / tcols;
        tc = t - tr * tcols;
    }
    const int pix = mt * TM + ar;
    const bool pvalid = tile2d || pix < p.P;
    int b_, hi0, wi0;
    if (tile2d) {
        b_ = tb;
        hi0 = (tr * 8 + (ar >> 4)) * p.stride - p.pad;
        wi0 = (tc * 16 + (ar & 15)) * p.stride - p.pad;
    } else {
        const int pp = pvalid ? pix : 0;
        const int hw = p.Ho * p.Wo;
        b_ = pp / hw;
        const int rem = pp - b_ * hw;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        hi0 = ho * p.stride - p.pad;
        wi0 = wo * p.stride - p.pad;
    }
    const float sc = ci_scale(PAIRS ? *p.xbound : *p.amax_in);
    // descriptor from the image of the workgroup's first pixel on (the host checks that a tile's span of images stays below 2^31 bytes)
    const int b0 = __builtin_amdgcn_readfirstlane(tile2d ? tb : (int)(((int64_t)mt * TM) / (p.Ho * p.Wo)));
    const int64_t img = (int64_t)p.H * p.W * p.Cin;
    const int64_t xbytes = (int64_t)(p.B - b0) * img * 4;
    const __amdgpu_buffer_rsrc_t rsX = ci_rsrc((const char *)(p.x + b0 * img), STEM && xbytes > CI_OOB - 1 ? CI_OOB - 1 : xbytes);
    const int pixoff0 = ((b_ - b0) * p.H + hi0) * p.W + wi0;
    int coloff[STEM ? 16 : 1];
    if (STEM) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int slot = hf * 16 + j, wi = wi0 + slot / 3;
            coloff[j] = (slot < p.KW * 3 && wi >= 0 && wi < p.W) ? (wi0 * 3 + slot) * 4 : CI_OOB;
        }
    }
    const int a_chunk0 = (hf * 2) ^ ((ar >> 1) & 7), a_chunk1 = (hf * 2 + 1) ^ ((ar >> 1) & 7);
    const int a_chunk2 = (4 + hf * 2) ^ ((ar >> 1) & 7), a_chunk3 = (5 + hf * 2) ^ ((ar >> 1) & 7);

    // K block (kh, kw, cb) of the general form / kernel row kh of the stem -> one of two register sets (zeros outside the image)
    cf4 ra0[4], ra1[4];
    auto a_load = [&](cf4 (&areg)[4], int kh, int kw, int cb) {
        if (!STEM) {
            // buffer loads: a tap outside the image gets an offset past the descriptor's range and the hardware returns zeros -- nothing
            // touches the registers between the load and the split one K block later, so no wait lands in front of the products
            const int hi = hi0 + kh;
            const int wi = wi0 + kw;
            const bool in = pvalid && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
            const int voff = in ? (pixoff0 + kh * p.W + kw) * (p.Cin * 4) + hf * 64 : 0x7fffffff;
#pragma unroll
            for (int q = 0; q < 4; ++q) areg[q] = ci_bload16(rsX, voff + q * 16, cb * 128);
        } else {
            // stem: K block = kernel row kh; slot j = kw * 3 + c of it is float wi0 * 3 + j of image row hi0 + kh.  One 4-byte buffer load
            // per slot (the run starts at an odd float): a slot outside the row carries the sentinel in its column offset, a row outside
            // the image the sentinel in the row offset -- either way the offset is past the descriptor and the hardware returns zero
            const int hi = hi0 + kh;
            const bool rowin = pvalid && hi >= 0 && hi < p.H;
            const int rowoff = rowin ? ((b_ - b0) * p.H + hi) * p.W * 12 : CI_OOB;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsX, rowoff + coloff[q * 4 + e], 0, 0));
                areg[q] = (cf4){v[0], v[1], v[2], v[3]};
            }
        }
    };
    // split the registers into pairs and write them into stage `st`'s row image
    auto a_store = [&](const cf4 (&areg)[4], int st) {
        char *row = smem + st * STAGE + ar * CI_ROWB;
        unsigned hh[8], ll[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const cf4 v = areg[q] * sc;
            const __half2 h0 = __floats2half2_rn(v.x, v.y), h1 = __floats2half2_rn(v.z, v.w);
            const __half2 l0 = __floats2half2_rn(ci_sub_half<0>(v.x, h0), ci_sub_half<1>(v.y, h0));
            const __half2 l1 = __floats2half2_rn(ci_sub_half<0>(v.z, h1), ci_sub_half<1>(v.w, h1));
            hh[2 * q] = *(const unsigned *)&h0; hh[2 * q + 1] = *(const unsigned *)&h1;
            ll[2 * q] = *(const unsigned *)&l0; ll[2 * q + 1] = *(const unsigned *)&l1;
        }
        *(cu4 *)(row + (a_chunk0 << 4)) = (cu4){hh[0], hh[1], hh[2], hh[3]};
        *(cu4 *)(row + (a_chunk1 << 4)) = (cu4){hh[4], hh[5], hh[6], hh[7]};
        *(cu4 *)(row + (a_chunk2 << 4)) = (cu4){ll[0], ll[1], ll[2], ll[3]};
        *(cu4 *)(row + (a_chunk3 << 4)) = (cu4){ll[4], ll[5], ll[6], ll[7]};
    };

    // ---- AM = 2: the activation tile by LDS-DMA.  Chunk pch = i * 256 + tid -> tile row i * 32 + (tid >> 3), physical slot tid & 7 =
    // logical 16-byte chunk ^ swz(row) of the pixel's 128-byte (tap, channel block) run; a tap outside the image: an offset past the
    // descriptor, the DMA writes zeros
    int pa_off[PAIRS ? NLA : 1], pa_h[PAIRS ? NLA : 1], pa_w[PAIRS ? NLA : 1];
    const int pa_chunk = ((tid & 7) ^ ((tid >> 4) & 7)) << 4;
    if (PAIRS) {
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            const int64_t px = (int64_t)mt * TM + i * 32 + (tid >> 3);
            const int hw = p.Ho * p.Wo;
            const int pp = px < p.P ? (int)px : 0;
            const int bb = pp / hw, rem = pp - bb * hw;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            pa_h[i] = px < p.P ? ho * p.stride - p.pad : -(1 << 20);
            pa_w[i] = wo * p.stride - p.pad;
            pa_off[i] = (((bb - b0) * p.H + pa_h[i]) * p.W + pa_w[i]) * (p.Cin * 4) + pa_chunk;      // bytes
        }
    }
    auto a_dma = [&](int st, int kh, int kw, int cb) {
        const int tapoff = (kh * p.W + kw) * (p.Cin * 4);
#pragma unroll
        for (int i = 0; i < (PAIRS ? NLA : 0); ++i) {
            // (unsigned compares: one per coordinate; the tap's offset is wave-uniform; select, not branch)
            const bool in = (unsigned)(pa_h[i] + kh) < (unsigned)p.H && (unsigned)(pa_w[i] + kw) < (unsigned)p.W;
            const int off = pa_off[i] + tapoff;
            ci_blds16(rsX, in ? off : 0x7fffffff, cb * 128, smem + st * STAGE + i * (256 * 16) + wave * 1024);
        }
    };

    // ---- KWS: LDS row R = i * 32 + (tid >> 3) (i = 0..4, rows 130.. read nothing) holds centre pixel first - 1 + R at kernel row kh
    int ka_off[KWS ? KA_NP : 1], ka_h[KWS ? KA_NP : 1];
    if (KWS) {
#pragma unroll
        for (int i = 0; i < KA_NP; ++i) {
            const int R = i * 32 + (tid >> 3);
            const int64_t pc = (int64_t)mt * TM - 1 + R;
            const bool ok = R < TM + 2 && pc >= 0 && pc < p.P;
            const int hw = p.Ho * p.Wo;
            const int pp = ok ? (int)pc : 0;
            const int bb = pp / hw, rem = pp - bb * hw;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            ka_h[i] = ok ? ho - 1 : -(1 << 20);                                                        // + kh = the input row
            ka_off[i] = (((bb - b0) * p.H + ho - 1) * p.W + wo) * (p.Cin * 4) + pa_chunk;             // + kh W Cin 4: bytes
        }
    }
    auto ka_dma = [&](int abuf, int kh, int cb, int i0, int i1) {       // pieces i0 .. i1 - 1 of the block (spread over the three steps)
        const int rowoff = kh * p.W * (p.Cin * 4);
#pragma unroll
        for (int i = 0; i < (KWS ? KA_NP : 0); ++i) {
            if (i < i0 || i >= i1 || (i == KA_NP - 1 && wave != 0)) continue;      // (the last piece = rows TM .. TM + 7: one wave's 1 KB; the buffer ends there)
            const bool in = (unsigned)(ka_h[i] + kh) < (unsigned)p.H;
            ci_blds16(rsX, in ? ka_off[i] + rowoff : 0x7fffffff, cb * 128, smem + abuf * KA_BYTES + i * (256 * 16) + wave * 1024);
        }
    };

    // ---- KWS with float32 input (AM = 0): the (slab, kernel row) block goes through registers -- item 0: row tid >> 1, channels
    // 16 (tid & 1) .. + 15 (rows 0..127); item 1: rows 128..135 on the first 16 threads (every wave issues its loads: counted waits) --
    // and is split once per THREE steps instead of once per step
    cf4 kr[2][4];
    auto kf_load = [&](int kh, int cb) {
#pragma unroll
        for (int it = 0; it < ((KWS && AM == 0) ? 2 : 0); ++it) {
            const int R = it * 128 + (tid >> 1);
            const int64_t pc = (int64_t)mt * TM - 1 + R;
            const bool ok = (it == 0 || tid < 16) && pc >= 0 && pc < p.P;
            const int hw = p.Ho * p.Wo;
            const int pp = ok ? (int)pc : 0;
            const int bb = pp / hw, rem = pp - bb * hw;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            const int hi = ho - 1 + kh;
            const bool in = ok && (unsigned)hi < (unsigned)p.H;
            const int voff = in ? (((bb - b0) * p.H + hi) * p.W + wo) * (p.Cin * 4) + hf * 64 : 0x7fffffff;
#pragma unroll
            for (int q = 0; q < 4; ++q) kr[it][q] = ci_bload16(rsX, voff + q * 16, cb * 128);
        }
    };
    auto kf_store = [&](int abuf) {
#pragma unroll
        for (int it = 0; it < ((KWS && AM == 0) ? 2 : 0); ++it) {
            if (it == 1 && tid >= 16) continue;
            const int R = it * 128 + (tid >> 1);
            char *row = smem + abuf * KA_BYTES + R * CI_ROWB;
            const int sw = (R >> 1) & 7;
            unsigned hh[8], ll[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const cf4 v = kr[it][q] * sc;
                const __half2 h0 = __floats2half2_rn(v.x, v.y), h1 = __floats2half2_rn(v.z, v.w);
                const __half2 l0 = __floats2half2_rn(ci_sub_half<0>(v.x, h0), ci_sub_half<1>(v.y, h0));
                const __half2 l1 = __floats2half2_rn(ci_sub_half<0>(v.z, h1), ci_sub_half<1>(v.w, h1));
                hh[2 * q] = *(const unsigned *)&h0; hh[2 * q + 1] = *(const unsigned *)&h1;
                ll[2 * q] = *(const unsigned *)&l0; ll[2 * q + 1] = *(const unsigned *)&l1;
            }
            *(cu4 *)(row + (((hf * 2) ^ sw) << 4)) = (cu4){hh[0], hh[1], hh[2], hh[3]};
            *(cu4 *)(row + (((hf * 2 + 1) ^ sw) << 4)) = (cu4){hh[4], hh[5], hh[6], hh[7]};
            *(cu4 *)(row + (((4 + hf * 2) ^ sw) << 4)) = (cu4){ll[0], ll[1], ll[2], ll[3]};
            *(cu4 *)(row + (((5 + hf * 2) ^ sw) << 4)) = (cu4){ll[4], ll[5], ll[6], ll[7]};
        }
    };

    // ---- weight tile by LDS-DMA: chunk pch = i * 256 + tid -> row pch >> 3, physical slot pch & 7 = logical chunk slot ^ swz(row)
    const int pitchw = p.nk * CI_ROWB;
    const __amdgpu_buffer_rsrc_t rsB = ci_rsrc(p.w2 + (int64_t)nt * TN * pitchw, (int64_t)TN * pitchw);
    int voffB[NLB];
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
        const int pch = i * 256 + tid, r = pch >> 3, slot = pch & 7;
        voffB[i] = r * pitchw + ((slot ^ ((r >> 1) & 7)) << 4);
    }
    // (called for K positions 0, 1, 2, ... in order: position -> (slab, tap) by two counters; the weights' blocks are stored tap-major)
    int bw_tap = 0, bw_cb = 0;
    const int ntap = p.KH * p.KW;
    auto b_load = [&](int st, int k) {
        const int widx = (PAIRS || KWS) ? bw_tap * p.ncb + bw_cb : k;
        if ((PAIRS || KWS) && ++bw_tap == ntap) { bw_tap = 0; ++bw_cb; }
#pragma unroll
        for (int i = 0; i < NLB; ++i)
            ci_blds16(rsB, voffB[i], widx * CI_ROWB, smem + (KWS ? 2 * KA_BYTES + st * OPB : st * STAGE + OPA) + i * (256 * 16) + wave * 1024);
    };

    // ---- fragments: row * 128 + (chunk ^ swz) * 16, chunk = 4 lo + 2 s + h for the 16-channel K step s
    const int swz = (lane >> 1) & 7;
    int foff[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int lo = 0; lo < 2; ++lo) foff[s][lo] = ((4 * lo + 2 * s + h) ^ swz) << 4;
    const int arow0 = (wm * 64 + l31) * CI_ROWB;
    const int brow0 = (wn * TNW + l31) * CI_ROWB;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    // one K block: two 16-channel steps of three products per tile.  AM = 2: the fragments of BOTH steps are requested before the first
    // product (the scheduler left to itself reads three, waits, multiplies: an LDS round trip in front of every other MFMA) and the
    // products stay in front of the step's closing wait; AM = 0, 1: the scheduler is free to weave the next block's split into the
    // MFMA stream (pinning the order there costs: the stem 2.6 -> 10.4 ms)
    auto multiply = [&](const char *sA, bool products = true) {
        const char *sB = sA + OPA;
        if (PAIRS) {
            f16x8 fa[2][2][MT], fb[2][2][NT];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    fa[s][0][m] = *(const f16x8 *)(sA + arow0 + m * 32 * CI_ROWB + foff[s][0]);
                    fa[s][1][m] = *(const f16x8 *)(sA + arow0 + m * 32 * CI_ROWB + foff[s][1]);
                }
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    fb[s][0][n] = *(const f16x8 *)(sB + brow0 + n * 32 * CI_ROWB + foff[s][0]);
                    fb[s][1][n] = *(const f16x8 *)(sB + brow0 + n * 32 * CI_ROWB + foff[s][1]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!products) {                           // (measurement build: the fragments are still read and waited for)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) asm volatile("" :: "v"(fa[s][0][m]), "v"(fa[s][1][m]));
#pragma unroll
                    for (int n = 0; n < NT; ++n) asm volatile("" :: "v"(fb[s][0][n]), "v"(fb[s][1][n]));
                }
                return;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s][0][m], fb[s][0][n], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s][1][m], fb[s][0][n], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s][0][m], fb[s][1][n], acc[m][n], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
            return;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 fa[2][MT], fb[2][NT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                fa[0][m] = *(const f16x8 *)(sA + arow0 + m * 32 * CI_ROWB + foff[s][0]);
                fa[1][m] = *(const f16x8 *)(sA + arow0 + m * 32 * CI_ROWB + foff[s][1]);
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                fb[0][n] = *(const f16x8 *)(sB + brow0 + n * 32 * CI_ROWB + foff[s][0]);
                fb[1][n] = *(const f16x8 *)(sB + brow0 + n * 32 * CI_ROWB + foff[s][1]);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][m], fb[0][n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][m], fb[0][n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][m], fb[1][n], acc[m][n], 0, 0, 0);
                }
        }
    };
    int kh = 0, kw = 0, cb = 0;                        // the K block the loader is at
    auto advance = [&]() {
        if (STEM) { ++kh; return; }
        // the taps are the INNER loop: the nine (kh, kw) blocks of one 32-channel slab re-read the same pixels' 128-byte runs back to back
        // (L2 hits); with the slabs inside a tap, a tile's whole input went by between two reads of a run and every tap missed the
        // 4 MB L2 -- PMC: 3.2 GB of fabric traffic on layer4 against 0.2 GB of operands, the kernel at 1.35 GHz under the power cap
        // (AM = 2; the float32 form keeps the slabs inside a tap: it is one 64-channel layer of the trunk, and its tests pin that order's rounding)
        if (PAIRS) { if (++kw == p.KW) { kw = 0; if (++kh == p.KH) { kh = 0; ++cb; } } }
        else if (++cb == p.ncb) { cb = 0; if (++kw == p.KW) { kw = 0; ++kh; } }
    };
    // Pipeline: the activation block is fetched TWO K blocks ahead (two register sets, the loop unrolled by two so that the sets are
    // named at compile time), split and written into LDS one block ahead; the weight block comes one block ahead by LDS-DMA.  Requests of
    // a step are issued weights first, so the step's closing wait can be COUNTED -- vmcnt(4): the four activation loads of block k + 2
    // stay in flight across the raw barrier (a workgroup's K step is 12-24 MFMAs per wave: with the loads of block k + 1 issued at its
    // top, as in the first form, every step ended in an L2 round trip; the stem's sixteen 4-byte loads per block: vmcnt(16)).
    // s_waitcnt simm16 on gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14
    // The steady loop is written without conditions (the tail of up to three blocks is peeled, its flags literals): with `if (more)`
    // around the loads and the split, the compiler's wait-count model has to assume that a register set may still be in flight at the
    // loop header and puts vmcnt(0) in front of the products.
    auto step = [&](const bool more1, const bool more2, int k, int cur, const cf4 (&rnext)[4], cf4 (&rpref)[4]) {
        if (more1) b_load(cur ^ 1, k + 1);
        if (more2) { advance(); a_load(rpref, kh, kw, cb); }
        __builtin_amdgcn_sched_barrier(0);             // (the scheduler otherwise sinks the loads to the step's end: one block ahead again)
        multiply(smem + cur * STAGE);
        if (more1) a_store(rnext, cur ^ 1);
        if (more2) __builtin_amdgcn_s_waitcnt(STEM ? (0 | (7 << 4) | (0 << 8) | (1 << 14))      // vmcnt(16) lgkmcnt(0)
                                                   : (4 | (7 << 4) | (0 << 8)));                // vmcnt(4) lgkmcnt(0)
        else __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_s_barrier();
    };
    if (!PAIRS && !KWS) {
    a_load(ra0, kh, kw, cb);
    b_load(0, 0);
    a_store(ra0, 0);
    if (p.nk > 1) { advance(); a_load(ra1, kh, kw, cb); }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_s_barrier();
    int k = 0;
    for (; k + 3 < p.nk; k += 2) {
        step(true, true, k, 0, ra1, ra0);
        step(true, true, k + 1, 1, ra0, ra1);
    }
    const int left = p.nk - k;                         // 1 (nk = 1), 2 or 3
    if (left == 3) {
        step(true, true, k, 0, ra1, ra0);
        step(true, false, k + 1, 1, ra0, ra1);
        step(false, false, k + 2, 0, ra1, ra0);
    } else if (left == 2) {
        step(true, false, k, 0, ra1, ra0);
        step(false, false, k + 1, 1, ra0, ra1);
    } else {
        step(false, false, k, 0, ra1, ra0);
    }
    } else if (KWS) {
        // lane masks: output pixel first + wm 64 + m 32 + l31 sits at a row's first / last column -> its kw = 0 / kw = 2 tap is padding
        bool e0[MT], e2[MT];
        int arow_k[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int64_t px = (int64_t)mt * TM + wm * 64 + m * 32 + l31;
            const int wo = (int)(px % p.Wo);
            e0[m] = wo == 0; e2[m] = wo == p.Wo - 1;
            arow_k[m] = wm * 64 + m * 32 + l31;                            // + kw = the LDS row of tap kw
        }
        const int ngrp = p.ncb * 3;                                        // (slab, kernel row) groups, three steps (kw) each
        // group G = cb * 3 + kh -> activation buffer G & 1; step k = 3 G + kw -> weight buffer k & 1
        if (AM == 0) { kf_load(0, 0); b_load(0, 0); kf_store(0); }
        else { ka_dma(0, 0, 0, 0, KA_NP); b_load(0, 0); }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_s_barrier();
        int gkh = 0, gcb = 0;                                              // the group being multiplied
        for (int G = 0; G < ngrp; ++G) {
            int nkh = gkh + 1, ncb_ = gcb;
            if (nkh == 3) { nkh = 0; ++ncb_; }
            const bool moreG = G + 1 < ngrp;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int k = 3 * G + kw;
                __builtin_amdgcn_sched_barrier(0);
                if (AM == 0) {
                    // weights first, then (kw = 0) the next block's float32 loads: they stay in flight over this step's counted wait and
                    // the next step, and are split behind the products of kw = 2
                    if (k + 1 < p.nk) b_load((k + 1) & 1, k + 1);
                    if (kw == 0 && moreG) kf_load(nkh, ncb_);
                } else {
                    if (moreG) ka_dma((G + 1) & 1, nkh, ncb_, kw == 0 ? 0 : (kw == 1 ? KA_NP / 3 + (KA_NP % 3 ? 1 : 0) : 2 * (KA_NP / 3) + (KA_NP % 3)),
                                      kw == 0 ? KA_NP / 3 + (KA_NP % 3 ? 1 : 0) : (kw == 1 ? 2 * (KA_NP / 3) + (KA_NP % 3) : KA_NP));
                    if (k + 1 < p.nk) b_load((k + 1) & 1, k + 1);
                }
                // fragments of tap kw: activation rows shifted by kw, the chunk swizzle follows the physical row
                const char *sA = smem + (G & 1) * KA_BYTES, *sB = smem + 2 * KA_BYTES + (k & 1) * OPB;
                f16x8 fa[2][2][MT], fb[2][2][NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int row = arow_k[m] + kw, sw = (row >> 1) & 7;
                    const bool z = (kw == 0 && e0[m]) || (kw == 2 && e2[m]);
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int lo = 0; lo < 2; ++lo) {
                            cu4 v = *(const cu4 *)(sA + row * CI_ROWB + (((4 * lo + 2 * s2 + h) ^ sw) << 4));
                            if (kw != 1) v = z ? (cu4)(0u) : v;
                            fa[s2][lo][m] = *(const f16x8 *)&v;
                        }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        fb[s2][0][n] = *(const f16x8 *)(sB + brow0 + n * 32 * CI_ROWB + foff[s2][0]);
                        fb[s2][1][n] = *(const f16x8 *)(sB + brow0 + n * 32 * CI_ROWB + foff[s2][1]);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s2][0][m], fb[s2][0][n], acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s2][1][m], fb[s2][0][n], acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s2][0][m], fb[s2][1][n], acc[m][n], 0, 0, 0);
                        }
                __builtin_amdgcn_sched_barrier(0);
                if (AM == 0 && kw == 2 && moreG) kf_store((G + 1) & 1);
                if (AM == 0 && kw == 0 && moreG) __builtin_amdgcn_s_waitcnt(8 | (7 << 4) | (0 << 8));      // vmcnt(8): the eight float32 loads stay in flight
                else __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_s_barrier();
            }
            gkh = nkh; gcb = ncb_;
        }
    } else {
        // both operands by DMA into a ring of NST stages: block k + NST - 1 is requested at the top of step k, so a request has
        // NST - 2 whole steps plus this step's products to arrive; with three stages the step's closing wait is COUNTED (the newest request
        // stays in flight).  Measured on layer1 (64 -> 64 channels, 56 x 56): three stages at two workgroups per CU 1.52 ms, two stages
        // at three per CU 1.31 ms, the 256 x 64 tile with two stages 0.92 ms -- NST stays 2.
        constexpr int NDMA = NLA + NLB;                // DMA instructions per thread and block
        a_dma(0, kh, kw, cb);
        b_load(0, 0);
        if (NST == 3 && p.nk > 1) {
            advance();
            a_dma(1, kh, kw, cb);
            b_load(1, 1);
            __builtin_amdgcn_s_waitcnt(NDMA | (7 << 4) | (0 << 8));
        } else {
            __builtin_amdgcn_s_waitcnt(0);
        }
        __builtin_amdgcn_s_barrier();
        int cur = 0, nxt = NST - 1;                    // stage of block k, stage block k + NST - 1 goes to
#ifdef CSLAM_ABLATIONS
        const int dbg = p.dbg;
#else
        constexpr int dbg = 0;
#endif
        for (int k = 0; k < p.nk; ++k) {
            const bool more = k + NST - 1 < p.nk;
            if (more) {
                advance();
                if (!(dbg & 1) && (!(dbg & 128) || kw == 0)) a_dma(nxt, kh, kw, cb);
                if (!(dbg & 2)) b_load(nxt, k + NST - 1);
            }
            if (!(dbg & 32)) multiply(smem + cur * STAGE, !(dbg & 16));
                                                       // (ends in a sched_barrier: MFMAs are no memory operations, the scheduler sinks them
                                                       // below the wait and the barrier, and the requests then have no products to fly under)
            if (NST == 3 && more) __builtin_amdgcn_s_waitcnt(NDMA | (7 << 4) | (0 << 8));
            else if (dbg & 8) __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0) only (vmcnt 63, expcnt 7)
            else __builtin_amdgcn_s_waitcnt(0);
            if (!(dbg & 4)) __builtin_amdgcn_s_barrier();
            cur = cur + 1 == NST ? 0 : cur + 1;
            nxt = nxt + 1 == NST ? 0 : nxt + 1;
        }
    }

#ifdef CSLAM_ABLATIONS
    if (p.dbg & 64) return;
#endif
    // ---- epilogue: lane = output channel (128-byte runs per pixel), 16 pixels per accumulator tile
    const float inv = p.inv_sw / sc;
    float amax = 0.0f;
    if (tile2d) {
        // ReLU(conv + bias) of the tile -> LDS [pixel 128][channel TN] (the stages are free: the last step ended in a barrier), then
        // MaxPool2d(3, 2, 1): the tile owns pooled pixels (4 tr + pr, 8 tc + pc), pr 0..4, pc 0..8 -- the maximum over the window's
        // pixels INSIDE this tile; a window that lies in one tile is stored, the others (first pooled row / column of a tile and the
        // row / column it hands to its neighbours) are completed with atomic maxima on the bit patterns (values >= 0; y starts at zero)
        float *T = (float *)smem;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int cl = wn * TNW + n * 32 + l31;
            const float bv = p.bias ? p.bias[nt * TN + cl] : 0.0f;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float v = fmaxf(acc[m][n][r] * inv + bv, 0.0f);
                    T[row * TN + cl] = v;
                    amax = fmaxf(amax, v);
                }
        }
        __syncthreads();
        const int Hp = p.Ho >> 1, Wp = p.Wo >> 1;
        for (int it = tid; it < 45 * TN; it += 256) {
            const int ch = it & (TN - 1), pp = it / TN;
            const int pr = pp / 9, pc = pp - pr * 9;
            const int gp = tr * 4 + pr, gq = tc * 8 + pc;
            if (gp >= Hp || gq >= Wp) continue;
            // nine unconditional reads: a window row / column outside the tile is clamped onto one inside (a duplicate does not change a
            // maximum) -- with `continue` around them every read sat in a block of its own behind its own LDS round trip
            float mx = 0.0f;
#pragma unroll
            for (int dr = -1; dr <= 1; ++dr) {
                const int lr = min(max(2 * pr + dr, 0), 7);
#pragma unroll
                for (int dc = -1; dc <= 1; ++dc) {
                    const int lc = min(max(2 * pc + dc, 0), 15);
                    mx = fmaxf(mx, T[(lr * 16 + lc) * TN + ch]);
                }
            }
            float *dst = p.y + (((int64_t)tb * Hp + gp) * Wp + gq) * p.Cout + nt * TN + ch;
            const bool whole = (pr >= 1 || tr == 0) && pr <= 3 && (pc >= 1 || tc == 0) && pc <= 7;
            if (whole) *dst = mx;
            else atomicMax((unsigned *)dst, __float_as_uint(mx));
        }
    } else {
        // Pair-format tensors: a pixel's 32-channel block is [hi 32 | lo 32] halfs.  Lane l31 = channel: after one exchange with lane
        // l31 ^ 1 the even lane holds (hi_l, hi_l+1) and writes / reads that dword of the hi run, the odd lane (lo_l-1, lo_l) of the lo run
        // -- 4-byte accesses, a wave covers the 128-byte run of two pixels per instruction.
        const bool odd = l31 & 1;
        const int pair_off = odd ? 64 + (l31 - 1) * 2 : l31 * 2;                   // byte offset of this lane's dword inside the block
        float s_out = 0.0f, inv_sres = 0.0f;
        if (p.out_pairs) {
            const float xmax = __uint_as_float(*p.amax_in);
            const float rmax = p.res ? __uint_as_float(*p.res_bound) : 0.0f;
            const float bound = (xmax * p.wl1 + p.bmax + rmax) * 1.001f;          // >= max |y| whatever the rounding of the products
            s_out = ci_scale(__float_as_uint(bound));
            if (mt == 0 && nt == 0 && tid == 0) *p.bound_out = __float_as_uint(bound);
        }
        if (p.res && p.res_pairs) inv_sres = 1.0f / ci_scale(*p.res_bound);
        // Stores and shortcut loads go through buffer descriptors that begin at the tile's first pixel and end with the tensor: a row
        // beyond the last pixel is out of range -- the load returns zero, the store is dropped -- so the epilogue has no branch (with
        // `if (pixel < P)` around every access the compiler serialised 64 load -> wait -> store sequences per lane: 0.7 ms of layer1's
        // 1.4 ms with a shortcut)
        const int64_t tile0 = (int64_t)mt * TM * p.Cout;                       // the tile's first value (both formats: 4 bytes per value)
        const int64_t rows_left = (int64_t)p.P - (int64_t)mt * TM;
        const int64_t tile_bytes = (rows_left < TM ? rows_left : TM) * p.Cout * 4;
        const __amdgpu_buffer_rsrc_t rsY = ci_rsrc((const char *)(p.y + tile0), tile_bytes);
        const __amdgpu_buffer_rsrc_t rsR = ci_rsrc((const char *)((p.res ? p.res : p.y) + tile0), tile_bytes);
        const float floor_ = p.relu ? 0.0f : -INFINITY;
        // (the operand formats as compile-time constants of six copies of the loop: as runtime flags they left a branch per element)
        auto epilogue = [&](auto res_c, auto outp_c) {
        constexpr int RES = decltype(res_c)::value;                           // shortcut: 0 none, 1 float32, 2 pair format
        constexpr bool OUTP = decltype(outp_c)::value;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int co0 = nt * TN + wn * TNW + n * 32;                      // the tile's 32 channels = one channel block
            const int co = co0 + l31;
            const float bv = p.bias ? p.bias[co] : 0.0f;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float rv[16];
                if (RES) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const int off = RES == 2 ? (row * p.Cout + co0) * 4 + pair_off : (row * p.Cout + co) * 4;
                        rv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsR, off, 0, 0));
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const bool live = row < rows_left;
                    float v = acc[m][n][r] * inv + bv;
                    if (RES) {
                        if (RES == 2) {
                            const unsigned wv = __float_as_uint(rv[r]);
                            const unsigned ov = (unsigned)__builtin_amdgcn_update_dpp(0, (int)wv, 0xB1, 0xF, 0xF, true);   // lane ^ 1
                            const unsigned hb = odd ? ov >> 16 : wv & 0xffffu, lb = odd ? wv >> 16 : ov & 0xffffu;
                            const unsigned short hs = (unsigned short)hb, ls = (unsigned short)lb;
                            v += (__half2float(*(const __half *)&hs) + __half2float(*(const __half *)&ls)) * inv_sres;
                        } else {
                            v += rv[r];
                        }
                    }
                    v = fmaxf(v, floor_);
                    amax = fmaxf(amax, live ? fabsf(v) : 0.0f);
                    if (OUTP) {
                        const float u = v * s_out;
                        const __half hu = __float2half_rn(u);
                        const __half lu = __float2half_rn(u - __half2float(hu));
                        const unsigned mine = (unsigned)*(const unsigned short *)&hu | ((unsigned)*(const unsigned short *)&lu << 16);
                        const unsigned oth = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xF, 0xF, true);
                        const unsigned wv = odd ? (oth >> 16) | (mine & 0xffff0000u) : (mine & 0xffffu) | (oth << 16);
                        __builtin_amdgcn_raw_buffer_store_b32(wv, rsY, (row * p.Cout + co0) * 4 + pair_off, 0, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsY, (row * p.Cout + co) * 4, 0, 0);
                    }
                }
            }
        }
        };
        typedef std::integral_constant<int, 0> R0; typedef std::integral_constant<int, 1> R1; typedef std::integral_constant<int, 2> R2;
        const int res_mode = !p.res ? 0 : (p.res_pairs ? 2 : 1);
        if (p.out_pairs) {
            if (res_mode == 0) epilogue(R0(), std::true_type()); else if (res_mode == 1) epilogue(R1(), std::true_type()); else epilogue(R2(), std::true_type());
        } else {
            if (res_mode == 0) epilogue(R0(), std::false_type()); else if (res_mode == 1) epilogue(R1(), std::false_type()); else epilogue(R2(), std::false_type());
        }
    }
    if (p.amax_out) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
        if (lane == 0 && __float_as_uint(amax) > *(volatile unsigned *)p.amax_out) atomicMax(p.amax_out, __float_as_uint(amax));
    }
}

// ---- The ResNet stem (7x7 / 2, pad 3, 3 -> 64 channels) with its MaxPool2d(3, 2, 1), patch form --------------------------------------
// The implicit-GEMM stem above gathers every K block of every pixel from global memory (sixteen 4-byte loads per thread and kernel row)
// and splits it again: 0.18 of the matrix peak.  Here a persistent workgroup (one per CU, eight waves) keeps the WEIGHTS in LDS for the
// whole launch (57 KB of B fragments in lane order), reads the 21 x 37-pixel image patch of an 8 x 16-pixel output tile ONCE, converts it
// to hi and lo planes once, and the MFMA A fragments are plain 4-byte LDS reads of those planes: pixel (r, c), kernel row kh, K group g
// = 8 consecutive halfs from float 6 c + 8 g of patch row 2 r + kh (the run of 21 floats (kw, ci) the K block is made of; slots 21..31
// are masked to zero).  No per-K-block staging, no barrier inside a tile's K loop.  Same products in the same order as
// conv_igemm_h2_kernel<.., 1>: the results are bit-equal.
// Two GROUPS of four waves share the weights and alternate: while one group runs the K loop of its tile (matrix pipe), the other pools
// and stores its previous tile (LDS and vector memory) -- with four waves alone (one per SIMD) every LDS round trip of the K loop and
// the whole pooling, which is as long as the K loop, were exposed: 3.7 ms against the implicit-GEMM form's 2.55.  A group's wave w: tile
// rows 2 w, 2 w + 1 (32 pixels) x 64 channels; accumulators 2 x 16.
#ifdef CSLAM_ABLATIONS
#define CI_KW_SHARING (!getenv("CSLAM_CI_KWS_OFF"))     // measurement build: A/B partner of the shared-row form
#else
#define CI_KW_SHARING true
#endif
#ifdef CSLAM_ABLATIONS
#define CI_TM256 (getenv("CSLAM_CI_TM256") != nullptr)  // measurement build only: the 256-pixel shared-row tiles (measured slower, below)
#else
#define CI_TM256 false
#endif
#ifdef CSLAM_ABLATIONS
#define CI_STEM_PATCH_FORM (!getenv("CSLAM_SP_OFF"))   // measurement build: the implicit-GEMM pooled form as the A/B partner
#else
#define CI_STEM_PATCH_FORM true
#endif
#define SP_PPH 112                      // halfs per patch row and plane (111 used; a padding group's reads may run into the next row).  112 halfs = 56 dwords:
                                        // a fragment read's two pixel rows (2 patch rows = 112 dwords apart) then fall on complementary banks
#define SP_PLANE (21 * SP_PPH * 2 + 64) // bytes of one plane (+ the overhang of the last row's reads)
#define SP_WBYTES (7 * 2 * 2 * 2 * 64 * 16)
#define SP_GROUP (4 * SP_PLANE + 128 * 64 * 4)      // a group's two patch buffers (hi + lo planes each) and its pooling tile
#define SP_LDS (SP_WBYTES + 2 * SP_GROUP)           // 163 712 of the CU's 163 840 bytes
struct StemPatchArgs {
    const float *x; const char *w2; const float *bias; float *y;
    int B, H, W, Ho, Wo, ntiles, tcols, tpi, n_xcd;
    const unsigned *amax_in; float inv_sw; unsigned *amax_out;
    int dbg;                           // measurement build (CSLAM_SP_DBG; WRONG results, timing): 1 no pooling phase, 2 no K loop, 4 no patch
                                       // split, 8 no patch requests, 16 stores instead of the atomic maxima, 32 neither
};
__global__ __launch_bounds__(512, 1) void conv_stem_pool_patch_kernel(StemPatchArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3, gt = tid & 255;
    const int h = lane >> 5, l31 = lane & 31;
#ifdef CSLAM_ABLATIONS
    const int dbg = p.dbg;
#else
    constexpr int dbg = 0;
#endif
    char *s_w = smem;                                  // [kh 7][s 2][n 2][hi | lo][lane 64] x 16 bytes
    char *s_patch = smem + SP_WBYTES + grp * SP_GROUP; // [buffer 2][hi | lo] planes of this group
    float *T = (float *)(s_patch + 4 * SP_PLANE);
    const float sc = ci_scale(*p.amax_in);
    const float inv = p.inv_sw / sc;
    // ---- weights: global rows [co][kh][hi 32 | lo 32] -> fragment order
    for (int e = tid; e < 7 * 2 * 2 * 2 * 64; e += 512) {
        const int ln = e & 63, hl = (e >> 6) & 1, n = (e >> 7) & 1, ss = (e >> 8) & 1, kh = e >> 9;
        const int co = n * 32 + (ln & 31), hh = ln >> 5;
        *(cu4 *)(s_w + e * 16) = *(const cu4 *)(p.w2 + ((size_t)(co * 7 + kh) * 128 + hl * 64 + (16 * ss + 8 * hh) * 2));
    }
    // ---- tiles: XCD b % n_xcd walks a contiguous range of tiles; (slot b / n_xcd, group) takes every (2 slots)-th of it
    const int xcd = blockIdx.x % p.n_xcd, slot = blockIdx.x / p.n_xcd, nslot = gridDim.x / p.n_xcd;
    const int per = (p.ntiles + p.n_xcd - 1) / p.n_xcd;
    const int t_lo = xcd * per, t_hi = min(p.ntiles, t_lo + per);
    const int step = 2 * nslot;
    const int64_t img = (int64_t)p.H * p.W * 3;
    float raw[10];
    // element pair e2 = gt + 256 j (j < 5) of the patch: row e2 / 56, floats 2 (e2 % 56) and + 1 (float 111 does not exist: zero) -- a
    // thread splits two neighbouring values and writes ONE dword per plane (2-byte stores of neighbouring lanes fall on the same bank:
    // SQ_LDS_BANK_CONFLICT was 22 % of the kernel's LDS cycles)
    auto patch_request = [&](int tile) {               // 21 rows x 112 floats of the tile's patch -> registers (zeros outside the image)
        const int tb = tile / p.tpi, t = tile - tb * p.tpi, tr = t / p.tcols, tc = t - tr * p.tcols;
        const __amdgpu_buffer_rsrc_t rs = ci_rsrc((const char *)(p.x + tb * img), img * 4);
        const int gy0 = tr * 16 - 3, gx0 = (tc * 32 - 3) * 3;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int e2 = gt + j * 256;
            const int pr = e2 / 56, pf = 2 * (e2 - pr * 56);
            const int gy = gy0 + pr, gx = gx0 + pf;
            const bool rowin = e2 < 21 * 56 && (unsigned)gy < (unsigned)p.H;
            const bool in0 = rowin && (unsigned)gx < (unsigned)(p.W * 3), in1 = rowin && pf + 1 < 111 && (unsigned)(gx + 1) < (unsigned)(p.W * 3);
            const int off = (gy * p.W * 3 + gx) * 4;
            raw[2 * j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, in0 ? off : 0x7fffffff, 0, 0));
            raw[2 * j + 1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, in1 ? off + 4 : 0x7fffffff, 0, 0));
        }
    };
    auto patch_store = [&](int buf) {                  // split into the hi and lo planes (the split of conv_igemm's a_store)
        char *hp = s_patch + buf * 2 * SP_PLANE, *lp = hp + SP_PLANE;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int e2 = gt + j * 256;
            if (e2 < 21 * 56) {
                const int pr = e2 / 56, pf = 2 * (e2 - pr * 56);
                const float v0 = raw[2 * j] * sc, v1 = raw[2 * j + 1] * sc;
                const __half2 hv = __floats2half2_rn(v0, v1);
                const __half2 lv = __floats2half2_rn(v0 - __low2float(hv), v1 - __high2float(hv));
                *(__half2 *)(hp + (pr * SP_PPH + pf) * 2) = hv;
                *(__half2 *)(lp + (pr * SP_PPH + pf) * 2) = lv;
            }
        }
    };
    // fragment addresses: pixel (r = 2 w4 + (l31 >> 4), c = l31 & 15); K step s, half h -> group g = 2 s + h
    const int a_base = ((2 * (2 * w4 + (l31 >> 4))) * SP_PPH + 6 * (l31 & 15) + 8 * h) * 2;
    const bool hz = h;                                 // K step 1: group 3 (h = 1) is all padding, group 2 keeps slots 16..20
    float amax = 0.0f;
    float bv[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) bv[n] = p.bias ? p.bias[n * 32 + l31] : 0.0f;
    int tile = t_lo + 2 * slot + grp;
    int ntl = tile < t_hi ? (t_hi - tile + step - 1) / step : 0;           // this group's tiles; the other group's count differs by <= 1
    const int tile_o = t_lo + 2 * slot + (grp ^ 1);
    const int ntl_o = tile_o < t_hi ? (t_hi - tile_o + step - 1) / step : 0;
    const int nphase = 2 * (ntl > ntl_o ? ntl : ntl_o) + 2;
    if (ntl > 0) patch_request(tile);
    __syncthreads();                                   // (the weights)
    if (ntl > 0) patch_store(0);
    __syncthreads();
    int buf = 0, pool_tile = -1;
    // phase ph: the group with (ph & 1) == grp multiplies its next tile, the other pools the tile it multiplied one phase earlier
    for (int ph = 0; ph < nphase; ++ph) {
        if ((ph & 1) == grp) {
            if (tile < t_hi) {
                const int nxt = tile + step;
                if (nxt < t_hi && !(dbg & 8)) patch_request(nxt);
                f32x16 acc[2];
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
                const char *hp = s_patch + buf * 2 * SP_PLANE + a_base, *lp = hp + SP_PLANE;
                // 14 steps (kernel row kh, K step ss); the fragments of step i + 1 are requested before the products of step i (two
                // register sets): with one wave per SIMD in this phase an LDS round trip in front of every three MFMAs is not hidden
                cu4 ah[2], al[2];
                f16x8 fbh[2][2], fbl[2][2];
                auto load_step = [&](int st, int set) {
                    const int kh = st >> 1, ss = st & 1;
                    const int off = (kh * SP_PPH + 16 * ss) * 2;
                    if (ss == 0) {
                        ah[set] = (cu4){*(const unsigned *)(hp + off), *(const unsigned *)(hp + off + 4), *(const unsigned *)(hp + off + 8), *(const unsigned *)(hp + off + 12)};
                        al[set] = (cu4){*(const unsigned *)(lp + off), *(const unsigned *)(lp + off + 4), *(const unsigned *)(lp + off + 8), *(const unsigned *)(lp + off + 12)};
                    } else {
                        const unsigned h0 = *(const unsigned *)(hp + off), h1 = *(const unsigned *)(hp + off + 4), h2 = *(const unsigned *)(hp + off + 8);
                        const unsigned l0 = *(const unsigned *)(lp + off), l1 = *(const unsigned *)(lp + off + 4), l2 = *(const unsigned *)(lp + off + 8);
                        ah[set] = (cu4){hz ? 0u : h0, hz ? 0u : h1, hz ? 0u : (h2 & 0xffffu), 0u};
                        al[set] = (cu4){hz ? 0u : l0, hz ? 0u : l1, hz ? 0u : (l2 & 0xffffu), 0u};
                    }
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const char *wb = s_w + ((((kh * 2 + ss) * 2 + n) * 2) * 64 + lane) * 16;
                        fbh[set][n] = *(const f16x8 *)wb;
                        fbl[set][n] = *(const f16x8 *)(wb + 64 * 16);
                    }
                };
                if (!(dbg & 2)) {
                    load_step(0, 0);
#pragma unroll
                    for (int st = 0; st < 14; ++st) {
                        const int set = st & 1;
                        if (st + 1 < 14) load_step(st + 1, set ^ 1);
                        const f16x8 fah = *(const f16x8 *)&ah[set], fal = *(const f16x8 *)&al[set];
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, fbh[set][n], acc[n], 0, 0, 0);
                            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal, fbh[set][n], acc[n], 0, 0, 0);
                            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, fbl[set][n], acc[n], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (nxt < t_hi && !(dbg & 4)) patch_store(buf ^ 1);
                // ReLU(conv + bias) -> T [pixel 128][channel 64]: pooled one phase later
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = w4 * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const float v = fmaxf(acc[n][r] * inv + bv[n], 0.0f);
                        T[row * 64 + n * 32 + l31] = v;
                        amax = fmaxf(amax, v);
                    }
                pool_tile = tile;
                tile = nxt;
                buf ^= 1;
            }
        } else if (pool_tile >= 0 && !(dbg & 1)) {
            // the pooling of conv_igemm_h2_kernel's pooled form, on this group's 256 threads
            const int tb = pool_tile / p.tpi, t = pool_tile - tb * p.tpi, tr = t / p.tcols, tc = t - tr * p.tcols;
            const int Hp = p.Ho >> 1, Wp = p.Wo >> 1;
            // 45 pooled positions x 64 channels on 256 threads: a wave takes positions w4, w4 + 4, ... (wave-uniform: scalar index
            // arithmetic), the lane is the channel; unrolled so that the reads of several positions are in flight together
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int pp = w4 + 4 * k, ch = lane;
                if (pp >= 45) break;
                const int pr = pp / 9, pc = pp - pr * 9;
                const int gp = tr * 4 + pr, gq = tc * 8 + pc;
                if (gp >= Hp || gq >= Wp) continue;
                // nine unconditional reads: a window row / column outside the tile is clamped onto one inside (a duplicate does not change
                // a maximum) -- with `continue` around them every read sat in a block of its own behind its own LDS round trip
                float mx = 0.0f;
#pragma unroll
                for (int dr = -1; dr <= 1; ++dr) {
                    const int lr = min(max(2 * pr + dr, 0), 7);
#pragma unroll
                    for (int dc = -1; dc <= 1; ++dc) {
                        const int lc = min(max(2 * pc + dc, 0), 15);
                        mx = fmaxf(mx, T[(lr * 16 + lc) * 64 + ch]);
                    }
                }
                float *dst = p.y + (((int64_t)tb * Hp + gp) * Wp + gq) * 64 + ch;
                const bool whole = (pr >= 1 || tr == 0) && pr <= 3 && (pc >= 1 || tc == 0) && pc <= 7;
                if (whole || (dbg & 16)) *dst = mx;
                else if (!(dbg & 32)) atomicMax((unsigned *)dst, __float_as_uint(mx));
            }
            pool_tile = -1;
        }
        __syncthreads();
    }
    if (p.amax_out) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
        if (lane == 0 && __float_as_uint(amax) > *(volatile unsigned *)p.amax_out) atomicMax(p.amax_out, __float_as_uint(amax));
    }
}

// zero fill of the pooled map's positions that the atomic maxima complete: pooled rows 4 k and pooled columns 8 k (the windows that reach
// into the tile above / to the left); every other position is written by a plain store.  34 % of the map instead of all of it.
__global__ __launch_bounds__(256) void ci_zero_kernel(float4 *__restrict__ y, int Hp, int Wp, int c4) {
    const int64_t row = blockIdx.x;                                    // (image, pooled row): one workgroup per row of the pooled map
    const bool all = ((int)(row % Hp) & 3) == 0;
    float4 *r = y + row * Wp * c4;
    for (int i = threadIdx.x; i < Wp * c4; i += 256)
        if (all || ((i / c4) & 7) == 0) r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

/* y = act(conv(x, w) + bias (+ res)) for x [B,H,W,Cin] NHWC float32 -> y [B,Ho,Wo,Cout] NHWC float32, Ho = (H + 2 pad - KH) / stride + 1.
 * d_w2: the weights as fp16 pairs, `igemm_pair_weights(weight)` of cslam_amd/vpr/winograd.py: rows = output channels, every K block of
 * 32 = [hi 32 | lo 32]; general form (Cin a multiple of 32) K blocks in (kh, kw, Cin / 32) order; stem form (Cin = 3, 3 KW <= 32) one
 * block per kernel row, slot kw * 3 + c.  inv_sw = 1 / (the power of two the weights were scaled by).  d_amax_in: 4-byte device slot,
 * float bits of (a bound of) max |x|, > 0; d_amax_out (optional): slot that receives max |y| (atomic maximum: zero it first).
 * Replaces torch.nn.functional.conv2d for the layers named at the top of this file (cslam/vpr/cosplace_utils/network.py:38-68). */
struct ConvIgemmFormats {               // pair-format operands of a launch (all off: float32 everywhere)
    int x_pairs; const unsigned *xbound;
    int res_pairs; const unsigned *res_bound;
    int out_pairs; float wl1, bmax; unsigned *bound_out;
};
static int conv_igemm_launch(const float *d_x, const void *d_w2, const float *d_bias, const float *d_res, int B, int H, int W,
                             int Cin, int Cout, int KH, int KW, int stride, int pad, int relu, int pool, const unsigned *d_amax_in,
                             float inv_sw, unsigned *d_amax_out, float *d_y, void *stream, const ConvIgemmFormats &f) {
    PTR_DEVICE(d_x);
    ARG_CHECK(d_x && d_w2 && d_y && d_amax_in, "NULL argument");
    ARG_CHECK(B >= 1 && H >= 1 && W >= 1 && KH >= 1 && KW >= 1 && stride >= 1 && pad >= 0, "bad geometry");
    const bool stem = Cin == 3;
    ARG_CHECK(stem ? (3 * KW <= 32) : (Cin % 32 == 0 && Cin >= 32), "Cin must be a multiple of 32, or 3 with a kernel row of at most 10 taps");
    ARG_CHECK(Cout % 64 == 0 && Cout >= 64, "Cout must be a multiple of 64");
    ARG_CHECK(H + 2 * pad >= KH && W + 2 * pad >= KW, "kernel larger than the padded map");
    ConvIgemmArgs a;
    a.x = d_x; a.w2 = (const char *)d_w2; a.bias = d_bias; a.res = d_res; a.y = d_y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
    a.Ho = (H + 2 * pad - KH) / stride + 1; a.Wo = (W + 2 * pad - KW) / stride + 1;
    const int64_t P = (int64_t)B * a.Ho * a.Wo;
    ARG_CHECK(P < (1ll << 31) && P * Cout < (1ll << 40), "too many output pixels for one launch");
    a.P = (int)P;
    a.pool = pool;
    if (pool) {
        ARG_CHECK(stem && relu && !d_res, "the fused MaxPool2d(3, 2, 1) exists for the stem form with ReLU and without a shortcut");
        ARG_CHECK(a.Ho % 8 == 0 && a.Wo % 16 == 0, "the fused MaxPool2d needs an output map of 8 x 16-pixel tiles");
    }
    // the activation descriptor starts at the image of a tile's first pixel: the tile's (at most 256) pixels then span this many images
    ARG_CHECK((int64_t)(256 / (a.Ho * a.Wo) + 2) * H * W * Cin * 4 < (stem ? CI_OOB - 1 : 0x7fffffffll), "image too large for 32-bit activation offsets");
    a.ncb = stem ? 0 : Cin / 32;
    a.nk = stem ? KH : KH * KW * (Cin / 32);
    a.relu = relu; a.amax_in = d_amax_in; a.inv_sw = inv_sw; a.amax_out = d_amax_out;
    a.xbound = f.xbound; a.out_pairs = f.out_pairs; a.wl1 = f.wl1; a.bmax = f.bmax; a.bound_out = f.bound_out;
    a.res_pairs = f.res_pairs; a.res_bound = f.res_bound;
    a.dbg = 0;
#ifdef CSLAM_ABLATIONS
    if (const char *e = getenv("CSLAM_CI_DBG")) a.dbg = atoi(e);
#endif
    ARG_CHECK(!f.x_pairs || (!stem && f.xbound), "pair-format input needs its bound slot (and is not the stem's format)");
    ARG_CHECK(!f.out_pairs || (!pool && f.bound_out && f.wl1 >= 0.0f && f.bmax >= 0.0f), "pair-format output needs wl1, bmax and the bound slot");     // (an all-zero folded convolution has wl1 = 0: the bound bmax + max|res| still holds)
    ARG_CHECK(!(d_res && (f.res_pairs || f.out_pairs)) || f.res_bound, "the shortcut's bound slot is missing");
    const int tn = Cout % 128 == 0 ? 128 : 64;
    const int am = stem ? 1 : (f.x_pairs ? 2 : 0);
    // three taps, one activation block: the pair-format layers, and the float32-input layer that opens the pair-format chain (the plain
    // float32 entry keeps its K order: its tests pin that rounding)
    const bool kws = CI_KW_SHARING && (am == 2 || (am == 0 && f.out_pairs)) && KH == 3 && KW == 3 && stride == 1 && pad == 1;
    // 256-pixel tiles: 64-channel layers without shared rows (the shared-row 128 x 64 form, three per CU, is +0.7 % over it).  For the
    // pair-format 128-channel-tile layers with shared rows (one workgroup per CU, 8 accumulator tiles per wave, 0.66 of the L2 -> LDS
    // bytes per product) they were measured and lost: layer2 0.73 against 0.69 ms, layer3 0.67 / 0.63, layer4 0.62 / 0.60, C2 75k
    // against 82k keyframes/s (profiles/r06_f_igemm_tm256_rejected.log) -- a lone workgroup has nobody to overlap its request phase
    // with.  The form stays in the measurement build (CSLAM_CI_TM256=1).
    const bool big = kws && am == 2 && tn == 128 && CI_TM256 && P >= 256ll * 64;
    const int tm = big || (am == 2 && tn == 64 && !kws) ? 256 : 128;
    a.n_mt = (int)ceil_div64(P, tm); a.ntb = Cout / tn;                                      // (pooled form: P / 128 tiles exactly)
    a.nx = cslam_cu_count() % 8 == 0 ? 8 : 1;
    a.mt_per_x = (int)ceil_div64(a.n_mt, a.nx);
    const dim3 grid((unsigned)((int64_t)a.mt_per_x * a.nx * a.ntb)), blk(256);
    const int lds = kws ? 2 * (tm + 8) * CI_ROWB + 2 * tn * CI_ROWB : 2 * (tm + tn) * CI_ROWB;   // >= 128 x tn floats, the pooled form's tile
    hipStream_t st = (hipStream_t)stream;
    const bool patch_form = CI_STEM_PATCH_FORM && pool && Cout == 64 && KH == 7 && KW == 7 && stride == 2 && pad == 3 && (int64_t)H * W * 12 < 0x7fffffffll;
    if (pool) {
        // (a kernel, not hipMemsetAsync: as a memset NODE of a captured graph the fill did not precede the convolution on replay --
        // tests/test_heads_gpu.py::test_online_hip_graph_replay_equals_plain_launches caught the second replay 3e-2 off)
        ARG_CHECK((int64_t)B * (a.Ho / 2) < (1ll << 31), "too many pooled rows for one launch");
        hipLaunchKernelGGL(ci_zero_kernel, dim3((unsigned)(B * (a.Ho / 2))), dim3(256), 0, st, (float4 *)d_y, a.Ho / 2, a.Wo / 2, Cout / 4);
    }
    if (patch_form) {
        StemPatchArgs sp;
        sp.x = d_x; sp.w2 = (const char *)d_w2; sp.bias = d_bias; sp.y = d_y;
        sp.B = B; sp.H = H; sp.W = W; sp.Ho = a.Ho; sp.Wo = a.Wo;
        sp.tcols = a.Wo / 16; sp.tpi = (a.Ho / 8) * sp.tcols; sp.ntiles = B * sp.tpi;
        sp.amax_in = d_amax_in; sp.inv_sw = inv_sw; sp.amax_out = d_amax_out;
        sp.dbg = 0;
#ifdef CSLAM_ABLATIONS
        if (const char *e = getenv("CSLAM_SP_DBG")) sp.dbg = atoi(e);
#endif
        const int n_cu = cslam_cu_count();
        ARG_CHECK(n_cu > 0, "no HIP device");
        sp.n_xcd = n_cu % 8 == 0 ? 8 : 1;
        int nb = (sp.ntiles + 1) / 2 < n_cu ? (sp.ntiles + 1) / 2 : n_cu;       // a workgroup = two groups of four waves = two tiles at a time
        if (nb >= sp.n_xcd) nb -= nb % sp.n_xcd; else sp.n_xcd = 1;
        static DeviceOnce once_sp; int once_dev_sp;
        if (once_sp.todo(&once_dev_sp)) {
            HIP_TRY(hipFuncSetAttribute((const void *)conv_stem_pool_patch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS));
            once_sp.done(once_dev_sp); }
        hipLaunchKernelGGL(conv_stem_pool_patch_kernel, dim3(nb), dim3(512), SP_LDS, st, sp);
        HIP_TRY(hipGetLastError());
        return CSLAM_OK;
    }
#define CI_LAUNCH(TM_, TN_, ST_) do { \
        static DeviceOnce once; int once_dev; \
        if (once.todo(&once_dev)) { \
            HIP_TRY(hipFuncSetAttribute((const void *)conv_igemm_h2_kernel<TM_, TN_, ST_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
            once.done(once_dev); } \
        hipLaunchKernelGGL((conv_igemm_h2_kernel<TM_, TN_, ST_>), grid, blk, lds, st, a); } while (0)
#define CI_LAUNCH_K(TM_, TN_, AM_) do { \
        static DeviceOnce once; int once_dev; \
        if (once.todo(&once_dev)) { \
            HIP_TRY(hipFuncSetAttribute((const void *)conv_igemm_h2_kernel<TM_, TN_, AM_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
            once.done(once_dev); } \
        hipLaunchKernelGGL((conv_igemm_h2_kernel<TM_, TN_, AM_, true>), grid, blk, lds, st, a); } while (0)
    if (stem) { if (tn == 128) CI_LAUNCH(128, 128, 1); else CI_LAUNCH(128, 64, 1); }
#ifdef CSLAM_ABLATIONS
    else if (f.x_pairs && kws && big) CI_LAUNCH_K(256, 128, 2);
#endif
    else if (f.x_pairs && kws) { if (tn == 128) CI_LAUNCH_K(128, 128, 2); else CI_LAUNCH_K(128, 64, 2); }
    else if (kws) { if (tn == 128) CI_LAUNCH_K(128, 128, 0); else CI_LAUNCH_K(128, 64, 0); }
    else if (f.x_pairs) { if (tn == 128) CI_LAUNCH(128, 128, 2); else CI_LAUNCH(256, 64, 2); }
    else { if (tn == 128) CI_LAUNCH(128, 128, 0); else CI_LAUNCH(128, 64, 0); }
#undef CI_LAUNCH
#undef CI_LAUNCH_K
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}
CSLAM_API int cslam_conv_igemm_h2_dev(const float *d_x, const void *d_w2, const float *d_bias, const float *d_res, int B, int H, int W,
                                      int Cin, int Cout, int KH, int KW, int stride, int pad, int relu, const unsigned *d_amax_in,
                                      float inv_sw, unsigned *d_amax_out, float *d_y, void *stream) {
    return conv_igemm_launch(d_x, d_w2, d_bias, d_res, B, H, W, Cin, Cout, KH, KW, stride, pad, relu, 0, d_amax_in, inv_sw, d_amax_out,
                             d_y, stream, ConvIgemmFormats{0, nullptr, 0, nullptr, 0, 0.0f, 0.0f, nullptr});
}
/* The same convolution between PAIR-FORMAT activations: a tensor [B,H,W,C] (C a multiple of 32) whose every (pixel, 32-channel block) is
 * 128 bytes [hi 32 | lo 32] fp16 of s x -- s = the power of two that brings the tensor's 4-byte BOUND slot into [2^13, 2^14) -- the
 * row image the kernel's LDS stages hold, written once by the producing layer's epilogue instead of being re-derived per tap and
 * output tile by every consumer (same bytes as float32).  x_pairs: d_x is such a tensor and d_xbound its slot (d_amax_in then is the
 * MEASURED max |x| the producer left); res_pairs: d_res likewise with d_res_bound (float32 d_res with out_pairs: d_res_bound = a bound
 * of max |res|); out_pairs: d_y leaves in the format, scaled for the bound max|x| wl1 + bmax (+ max|res|) -- wl1 = max over output
 * channels of sum |w|, bmax = max |bias|, both of the float32 weights -- which goes to d_bound_out.  Any flag may be 0 (float32). */
CSLAM_API int cslam_conv_igemm_h2p_dev(const void *d_x, int x_pairs, const unsigned *d_xbound, const void *d_w2, const float *d_bias,
                                       const void *d_res, int res_pairs, const unsigned *d_res_bound, int B, int H, int W, int Cin,
                                       int Cout, int KH, int KW, int stride, int pad, int relu, const unsigned *d_amax_in, float inv_sw,
                                       float wl1, float bmax, unsigned *d_amax_out, int out_pairs, unsigned *d_bound_out, void *d_y,
                                       void *stream) {
    ARG_CHECK(Cin != 3, "the stem reads float32 frames: cslam_conv_igemm_h2_dev / cslam_conv_stem_pool_igemm_h2_dev");
    return conv_igemm_launch((const float *)d_x, d_w2, d_bias, (const float *)d_res, B, H, W, Cin, Cout, KH, KW, stride, pad, relu, 0,
                             d_amax_in, inv_sw, d_amax_out, (float *)d_y, stream,
                             ConvIgemmFormats{x_pairs, d_xbound, res_pairs, d_res_bound, out_pairs, wl1, bmax, d_bound_out});
}
/* The stem with its pooling: y = MaxPool2d(3, 2, 1)(ReLU(conv(x, w) + bias)) for x [B,H,W,3] -> y [B,Ho/2,Wo/2,Cout] (Ho a multiple of 8,
 * Wo of 16: ResNet's 7x7 / 2 stem on 224 x 224 frames gives 112 x 112).  The un-pooled map never exists in HBM; d_amax_out receives max
 * of the UN-pooled map (a bound of the pooled one).  ResNet's own geometry (7x7, stride 2, pad 3, 64 channels) runs the persistent
 * patch-form kernel (conv_stem_pool_patch_kernel), any other one the implicit-GEMM form with 8 x 16-pixel tiles: the same products in the
 * same order.  conv1 + bn1 + relu + maxpool of the torchvision ResNet trunk cslam/vpr/cosplace_utils/network.py:38-68 builds. */
CSLAM_API int cslam_conv_stem_pool_igemm_h2_dev(const float *d_x, const void *d_w2, const float *d_bias, int B, int H, int W, int Cout,
                                                int KH, int KW, int stride, int pad, const unsigned *d_amax_in, float inv_sw,
                                                unsigned *d_amax_out, float *d_y, void *stream) {
    return conv_igemm_launch(d_x, d_w2, d_bias, nullptr, B, H, W, 3, Cout, KH, KW, stride, pad, 1, 1, d_amax_in, inv_sw, d_amax_out,
                             d_y, stream, ConvIgemmFormats{0, nullptr, 0, nullptr, 0, 0.0f, 0.0f, nullptr});
}
