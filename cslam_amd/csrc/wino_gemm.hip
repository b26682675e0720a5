// wino_gemm.hip -- the 36 per-frequency products of the F(4x4, 3x3) Winograd convolution as ONE persistent kernel on the
// fp16 matrix pipe with fp32-grade results (gfx950), and the input transform that feeds it.
//
// Replaces, for the wide layers of the VGG-16 trunk (conv2_2 ... conv5_3, cslam/vpr/netvlad.py:163-171,227 through
// torchvision's Conv2d), the library GEMM between wino4_input_* and wino4_output_* (winograd.hip):
//     M[xi] = V[xi] U[xi],  xi = 0..35,  V[xi] [T tiles, Cin],  U[xi] [Cin, Cout],  M[xi] [T, Cout] float32.
// A float times a power of two splits exactly into an fp16 pair hi + lo (11 + 11 significant bits); fp16 x fp16 products
// are exact in the MFMA's fp32 accumulator, so  V U = vh uh + vl uh + vh ul  (the dropped vl ul is 2^-22 of the product)
// is an fp32-grade product at the fp16 MFMA rate.  Round 1 ran this as one library GEMM over K' = 3 Cin with operands
// [vh | vl | vh] x [uh ; uh ; ul]: V was written and read at 1.5x its fp32 size.  Here the operands are stored ONCE as
// pairs -- 4 bytes per value, the fp32 size -- and every fragment read from LDS feeds two of the three products:
//     acc += vh uh;  acc += vl uh;  acc += vh ul        (v_mfma_f32_32x32x16_f16, fp32 accumulate)
//
// Layout (both operands, K-major rows):  row r of V2[xi] / U2[xi] = Cin/32 blocks of 128 bytes, block kb =
// [hi of channels 32 kb .. 32 kb + 31 | lo of the same channels] as fp16.  V2 rows are tiles, U2 rows are OUTPUT channels
// (U transposed), so one loader, one LDS image and one fragment addressing serve both.  A K stage = one block of every
// row of the workgroup tile = [256 | TN rows][128 B], moved by global_load_lds_dwordx4 into an XOR-swizzled image (the
// 16-byte chunk index ^ (row >> 1 & 7), on the per-lane SOURCE address and on the read: conflict-free ds_read_b128, as in
// sim_topk_mfma.hip), double-buffered, the next stage's loads issued between this stage's MFMAs.
//
// Work: one item = (xi, 256-tile row block, TN-channel column block); a persistent workgroup (8 waves as 2 x 4, wave tile
// 128 x TN/4 = 4 x {1,2} MFMA tiles) walks its share of the item list with the K loops of consecutive items fused, so the
// first stage of the next item lands while this item's 256 x TN results are stored.  Items are xi-major and every XCD
// owns a contiguous run of them: the U2[xi] it needs (<= 1 MB) stays in its L2, V2 is streamed once per column block.
// (Measured and rejected: the weight fragments loaded from L2 straight into registers with V2 alone in LDS -- 0.94 vs 0.73 ms
// on conv4_2, profiles/r02_v20_perf_wino_gemm_registers_rejected.log; LDS bandwidth is not what bounds the 512-channel layers.)
// Bound: HBM for Cin <= 256 (V2 in + M out = 4 (Cin + Cout) bytes per tile row and frequency against 6 Cin Cout flop),
// about balanced at 512 x 512.
#include "common.h"
#include <type_traits>
#include <hip/hip_fp16.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float wf4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned wu2 __attribute__((ext_vector_type(2)));

#define WG_TM 256                      // tile rows per workgroup
#define WG_ROWB 128                    // bytes of one K block of one row: 32 hi + 32 lo halfs

__device__ __forceinline__ void wg_glds16(const char *g, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
// LDS-DMA in the MUBUF encoding (round 4; sim_topk_pair.hip has the why: hipcc counts LDS reads again, the K offset travels in an
// SGPR, the lane's offset is one register, rows beyond `bytes` read as zero)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_rsrc(const char *base, int64_t bytes) {
    const uint64_t a = (uint64_t)base;
    const uint64_t u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)a);
    const int n = __builtin_amdgcn_readfirstlane((int)(bytes > 0x7fffffff ? 0x7fffffff : bytes));
    return __builtin_amdgcn_make_buffer_rsrc((void *)u, 0, n, 0x00020000);
}
__device__ __forceinline__ void wg_blds16(__amdgpu_buffer_rsrc_t rs, int voff, int soff, char *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)lds_wave_base, 16, voff, soff, 0, 0);
}

struct WinoGemmArgs {
    const char *V2;        // [36][T][Cin/32][128 B]
    const char *U2;        // [36][Cout][Cin/32][128 B]
    float *M;              // [36][T][Cout]
    int T, Cin, Cout;
    int nk;                // Cin / 32
    int n_mt, n_nt;        // row blocks = ceil(T / 256), column blocks = Cout / TN
    int n_items;           // nxi * n_mt * n_nt
    int nxi;               // independent products in the launch: 36 Winograd frequencies, or the K splits of a projection
};

// s_waitcnt with only the vector-memory counter: gfx9 simm16 = vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14
#define WG_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | (7 << 4) | (15 << 8))

// TM x TN = workgroup tile (tile rows x output channels), 8 waves as 2 x 4, wave tile TM/2 x TN/4 = MT x NT MFMA tiles.
// NS = LDS stages: 2 = double buffer (the loads of stage t+1 are issued during stage t and drained at its end);
// 3 = ring, loads two stages ahead, `s_waitcnt vmcnt(L)` (L = loads per thread per stage) leaves the newest stage in
// flight across the raw s_barrier -- the loads of a stage then have a whole stage of MFMAs to land.
// A finished item's results are stored right AFTER the barrier that ends its last stage, so the stores drain under the
// next stage's MFMAs instead of in front of a wait.
// DBG: timing-only ablations (1 = no stores, 2 = no loads after the prologue), CSLAM_WGEMM_DBG; never the product path.
template <int TM, int TN, int NS, int DBG, int WM>
__global__ __launch_bounds__(512, 2) void wino_gemm_h2_kernel(WinoGemmArgs p) {
    // the eight waves as WM x WN over the tile: wave tile 32 MT x 32 NT (WM = 2: 128 x 32 at 256 x 128 -- ten 16-byte fragment
    // reads per twelve MFMAs; WM = 4: 64 x 64 -- eight)
    constexpr int WN = 8 / WM;
    constexpr int MT = TM / (32 * WM), NT = TN / (32 * WN);
    constexpr int OPA = TM * WG_ROWB, OPB = TN * WG_ROWB;
    constexpr int STAGE = OPA + OPB;
    constexpr int NLA = TM * 8 / 512, NLB = TN * 8 / 512;   // 16-byte chunks per thread per stage
    constexpr int L = NLA + NLB;
    static_assert(NLA >= 2 && NLB >= 2 && (NS == 2 || NS == 3), "tile / ring shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5, l31 = lane & 31;

    // this workgroup's items: XCD x (= blockIdx % 8) owns the contiguous run [x I / 8, (x + 1) I / 8) of the xi-major list,
    // its workgroups take them round-robin
    const int xcd = blockIdx.x & 7, j8 = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int q8 = p.n_items >> 3, r8 = p.n_items & 7;
    const int x_beg = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int x_cnt = q8 + (xcd < r8 ? 1 : 0);
    const int n_mine = j8 < x_cnt ? (x_cnt - j8 + per_xcd - 1) / per_xcd : 0;
    if (n_mine == 0) return;

    const int pitch = p.nk * WG_ROWB;                  // bytes per row (both operands)
    // loader geometry: chunk pch = i * 512 + tid -> row pch >> 3, physical slot pch & 7 holding logical chunk slot ^ swz(row):
    // the lane's byte offset inside the item's row block of either operand
    int voffA[NLA], voffB[NLB];
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
        const int pch = i * 512 + tid, r = pch >> 3, slot = pch & 7;
        voffA[i] = r * pitch + ((slot ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
        const int pch = i * 512 + tid, r = pch >> 3, slot = pch & 7;
        voffB[i] = r * pitch + ((slot ^ ((r >> 1) & 7)) << 4);
    }
    const int wave_chunk = wave * 1024;

    struct Item { int xi, mt, nt; };
    auto decode = [&](int k) {
        const int it = x_beg + j8 + k * per_xcd;
        Item c;
        c.nt = it % p.n_nt;
        const int rest = it / p.n_nt;
        c.mt = rest % p.n_mt;
        c.xi = rest / p.n_mt;
        return c;
    };
    // buffer resources over the row blocks of the item the loader is in (rebuilt only at item boundaries, on the scalar unit);
    // tile rows beyond T read as zero and are never stored
    __amdgpu_buffer_rsrc_t rsA, rsB;
    auto point_at = [&](const Item &c) {
        int64_t rows = (int64_t)p.T - (int64_t)c.mt * TM;
        if (rows > TM) rows = TM;
        rsA = wg_rsrc(p.V2 + ((int64_t)c.xi * p.T + (int64_t)c.mt * TM) * pitch, rows * pitch);
        rsB = wg_rsrc(p.U2 + ((int64_t)c.xi * p.Cout + (int64_t)c.nt * TN) * pitch, (int64_t)TN * pitch);
    };
    auto load_part_a = [&](int stage, int kt, int i) {
        wg_blds16(rsA, voffA[i], kt * WG_ROWB, smem + stage * STAGE + i * (512 * 16) + wave_chunk);
    };
    auto load_part_b = [&](int stage, int kt, int i) {
        wg_blds16(rsB, voffB[i], kt * WG_ROWB, smem + stage * STAGE + OPA + i * (512 * 16) + wave_chunk);
    };

    // fragment read offsets: row * 128 + ((chunk) ^ swz) * 16, chunk = 4 * lo + 2 * s + h for K step s of the stage
    const int swz = (lane >> 1) & 7;
    int foff[2][2];                                    // [s][lo]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int lo = 0; lo < 2; ++lo) foff[s][lo] = ((4 * lo + 2 * s + h) ^ swz) << 4;
    const int arow0 = (wm * (32 * MT) + l31) * WG_ROWB;
    const int brow0 = (wn * (32 * NT) + l31) * WG_ROWB;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    // the loader runs NS - 1 stages ahead of the MFMAs: (l_item, l_kt) = the next stage to fetch
    const int total = n_mine * p.nk;
    int l_item = 0, l_kt = 0, l_stage = 0, fetched = 0;
    Item cur_item = decode(0);
    point_at(cur_item);
    auto advance_loader = [&]() {
        ++fetched;
        l_stage = l_stage + 1 == NS ? 0 : l_stage + 1;
        if (++l_kt == p.nk) {
            l_kt = 0; ++l_item;
            if (fetched < total) point_at(decode(l_item));
        }
    };
#pragma unroll 1
    for (int pre = 0; pre < NS - 1; ++pre) {
        if (fetched < total) {
#pragma unroll
            for (int i = 0; i < NLA; ++i) load_part_a(l_stage, l_kt, i);
#pragma unroll
            for (int i = 0; i < NLB; ++i) load_part_b(l_stage, l_kt, i);
            advance_loader();
        }
    }
    if (NS == 2 || total == 1) __builtin_amdgcn_s_waitcnt(0); else WG_VMCNT(L);   // stage 0 landed
    __syncthreads();

    int k_item = 0, kt = 0, cur = 0;
    bool store_pending = false;
    Item st_item = cur_item;
    for (int it = 0; it < total; ++it) {
        if (store_pending) {
            // ---- the item finished in the previous stage: store the 32 MT x 32 NT wave tile (lane = column, 128-byte runs per
            // row) and clear; issued behind the barrier, drains under this stage's MFMAs
            const int row_base = st_item.mt * TM + wm * (32 * MT) + 4 * h;
            const int col = st_item.nt * TN + wn * (32 * NT) + l31;
            float *mo = p.M + ((int64_t)st_item.xi * p.T + row_base) * p.Cout + col;
            const bool full = st_item.mt * TM + wm * (32 * MT) + 32 * MT <= p.T;    // wave-uniform: no per-row test
            const bool st_on = DBG != 1 || p.T < 0;                               // DBG 1: stores compiled, never executed
            if (full) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            if (st_on) __builtin_nontemporal_store(acc[m][n][r], mo + (int64_t)(m * 32 + (r & 3) + 8 * (r >> 2)) * p.Cout + 32 * n);
            } else {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ro = m * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            if (st_on && row_base + ro < p.T) __builtin_nontemporal_store(acc[m][n][r], mo + (int64_t)ro * p.Cout + 32 * n);
                    }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
            store_pending = false;
        }
        const bool fetch = (DBG != 2) && fetched < total;    // wave-uniform
        const char *sA = smem + cur * STAGE;
        const char *sB = sA + OPA;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                ah[m] = *(const f16x8 *)(sA + arow0 + m * 32 * WG_ROWB + foff[s][0]);
                al[m] = *(const f16x8 *)(sA + arow0 + m * 32 * WG_ROWB + foff[s][1]);
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                bh[n] = *(const f16x8 *)(sB + brow0 + n * 32 * WG_ROWB + foff[s][0]);
                bl[n] = *(const f16x8 *)(sB + brow0 + n * 32 * WG_ROWB + foff[s][1]);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[n], acc[m][n], 0, 0, 0);
            // half of the fetched stage's LDS-DMA per K step, issued between this step's MFMAs (sim_topk_mfma.hip: a
            // global_load_lds costs 60-180 issue cycles; back to back after the barrier they idle the matrix pipe)
            if (fetch) {
#pragma unroll
                for (int i = s * (NLA / 2); i < (s + 1) * (NLA / 2); ++i) load_part_a(l_stage, l_kt, i);
#pragma unroll
                for (int i = s * (NLB / 2); i < (s + 1) * (NLB / 2); ++i) load_part_b(l_stage, l_kt, i);
            }
        }
        if (fetch) advance_loader();

        if (kt == p.nk - 1) { store_pending = true; st_item = cur_item; }
        // the next stage must have landed: with the double buffer that is everything; with the ring only the loads issued
        // during THIS stage may stay in flight (loads return in order; stores issued at the top of the stage are older)
        if (NS == 2 || !fetch) __builtin_amdgcn_s_waitcnt(0); else WG_VMCNT(L);
        __builtin_amdgcn_s_barrier();
        cur = cur + 1 == NS ? 0 : cur + 1;
        if (++kt == p.nk) { kt = 0; ++k_item; if (it + 1 < total) cur_item = decode(k_item); }
    }
    if (store_pending) {
        const int row_base = st_item.mt * TM + wm * (32 * MT) + 4 * h;
        const int col = st_item.nt * TN + wn * (32 * NT) + l31;
        float *mo = p.M + ((int64_t)st_item.xi * p.T + row_base) * p.Cout + col;
        const bool st_on = DBG != 1 || p.T < 0;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = m * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    if (st_on && row_base + ro < p.T) __builtin_nontemporal_store(acc[m][n][r], mo + (int64_t)ro * p.Cout + 32 * n);
            }
    }
}

// ---- the same products on FOUR waves of 512 registers ("big" form, round 4; CSLAM_WGEMM_CFG=6; the default for Cin >= 512) -------
// What bounds the eight-wave kernel above is LDS volume: with 64 accumulator registers per lane a wave reads 8 fragments (8 KB) per
// 12 MFMAs -- per stage and CU 128 KB of reads + 48 KB of DMA writes = 1400 LDS cycles under 1536 matrix cycles, and the counters
// show the two units busy one after the other.  One wave per SIMD may use the whole register file: wave tile 128 x 128 = 256
// accumulators (AGPRs), 16 fragments per 48 MFMAs -- half the LDS bytes per MFMA (PMC: SQ_LDS_IDX_ACTIVE halves, matrix pipe busy
// 0.53 -> 0.62 of its cycles on conv4_2).  With nobody else on the SIMD to cover a wave's waits everything is software-pipelined
// inside the wave: fragments in two register sets (K step 1 read under the MFMAs of K step 0, the stage's last K step multiplied
// BEHIND the barrier), the next stage's 16 LDS-DMA requests spread over the first half of the stage, an item's first K step starts
// its accumulators from the constant 0.  256 x 256 tiles, double-buffered 64 KB stages; Cout a multiple of 256.
// Measured and not kept: stages of ONE K step in a ring of four (requests three stages ahead): the 256 stores of a tile cannot be
// counted in a 6-bit vmcnt, requests return in order, so whatever is requested behind the stores is confirmed only when they have
// drained -- with twice the barriers it came out 7 % slower than this form (profiles/r04_v37_perf_wino_gemm_big.log).
template <int DBG>
__global__ __launch_bounds__(256, 1) void wino_gemm_h2_big_kernel(WinoGemmArgs p) {
    constexpr int TM = 256, TN = 256, MT = 4, NT = 4;
    constexpr int OPA = TM * WG_ROWB, OPB = TN * WG_ROWB;
    constexpr int STAGE = OPA + OPB;
    constexpr int NLA = TM * 8 / 256, NLB = TN * 8 / 256;   // 8 + 8 sixteen-byte chunks per thread per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;

    const int xcd = blockIdx.x & 7, j8 = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int q8 = p.n_items >> 3, r8 = p.n_items & 7;
    const int x_beg = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int x_cnt = q8 + (xcd < r8 ? 1 : 0);
    const int n_mine = j8 < x_cnt ? (x_cnt - j8 + per_xcd - 1) / per_xcd : 0;
    if (n_mine == 0) return;

    const int pitch = p.nk * WG_ROWB;
    // chunk pch = i * 256 + tid -> row pch >> 3, physical slot pch & 7 holding logical chunk slot ^ swz(row); both operands alike
    // Part i of a stage is 32 rows below part 0 (the swizzle term repeats every 16 rows): ONE per-lane offset.  The V2 operand's
    // row step is added per request (the buffer's range check, which drops the rows beyond T, sees voffset only); the U2 operand's
    // tile is always whole, its row step rides in the scalar offset.  (Eight per-lane offsets held across the K loop were what
    // pushed this 256 + 256 register kernel to an 8-byte spill.)
    int voff0;
    const int step32 = 32 * pitch;
    const int wave_chunk = wave * 1024;

    struct Item { int xi, mt, nt; };
    auto decode = [&](int k) {
        const int it = x_beg + j8 + k * per_xcd;
        Item c;
        c.nt = it % p.n_nt;
        const int rest = it / p.n_nt;
        c.mt = rest % p.n_mt;
        c.xi = rest / p.n_mt;
        return c;
    };
    __amdgpu_buffer_rsrc_t rsA, rsB;
    auto point_at = [&](const Item &c) {
        int64_t rows = (int64_t)p.T - (int64_t)c.mt * TM;
        if (rows > TM) rows = TM;
        rsA = wg_rsrc(p.V2 + ((int64_t)c.xi * p.T + (int64_t)c.mt * TM) * pitch, rows * pitch);
        rsB = wg_rsrc(p.U2 + ((int64_t)c.xi * p.Cout + (int64_t)c.nt * TN) * pitch, (int64_t)TN * pitch);
    };
    auto load_part = [&](int stage, int kt, int i) {          // parts 0..7: A, 8..15: B
        if (i < NLA) wg_blds16(rsA, voff0 + i * step32, kt * WG_ROWB, smem + stage * STAGE + i * (256 * 16) + wave_chunk);
        else wg_blds16(rsB, voff0, kt * WG_ROWB + (i - NLA) * step32, smem + stage * STAGE + OPA + (i - NLA) * (256 * 16) + wave_chunk);
    };

    // The per-lane address constants of the K loop (request offset, fragment offsets) are re-derived at the top of every item from
    // a thread id the compiler cannot see through: hoisted out of the item loop they stay live across the item's 256-store
    // epilogue, where the accumulators pass through the arch VGPRs, and one of them was spilled (8 bytes of scratch, round 4).
    int foff[2][2];                                    // [s][lo]
    int arow0, brow0;
    auto lane_constants = [&]() {
        int t = tid;
        asm volatile("" : "+v"(t));
        const int ln = t & 63, hh = ln >> 5, l31_ = ln & 31, swz = (ln >> 1) & 7;
        const int r = t >> 3, slot = t & 7;
        voff0 = r * pitch + ((slot ^ ((r >> 1) & 7)) << 4);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int lo = 0; lo < 2; ++lo) foff[s][lo] = ((4 * lo + 2 * s + hh) ^ swz) << 4;
        arow0 = (wm * 128 + l31_) * WG_ROWB;
        brow0 = (wn * 128 + l31_) * WG_ROWB;
    };
    lane_constants();

    f32x16 acc[MT][NT];
    f16x8 fa[2][2][MT], fb[2][2][NT];                         // [register set][hi | lo][tile]
    auto read_frags = [&](int u, const char *sA, const char *sB, int s) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            fa[u][0][m] = *(const f16x8 *)(sA + arow0 + m * 32 * WG_ROWB + foff[s][0]);
            fa[u][1][m] = *(const f16x8 *)(sA + arow0 + m * 32 * WG_ROWB + foff[s][1]);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            fb[u][0][n] = *(const f16x8 *)(sB + brow0 + n * 32 * WG_ROWB + foff[s][0]);
            fb[u][1][n] = *(const f16x8 *)(sB + brow0 + n * 32 * WG_ROWB + foff[s][1]);
        }
    };
    // 48 MFMAs: hi.hi, lo.hi, hi.lo over the 16 tiles; ZERO = an item's first K step: the accumulators start from the constant 0 (no
    // 256 register writes per item to clear them)
    auto multiply = [&](int u, auto zero_tag) {
        constexpr bool ZERO = decltype(zero_tag)::value;
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[u][0][m], fb[u][0][n], ZERO ? z : acc[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[u][1][m], fb[u][0][n], acc[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[u][0][m], fb[u][1][n], acc[m][n], 0, 0, 0);
    };

    // loader: one stage ahead (double buffer).  Which stage that is follows from the K loop's position: the same item's next K block,
    // or -- in an item's last stage -- the next item's first (behind the workgroup's last item: its own last block once more, into the
    // idle buffer).  The resources move to the next item right before an item's last stage.
    point_at(decode(0));
#pragma unroll
    for (int i = 0; i < NLA + NLB; ++i) load_part(0, 0, i);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();

    int cur = 0;
    auto stage = [&](auto first_tag, auto last_tag, int kt_load) {
        constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
        const char *sA = smem + cur * STAGE;
        const char *sB = sA + OPA;
        // ---- behind the barrier: K step 0's fragments, ALL of the next stage's requests (they have the rest of this stage to land: the
        // wait at its end is vmcnt(0)), the previous stage's K step 1
        read_frags(0, sA, sB, 0);
        if (DBG != 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) load_part(cur ^ 1, kt_load, i);
        }
        if constexpr (!FIRST) {
            multiply(1, std::false_type{});
            __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);      // DS read
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read (LDS-DMA)
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- K step 0 under the reads of K step 1
        read_frags(1, sA, sB, 1);
        multiply(0, first_tag);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (LAST) multiply(1, std::false_type{});
        __builtin_amdgcn_s_waitcnt(0);                               // the next stage has landed, my reads of this one are done
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        cur ^= 1;
    };
    for (int k_item = 0; k_item < n_mine; ++k_item) {
        if (k_item > 0) lane_constants();
        const Item c = decode(k_item);
        const bool more = k_item + 1 < n_mine;
        const int kt_after = more ? 0 : p.nk - 1;                    // what an item's last stage requests
        if (p.nk == 1) {
            if (more) point_at(decode(k_item + 1));
            stage(std::true_type{}, std::true_type{}, kt_after);
        } else {
            stage(std::true_type{}, std::false_type{}, 1);
            for (int kt = 1; kt + 1 < p.nk; ++kt) stage(std::false_type{}, std::false_type{}, kt + 1);
            if (more) point_at(decode(k_item + 1));
            stage(std::false_type{}, std::true_type{}, kt_after);
        }
        // ---- the finished item: lane = column, 128-byte runs per row; issued behind the barrier, drains under the next item's first
        // stage.  Buffer stores: the resource spans the wave tile's rows that exist (a store to a row beyond T is out of range and
        // dropped by the hardware: no predicate, no second code path), the lane's offset is one register, a row's offset a scalar
        // (as 64-bit per-lane addresses the 256 stores of a tile cost two VALU instructions each and spilled)
        {
            const int row0 = c.mt * TM + wm * 128;
            int64_t rows = (int64_t)p.T - row0;
            if (rows < 0) rows = 0;
            const int col0 = c.nt * TN + wn * 128;
            const __amdgpu_buffer_rsrc_t rsM = wg_rsrc((const char *)(p.M + ((int64_t)c.xi * p.T + row0) * p.Cout + col0),
                                                       rows > 0 ? rows * p.Cout * 4 - (int64_t)col0 * 4 : 0);   // a wave tile beyond T: nothing is in range
            const int st_voff = (4 * h * p.Cout + l31) * 4;
            const bool st_on = DBG != 1 || p.T < 0;
            int row_bytes = p.Cout * 4;
            asm volatile("" : "+s"(row_bytes));                  // opaque per item: the 256 row offsets are loop invariants otherwise, hoisted and spilled
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        if (st_on) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[m][n][r]), rsM, st_voff, (m * 32 + (r & 3) + 8 * (r >> 2)) * row_bytes + 128 * n, 2);
        }
    }
}

// Measured and removed (round 4, profiles/r04_v43_perf_wino_gemm_big2.log): the four-wave kernel with TWO accumulator sets of 128
// (256 x 128 tiles, ring of three) -- item k accumulates in set k & 1 while the tile of item k - 1 leaves 8-32 buffer stores per stage,
// counted in the stage wait, so that no stage waits for a whole tile to drain.  0.745 against 0.659 ms on conv4_2, 0.871 against 0.828
// on conv3_2: its 12 fragment reads per 24 MFMAs (8 per 48 above) cost more than the smoother stores gave.  (Found on the way: a
// run-time choice between the two sets in ONE merged block -- the compiler's tail merging of the two final-tile stores -- turns the
// accumulators into stack objects; buffer stores with the row offset in an SGPR need no per-store address registers and drop rows
// beyond the resource's range in hardware.)

// ---- the same products with the COLUMN half of the output transform folded in ("Z form") ------------------------------
// Y = A^T M A per tile and channel; Z_i[q] = sum_j M[6 i + j] A^T[q][j] (q = 0..3) needs the six frequencies of ONE row i of
// the 6 x 6 frequency grid only.  A work item here = (row i, 128-tile row block, 128-channel column block): the K loops of
// its six frequencies run back to back, each finished 128 x 128 product is folded into four register-resident Z planes
// (a few hundred v_fma per wave next to thousands of MFMA cycles) and only those leave the chip:
//     Z [24][T][Cout] float32 (plane 4 i + q)  =  2/3 of M's bytes, written here and read by wino4_output_z_kernel,
// which finishes with Y[p][q] = sum_i A^T[p][i] Z_i[q].  Same operands, same loader, same LDS image and fragment addressing
// as wino_gemm_h2_kernel; 128 x 128 tiles (8 waves as 2 x 4, wave tile 64 x 32: 32 accumulator + 128 Z registers), ring of
// three 32 KB stages.  Used where the product is HBM-bound (Cin <= 256: V2 in + M out), DESIGN.md section 3.6.
__constant__ float WZ_AT[6][4] = {{1.f, 0.f, 0.f, 0.f}, {1.f, 1.f, 1.f, 1.f}, {1.f, -1.f, 1.f, -1.f},
                                  {1.f, 2.f, 4.f, 8.f}, {1.f, -2.f, 4.f, -8.f}, {0.f, 0.f, 0.f, 1.f}};

template <int NS>
__global__ __launch_bounds__(512, 2) void wino_zgemm_h2_kernel(WinoGemmArgs p) {
    constexpr int TM = 128, TN = 128, MT = 2;
    constexpr int OPA = TM * WG_ROWB, OPB = TN * WG_ROWB;
    constexpr int STAGE = OPA + OPB;
    constexpr int NLA = TM * 8 / 512, NLB = TN * 8 / 512;   // 2 + 2 sixteen-byte chunks per thread per stage
    constexpr int L = NLA + NLB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int h = lane >> 5, l31 = lane & 31;

    const int xcd = blockIdx.x & 7, j8 = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int q8 = p.n_items >> 3, r8 = p.n_items & 7;
    const int x_beg = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int x_cnt = q8 + (xcd < r8 ? 1 : 0);
    const int n_mine = j8 < x_cnt ? (x_cnt - j8 + per_xcd - 1) / per_xcd : 0;
    if (n_mine == 0) return;

    const int64_t pitch = (int64_t)p.nk * WG_ROWB;
    int rowA[NLA], offA[NLA], rowB[NLB], offB[NLB];
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
        const int pch = i * 512 + tid, r = pch >> 3, slot = pch & 7;
        rowA[i] = r; offA[i] = (slot ^ ((r >> 1) & 7)) << 4;
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
        const int pch = i * 512 + tid, r = pch >> 3, slot = pch & 7;
        rowB[i] = r; offB[i] = (slot ^ ((r >> 1) & 7)) << 4;
    }
    const int wave_chunk = wave * 1024;

    struct Item { int gi, mt, nt; };                         // gi = row of the frequency grid (0..5)
    auto decode = [&](int k) {
        const int it = x_beg + j8 + k * per_xcd;
        Item c;
        c.nt = it % p.n_nt;
        const int rest = it / p.n_nt;
        c.mt = rest % p.n_mt;
        c.gi = rest / p.n_mt;
        return c;
    };
    const char *gA[NLA], *gB[NLB];
    auto point_at = [&](const Item &c, int j) {              // K block 0 of frequency 6 gi + j of the item
        const int xi = 6 * c.gi + j;
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            int64_t row = (int64_t)c.mt * TM + rowA[i];
            if (row > p.T - 1) row = p.T - 1;
            gA[i] = p.V2 + ((int64_t)xi * p.T + row) * pitch + offA[i];
        }
#pragma unroll
        for (int i = 0; i < NLB; ++i)
            gB[i] = p.U2 + ((int64_t)xi * p.Cout + (int64_t)c.nt * TN + rowB[i]) * pitch + offB[i];
    };
    auto load_part_a = [&](int stage, int kt, int i) {
        wg_glds16(gA[i] + kt * WG_ROWB, smem + stage * STAGE + i * (512 * 16) + wave_chunk);
    };
    auto load_part_b = [&](int stage, int kt, int i) {
        wg_glds16(gB[i] + kt * WG_ROWB, smem + stage * STAGE + OPA + i * (512 * 16) + wave_chunk);
    };

    const int swz = (lane >> 1) & 7;
    int foff[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int lo = 0; lo < 2; ++lo) foff[s][lo] = ((4 * lo + 2 * s + h) ^ swz) << 4;
    const int arow0 = (wm * (TM / 2) + l31) * WG_ROWB;
    const int brow0 = (wn * 32 + l31) * WG_ROWB;

    f32x16 acc[MT], Z[4][MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[m][r] = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) Z[q][m][r] = 0.0f;
        }

    // the loader runs NS - 1 stages ahead: (l_item, l_j, l_kt) = the next stage to fetch; a stage = one K block of one of the
    // item's six frequencies
    const int total = n_mine * 6 * p.nk;
    int l_item = 0, l_j = 0, l_kt = 0, l_stage = 0, fetched = 0;
    Item l_cur = decode(0);
    Item cur_item = l_cur;
    point_at(l_cur, 0);
    auto advance_loader = [&]() {
        ++fetched;
        l_stage = l_stage + 1 == NS ? 0 : l_stage + 1;
        if (++l_kt == p.nk) {
            l_kt = 0;
            if (++l_j == 6) { l_j = 0; ++l_item; if (fetched < total) l_cur = decode(l_item); }
            if (fetched < total) point_at(l_cur, l_j);
        }
    };
#pragma unroll 1
    for (int pre = 0; pre < NS - 1; ++pre) {
        if (fetched < total) {
#pragma unroll
            for (int i = 0; i < NLA; ++i) load_part_a(l_stage, l_kt, i);
#pragma unroll
            for (int i = 0; i < NLB; ++i) load_part_b(l_stage, l_kt, i);
            advance_loader();
        }
    }
    if (NS == 2 || total == 1) __builtin_amdgcn_s_waitcnt(0); else WG_VMCNT(L);
    __syncthreads();

    int k_item = 0, jf = 0, kt = 0, cur = 0;
    bool store_pending = false;
    Item st_item = cur_item;
    auto store_z = [&](const Item &c) {
        // lane = column, 128-byte runs per row (as wino_gemm_h2_kernel); four planes 4 gi + q
        const int row_base = c.mt * TM + wm * (TM / 2) + 4 * h;
        const int col = c.nt * TN + wn * 32 + l31;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float *zo = p.M + ((int64_t)(4 * c.gi + q) * p.T + row_base) * p.Cout + col;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = m * 32 + (r & 3) + 8 * (r >> 2);
                    if (row_base + ro < p.T) __builtin_nontemporal_store(Z[q][m][r], zo + (int64_t)ro * p.Cout);
                    Z[q][m][r] = 0.0f;
                }
        }
    };
    for (int it = 0; it < total; ++it) {
        if (store_pending) { store_z(st_item); store_pending = false; }
        const bool fetch = fetched < total;
        const char *sA = smem + cur * STAGE;
        const char *sB = sA + OPA;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 ah[MT], al[MT], bh, bl;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                ah[m] = *(const f16x8 *)(sA + arow0 + m * 32 * WG_ROWB + foff[s][0]);
                al[m] = *(const f16x8 *)(sA + arow0 + m * 32 * WG_ROWB + foff[s][1]);
            }
            bh = *(const f16x8 *)(sB + brow0 + foff[s][0]);
            bl = *(const f16x8 *)(sB + brow0 + foff[s][1]);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh, acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh, acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl, acc[m], 0, 0, 0);
            if (fetch) {
#pragma unroll
                for (int i = s * (NLA / 2); i < (s + 1) * (NLA / 2); ++i) load_part_a(l_stage, l_kt, i);
#pragma unroll
                for (int i = s * (NLB / 2); i < (s + 1) * (NLB / 2); ++i) load_part_b(l_stage, l_kt, i);
            }
        }
        if (fetch) advance_loader();

        if (kt == p.nk - 1) {
            // frequency 6 gi + jf is complete: fold it into the four Z planes (coefficients A^T[q][jf], wave-uniform)
            const float c0 = WZ_AT[jf][0], c1 = WZ_AT[jf][1], c2 = WZ_AT[jf][2], c3 = WZ_AT[jf][3];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float a = acc[m][r];
                    Z[0][m][r] = fmaf(c0, a, Z[0][m][r]);
                    Z[1][m][r] = fmaf(c1, a, Z[1][m][r]);
                    Z[2][m][r] = fmaf(c2, a, Z[2][m][r]);
                    Z[3][m][r] = fmaf(c3, a, Z[3][m][r]);
                    acc[m][r] = 0.0f;
                }
            if (jf == 5) { store_pending = true; st_item = cur_item; }
        }
        if (NS == 2 || !fetch) __builtin_amdgcn_s_waitcnt(0); else WG_VMCNT(L);
        __builtin_amdgcn_s_barrier();
        cur = cur + 1 == NS ? 0 : cur + 1;
        if (++kt == p.nk) {
            kt = 0;
            if (++jf == 6) { jf = 0; ++k_item; if (it + 1 < total) cur_item = decode(k_item); }
        }
    }
    if (store_pending) store_z(st_item);
}

// ---- input transform into the pair layout ------------------------------------------------------------------------------
__device__ __forceinline__ float wg_h_scale(unsigned amax_bits) {       // = wino_h3_scale (winograd.hip)
    const float a = fminf(fmaxf(__uint_as_float(amax_bits), 1e-30f), 1e30f);
    int e;
    (void)frexpf(327.68f / a, &e);
    return ldexpf(1.0f, e - 1);
}
__device__ __forceinline__ void wg_bt4(wf4 &d0, wf4 &d1, wf4 &d2, wf4 &d3, wf4 &d4, wf4 &d5) {   // = wino4_bt4
    const wf4 r0 = 4.0f * d0 - 5.0f * d2 + d4;
    const wf4 r1 = -4.0f * (d1 + d2) + d3 + d4;
    const wf4 r2 = 4.0f * (d1 - d2) - d3 + d4;
    const wf4 r3 = 2.0f * (d3 - d1) - d2 + d4;
    const wf4 r4 = 2.0f * (d1 - d3) - d2 + d4;
    const wf4 r5 = 4.0f * d1 - 5.0f * d3 + d5;
    d0 = r0; d1 = r1; d2 = r2; d3 = r3; d4 = r4; d5 = r5;
}

template <int HI>
__device__ __forceinline__ float wg_sub_half(float v, __half2 h) {       // v - (float)half HI of the packed pair h: one v_fma_mix_f32
    float d;
    const unsigned hb = *(const unsigned *)&h;
    if (HI) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hb), "v"(v));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hb), "v"(v));
    return d;
}

// V = B^T d B of every 6 x 6 input tile of x [B,H,W,C] (NHWC, times the power-of-two scale derived from *amax), split into
// fp16 pairs and stored as V2 [36][T][C/32][hi 32 | lo 32].  One thread = one tile x 4 consecutive channels: 16-byte
// loads; per frequency the hi and the lo halves of its 4 channels (8 bytes each) -- lanes 2k / 2k + 1 trade them (DPP) so that
// the even lane stores 16 bytes of hi halves and the odd lane 16 bytes of lo halves (PAIR16), instead of two 8-byte stores each.
template <bool PAIR16>
__global__ __launch_bounds__(256) void wino4_input_h2_kernel(const float *__restrict__ x, int B, int H, int W, int C,
                                                             const unsigned *__restrict__ amax, __half *__restrict__ V2) {
    const int c4n = C >> 2;
    const int64_t bid = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-contiguous tiles
    const int64_t gid = bid * 256 + threadIdx.x;
    const int TH = (H + 3) >> 2, TW = (W + 3) >> 2;
    const int64_t T = (int64_t)B * TH * TW;
    if (gid >= T * c4n) return;
    const int c4 = (int)(gid % c4n);
    const int64_t t = gid / c4n;
    const int tj = (int)(t % TW);
    const int ti = (int)((t / TW) % TH);
    const int b = (int)(t / ((int64_t)TW * TH));
    const int h0 = 4 * ti - 1, w0 = 4 * tj - 1;
    const float sc = wg_h_scale(*amax);
    wf4 d[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int hh = h0 + i;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int ww = w0 + j;
            const bool in = (hh >= 0) & (hh < H) & (ww >= 0) & (ww < W);
            const wf4 v = *((const wf4 *)(x + (((int64_t)b * H + (in ? hh : 0)) * W + (in ? ww : 0)) * C) + c4);
            d[i][j] = in ? v * sc : (wf4)(0.0f);
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) wg_bt4(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);
#pragma unroll
    for (int i = 0; i < 6; ++i) wg_bt4(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5]);
    const int64_t plane = T * 2 * C;                                  // halfs per frequency
    const int c = 4 * c4;
    __half *o = V2 + t * 2 * C + (c >> 5) * 64 + (c & 31);
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const wf4 v = d[i][j];
            const __half2 h0v = __floats2half2_rn(v.x, v.y), h1v = __floats2half2_rn(v.z, v.w);
            // v - (float)hi read in place out of the packed register (one v_fma_mix_f32 per value; through __half22float2 hipcc rounds
            // the value once more with a scalar conversion, converts back and subtracts)
            const __half2 l0v = __floats2half2_rn(wg_sub_half<0>(v.x, h0v), wg_sub_half<1>(v.y, h0v));
            const __half2 l1v = __floats2half2_rn(wg_sub_half<0>(v.z, h1v), wg_sub_half<1>(v.w, h1v));
            wu2 hi, lo;
            hi.x = *(const unsigned *)&h0v; hi.y = *(const unsigned *)&h1v;
            lo.x = *(const unsigned *)&l0v; lo.y = *(const unsigned *)&l1v;
            __half *q = o + (int64_t)(6 * i + j) * plane;
            if (PAIR16) {
                // 16-byte stores: lanes 2k / 2k + 1 hold channels 8k .. 8k + 7 of this block between them; the even lane stores
                // the hi halves of all eight (its own + the partner's), the odd lane the lo halves
                const bool odd = threadIdx.x & 1;
                wu2 give = odd ? hi : lo, got;
                got.x = (unsigned)__builtin_amdgcn_mov_dpp((int)give.x, 0xB1, 0xF, 0xF, true);       // quad_perm [1, 0, 3, 2]
                got.y = (unsigned)__builtin_amdgcn_mov_dpp((int)give.y, 0xB1, 0xF, 0xF, true);
                typedef unsigned wu4 __attribute__((ext_vector_type(4)));
                wu4 v16;
                if (odd) { v16.x = got.x; v16.y = got.y; v16.z = lo.x; v16.w = lo.y; }
                else { v16.x = hi.x; v16.y = hi.y; v16.z = got.x; v16.w = got.y; }
                __builtin_nontemporal_store(v16, (wu4 *)(odd ? q + 32 - 4 : q));
            } else {
                __builtin_nontemporal_store(hi, (wu2 *)q);
                __builtin_nontemporal_store(lo, (wu2 *)(q + 32));
            }
        }
}

CSLAM_API int cslam_wino4_input_h2_dev(const float *d_x, int B, int H, int W, int C, const unsigned *d_amax, void *d_V2,
                                       void *stream) {
    PTR_DEVICE(d_x);
    ARG_CHECK(d_x && d_V2 && d_amax, "NULL argument");
    ARG_CHECK(B >= 1 && H >= 1 && W >= 1, "empty map");
    ARG_CHECK(C >= 32 && (C % 32) == 0, "C must be a multiple of 32");
    const int64_t n4 = (int64_t)B * ((H + 3) / 4) * ((W + 3) / 4) * (C / 4);
    ARG_CHECK(ceil_div64(n4, 256) < (1LL << 31), "too many tiles for one launch");
    // 16-byte stores through a lane-pair exchange (two 8-byte stores per thread and frequency: 5.0 -> 5.3 TB/s on the trunk's
    // shapes, profiles/r02_v28_input_transform_pair16_ab.log)
    hipLaunchKernelGGL(wino4_input_h2_kernel<true>, dim3((unsigned)round_up64(ceil_div64(n4, 256), 8)), dim3(256), 0,
                       (hipStream_t)stream, d_x, B, H, W, C, d_amax, (__half *)d_V2);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

// ---- diagnostics: per-launch timing of the trunk's products inside a run (include/cslam_hip_experimental.h) -----------------
// bench.py prices the step's largest consumer from the launches of the timed steps themselves: while enabled, every product
// launch is bracketed by two HIP events ON ITS OWN STREAM and its algorithmic flop / bytes are noted; cslam_trunk_timing_read
// waits for the recorded events, adds everything up and clears the log.  Off (the default) it costs one relaxed load.
namespace {
struct TimedLaunch { hipEvent_t e0, e1; double flop16, bytes; int cin; };
struct TrunkTiming {
    std::atomic<int> on{0};
    std::mutex mu;                      // entry points are called from several host threads (extraction lanes): the log is shared
    int n = 0;
    TimedLaunch log[512];
    bool made[512] = {};
} g_tt;
}
static void tt_begin(hipStream_t st, int &slot) {
    slot = -1;
    if (!g_tt.on.load(std::memory_order_relaxed)) return;
    // an event recorded on a capturing stream becomes a graph node: the later hipEventSynchronize would fail
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return; }
    std::lock_guard<std::mutex> lock(g_tt.mu);
    if (g_tt.n >= 512) return;
    slot = g_tt.n++;                    // the slot is this launch's from here on
    g_tt.log[slot].cin = -1;            // not complete yet (cslam_trunk_timing_read skips it)
    if (!g_tt.made[slot]) {
        if (hipEventCreate(&g_tt.log[slot].e0) != hipSuccess || hipEventCreate(&g_tt.log[slot].e1) != hipSuccess) { slot = -1; return; }
        g_tt.made[slot] = true;
    }
    (void)hipEventRecord(g_tt.log[slot].e0, st);
}
static void tt_end(hipStream_t st, int slot, const WinoGemmArgs &a, int planes_out) {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lock(g_tt.mu);
    (void)hipEventRecord(g_tt.log[slot].e1, st);
    g_tt.log[slot].flop16 = 3.0 * 2.0 * 36.0 * a.T * (double)a.Cin * a.Cout * (a.nxi == 36 || a.nxi == 6 ? 1.0 : a.nxi / 36.0);
    g_tt.log[slot].bytes = 36.0 * a.T * a.Cin * 4.0 + (double)planes_out * a.T * a.Cout * 4.0 + 36.0 * a.Cin * (double)a.Cout * 4.0;
    g_tt.log[slot].cin = a.Cin;
}
CSLAM_API int cslam_trunk_timing(int enable) {
    std::lock_guard<std::mutex> lock(g_tt.mu);
    g_tt.n = 0;
    g_tt.on.store(enable ? 1 : 0, std::memory_order_relaxed);
    return CSLAM_OK;
}
/* out[0..3] = launches / ms / fp16 flop / algorithmic HBM bytes of the products with Cin <= 256 (HBM-bound), out[4..7] = the
 * same for Cin > 256 (matrix-pipe-bound), since the last read; clears the log. */
CSLAM_API int cslam_trunk_timing_read(double out[8]) {
    ARG_CHECK(out, "NULL argument");
    for (int i = 0; i < 8; ++i) out[i] = 0.0;
    std::lock_guard<std::mutex> lock(g_tt.mu);
    for (int i = 0; i < g_tt.n; ++i) {
        float ms = 0.0f;
        if (g_tt.log[i].cin < 0) continue;           // begun, never ended (its launch failed)
        HIP_TRY(hipEventSynchronize(g_tt.log[i].e1));
        HIP_TRY(hipEventElapsedTime(&ms, g_tt.log[i].e0, g_tt.log[i].e1));
        const int o = g_tt.log[i].cin <= 256 ? 0 : 4;
        out[o] += 1.0; out[o + 1] += ms; out[o + 2] += g_tt.log[i].flop16; out[o + 3] += g_tt.log[i].bytes;
    }
    g_tt.n = 0;
    return CSLAM_OK;
}

template <int TM, int TN, int NS, int WM = 2>
static int wino_gemm_launch(WinoGemmArgs a, int dbg, hipStream_t st) {
    int n_cu = cslam_cu_count();
    ARG_CHECK(n_cu > 0, "no HIP device");
    if (n_cu < 8) n_cu = 8;
    a.n_mt = (int)ceil_div64(a.T, TM);
    a.n_nt = a.Cout / TN;
    const int64_t items = (int64_t)a.nxi * a.n_mt * a.n_nt;
    ARG_CHECK(items < (1LL << 31), "too many work items");
    a.n_items = (int)items;
    constexpr int lds = NS * (TM + TN) * WG_ROWB;
    int grid = n_cu - n_cu % 8;                                      // one persistent workgroup per CU, 8 | grid
    if ((int64_t)grid > round_up64(a.n_items, 8)) grid = (int)round_up64(a.n_items, 8);
    static DeviceOnce once;
    int once_dev;
    if (once.todo(&once_dev)) {
        HIP_TRY(hipFuncSetAttribute((const void *)wino_gemm_h2_kernel<TM, TN, NS, 0, WM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#ifdef CSLAM_ABLATIONS
        HIP_TRY(hipFuncSetAttribute((const void *)wino_gemm_h2_kernel<TM, TN, NS, 1, WM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void *)wino_gemm_h2_kernel<TM, TN, NS, 2, WM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#endif
        once.done(once_dev);
    }
    int tslot;
    tt_begin(st, tslot);
#ifdef CSLAM_ABLATIONS
    if (dbg == 1) hipLaunchKernelGGL((wino_gemm_h2_kernel<TM, TN, NS, 1, WM>), dim3(grid), dim3(512), lds, st, a);
    else if (dbg == 2) hipLaunchKernelGGL((wino_gemm_h2_kernel<TM, TN, NS, 2, WM>), dim3(grid), dim3(512), lds, st, a);
    else
#endif
    hipLaunchKernelGGL((wino_gemm_h2_kernel<TM, TN, NS, 0, WM>), dim3(grid), dim3(512), lds, st, a);
    if (a.nxi == 36) tt_end(st, tslot, a, 36);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

static int wino_gemm_big_launch(WinoGemmArgs a, int dbg, hipStream_t st) {
    int n_cu = cslam_cu_count();
    ARG_CHECK(n_cu > 0, "no HIP device");
    if (n_cu < 8) n_cu = 8;
    a.n_mt = (int)ceil_div64(a.T, 256);
    a.n_nt = a.Cout / 256;
    const int64_t items = (int64_t)a.nxi * a.n_mt * a.n_nt;
    ARG_CHECK(items < (1LL << 31), "too many work items");
    a.n_items = (int)items;
    constexpr int lds = 2 * (256 + 256) * WG_ROWB;                    // two 64 KB stages
    int grid = n_cu - n_cu % 8;
    if ((int64_t)grid > round_up64(a.n_items, 8)) grid = (int)round_up64(a.n_items, 8);
    static DeviceOnce once;
    int once_dev;
    if (once.todo(&once_dev)) {
        HIP_TRY(hipFuncSetAttribute((const void *)wino_gemm_h2_big_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        once.done(once_dev);
    }
    int tslot;
    tt_begin(st, tslot);
#ifdef CSLAM_ABLATIONS
    if (dbg == 1 || dbg == 2) {                                     // timing-only: 1 = no stores, 2 = no requests after the prologue
        if (dbg == 1) { HIP_TRY(hipFuncSetAttribute((const void *)wino_gemm_h2_big_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                        hipLaunchKernelGGL((wino_gemm_h2_big_kernel<1>), dim3(grid), dim3(256), lds, st, a); }
        else { HIP_TRY(hipFuncSetAttribute((const void *)wino_gemm_h2_big_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
               hipLaunchKernelGGL((wino_gemm_h2_big_kernel<2>), dim3(grid), dim3(256), lds, st, a); }
    } else
#endif
    hipLaunchKernelGGL((wino_gemm_h2_big_kernel<0>), dim3(grid), dim3(256), lds, st, a);
    if (a.nxi == 36) tt_end(st, tslot, a, 36);
    HIP_TRY(hipGetLastError());
    (void)dbg;
    return CSLAM_OK;
}

CSLAM_API int cslam_wino_gemm_h2_dev(const void *d_V2, const void *d_U2, int64_t T, int Cin, int Cout, float *d_M,
                                     void *stream) {
    PTR_DEVICE(d_V2);
    ARG_CHECK(d_V2 && d_U2 && d_M, "NULL argument");
    ARG_CHECK(T >= 1 && T < (1LL << 31), "T out of range");
    ARG_CHECK(Cin >= 32 && (Cin % 32) == 0, "Cin must be a multiple of 32");
    ARG_CHECK(Cout >= 128 && (Cout % 128) == 0, "Cout must be a multiple of 128");
    WinoGemmArgs a;
    a.V2 = (const char *)d_V2; a.U2 = (const char *)d_U2; a.M = d_M;
    a.T = (int)T; a.Cin = Cin; a.Cout = Cout; a.nk = Cin / 32;
    a.n_mt = a.n_nt = a.n_items = 0;
    a.nxi = 36;
    // read on every call (~100 ns) so that tests can switch shapes inside one process
    const char *c = getenv("CSLAM_WGEMM_CFG");
    int dbg = 0;
#ifdef CSLAM_ABLATIONS
    if (const char *e = getenv("CSLAM_WGEMM_DBG")) dbg = atoi(e);   // timing-only ablations (wrong results): measurement build only
#endif
    const int cfg = c ? atoi(c) : 0;            // force a tile / ring shape (1..5 below); 0 = default
    hipStream_t st = (hipStream_t)stream;
    // default (cfg 5): 256 x 128 tiles, ring of three stages, the eight waves as 4 x 2 = 64 x 64 wave tiles: 8 fragment reads per 12
    // MFMAs (2 x 4 waves = 128 x 32 wave tiles, cfg 2, the default until round 4: 10).  With `global_load_lds` the two were the same
    // speed (profiles/r03_v33_perf_wino_gemm_wave_grid.log); with the buffer form of the LDS-DMA, under which hipcc counts LDS reads
    // instead of draining them, cfg 5 is 4-6 % faster on every layer (profiles/r04_v22_perf_wino_gemm_shapes_mubuf.log): what is left
    // is LDS bandwidth -- per stage and CU 128 KB of fragment reads + 48 KB of DMA writes = 1400 LDS cycles under 1536 matrix cycles.
    const bool wide = (Cout % 256) == 0;
    int use = cfg;
    // 0 / invalid: the default -- the four-wave form from 256 input channels on (conv3_2 ... conv5_3: 3-7 % per layer at Cin = 512, a
    // tie at 256 -> 256, slower at 128 -> 256), 64 x 64 wave tiles on eight waves elsewhere.  Whole trunk passes, interleaved:
    // 15.25 ms against 15.47 with the eight-wave form everywhere (profiles/r04_v33_perf_wino_gemm_big.log, r04_v39_trunk_gemm_cfg_ab.log)
    if (use < 1 || use > 6 || ((use == 1 || use == 3 || use == 6) && !wide)) use = (wide && Cin >= 256) ? 6 : 5;
    if (use == 6) return wino_gemm_big_launch(a, dbg, st);          // four waves of 512 registers, 256 x 256 tiles
    switch (use) {
    case 1: return wino_gemm_launch<256, 256, 2>(a, dbg, st);      // double buffer, 256 x 256
    case 2: return wino_gemm_launch<256, 128, 3>(a, dbg, st);      // ring of 3, 256 x 128
    case 3: return wino_gemm_launch<128, 256, 3>(a, dbg, st);      // ring of 3, 128 x 256
    case 5: return wino_gemm_launch<256, 128, 3, 4>(a, dbg, st);   // ring of 3, 256 x 128, 64 x 64 wave tiles
    default: return wino_gemm_launch<256, 128, 2>(a, dbg, st);     // double buffer, 256 x 128 (round-2 first form)
    }
}

/* The same kernel as a general batch of `nxi` independent pair products M[i] = A2[i] B2[i]^T (rows x K) x (N x K): used by the
 * PCA projection of the NetVLAD head (csrc/gemm_nt.hip: its K = 32768 is cut into nxi splits whose partial products the
 * epilogue sums).  Layouts as cslam_wino_gemm_h2_dev with 36 -> nxi. */
int cslam_pair_gemm_launch(const void *d_A2, const void *d_B2, int nxi, int64_t T, int K, int N, float *d_M, hipStream_t st) {
    ARG_CHECK(d_A2 && d_B2 && d_M, "NULL argument");
    ARG_CHECK(nxi >= 1 && T >= 1 && T < (1LL << 31), "nxi / T out of range");
    ARG_CHECK(K >= 32 && (K % 32) == 0, "K must be a multiple of 32");
    ARG_CHECK(N >= 128 && (N % 128) == 0, "N must be a multiple of 128");
    WinoGemmArgs a;
    a.V2 = (const char *)d_A2; a.U2 = (const char *)d_B2; a.M = d_M;
    a.T = (int)T; a.Cin = K; a.Cout = N; a.nk = K / 32;
    a.n_mt = a.n_nt = a.n_items = 0;
    a.nxi = nxi;
    return wino_gemm_launch<256, 128, 3, 4>(a, 0, st);
}

/* The Z form (wino_zgemm_h2_kernel above): d_Z [24][T][Cout] float32, plane 4 i + q = sum_j (V U)[6 i + j] A^T[q][j]; finished by
 * cslam_wino4_output_z_dev (winograd.hip).  Cin a multiple of 32, Cout of 128. */
CSLAM_API int cslam_wino_zgemm_h2_dev(const void *d_V2, const void *d_U2, int64_t T, int Cin, int Cout, float *d_Z, void *stream) {
    PTR_DEVICE(d_V2);
    ARG_CHECK(d_V2 && d_U2 && d_Z, "NULL argument");
    ARG_CHECK(T >= 1 && T < (1LL << 31), "T out of range");
    ARG_CHECK(Cin >= 32 && (Cin % 32) == 0, "Cin must be a multiple of 32");
    ARG_CHECK(Cout >= 128 && (Cout % 128) == 0, "Cout must be a multiple of 128");
    WinoGemmArgs a;
    a.V2 = (const char *)d_V2; a.U2 = (const char *)d_U2; a.M = d_Z;
    a.T = (int)T; a.Cin = Cin; a.Cout = Cout; a.nk = Cin / 32;
    a.nxi = 6;
    a.n_mt = (int)ceil_div64(T, 128);
    a.n_nt = Cout / 128;
    const int64_t items = (int64_t)6 * a.n_mt * a.n_nt;
    ARG_CHECK(items < (1LL << 31), "too many work items");
    a.n_items = (int)items;
    int n_cu = cslam_cu_count();
    ARG_CHECK(n_cu > 0, "no HIP device");
    if (n_cu < 8) n_cu = 8;
    // ring of three 32 KB stages, one workgroup per CU (a double buffer, two workgroups per CU: 1.50-1.54 ms against 1.30 on
    // conv2_2, profiles/r03_v9_zgemm_variants.log)
    const int lds = 3 * (128 + 128) * WG_ROWB;
    int grid = n_cu - n_cu % 8;
    if ((int64_t)grid > round_up64(a.n_items, 8)) grid = (int)round_up64(a.n_items, 8);
    static DeviceOnce once;
    int once_dev;
    if (once.todo(&once_dev)) {
        HIP_TRY(hipFuncSetAttribute((const void *)wino_zgemm_h2_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (128 + 128) * WG_ROWB));
        once.done(once_dev);
    }
    int tslot;
    tt_begin((hipStream_t)stream, tslot);
    hipLaunchKernelGGL((wino_zgemm_h2_kernel<3>), dim3(grid), dim3(512), lds, (hipStream_t)stream, a);
    tt_end((hipStream_t)stream, tslot, a, 24);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}
