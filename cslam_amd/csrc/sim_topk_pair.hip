// sim_topk_pair.hip -- the candidate stage of the batched cosine top-k (sim_topk_mfma.hip's stage 1) on the fp16 matrix pipe.
//
// Replaces, for a batch of queries, the scoring loop of cslam/nns_matching.py:55-61 exactly as sim_topk_mfma_kernel does --
// S^T tile = Bank_tile x Query_tile^T, queries as the MFMA B operand so that a lane's accumulators belong to ONE query and
// the running candidate list of that query lives in the lane's registers -- but with both operands as exact fp16 pairs:
// a float times a power of two splits exactly into hi + lo (11 + 11 significant bits), fp16 x fp16 products are exact in the
// MFMA's fp32 accumulator, and
//     q.b  =  qh.bh + ql.bh + qh.bl        (v_mfma_f32_32x32x16_f16 x 3; the dropped ql.bl is 2^-22 of the product)
// is an fp32-grade dot product at 16/3 of the f32-input MFMA rate.  Stage 1 is a FILTER: stage 2 (rescore_kernel) re-scores
// the contenders in float64 with the reference formula and certifies with a rigorous bound on |key - exact| that no row
// outside the candidate set can reach the k-th place (uncertified queries fall back to the exact scan) -- so the result is
// bit-identical to the float64 oracle whatever the candidate stage computes, as long as the bound handed to stage 2 holds.
// That bound for this kernel (pair_err_bound below):
//     representation   each operand value w = s v: |w - (hi + lo)| <= 2^-22 |w| + 2^-25      (two roundings to fp16)
//                      float64 / float32 query -> float32 before the split: 2^-24
//     dropped product  |ql bl| <= 2^-22 |q b| (1 + 2^-10)
//     accumulation     3 kd exact products summed in fp32 in an unspecified order, every partial sum rounded OR truncated
//                      (the MFMA's internal adder is not documented: unit 2^-23 instead of 2^-24)
//     key              two roundings: (acc * invs[row]) * qinvs[query]; invs carries 1/||b|| rounded to float32
// Operand layout (bank copy `rows2`, query copy in the workspace): row = kd/32 blocks of 128 bytes, block = [hi of 32
// channels | lo of the same] -- the layout of wino_gemm.hip's operands, same loader, same XOR-swizzled LDS image, same
// fragment addressing.  Work decomposition, candidate lists, drop bounds and block merge are those of sim_topk_mfma_kernel.
#include <stdlib.h>
#include <hip/hip_fp16.h>
#include "bank.h"
#include "sim_topk.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define PK_ROWB 128     // bytes of one 32-channel block of one row

__device__ __forceinline__ void pk_glds16(const char *g, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}


// ---- tile epilogue: lane-local candidate update from the finished 32 MT x 64 wave tile, then clear the accumulators
// (a lane's 16 accumulator registers of a tile belong to ONE query column: no cross-lane traffic)
template <int MT, int KPL>
__device__ __forceinline__ void pair_tile_epilogue(f32x16 (&acc)[MT][2], float (&lk)[2][KPL], int (&li)[2][KPL], const int (&lim)[2],
                                                   const float (&qmul)[2], const float *__restrict__ invs, int n_rows, int row_base) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        float inv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = row_base + m * 32 + (r & 3) + 8 * (r >> 2);
            inv[r] = invs[row < n_rows ? row : n_rows - 1];
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            f32x16 keys;
            bool any = false;
            const float thr = lk[n][KPL - 1];
            const int rel_lim = lim[n] - (row_base + m * 32);   // row < lim  <=>  rowoff < rel_lim
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float kx = (acc[m][n][r] * inv[r]) * qmul[n];
                keys[r] = kx;
                bool ok = ((r & 3) + 8 * (r >> 2)) < rel_lim;
                any |= ok && !(kx <= thr);     // NaN passes (ranks first)
                acc[m][n][r] = 0.0f;
            }
            if (__any(any)) {
#pragma unroll 1
                for (int r = 0; r < 16; ++r) {
                    float ck = keys[r];        // uniform dynamic index
                    int roff = (r & 3) + 8 * (r >> 2);
                    ck = (ck != ck) ? INFINITY : ck;
                    bool ins = (roff < rel_lim) && (ck > lk[n][KPL - 1]);
                    if (__any(ins)) {
                        ck = ins ? ck : -INFINITY;
                        int ci = row_base + m * 32 + roff;
#pragma unroll
                        for (int j = 0; j < KPL; ++j) {
                            bool sw = ck > lk[n][j];
                            float tk = sw ? lk[n][j] : ck;
                            int ti = sw ? li[n][j] : ci;
                            lk[n][j] = sw ? ck : lk[n][j];
                            li[n][j] = sw ? ci : li[n][j];
                            ck = tk; ci = ti;
                        }
                    }
                }
            }
        }
    }
}

// ---- block merge: 4 lists per query (2 row-halves of the wave x 2 waves along the bank axis) -> the best KP of them, plus the
// bound on everything dropped (see sim_topk_mfma_kernel); the LDS of the K loop is reused
template <int T_, int KPL>
__device__ __forceinline__ void pair_block_merge(char *smem, float (&lk)[2][KPL], int (&li)[2][KPL], int wn, int wm, int h, int l31,
                                                 int tid, int qt, int seg, const PairArgs &p) {
    __syncthreads();
    float *mk = (float *)smem;                       // [T_][4][KPL]
    int *mi = (int *)(smem + T_ * 4 * KPL * 4);      // [T_][4][KPL]
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        int qcol = wn * 64 + n * 32 + l31;
        int src = wm * 2 + h;
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            mk[(qcol * 4 + src) * KPL + j] = lk[n][j];
            mi[(qcol * 4 + src) * KPL + j] = li[n][j];
        }
    }
    __syncthreads();
    if (tid < T_) {
        const float *k0 = mk + (tid * 4) * KPL;
        const int *i0 = mi + (tid * 4) * KPL;
        int p0 = 0, p1 = 0, p2 = 0, p3 = 0;
        size_t o = ((size_t)(qt * T_ + tid) * p.nseg + seg) * SIM_KP;
        float bound = -INFINITY;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (i0[s * KPL + KPL - 1] >= 0) bound = fmaxf(bound, k0[s * KPL + KPL - 1]);   // full lane list
        for (int j = 0; j < SIM_KP; ++j) {
            float c0 = p0 < KPL ? k0[p0] : -INFINITY;
            float c1 = p1 < KPL ? k0[KPL + p1] : -INFINITY;
            float c2 = p2 < KPL ? k0[2 * KPL + p2] : -INFINITY;
            float c3 = p3 < KPL ? k0[3 * KPL + p3] : -INFINITY;
            int best = 0; float bk = c0;
            if (c1 > bk) { bk = c1; best = 1; }
            if (c2 > bk) { bk = c2; best = 2; }
            if (c3 > bk) { bk = c3; best = 3; }
            int bi;
            if (best == 0) { bi = p0 < KPL ? i0[p0] : -1; ++p0; }
            else if (best == 1) { bi = i0[KPL + p1]; ++p1; }
            else if (best == 2) { bi = i0[2 * KPL + p2]; ++p2; }
            else { bi = i0[3 * KPL + p3]; ++p3; }
            p.part_key[o + j] = bk;
            p.part_idx[o + j] = (bk == -INFINITY) ? -1 : bi;
        }
        float c0 = p0 < KPL ? k0[p0] : -INFINITY, c1 = p1 < KPL ? k0[KPL + p1] : -INFINITY;
        float c2 = p2 < KPL ? k0[2 * KPL + p2] : -INFINITY, c3 = p3 < KPL ? k0[3 * KPL + p3] : -INFINITY;
        bound = fmaxf(fmaxf(bound, fmaxf(c0, c1)), fmaxf(c2, c3));
        p.part_bound[(size_t)(qt * T_ + tid) * p.nseg + seg] = bound;
    }
}

// T_ = tile edge (bank rows = queries per tile), MT = 32-row MFMA tiles per wave along the bank axis (wave tile =
// 32 MT x 64), KPL = per-lane candidate list length.  Waves: 2 along the bank axis x (T_/64) along the query axis.
// DBG != 0: TIMING-ONLY ablations (wrong results): 1 = no global loads after the first stage.
// LDM: placement of the next stage's LDS-DMA requests: 1 = all of them among the MFMAs of the first K step (default),
// 0 = half behind each K step's MFMAs (round 3's first form; CSLAM_PAIR_LDM=0 for A/B runs)
template <int T_, int MT, int KPL, int DBG, int LDM>
__global__ __launch_bounds__(T_ * 2, 2) void sim_topk_pair_kernel(PairArgs p) {
    constexpr int NTHR = T_ * 2;
    constexpr int NWN = T_ / 64;
    constexpr int OPB = T_ * PK_ROWB;            // bytes of one operand tile (one 32-channel block of T_ rows) in LDS
    constexpr int STAGE = 2 * OPB;
    constexpr int NLD = T_ * 8 / NTHR;           // 16-byte chunks per thread per operand (= 4)
    static_assert(T_ == 64 * MT, "wave tile must cover half the bank tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int h = lane >> 5, l31 = lane & 31;

    // XCD-aware work-item mapping (see sim_topk_mfma_kernel): block b runs on XCD b % 8, each XCD gets a contiguous run of the
    // patch-major item list
    const int T = p.nqt * p.nseg;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, j8 = bid >> 3, q8 = T >> 3, r8 = T & 7;
    const int item = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j8;
    const int packed = p.item_map[item];
    const int qt = packed >> 16, seg = packed & 0xffff;

    int t_beg = seg * p.tps;
    int t_end = t_beg + p.tps;
    if (t_end > p.n_btiles) t_end = p.n_btiles;
    {
        int ml = p.qt_maxlim[qt];
        int te = (ml + T_ - 1) / T_;
        if (t_end > te) t_end = te;
    }

    float lk[2][KPL]; int li[2][KPL];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int j = 0; j < KPL; ++j) { lk[n][j] = -INFINITY; li[n][j] = -1; }
    int lim[2];
    float qmul[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        lim[n] = p.lim[qt * T_ + wn * 64 + n * 32 + l31];
        qmul[n] = p.qinvs[qt * T_ + wn * 64 + n * 32 + l31];
    }

    const int ntiles = t_end - t_beg;
    if (ntiles > 0) {
        // ---- loader: chunk pch = i*NTHR + tid -> tile row pch >> 3, physical 16-byte slot pch & 7 holding logical chunk
        // slot ^ ((row >> 1) & 7) of the row's 128-byte block
        const char *gB[NLD];
        int rowA[NLD], offA[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int pch = i * NTHR + tid;
            const int r = pch >> 3, slot = pch & 7;
            const int c = slot ^ ((r >> 1) & 7);
            rowA[i] = r; offA[i] = c << 4;
            gB[i] = p.q2 + ((int64_t)qt * T_ + r) * p.ldq2 + (c << 4);      // the query copy is padded to whole tiles
        }
        const int wave_chunk = wave * 1024;

        auto stage_load_part = [&](int stage, int tile, int kt, int i) {
            char *sA = smem + stage * STAGE;
            char *sB = sA + OPB;
            int64_t brow = (int64_t)tile * T_ + rowA[i];
            if (brow > p.n_rows - 1) brow = p.n_rows - 1;
            pk_glds16(p.bank2 + brow * p.ldb2 + kt * PK_ROWB + offA[i], sA + i * (NTHR * 16) + wave_chunk);
            pk_glds16(gB[i] + kt * PK_ROWB, sB + i * (NTHR * 16) + wave_chunk);
        };

        // fragment read offsets: row * 128 + (chunk ^ swz) * 16, chunk = 4 lo + 2 s + h for K step s (16 channels) of the stage
        const int swz = (lane >> 1) & 7;
        int foff[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int lo = 0; lo < 2; ++lo) foff[s][lo] = ((4 * lo + 2 * s + h) ^ swz) << 4;
        const int arow0 = (wm * 32 * MT + l31) * PK_ROWB;
        const int brow0 = (wn * 64 + l31) * PK_ROWB;

        f32x16 acc[MT][2];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

        const int total = ntiles * p.nkt;
#pragma unroll
        for (int i = 0; i < NLD; ++i) stage_load_part(0, t_beg, 0, i);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();

        int tile = t_beg, kt = 0, cur = 0;
        for (int it = 0; it < total; ++it) {
            int nkt_ = kt + 1, ntile = tile;
            if (nkt_ == p.nkt) { nkt_ = 0; ntile = tile + 1; }
            // branch-free prefetch: the very last stage re-fetches its own block into the idle buffer, which nobody reads
            const int ltile = (it + 1 < total) ? ntile : tile, lkt = (it + 1 < total) ? nkt_ : kt;

            const char *sA = smem + cur * STAGE;
            const char *sB = sA + OPB;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f16x8 ah[MT], al[MT], bh[2], bl[2];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    ah[m] = *(const f16x8 *)(sA + arow0 + m * 32 * PK_ROWB + foff[s][0]);
                    al[m] = *(const f16x8 *)(sA + arow0 + m * 32 * PK_ROWB + foff[s][1]);
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    bh[n] = *(const f16x8 *)(sB + brow0 + n * 32 * PK_ROWB + foff[s][0]);
                    bl[n] = *(const f16x8 *)(sB + brow0 + n * 32 * PK_ROWB + foff[s][1]);
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[n], acc[m][n], 0, 0, 0);
                // The whole next stage's LDS-DMA is issued during the FIRST K step, one request between every few of its MFMAs
                // (a global_load_lds costs 60-180 issue cycles: back to back after the barrier they idle the matrix pipe,
                // sim_topk_mfma.hip), so that the second K step's MFMAs cover the L2 round trip.  Round 3's first form issued half
                // of them behind the LAST MFMAs of the stage and waited for them at once: a third of the wave cycles parked at
                // that s_waitcnt (SQ_WAIT_ANY / SQ_WAVE_CYCLES = 0.33, profiles/r03_v6_pmc_match_summary.json).
                if (LDM == 0) {
                    if (DBG != 1) {
#pragma unroll
                        for (int i = s * (NLD / 2); i < (s + 1) * (NLD / 2); ++i) stage_load_part(cur ^ 1, ltile, lkt, i);
                    }
                } else if (s == 0) {
                    if (DBG != 1) {
#pragma unroll
                        for (int i = 0; i < NLD; ++i) stage_load_part(cur ^ 1, ltile, lkt, i);
                    }
                    constexpr int G = MT * 2 * 3;                    // MFMAs of a K step: 24 | 12
                    constexpr int NL = 2 * NLD;                      // LDS-DMA requests of a stage: 8
                    constexpr int PER = G / NL > 0 ? G / NL : 1;
#pragma unroll
                    for (int i = 0; i < NL; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);      // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);        // VMEM read (LDS-DMA)
                    }
                    if (G - PER * NL > 0) __builtin_amdgcn_sched_group_barrier(0x008, G - PER * NL > 0 ? G - PER * NL : 1, 0);
                }
            }

            if (kt == p.nkt - 1)
                pair_tile_epilogue<MT, KPL>(acc, lk, li, lim, qmul, p.invs, p.n_rows, tile * T_ + wm * 32 * MT + 4 * h);

            __builtin_amdgcn_s_waitcnt(0);       // next stage landed (vmcnt(0)), this stage's fragment reads done
            __syncthreads();
            cur ^= 1;
            kt = nkt_; tile = ntile;
        }
    }

    pair_block_merge<T_, KPL>(smem, lk, li, wn, wm, h, l31, tid, qt, seg, p);
}

// Measured and rejected (round 3, profiles/r03_v13_pair_pingpong_rejected.log; code removed): a "ping-pong" form of the 256 x 256
// kernel -- a K step as two phases per wave (R: its 12 fragment reads; C: its 24 MFMAs), raw s_barriers between phases, waves 4..7
// one barrier behind waves 0..3 for the whole K loop so that on every SIMD one wave reads while the other feeds the matrix pipe,
// ring of four 32 KB half-stages with counted vmcnt(4).  Correct both times (bit-identical results in every round, the suite green
// with it), and slower both times: with the LDS-DMA requests between the MFMAs of C 269 ms against 223 ms on the 100k x 100k launch
// (ONE wave issuing into the pipe: every request's 60-180-cycle issue slot idles it); with the requests in R 237 ms against 225 ms --
// a wave's non-MFMA issue work per K step (12 ds_read_b128, 4 LDS-DMA requests, waits, two barriers: ~1000 cycles) is longer than
// its partner's 768 cycles of MFMAs, so R bounds the phase, where the lockstep form's two waves fill each other's issue gaps at
// instruction granularity (matrix pipe busy 0.60).

// ---- query preparation: exact fp16 pairs of every query (times its own power-of-two scale), the factor that removes that
// scale from the keys, per-query row limits, per-tile max limit.  One workgroup per (padded) query row.
template <typename QS>
__global__ __launch_bounds__(256) void pair_prep_kernel(const QS *__restrict__ q, int64_t ldq, int nq, int dim, int kd,
                                                        char *__restrict__ q2, int64_t ldq2, float *__restrict__ qinvs,
                                                        const int64_t *__restrict__ row_limit, int n_rows,
                                                        int *__restrict__ lim, int *__restrict__ qt_maxlim, int tile) {
    const int row = blockIdx.x;
    __shared__ float s_max[4];
    __shared__ int s_fin[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool real = row < nq;
    float amax = 0.0f;
    bool finite = true;
    if (real)
        for (int c = threadIdx.x; c < dim; c += 256) {
            const float v = (float)q[(size_t)row * ldq + c];
            finite &= (v - v) == 0.0f;
            amax = fmaxf(amax, fabsf(v));
        }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    finite = __all(finite);
    if (lane == 0) { s_max[wave] = amax; s_fin[wave] = finite; }
    __syncthreads();
    amax = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    finite = s_fin[0] && s_fin[1] && s_fin[2] && s_fin[3];
    int e = 0;
    if (amax > 0.0f) (void)frexpf(amax, &e);
    // a query the scale cannot serve (non-finite entries, magnitudes outside [2^-100, 2^100]): NaN factor -> NaN keys ->
    // stage 2 cannot certify it -> exact float64 scan, like every other uncertified query
    const bool servable = finite && (amax == 0.0f || (e > -100 && e < 100));
    const float sc = (amax > 0.0f && servable) ? ldexpf(1.0f, 15 - e) : 1.0f;
    char *d2 = q2 + (size_t)row * ldq2;
    for (int c = 4 * threadIdx.x; c < kd; c += 1024) {
        unsigned hi[2], lo[2];
        float w[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) w[t] = (real && servable && c + t < dim) ? (float)q[(size_t)row * ldq + c + t] * sc : 0.0f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const __half2 hh = __floats2half2_rn(w[2 * t], w[2 * t + 1]);
            const float2 f = __half22float2(hh);
            const __half2 ll = __floats2half2_rn(w[2 * t] - f.x, w[2 * t + 1] - f.y);
            hi[t] = *(const unsigned *)&hh; lo[t] = *(const unsigned *)&ll;
        }
        char *blk = d2 + (c >> 5) * 128 + (c & 31) * 2;
        *(uint2 *)blk = make_uint2(hi[0], hi[1]);
        *(uint2 *)(blk + 64) = make_uint2(lo[0], lo[1]);
    }
    if (threadIdx.x == 0) {
        qinvs[row] = real ? (servable ? 1.0f / sc : NAN) : 0.0f;
        int l = 0;
        if (real) {
            int64_t v = row_limit ? row_limit[row] : n_rows;
            if (v > n_rows) v = n_rows;
            if (v < 0) v = 0;
            l = (int)v;
        }
        lim[row] = l;
        atomicMax(&qt_maxlim[row / tile], l);
    }
}

// rigorous bound on |pair key - exact key| / ||q|| (header comment); kd = padded dimension
double pair_err_bound(int kd) {
    const double u16 = 1.0 / 4194304.0;                        // 2^-22: value -> hi + lo
    const double rep = 2.0 * (u16 + 5.9604644775390625e-08);   // both operands (+ the query's float64 -> float32 rounding)
    const double dropped = u16 * (1.0 + 1.0 / 1024.0);
    const double accum = 1.002 * (3.0 * kd + 64.0) * 1.1920928955078125e-07;   // unit 2^-23: rounding or truncation
    const double keyr = 3.0 * 5.9604644775390625e-08;          // invs rounding + two multiplications
    const double floor_ = sqrt((double)kd) * 3.637978807091713e-12;            // sqrt(kd) 2^-38: subnormal-lo floor
    return 1.0625 * (rep + dropped + accum + keyr + floor_);
}

template <int T_, int MT, int KPL>
static int launch_pair(const PairArgs &a, int dbg, hipStream_t st) {
    constexpr int lds = 2 * 2 * T_ * PK_ROWB;
    static DeviceOnce once;
    int once_dev;
    if (once.todo(&once_dev)) {
        HIP_TRY(hipFuncSetAttribute((const void *)sim_topk_pair_kernel<T_, MT, KPL, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void *)sim_topk_pair_kernel<T_, MT, KPL, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void *)sim_topk_pair_kernel<T_, MT, KPL, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        once.done(once_dev);
    }
    const char *le = getenv("CSLAM_PAIR_LDM");
    const int ldm = (le && le[0] == '0') ? 0 : 1;
    const dim3 grid(a.nqt * a.nseg), blk(T_ * 2);
    if (dbg == 1) hipLaunchKernelGGL((sim_topk_pair_kernel<T_, MT, KPL, 1, 1>), grid, blk, lds, st, a);
    else if (ldm == 0) hipLaunchKernelGGL((sim_topk_pair_kernel<T_, MT, KPL, 0, 0>), grid, blk, lds, st, a);
    else hipLaunchKernelGGL((sim_topk_pair_kernel<T_, MT, KPL, 0, 1>), grid, blk, lds, st, a);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

int pair_stage1_launch(const PairArgs &a, int tile, int dbg, hipStream_t st) {
    return tile == 256 ? launch_pair<256, 4, 8>(a, dbg, st) : launch_pair<128, 2, 16>(a, dbg, st);
}

int pair_prep_launch(const void *d_q, int q_dtype, int64_t ldq, int nq, int dim, int kd, char *q2, int64_t ldq2, float *qinvs,
                     const int64_t *d_row_limit, int n_rows, int *lim, int *qtm, int nq_pad, int tile, hipStream_t st) {
    if (q_dtype == CSLAM_F32)
        hipLaunchKernelGGL(pair_prep_kernel<float>, dim3(nq_pad), dim3(256), 0, st, (const float *)d_q, ldq, nq, dim, kd, q2, ldq2,
                           qinvs, d_row_limit, n_rows, lim, qtm, tile);
    else
        hipLaunchKernelGGL(pair_prep_kernel<double>, dim3(nq_pad), dim3(256), 0, st, (const double *)d_q, ldq, nq, dim, kd, q2, ldq2,
                           qinvs, d_row_limit, n_rows, lim, qtm, tile);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}
