// sim_topk_mfma.hip -- batched all-pairs cosine similarity + top-k on gfx950 matrix cores.
//
// Replaces, for a batch of queries, the loop of cslam/nns_matching.py:55-61
//   for i in range(n): sim[i] = 1 - cosine(query, data[i]);  argsort(sim)[::-1][:k]
// (and its callers lcsm.py:45-47, 66, 75-76) without ever materialising the
// nq x n similarity matrix.
//
// Stage 1  sim_topk_mfma_kernel   S^T tile = Bank_tile x Query_tile^T in exact-f32 MFMA
//          (v_mfma_f32_32x32x2_f32), K streamed in 32-float steps through double-buffered LDS
//          filled by global_load_lds (16 B/lane, XOR-swizzled source so ds_read_b128 fragment
//          reads are bank-conflict free).  Queries are the MFMA "B" operand, so a lane's 16
//          accumulator registers all belong to ONE query: the running candidate list of that
//          query lives in that lane's registers and the epilogue is lane-local (no cross-lane
//          traffic, no score matrix in memory).  One workgroup = one (query tile, bank segment).
//          Two tile shapes (template): 128x128 (4 waves, 2 workgroups/CU) and 256x256 (8 waves,
//          1 workgroup/CU: half the L2->LDS traffic per flop, twice the prefetch distance).
// Stage 2  rescore_kernel         one wave per query: merge the per-segment candidate lists,
//          re-score the contenders in float64 with the reference formula, order them
//          exactly, and certify with a rigorous f32 error bound that no row outside the
//          candidate set can reach the k-th place.  Uncertified queries (never seen on
//          random data) are re-done by the exact scan kernel (bank.hip).
#include <stdlib.h>
#include "bank.h"
#include "sim_topk.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define TK 32           // floats per K step
#define KP SIM_KP       // merged candidate list length per (query, segment)

__device__ __forceinline__ void glds16(const float *g, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

struct MfmaArgs {
    const float *bank; int64_t ldb;      // bank rows, pitch in floats
    const float *invn; int n_rows;
    const float *q; int64_t ldq; int nq; // f32 queries (rows clamped to nq-1 by the loader)
    const int *lim;                      // [nqt*TN] visible-row limit per query (0 for padding)
    const int *qt_maxlim;                // [nqt]
    int nkt;                             // K steps = kd / 32
    int nqt, nseg, tps, n_btiles;
    float *part_key; int *part_idx;      // [nqt*TN][nseg][KP] merged candidates, sorted
    const int *item_map;                 // [nqt*nseg] (query tile << 16 | segment) in patch-major order
    float *part_bound;                   // [nqt*TN][nseg] upper bound of the key of every row of the
                                         // segment that is NOT in the merged list (-inf: none dropped)
};

// T_ = tile edge (bank rows = queries per tile), MT = 32-row MFMA tiles per wave along the bank
// axis (wave tile = 32*MT x 64), KPL = per-lane candidate list length.
// Waves: 2 along the bank axis x (T_/64) along the query axis.
// DBG != 0 are TIMING-ONLY ablations (wrong results): 1 = no global loads after the first K step,
// 2 = no per-step wait/barrier.  Selected with CSLAM_MFMA_DBG; never used by the product path.
template <int T_, int MT, int KPL, int DBG, int ILV>
__global__ __launch_bounds__(T_ * 2, 2) void sim_topk_mfma_kernel(MfmaArgs p) {
    constexpr int NTHR = T_ * 2;                 // 256 or 512 threads
    constexpr int NWN = T_ / 64;                 // waves along the query axis
    constexpr int OPB = T_ * TK * 4;             // bytes of one operand tile in LDS
    constexpr int STAGE = 2 * OPB;
    constexpr int NLD = T_ * 8 / NTHR;           // 16-byte chunks per thread per operand (= 4)
    static_assert(T_ == 64 * MT, "wave tile must cover half the bank tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int h = lane >> 5, l31 = lane & 31;

    // XCD-aware work-item mapping: the hardware places block b on XCD b % 8; give each XCD a
    // contiguous run of the item list.  The list is patch-major over the (query tile, segment)
    // grid (host-built table), so the ~32/64 workgroups resident on an XCD form a roughly square
    // patch: ~sqrt as many distinct bank/query streams per L2 as a row of the grid would give
    // (speed only; any mapping is correct).
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

    // per-lane candidate lists for the two query columns this lane owns
    float lk[2][KPL]; int li[2][KPL];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int j = 0; j < KPL; ++j) { lk[n][j] = -INFINITY; li[n][j] = -1; }
    int lim[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) lim[n] = p.lim[qt * T_ + wn * 64 + n * 32 + l31];

    const int ntiles = t_end - t_beg;
    if (ntiles > 0) {
        // ---- loader: thread handles LDS chunks pch = i*NTHR + tid; chunk pch -> tile row pch>>3,
        // physical 16-B slot pch&7 holding logical chunk slot ^ ((row>>1)&7)
        const float *gB[NLD];
        int rowA[NLD], colc[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int pch = i * NTHR + tid;
            int r = pch >> 3, slot = pch & 7;
            int c = slot ^ ((r >> 1) & 7);
            rowA[i] = r; colc[i] = c * 4;
            int64_t qrow = (int64_t)qt * T_ + r;
            if (qrow > p.nq - 1) qrow = p.nq - 1;
            gB[i] = p.q + qrow * p.ldq + c * 4;
        }
        const int wave_chunk = wave * 1024;          // this wave's 1 KiB slice of each NTHR*16-byte group

        auto stage_load_part = [&](int stage, int tile, int kt, int i) {   // 2 LDS-DMA instructions
            char *sA = smem + stage * STAGE;
            char *sB = sA + OPB;
            int64_t brow = (int64_t)tile * T_ + rowA[i];
            if (brow > p.n_rows - 1) brow = p.n_rows - 1;
            const float *ga = p.bank + brow * p.ldb + (kt * TK + colc[i]);
            glds16(ga, sA + i * (NTHR * 16) + wave_chunk);
            glds16(gB[i] + kt * TK, sB + i * (NTHR * 16) + wave_chunk);
        };
        auto stage_load = [&](int stage, int tile, int kt) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) stage_load_part(stage, tile, kt, i);
        };

        // ---- fragment read offsets (bytes) within a tile: row*128 + ((2j+h) ^ swz)*16
        const int swz = (lane >> 1) & 7;
        int foff[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) foff[j] = (((2 * j + h) ^ swz) << 4);
        const int arow0 = (wm * 32 * MT + l31) * 128;
        const int brow0 = (wn * 64 + l31) * 128;

        f32x16 acc[MT][2];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

        const int total = ntiles * p.nkt;
        stage_load(0, t_beg, 0);
        __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0)
        __syncthreads();

        int tile = t_beg, kt = 0, cur = 0;
        for (int it = 0; it < total; ++it) {
            // prefetch the next K step (possibly of the next bank tile) into the other stage
            int nkt_ = kt + 1, ntile = tile;
            if (nkt_ == p.nkt) { nkt_ = 0; ntile = tile + 1; }
            const bool pf = (DBG != 1) && (it + 1 < total);
            if (ILV == 0 && pf) stage_load(cur ^ 1, ntile, nkt_);
            // interleaved form: branch-free (a branch would split the scheduling region); the very last
            // step re-fetches its own tile into the idle stage, which nobody reads
            const int ltile = (it + 1 < total) ? ntile : tile, lkt = (it + 1 < total) ? nkt_ : kt;

            const char *sA = smem + cur * STAGE;
            const char *sB = sA + OPB;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 a[MT], b[2];
#pragma unroll
                for (int m = 0; m < MT; ++m) a[m] = *(const f32x4 *)(sA + arow0 + m * 32 * 128 + foff[j]);
#pragma unroll
                for (int n = 0; n < 2; ++n) b[n] = *(const f32x4 *)(sB + brow0 + n * 32 * 128 + foff[j]);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][t], b[n][t], acc[m][n], 0, 0, 0);
                if (ILV == 1) {
                    // one quarter of the next stage's LDS-DMA per K group, issued BETWEEN this group's MFMAs:
                    // a global_load_lds costs ~60-180 issue cycles; eight of them back to back at the top of
                    // the step (both waves of a SIMD do that right after the barrier) leave the matrix pipe
                    // idle ~1.3k cycles per step.  Behind an MFMA the issue hides in its 64-cycle shadow.
                    if (DBG != 1) stage_load_part(cur ^ 1, ltile, lkt, j);
                    constexpr int G = MT * 2 * 4;                 // MFMAs per K group (16 or 32)
                    __builtin_amdgcn_sched_group_barrier(0x008, G / 4, 0);      // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);          // VMEM read (LDS-DMA)
                    __builtin_amdgcn_sched_group_barrier(0x008, G / 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, G / 2, 0);
                }
            }

            if (kt == p.nkt - 1) {
                // ---- tile epilogue: lane-local candidate update, then clear the accumulators
                const int row_base = tile * T_ + wm * 32 * MT + 4 * h;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    float inv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int row = row_base + m * 32 + (r & 3) + 8 * (r >> 2);
                        inv[r] = p.invn[row < p.n_rows ? row : p.n_rows - 1];
                    }
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        f32x16 keys;
                        bool any = false;
                        const float thr = lk[n][KPL - 1];
                        const int rel_lim = lim[n] - (row_base + m * 32);   // row < lim  <=>  rowoff < rel_lim
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float kx = acc[m][n][r] * inv[r];
                            keys[r] = kx;
                            bool ok = ((r & 3) + 8 * (r >> 2)) < rel_lim;
                            any |= ok && !(kx <= thr);     // NaN passes (ranks first)
                            acc[m][n][r] = 0.0f;
                        }
                        if (__any(any)) {
#pragma unroll 1
                            for (int r = 0; r < 16; ++r) {
                                float ck = keys[r];        // uniform dynamic index -> s_set_gpr_idx
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

            if (DBG != 2) {
                __builtin_amdgcn_s_waitcnt(0);   // next stage landed (vmcnt(0))
                __syncthreads();
            }
            cur ^= 1;
            kt = nkt_; tile = ntile;
        }
    }

    // ---- block merge: 4 lists per query (2 row-halves of the wave x 2 waves along the bank axis)
    // -> the best KP of them, plus the bound on everything dropped: a full lane list may have
    // dropped rows no better than its last key; the merge drops entries no better than its last.
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
        size_t o = ((size_t)(qt * T_ + tid) * p.nseg + seg) * KP;
        float bound = -INFINITY;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (i0[s * KPL + KPL - 1] >= 0) bound = fmaxf(bound, k0[s * KPL + KPL - 1]);   // full lane list
        for (int j = 0; j < KP; ++j) {
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
        // heads left after taking KP entries are the best entries the merge dropped
        float c0 = p0 < KPL ? k0[p0] : -INFINITY, c1 = p1 < KPL ? k0[KPL + p1] : -INFINITY;
        float c2 = p2 < KPL ? k0[2 * KPL + p2] : -INFINITY, c3 = p3 < KPL ? k0[3 * KPL + p3] : -INFINITY;
        bound = fmaxf(fmaxf(bound, fmaxf(c0, c1)), fmaxf(c2, c3));
        p.part_bound[(size_t)(qt * T_ + tid) * p.nseg + seg] = bound;
    }
}

// ---- query preparation: float32 padded copy, per-query row limits, per-tile max limit ----
template <typename QS>
__global__ void mfma_prep_kernel(const QS *__restrict__ q, int64_t ldq, int nq, int dim, int kd, int ld,
                                 float *__restrict__ q32, const int64_t *__restrict__ row_limit,
                                 int n_rows, int *__restrict__ lim, int *__restrict__ qt_maxlim, int nq_pad,
                                 int tile) {
    const int row = blockIdx.x;
    if (row >= nq_pad) return;
    if (q32) {
        for (int c = threadIdx.x; c < kd; c += blockDim.x)
            q32[(size_t)row * ld + c] = (row < nq && c < dim) ? (float)q[(size_t)row * ldq + c] : 0.0f;
    }
    if (threadIdx.x == 0) {
        int l = 0;
        if (row < nq) {
            int64_t v = row_limit ? row_limit[row] : n_rows;
            if (v > n_rows) v = n_rows;
            if (v < 0) v = 0;
            l = (int)v;
        }
        lim[row] = l;
        atomicMax(&qt_maxlim[row / tile], l);
    }
}

// ---- stage 2: merge, float64 re-score, exact order, certificate ---------------------------
template <typename QS>
__global__ __launch_bounds__(256) void rescore_kernel(
    const float *__restrict__ bank, int64_t pitch, int ld, const double *__restrict__ vv,
    const QS *__restrict__ q, int64_t ldq, int dim, int nq,
    const float *__restrict__ part_key, const int *__restrict__ part_idx,
    const float *__restrict__ part_bound, int nseg,
    int k, double err_bound,
    int64_t *__restrict__ out_idx, double *__restrict__ out_sim, int32_t *__restrict__ out_cnt,
    int *__restrict__ flag_list, int *__restrict__ flag_count,
    const int *__restrict__ qt_nseg, const int *__restrict__ qt_segoff, int seg_tile) {
    const int lane = threadIdx.x & 63;
    const int qn = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qn >= nq) return;
    const QS *qp = q + (size_t)qn * ldq;
    // lists of this query: `nseg` of them from list number `l0` on.  Uniform layout [nq][nseg] (one workgroup per work item), or
    // the persistent stage's per-query-tile counts (sim_topk_ring.hip: a tile's walk may be cut into several runs)
    size_t l0 = (size_t)qn * nseg;
    if (qt_nseg) {
        const int qt = qn / seg_tile;
        nseg = qt_nseg[qt];
        l0 = (size_t)qt_segoff[qt] * seg_tile + (size_t)(qn - qt * seg_tile) * nseg;
    }

    // uu = q.q (float64)
    double uu = 0.0;
    for (int c = lane; c < dim; c += 64) { double x = (double)qp[c]; uu += x * x; }
    uu = wave_allreduce_sum(uu);
    const double qnorm = sqrt(uu);

    // 1. t32 = upper bound on the f32 key of every row NOT kept: the per-segment bounds of stage 1 ...
    double t32 = -INFINITY;
    for (int s = lane; s < nseg; s += 64) {
        double bnd = (double)part_bound[l0 + s];
        t32 = bnd > t32 ? bnd : t32;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { double o = __shfl_xor(t32, off, 64); t32 = o > t32 ? o : t32; }
    //    ... and whatever the merge of the per-segment lists below drops.
    WaveList cand; cand.init();
    const float *pk = part_key + l0 * KP;
    const int *pi = part_idx + l0 * KP;
    const int total = nseg * KP;
    for (int base = 0; base < total; base += 64) {
        int e = base + lane;
        double ck = e < total ? (double)pk[e] : -INFINITY;
        int ci = e < total ? pi[e] : -1;
        // only entries that can enter the current 64-entry list need the serial insert
        double lastk = cand.key_at(63);
        int lasti = cand.idx_at(63);
        bool enters = ci >= 0 && (lasti < 0 || ranks_before(ck, ci, lastk, lasti));
        double dropk = (ci >= 0 && !enters) ? ck : -INFINITY;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { double o = __shfl_xor(dropk, off, 64); dropk = o > dropk ? o : dropk; }
        t32 = dropk > t32 ? dropk : t32;
        unsigned long long mask = __ballot(enters);
        while (mask) {
            int src = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            double k2 = __shfl(ck, src, 64);
            int i2 = __shfl(ci, src, 64);
            lastk = cand.key_at(63);
            lasti = cand.idx_at(63);
            if (lasti >= 0) {       // list full: something falls off
                if (ranks_before(k2, i2, lastk, lasti)) t32 = lastk > t32 ? lastk : t32;
                else { t32 = k2 > t32 ? k2 : t32; continue; }
            }
            cand.insert(k2, i2, lane);
        }
    }
    unsigned long long cmask = __ballot(cand.idx >= 0);
    const int ncand = __popcll(cmask);
    // contenders: the candidates (at most the 64 the merged list holds) whose f32 key is within 2e of the k-th f32 key.
    // (Round 2 capped them at KP = 16: enough for the f32-input stage's bound, 2e = 5e-4 at 4096-D; with the fp16-pair stage's
    // 2e = 3e-3 a 100k-row random bank has ~4 rows in that window on average and 0.06 % of the queries had more than 11 --
    // each of them a needless trip through the exact-scan fallback.)
    const double e_abs = err_bound * qnorm;                 // error bound in key units (dot / ||b||)
    const int kth = (k - 1 < ncand - 1) ? k - 1 : ncand - 1;
    const double ck_k = ncand > 0 ? cand.key_at(kth > 0 ? kth : 0) : -INFINITY;
    int nres = 0;
    {
        bool want = cand.idx >= 0 && (cand.key >= ck_k - 2.0 * e_abs || !(e_abs == e_abs));
        unsigned long long wm_ = __ballot(want);
        nres = __popcll(wm_);            // candidates are sorted, so `want` lanes are a prefix
        if (nres < ncand) {              // everything after the prefix is treated as "not kept"
            double nk = cand.key_at(nres);
            t32 = nk > t32 ? nk : t32;
        }
    }

    // 2. exact float64 scores of the contenders, exact order
    WaveList ex; ex.init();
    const int nchunk = ld >> 2;
    for (int c0 = 0; c0 < nres; ++c0) {
        const int row = cand.idx_at(c0);
        const float4 *rp = (const float4 *)(bank + (size_t)row * pitch);
        double acc = 0.0;
#pragma unroll 4
        for (int c = lane; c < nchunk; c += 64) {
            float4 b = rp[c];
            const int col = c * 4;
            double q0 = col < dim ? (double)qp[col] : 0.0;
            double q1 = col + 1 < dim ? (double)qp[col + 1] : 0.0;
            double q2 = col + 2 < dim ? (double)qp[col + 2] : 0.0;
            double q3 = col + 3 < dim ? (double)qp[col + 3] : 0.0;
            acc += (double)b.x * q0;
            acc += (double)b.y * q1;
            acc += (double)b.z * q2;
            acc += (double)b.w * q3;
        }
        double uv = wave_allreduce_sum(acc);
        double key = rank_key(sim_from_dots(uv, uu, vv[row]));
        ex.insert(key, row, lane);
    }

    // 3. certificate: every row outside the contenders has f32 key <= t32, hence exact
    //    normalised score <= t32/||q|| + err_bound; the k-th exact score must beat that.
    const int nout = k < nres ? k : nres;
    bool certified;
    if (t32 == -INFINITY) certified = true;                       // nothing was left out
    else if (nres < k) certified = false;
    else {
        // sim = clamp(ratio, -1, 1) is monotone in ratio; rows left out have ratio <= bound
        double xk = ex.key_at(k - 1);
        double bound = t32 / qnorm + err_bound;
        if (bound < -1.0) bound = -1.0;
        certified = xk > bound;                                   // false when anything is NaN
    }

    if (lane < k) {
        size_t o = (size_t)qn * k + lane;
        bool v = lane < nout && ex.idx >= 0;
        out_idx[o] = v ? (int64_t)ex.idx : -1;
        out_sim[o] = (v && ex.key != INFINITY) ? ex.key : NAN;
    }
    if (lane == 0) {
        out_cnt[qn] = nout;
        if (!certified) { int s = atomicAdd(flag_count, 1); flag_list[s] = qn; }
    }
}

template <int T_, int MT, int KPL>
static int launch_stage1(const MfmaArgs &a, int dbg, int ilv, hipStream_t st) {
    constexpr int lds = 2 * 2 * T_ * TK * 4;
    static DeviceOnce once;                                             // one latch per template instance and device
    int once_dev;
    if (once.todo(&once_dev)) {
        HIP_TRY(hipFuncSetAttribute((const void *)sim_topk_mfma_kernel<T_, MT, KPL, 0, 0>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void *)sim_topk_mfma_kernel<T_, MT, KPL, 0, 1>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#ifdef CSLAM_ABLATIONS
        HIP_TRY(hipFuncSetAttribute((const void *)sim_topk_mfma_kernel<T_, MT, KPL, 1, 0>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void *)sim_topk_mfma_kernel<T_, MT, KPL, 2, 0>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#endif
        once.done(once_dev);
    }
    const dim3 grid(a.nqt * a.nseg), blk(T_ * 2);
#ifdef CSLAM_ABLATIONS
    if (dbg == 1) hipLaunchKernelGGL((sim_topk_mfma_kernel<T_, MT, KPL, 1, 0>), grid, blk, lds, st, a);
    else if (dbg == 2) hipLaunchKernelGGL((sim_topk_mfma_kernel<T_, MT, KPL, 2, 0>), grid, blk, lds, st, a);
    else
#endif
    if (ilv) hipLaunchKernelGGL((sim_topk_mfma_kernel<T_, MT, KPL, 0, 1>), grid, blk, lds, st, a);
    else hipLaunchKernelGGL((sim_topk_mfma_kernel<T_, MT, KPL, 0, 0>), grid, blk, lds, st, a);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

int mfma_search_enqueue(cslam_bank *b, const void *d_q, int q_dtype, int64_t ldq, int64_t nq, int k,
                        const int64_t *d_row_limit, int64_t *d_out_idx, double *d_out_sim,
                        int32_t *d_out_cnt, hipStream_t st) {
    static int dbg = -1, tile_env = -1;
    // read on every call (tests and A/B runs switch it inside one process): "f32" = the f32-input MFMA candidate stage
    // (round 1-2's kernel), "pair" = exact fp16 pairs, three products (round 3's); default ("h1") = ONE fp16 product on the hi
    // halves (round 4, sim_topk_pair.hip NPROD = 1: same rigorous bound as the pair stage at a third of the matrix work)
    const char *s1 = getenv("CSLAM_MFMA_STAGE1");
    int nprod = (s1 && s1[0] == 'f') ? 0 : ((s1 && s1[0] == 'p') ? 3 : 1);
    // Clustered banks (many near-duplicates inside the fp16 stages' re-scoring window, 2 x 1.57e-3 at 4096-D) overflow the 64
    // contenders stage 2 re-scores and send their queries through the exact scan: results stay exact, throughput does not.  When
    // the last search left more than 1/32 of its queries uncertified, the next searches of this bank take the f32-input stage
    // (window 2 x 2.6e-4: six times fewer contenders), then the fp16 stage is tried again: 8 searches the first time, twice as many
    // after every retry that overflows again (up to 1024: a permanently clustered bank pays one exact-scan fallback per 1024
    // searches), back to 8 once a retry certifies.  An explicit CSLAM_MFMA_STAGE1 switches this off (stage_pinned in
    // cslam_bank_last_stage).
    b->stage_pinned = s1 != nullptr;
    if (!s1 && b->f32_backoff > 0) { nprod = 0; --b->f32_backoff; }
    b->last_nprod = nprod;
    const int stage1_pair = nprod != 0;
    if (dbg < 0) {
        dbg = 0;
#ifdef CSLAM_ABLATIONS
        const char *v = getenv("CSLAM_MFMA_DBG");       // timing-only ablations (wrong results), see the kernel: never in the default build
        dbg = v ? atoi(v) : 0;
        if (dbg < 0 || dbg > 4095) dbg = 0;          // 3..5: the persistent stage only (sim_topk_ring.hip)
#endif
        const char *t = getenv("CSLAM_MFMA_TILE");      // 128 | 256 (default chosen below)
        tile_env = t ? atoi(t) : 0;
    }
    const int ld = b->ld, kd = b->kd;
    // tile shape: 256x256 halves the operand traffic per flop and is faster from nq = 512 upwards
    // (measured at 100k rows: 76.6 vs 73.9 % of peak at nq = 1024, 87.1 vs 80.5 % at 4096); below that
    // the 128x128 tile wastes less query padding
    int tile = (nq > 256 && b->n >= 1024) ? 256 : 128;
    if (tile_env == 128 || tile_env == 256) tile = tile_env;
    const int nqt = (int)ceil_div64(nq, tile);
    const int nq_pad = nqt * tile;
    ARG_CHECK(nqt < 32768, "too many queries for one call (item map packs the query tile in 15 bits): split the batch");
    const int n_btiles = (int)ceil_div64(b->n, tile);
    // Segment count: all work items take the same time and the chip holds S workgroups, so the
    // grid runs in ceil(T/S) rounds; pick the split whose last round is fullest, discounting the
    // fixed per-item cost (prologue, list merge ~ 0.3 tile-times) and preferring fewer segments
    // (shorter candidate merge in rescore_kernel).
    const int slots = (tile == 256 ? 1 : 2) * b->num_cu;
    int nseg = 1, tps = n_btiles;
    // The one-product stage on 256 x 256 tiles runs as a persistent kernel on a static XCD-aligned schedule (sim_topk_ring.hip);
    // variant bit 0 = tile-start rendezvous of a patch, bit 1 = static priority for the younger waves.
    bool ring = nprod == 1 && tile == 256 && b->num_cu >= 8;
    int ring_variant = 0;
#ifdef CSLAM_ABLATIONS
    if (const char *r = getenv("CSLAM_MFMA_RING")) {     // measurement build: -1 = the one-workgroup-per-item kernel, 0..3 = variant
        const int v = atoi(r);
        if (v < 0) ring = false; else ring_variant = v & 15;
    }
#endif
    RingSchedule *rs = nullptr;
    bool want_xcc = false;
    int flow_w = 3, flow_bq = 0, flow_bb = 0;
#ifdef CSLAM_ABLATIONS
    want_xcc = getenv("CSLAM_RING_XCC") != nullptr;
    if (const char *e = getenv("CSLAM_RING_FLOW_W")) (void)sscanf(e, "%d,%d,%d", &flow_w, &flow_bq, &flow_bb);
#endif
    if (ring) {
        const int n_xcd = b->num_cu % 8 == 0 ? 8 : 1;
        const int wpx = b->num_cu / n_xcd;
        rs = &b->ring_sched;
        if (rs->nqt != nqt || rs->n_rows != (int)b->n || rs->n_xcd != n_xcd || rs->wpx != wpx) {
            ring_schedule_build(*rs, nqt, (int)b->n, n_xcd, wpx);
            // the packed candidate lists address a task's rows with 13 bits (sim_topk_pair_dev.h): the schedule cuts longer walks
            for (const RingTask &t : rs->tasks) ARG_CHECK(t.n_tiles <= 128, "internal: a ring task longer than 128 tiles");
        }
        nseg = 0;
        for (int v : rs->qt_nseg) nseg = v > nseg ? v : nseg;       // reported; the lists have per-query-tile counts
    }
    if (!ring) {
        double best = -1.0;
        int max_seg = (int)ceil_div64(slots, nqt);      // few query tiles: split finer to fill the chip
        if (max_seg < 64) max_seg = 64;
        if (max_seg > n_btiles) max_seg = n_btiles;
        for (int s = 1; s <= max_seg; ++s) {
            int t = (int)ceil_div64(n_btiles, s);
            int se = (int)ceil_div64(n_btiles, t);
            if (se != s) continue;
            double T = (double)nqt * se;
            double rounds = ceil(T / slots);
            // uneven segments (last one shorter) let workgroups that share a bank stream or a
            // query tile drift apart in time and lose their L2/MALL sharing: measured 82.8 % of
            // peak with 15 x 27-tile segments over 391 tiles vs 87.5 % with even splits
            double balance = (double)n_btiles / ((double)se * t);
            double eff = T / (rounds * slots) * balance * balance * ((double)t / (t + 0.3)) - 0.0005 * se;
            if (nprod == 1 && tile == 256) {
                // The one-product stage on 256 x 256 tiles, refitted on forced segment counts (profiles/r04_v71_match_nseg_sweep.log:
                // 100k rows x 16 384 / 49 152 / 100 000 queries, 1 ... 23 segments): a launch takes rounds x (t + R) tile-times with
                // R = min(8, 0.3 t) (at least 4 over several rounds) -- every round of work items costs several tile-times beyond its tiles (49 152 queries: 4
                // segments = 3 rounds of 98 tiles 35.8 ms, 8 = 6 rounds of 49 tiles 38.5, 17 = 13 rounds of 23 tiles 41.9), which the
                // 0.3 above (fitted on the f32 stage, whose tile-time is 16 x longer) does not see.  At least 4 segments once the
                // chip is full: fewer leave too few candidates per query for the certificate (2 segments: uncertified queries).
                double R = fmin(8.0, 0.3 * t);
                if (rounds > 1.0 && R < 4.0) R = 4.0;          // short items over many rounds were not measured: kept out of reach
                const double cost = rounds * (t + R) / (balance * balance) * (1.0 + 0.0005 * se);
                eff = (se < 4 && T >= slots) ? -1.0 : 1.0e6 / cost;
            }
            if (eff > best) { best = eff; nseg = se; tps = t; }
        }
    }

    // queries are used in place only when their pitch is not L2-set-aliasing (see bank.hip)
    const bool pair = stage1_pair != 0;
    if (nprod == 3) { int rcp = bank_pairs_ensure(b, st); if (rcp) return rcp; }
    const int64_t ldq2 = nprod == 1 ? b->ldh : b->ld2;          // the query copy has the layout (and pitch) of the bank copy
    const int kq2 = nprod == 1 ? b->kh : kd;
    const bool direct = !pair && q_dtype == CSLAM_F32 && b->dim == kd && (ldq % 4 == 0) && (ldq % 256 != 0) &&
                        (((uintptr_t)d_q) % 16 == 0);
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = (size_t)round_up64((int64_t)(off + bytes), 256); return o; };
    size_t o_q32 = carve(direct ? 0 : (pair ? (size_t)nq_pad * ldq2 : (size_t)nq_pad * ld * 4));
    size_t o_qs = carve(pair ? (size_t)nq_pad * 4 * (ring ? 3 : 1) : 0);        // 1 / s_query (+ the persistent stage's key factor and unit)
    size_t o_lim = carve((size_t)nq_pad * 4);
    size_t o_qtm = carve((size_t)nqt * 4);
    const size_t n_lists = ring ? (size_t)rs->total_lists * tile : (size_t)nq_pad * nseg;
    size_t o_pk = carve(n_lists * KP * 4);
    size_t o_pi = carve(n_lists * KP * 4);
    size_t o_pb = carve(n_lists * 4);
    size_t o_fl = carve((size_t)nq * 4);
    size_t o_fc = carve(256);
    size_t o_im = carve(ring ? 0 : (size_t)nqt * nseg * 4);
    const size_t o_rb = carve(ring ? rs->blob.size() : 0);           // the schedule's four tables, one upload
    const size_t o_rt = o_rb + (ring ? rs->off_tasks : 0), o_ro = o_rb + (ring ? rs->off_task_off : 0);
    const size_t o_rn = o_rb + (ring ? rs->off_qt_nseg : 0), o_rs = o_rb + (ring ? rs->off_qt_segoff : 0);
    size_t o_ry = carve(ring ? (size_t)rs->n_xcd * 32 * 4 : 0);
    size_t o_rx = carve(ring && want_xcc ? (size_t)b->num_cu * 4 : 0);
    size_t o_rz = carve(ring && want_xcc ? ((size_t)b->num_cu * 64 + 8 * 48 * 2 + 8) * 8 : 0);
    int rc = bank_ws_reserve(b, 0, off);
    if (rc) return rc;
    char *ws = b->ws[0];
    float *q32 = direct ? nullptr : (float *)(ws + o_q32);
    int *lim = (int *)(ws + o_lim);
    int *qtm = (int *)(ws + o_qtm);
    float *part_key = (float *)(ws + o_pk);
    int *part_idx = (int *)(ws + o_pi);
    float *part_bound = (float *)(ws + o_pb);
    int *flag_list = (int *)(ws + o_fl);
    int *flag_count = (int *)(ws + o_fc);
    int *item_map = (int *)(ws + o_im);
    if (ring) {
        HIP_TRY(hipMemcpyAsync(ws + o_rb, rs->blob.data(), rs->blob.size(), hipMemcpyHostToDevice, st));
        // (the progress lines of the flow control: measurement-build variants only -- the default launch passes prog = nullptr)
        if ((ring_variant & 13) && rs->wpx <= 32) HIP_TRY(hipMemsetAsync(ws + o_ry, 0, (size_t)rs->n_xcd * 32 * 4, st));
    } else {
        // patch-major order of the (query tile, segment) grid; patches of a x bseg items ~ the number
        // of workgroups resident per XCD
        const int per_xcd = slots / 8 > 0 ? slots / 8 : 1;
        int bseg = (int)floor(sqrt((double)per_xcd) + 0.5);
        if (bseg > nseg) bseg = nseg;
        if (bseg < 1) bseg = 1;
        int aq = (int)ceil_div64(per_xcd, bseg);
        if (aq < 1) aq = 1;
        if (b->item_map_host.size() != (size_t)nqt * nseg || b->item_map_key[0] != nqt || b->item_map_key[1] != nseg ||
            b->item_map_key[2] != aq || b->item_map_key[3] != bseg) {
            b->item_map_host.clear();
            b->item_map_host.reserve((size_t)nqt * nseg);
            for (int q0 = 0; q0 < nqt; q0 += aq)
                for (int s0 = 0; s0 < nseg; s0 += bseg)
                    for (int qq = q0; qq < q0 + aq && qq < nqt; ++qq)
                        for (int sg = s0; sg < s0 + bseg && sg < nseg; ++sg)
                            b->item_map_host.push_back((qq << 16) | sg);
            b->item_map_key[0] = nqt; b->item_map_key[1] = nseg; b->item_map_key[2] = aq; b->item_map_key[3] = bseg;
        }
        HIP_TRY(hipMemcpyAsync(item_map, b->item_map_host.data(), (size_t)nqt * nseg * 4, hipMemcpyHostToDevice, st));
    }

    HIP_TRY(hipMemsetAsync(qtm, 0, (size_t)nqt * 4, st));
    HIP_TRY(hipMemsetAsync(flag_count, 0, 4, st));
    if (pair) {
        rc = pair_prep_launch(d_q, q_dtype, ldq, (int)nq, b->dim, kq2, nprod, ws + o_q32, ldq2, (float *)(ws + o_qs), d_row_limit,
                              (int)b->n, lim, qtm, nq_pad, tile, ring ? (float *)(ws + o_qs) + nq_pad : nullptr, st);
        if (rc) return rc;
    } else if (q_dtype == CSLAM_F32)
        hipLaunchKernelGGL(mfma_prep_kernel<float>, dim3(nq_pad), dim3(256), 0, st, (const float *)d_q, ldq,
                           (int)nq, b->dim, kd, ld, q32, d_row_limit, (int)b->n, lim, qtm, nq_pad, tile);
    else
        hipLaunchKernelGGL(mfma_prep_kernel<double>, dim3(nq_pad), dim3(256), 0, st, (const double *)d_q, ldq,
                           (int)nq, b->dim, kd, ld, q32, d_row_limit, (int)b->n, lim, qtm, nq_pad, tile);
    if (!pair) HIP_TRY(hipGetLastError());

    MfmaArgs a;
    a.bank = b->rows; a.ldb = ld; a.invn = b->invn; a.n_rows = (int)b->n;
    a.q = direct ? (const float *)d_q : q32; a.ldq = direct ? ldq : ld; a.nq = direct ? (int)nq : nq_pad;
    a.lim = lim; a.qt_maxlim = qtm; a.nkt = kd / TK;
    a.nqt = nqt; a.nseg = nseg; a.tps = tps; a.n_btiles = n_btiles;
    a.part_key = part_key; a.part_idx = part_idx; a.part_bound = part_bound; a.item_map = item_map;
    if (b->ev_valid) HIP_TRY(hipEventRecord(b->ev0, st));
    // interleaved LDS-DMA issue: +3.4 points of peak on the 256 tile (one workgroup per CU: both waves of
    // a SIMD used to issue their 8 loads together right after the barrier); -0.5 on the 128 tile, whose
    // two independent workgroups per CU already overlap each other's issue slots
    const int ilv = tile == 256 ? 1 : 0;
    if (ring) {
        RingArgs ra;
        ra.bank2 = b->rowsh; ra.ldb2 = ldq2; ra.invs = b->invs; ra.n_rows = (int)b->n;
        ra.q2 = ws + o_q32; ra.ldq2 = ldq2; ra.qinvs = (const float *)(ws + o_qs) + nq_pad; ra.qunit = (const float *)(ws + o_qs) + 2 * nq_pad;
        ra.lim = lim; ra.qt_maxlim = qtm; ra.nkt = b->kh / 64;
        ra.nqt = nqt; ra.sb = rs->sb; ra.n_xcd = rs->n_xcd; ra.wpx = rs->wpx;
        ra.tasks = (const RingTask *)(ws + o_rt); ra.task_off = (const int *)(ws + o_ro);
        ra.qt_nseg = (const int *)(ws + o_rn); ra.qt_segoff = (const int *)(ws + o_rs);
        ra.part_key = part_key; ra.part_idx = part_idx; ra.part_bound = part_bound;
        ra.prog = ((ring_variant & 13) && rs->wpx <= 32) ? (int *)(ws + o_ry) : nullptr; ra.flow_w = flow_w; ra.flow_bias_q = flow_bq; ra.flow_bias_b = flow_bb;
        ra.xcc_out = want_xcc ? (int *)(ws + o_rx) : nullptr;
        ra.trace_out = want_xcc ? (long long *)(ws + o_rz) : nullptr;
        if (want_xcc) HIP_TRY(hipMemsetAsync(ws + o_rz, 0, ((size_t)b->num_cu * 64 + 8 * 48 * 2 + 8) * 8, st));
        rc = ring_stage1_launch(ra, ring_variant, dbg, st);
        if (want_xcc && rc == CSLAM_OK) {            // measurement build: print the placement (block b is assumed on XCD b % n_xcd)
            std::vector<int> hx(b->num_cu);
            HIP_TRY(hipStreamSynchronize(st));
            HIP_TRY(hipMemcpy(hx.data(), ws + o_rx, hx.size() * 4, hipMemcpyDeviceToHost));
            int ok = 0;
            for (int i = 0; i < b->num_cu; ++i) ok += (hx[i] & 15) == i % rs->n_xcd;
            fprintf(stderr, "[ring] XCC_ID == block %% %d for %d of %d workgroups; first 16:", rs->n_xcd, ok, b->num_cu);
            for (int i = 0; i < 16 && i < b->num_cu; ++i) fprintf(stderr, " %d", hx[i]);
            fprintf(stderr, "\n");
            // tile-start spread of every XCD's patch (first task): max - min of the 32 workgroups' wall clocks, per tile, in us
            std::vector<long long> tr((size_t)b->num_cu * 64);
            HIP_TRY(hipMemcpy(tr.data(), ws + o_rz, tr.size() * 8, hipMemcpyDeviceToHost));
            long long t0 = -1;
            for (int w = 0; w < b->num_cu; ++w) if (tr[(size_t)w * 64] > 0 && (t0 < 0 || tr[(size_t)w * 64] < t0)) t0 = tr[(size_t)w * 64];
            for (int x = 0; x < rs->n_xcd; x += 3) {
                fprintf(stderr, "[ring] XCD %d tile: start(us) spread(us):", x);
                for (int i = 0; i < 62; i += (i < 8 ? 1 : 6)) {
                    long long lo = -1, hi = -1;
                    for (int w = x; w < b->num_cu; w += rs->n_xcd) {
                        const long long v = tr[(size_t)w * 64 + i];
                        if (v <= 0) continue;
                        if (lo < 0 || v < lo) lo = v;
                        if (v > hi) hi = v;
                    }
                    if (lo > 0) fprintf(stderr, "  %d: %.0f %.1f", i, (lo - t0) * 0.01, (hi - lo) * 0.01);
                }
                long long to = 0;
                for (int w = x; w < b->num_cu; w += rs->n_xcd) to += tr[(size_t)w * 64 + 62];
                fprintf(stderr, "  pauses %lld\n", to);
            }
            for (int i = 1; i <= 3; i += 2) {
                fprintf(stderr, "[ring] XCD 0, start of tile %d by slot (qi-major, Sb = %d), us after the first:", i, rs->sb);
                for (int sl = 0; sl < rs->wpx; ++sl) fprintf(stderr, "%s%.0f", sl % rs->sb == 0 ? " | " : " ", (tr[(size_t)(sl * rs->n_xcd) * 64 + i] - t0) * 0.01);
                fprintf(stderr, "\n");
            }
            if (dbg == 33) {      // barrier stamps of workgroup 0: per stage, arrival of each wave relative to the first arrival, and the release
                std::vector<long long> bt(8 * 48 * 2 + 8);
                HIP_TRY(hipMemcpy(bt.data(), ws + o_rz + (size_t)b->num_cu * 64 * 8, bt.size() * 8, hipMemcpyDeviceToHost));
                fprintf(stderr, "[ring] HW_ID simd of waves 0..7:");
                for (int w = 0; w < 8; ++w) fprintf(stderr, " %lld", (bt[8 * 48 * 2 + w] >> 4) & 3);
                fprintf(stderr, "\n");
                for (int it2 = 8; it2 < 48; it2 += 3) {
                    long long first = -1, last = -1;
                    for (int w = 0; w < 8; ++w) { const long long v = bt[((size_t)w * 48 + it2) * 2]; if (first < 0 || v < first) first = v; if (v > last) last = v; }
                    const long long prev_rel = bt[((size_t)0 * 48 + it2 - 1) * 2 + 1];
                    fprintf(stderr, "[ring] stage %2d: length %5lld  arrivals", it2, last - prev_rel);
                    for (int w = 0; w < 8; ++w) fprintf(stderr, " %4lld", bt[((size_t)w * 48 + it2) * 2] - first);
                    fprintf(stderr, "  release after last arrival:");
                    for (int w = 0; w < 8; ++w) fprintf(stderr, " %3lld", bt[((size_t)w * 48 + it2) * 2 + 1] - last);
                    fprintf(stderr, "\n");
                }
            }
            long long elo = -1, ehi = -1;
            for (int w = 0; w < b->num_cu; ++w) { const long long v = tr[(size_t)w * 64 + 63]; if (v > 0) { if (elo < 0 || v < elo) elo = v; if (v > ehi) ehi = v; } }
            fprintf(stderr, "[ring] kernel exits between %.0f and %.0f us after the first tile start\n", (elo - t0) * 0.01, (ehi - t0) * 0.01);
        }
    } else if (pair) {
        PairArgs pa;
        pa.bank2 = nprod == 1 ? b->rowsh : b->rows2; pa.ldb2 = ldq2; pa.invs = b->invs; pa.n_rows = (int)b->n;
        pa.q2 = ws + o_q32; pa.ldq2 = ldq2; pa.qinvs = (const float *)(ws + o_qs);
        pa.lim = lim; pa.qt_maxlim = qtm; pa.nkt = nprod == 1 ? b->kh / 64 : kd / 32;
        pa.nqt = nqt; pa.nseg = nseg; pa.tps = tps; pa.n_btiles = n_btiles;
        pa.part_key = part_key; pa.part_idx = part_idx; pa.part_bound = part_bound; pa.item_map = item_map;
        rc = pair_stage1_launch(pa, tile, nprod, dbg, st);
    } else {
        rc = tile == 256 ? launch_stage1<256, 4, 8>(a, dbg, ilv, st) : launch_stage1<128, 2, 16>(a, dbg, ilv, st);
    }
    if (rc) return rc;
    if (b->ev_valid) HIP_TRY(hipEventRecord(b->ev1, st));

    // rigorous bound on |f32 key - exact| / ||q||: kd-term fma chain (gamma_kd), inv-norm
    // rounding, key multiply rounding, query f64->f32 rounding; 2^-24 unit roundoff.
    const double u = 5.9604644775390625e-08;
    // (the persistent stage's lists hold integer keys in units of <= 1.011 x 2^-16 ||q||, truncated: sim_topk_pair_dev.h)
    const double err_bound = (pair ? pair_err_bound(nprod == 1 ? b->kh : kd, nprod) : 1.0625 * ((double)kd + 8.0) * u) + (ring ? 1.02 / 65536.0 : 0.0);
    const unsigned rgrid = (unsigned)ceil_div64(nq, 4);
    if (q_dtype == CSLAM_F32)
        hipLaunchKernelGGL(rescore_kernel<float>, dim3(rgrid), dim3(256), 0, st, b->rows, (int64_t)ld, kd, b->vv,
                           (const float *)d_q, ldq, b->dim, (int)nq, part_key, part_idx, part_bound, nseg, k,
                           err_bound, d_out_idx, d_out_sim, d_out_cnt, flag_list, flag_count,
                           ring ? (const int *)(ws + o_rn) : nullptr, ring ? (const int *)(ws + o_rs) : nullptr, tile);
    else
        hipLaunchKernelGGL(rescore_kernel<double>, dim3(rgrid), dim3(256), 0, st, b->rows, (int64_t)ld, kd, b->vv,
                           (const double *)d_q, ldq, b->dim, (int)nq, part_key, part_idx, part_bound, nseg, k,
                           err_bound, d_out_idx, d_out_sim, d_out_cnt, flag_list, flag_count,
                           ring ? (const int *)(ws + o_rn) : nullptr, ring ? (const int *)(ws + o_rs) : nullptr, tile);
    HIP_TRY(hipGetLastError());

    // uncertified queries -> exact scan (needs the count on the host: one 4-byte readback into pinned memory, read by
    // mfma_search_finish after the caller's synchronisation -- one per search, or one for a whole group of searches)
    if (!b->h_nflag) HIP_TRY(hipHostMalloc((void **)&b->h_nflag, sizeof(int), hipHostMallocDefault));
    *b->h_nflag = 0;
    HIP_TRY(hipMemcpyAsync(b->h_nflag, flag_count, 4, hipMemcpyDeviceToHost, st));
    b->stats[2] = nseg; b->stats[3] = nqt;
    b->pending_flag_list = flag_list;
    b->pending_flag_count = flag_count;
    b->pending_dbg = dbg;
    b->dbg_part_key = part_key; b->dbg_part_idx = part_idx; b->dbg_nseg = nseg; b->dbg_nq = (int)nq; b->dbg_err_bound = err_bound;
    b->dbg_ring = ring;
    return CSLAM_OK;
}

int mfma_search_finish(cslam_bank *b, const void *d_q, int q_dtype, int64_t ldq, int k, const int64_t *d_row_limit,
                       int64_t *d_out_idx, double *d_out_sim, int32_t *d_out_cnt, hipStream_t st) {
    const int nflag = b->h_nflag ? *b->h_nflag : 0;        // valid once `st` has been synchronised
    b->stats[0] = nflag;
    if (b->last_nprod != 0 && !b->stage_pinned) {
        if ((int64_t)nflag * 32 > (int64_t)b->dbg_nq) {
            b->f32_backoff = b->f32_backoff_len;
            if (b->f32_backoff_len < 1024) b->f32_backoff_len *= 2;
        } else {
            b->f32_backoff_len = 8;
        }
    }
    if (nflag > 0 && b->pending_dbg == 0)
        return scan_search(b, d_q, q_dtype, ldq, b->pending_flag_list, nflag, k, d_row_limit, d_out_idx, d_out_sim,
                           d_out_cnt, st);
    return CSLAM_OK;
}

int mfma_search(cslam_bank *b, const void *d_q, int q_dtype, int64_t ldq, int64_t nq, int k,
                const int64_t *d_row_limit, int64_t *d_out_idx, double *d_out_sim,
                int32_t *d_out_cnt, hipStream_t st) {
    int rc = mfma_search_enqueue(b, d_q, q_dtype, ldq, nq, k, d_row_limit, d_out_idx, d_out_sim, d_out_cnt, st);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    return mfma_search_finish(b, d_q, q_dtype, ldq, k, d_row_limit, d_out_idx, d_out_sim, d_out_cnt, st);
}

/* diagnostics (include/cslam_hip_experimental.h): the candidate lists stage 1 of the last MFMA-mode search left in the bank's
 * workspace -- keys [nq][nseg][16] (float32, in units of q.b / ||b||), rows [nq][nseg][16] (-1 = empty) -- and the bound on
 * |key - exact| / ||q|| handed to the certificate.  Valid until the next search of the bank. */
CSLAM_API int cslam_debug_last_candidates(cslam_bank_t *b, int64_t nq, int *nseg, float *keys, int *rows, double *err_bound) {
    ARG_CHECK(b && nseg, "NULL argument");
    ARG_CHECK(b->dbg_part_key && nq >= 0 && nq <= b->dbg_nq, "no MFMA-mode search to report, or nq larger than its batch");
    BANK_DEVICE(b);
    *nseg = b->dbg_nseg;
    if (err_bound) *err_bound = b->dbg_err_bound;
    HIP_TRY(hipStreamSynchronize(b->last_stream));
    if (b->dbg_ring) {
        // the persistent stage keeps per-query-tile list counts: hand them out in the uniform layout, missing lists empty
        const RingSchedule &rs = b->ring_sched;
        const int T = 256, ns_out = b->dbg_nseg;
        std::vector<float> hk(keys ? (size_t)rs.total_lists * T * KP : 0);
        std::vector<int> hi(rows ? (size_t)rs.total_lists * T * KP : 0);
        if (keys) HIP_TRY(hipMemcpy(hk.data(), b->dbg_part_key, hk.size() * 4, hipMemcpyDeviceToHost));
        if (rows) HIP_TRY(hipMemcpy(hi.data(), b->dbg_part_idx, hi.size() * 4, hipMemcpyDeviceToHost));
        for (int64_t j = 0; j < nq; ++j) {
            const int qt = (int)(j / T), ns = rs.qt_nseg[qt];
            const size_t l0 = (size_t)rs.qt_segoff[qt] * T + (size_t)(j - (int64_t)qt * T) * ns;
            for (int l = 0; l < ns_out; ++l)
                for (int e = 0; e < KP; ++e) {
                    const size_t o = ((size_t)j * ns_out + l) * KP + e;
                    if (keys) keys[o] = l < ns ? hk[(l0 + l) * KP + e] : -INFINITY;
                    if (rows) rows[o] = l < ns ? hi[(l0 + l) * KP + e] : -1;
                }
        }
        return CSLAM_OK;
    }
    const size_t cnt = (size_t)nq * b->dbg_nseg * KP;
    if (keys) HIP_TRY(hipMemcpy(keys, b->dbg_part_key, cnt * 4, hipMemcpyDeviceToHost));
    if (rows) HIP_TRY(hipMemcpy(rows, b->dbg_part_idx, cnt * 4, hipMemcpyDeviceToHost));
    return CSLAM_OK;
}

CSLAM_API int cslam_bank_last_kernel_ms(cslam_bank_t *b, float *ms) {
    ARG_CHECK(b && ms, "NULL argument");
    *ms = -1.0f;
    if (!b->ev_valid || b->stats[1] != CSLAM_MODE_MFMA) return CSLAM_OK;
    BANK_DEVICE(b);
    HIP_TRY(hipEventSynchronize(b->ev1));
    HIP_TRY(hipEventElapsedTime(ms, b->ev0, b->ev1));
    return CSLAM_OK;
}
