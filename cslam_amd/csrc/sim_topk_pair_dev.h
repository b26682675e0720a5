// sim_topk_pair_dev.h -- device helpers shared by the candidate-stage kernels on the fp16 matrix pipe (sim_topk_pair.hip: one
// workgroup per (query tile, bank segment); sim_topk_ring.hip: persistent workgroups on a static, XCD-aligned schedule): the MUBUF
// LDS-DMA request, the lane-local candidate update of a finished tile, the block merge of the lane lists.
#pragma once
#include <hip/hip_fp16.h>
#include "bank.h"
#include "sim_topk.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define PK_ROWB 128     // bytes of one 32-channel block of one row

// LDS-DMA in the MUBUF encoding (`buffer_load_dwordx4 ... lds`), round 4.  `global_load_lds` is FLAT-encoded with an LDS operand:
// hipcc's waitcnt pass marks it "pending flat" and from then on turns every `lgkmcnt(N)` into `lgkmcnt(0)` -- a K step's MFMAs then
// wait for ALL fragment reads issued before them.  The buffer form carries no such mark (LDS reads are counted again), takes the
// stage's K offset in an SGPR and the lane's row offset in ONE register (a 64-bit address per request before), and clamps in
// hardware: rows beyond `num_records` read as zero.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pk_rsrc(const char *base, int64_t bytes) {
    const uint64_t a = (uint64_t)base;
    const uint64_t u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)a);
    const int n = __builtin_amdgcn_readfirstlane((int)(bytes > 0x7fffffff ? 0x7fffffff : bytes));
    return __builtin_amdgcn_make_buffer_rsrc((void *)u, 0, n, 0x00020000);
}
__device__ __forceinline__ void pk_blds16(__amdgpu_buffer_rsrc_t rs, int voff, int soff, char *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)lds_wave_base, 16, voff, soff, 0, 0);
}
// the same request with cache-policy bits (gfx940+: aux bit 0 = sc0, bit 1 = nt, bit 4 = sc1); measurement build of sim_topk_ring.hip
template <int AUX>
__device__ __forceinline__ void pk_blds16_aux(__amdgpu_buffer_rsrc_t rs, int voff, int soff, char *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)lds_wave_base, 16, voff, soff, 0, AUX);
}


// ---- tile epilogue: lane-local candidate update from the finished 32 MT x 64 wave tile, then clear the accumulators
// (a lane's 16 accumulator registers of a tile belong to ONE query column: no cross-lane traffic)
template <int MT, int KPL, int NTW>
__device__ __forceinline__ void pair_tile_epilogue(f32x16 (&acc)[MT][NTW], float (&lk)[NTW][KPL], int (&li)[NTW][KPL], const int (&lim)[NTW],
                                                   const float (&qmul)[NTW], const float *__restrict__ invs, int n_rows, int row_base) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        float inv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = row_base + m * 32 + (r & 3) + 8 * (r >> 2);
            inv[r] = invs[row < n_rows ? row : n_rows - 1];
        }
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
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

// ---- tile epilogue of the persistent stage (sim_topk_ring.hip), PACKED lists (round 6).  Same contract as pair_tile_epilogue -- every
// visible key of the finished tile is offered to the lane's sorted list, the accumulators are cleared.
//
// What the update cost before.  A key passes a list that has seen N rows with probability KPL / N, and the in-step launch's lists live six
// tiles: with 64 independent lists in a wave, SOME lane wanted an insertion at more than half of all row offsets, and every such offset
// cost the whole wave a predicated 8-step insertion of a (key, index) pair (~40 issue slots; round 5 batched lanes that want different
// offsets into one insertion, round 6 seeded the first tile and shared the threshold: the update was still 16 % of the in-step launch).
//
// Packed entries.  A list entry is ONE 32-bit integer: (-key in units of `unit`) << 13 | (position of the row in the task, 13 bits),
// smaller = better, RING_EMPTY = nothing.  Inserting x into the ascending list l is then l'[j] = median(l[j-1], l[j], x) for every j
// (l'[0] = min(l[0], x)): eight INDEPENDENT v_med3_i32 / v_min_i32, no predicate, no index to move, and a key that does not belong
// leaves the list unchanged -- so no per-lane mask either.  The only test left is wave-wide ("does any lane's key beat its threshold":
// one v_cmp against the list's last entry + a scalar branch).
//  * key units: the query's prep (pair_prep_kernel) delivers `qmul` = -2^(17 - Es) with 2^Es >= 1.01 ||s_q q||, so |key| < 2^18 always and
//    `unit` = 2^(Es - 17) / s_q (a power of two: packing and unpacking are exact) <= 1.011 x 2^-16 ||q||.  Truncation moves a key by less than
//    one unit: the host adds 1.02 x 2^-16 to the stage's error bound (pair_err_bound: 1.57e-3 at 4096-D; + 1 %), which covers the keys the
//    merge writes AND its drop bound (a dropped key's integer is >= the bound's).
//  * NaN keys (non-finite rows / unservable queries) must rank first: v_med3_f32 returns the MINIMUM of its other two operands when one
//    is NaN -- the clamp that bounds the integer delivers "-largest key", i.e. best, for free (pinned by the non-finite-row tests).
//  * positions: tile ordinal in the task (7 bits: ring_schedule_build cuts tasks at 128 tiles) << 6 | 32-row block << 4 | accumulator
//    register; the block merge turns them back into bank rows.
//  * 1 / (||row|| s_row) of the tile's 256 rows is in LDS (`s_inv`, fetched by LDS-DMA beside the tile's first K stage).
// Measured with these lists and dropped: a threshold shared by the four partial lists of a query column (two row halves of a wave x two
// waves along the bank axis; -4 % in step with the old insertion) buys nothing once an insertion costs eight instructions
// (profiles/r06_s_ring_packed_ab.log), and the seeded first tile has nothing left to save.
#define RING_IDX_BITS 13
#define RING_KF_MAX 262142           // |integer key| bound: (RING_KF_MAX << 13 | 8191) < RING_EMPTY
#define RING_EMPTY 0x7fffffff
__device__ __forceinline__ int ring_med3(int a, int b, int c) {
    int d;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// MASKED = some row of the wave's 32 MT rows may be invisible to some query of the wave (a causal limit, the task's end row): the tiles
// that are not -- all but the last one or two of a walk -- skip the two instructions per key of that test.
template <int MT, int KPL, int NTW, bool MASKED>
__device__ __forceinline__ void ring_tile_epilogue(f32x16 (&acc)[MT][NTW], int (&lp)[NTW][KPL], const int (&lim)[NTW], const float (&qmul)[NTW],
                                                   const float *s_inv, int row_base, int idx_base) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        f32x2 inv2[8];                                                              // rows 8 q + {0 1}, {2 3} of this lane's half
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *(const float4 *)(s_inv + m * 32 + 8 * q);
            inv2[2 * q] = f32x2{v.x, v.y}; inv2[2 * q + 1] = f32x2{v.z, v.w};
        }
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            const int rel_lim = lim[n] - (row_base + m * 32);   // row < lim  <=>  rowoff < rel_lim
            const f32x2 qm2 = f32x2{qmul[n], qmul[n]};
#pragma unroll
            for (int r2 = 0; r2 < 8; ++r2) {
                f32x2 k2 = (f32x2{acc[m][n][2 * r2], acc[m][n][2 * r2 + 1]} * inv2[r2]) * qm2;               // - key / unit
                asm("" : "+v"(k2));                  // (the pair is used AS a pair: hipcc keeps the two v_pk_mul_f32 instead of four v_mul_f32)
                acc[m][n][2 * r2] = 0.0f; acc[m][n][2 * r2 + 1] = 0.0f;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int r = 2 * r2 + e;
                    const float k = __builtin_amdgcn_fmed3f(k2[e], -(float)RING_KF_MAX, (float)RING_KF_MAX);   // NaN -> -RING_KF_MAX: ranks first
                    int x = (int)(((unsigned)(int)k << RING_IDX_BITS) | (unsigned)(idx_base + m * 16 + r));
                    if (MASKED) x = (((r & 3) + 8 * (r >> 2)) < rel_lim) ? x : RING_EMPTY;                      // rows the query may not see
                    if (__builtin_amdgcn_ballot_w64(x < lp[n][KPL - 1]) != 0) {                                  // wave-uniform
#pragma unroll
                        for (int j = KPL - 1; j > 0; --j) lp[n][j] = ring_med3(lp[n][j - 1], lp[n][j], x);
                        lp[n][0] = x < lp[n][0] ? x : lp[n][0];
                    }
                }
            }
        }
    }
}

// ---- block merge of the packed lists: 4 per query (2 row halves of a wave x 2 waves along the bank axis) -> the best SIM_KP of them as
// (key, bank row), plus the bound on everything dropped; the LDS of the K loop is reused.  Query `tid` of the tile writes its list at
// out_key / out_idx + (tid * qstride) * SIM_KP and its bound at out_bound[tid * qstride] (qstride = lists per query).
template <int T_, int MT, int KPL, int NTW>
__device__ __forceinline__ void ring_block_merge(char *smem, int (&lp)[NTW][KPL], int wn, int wm, int h, int l31, int tid, int row0, int stride_rows,
                                                 const float *__restrict__ qunit, float *__restrict__ out_key, int *__restrict__ out_idx,
                                                 float *__restrict__ out_bound, size_t qstride) {
    __syncthreads();
    int *mk = (int *)smem;                           // [T_][4][KPL]
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int qcol = wn * (32 * NTW) + n * 32 + l31;
        const int src = wm * 2 + h;
#pragma unroll
        for (int j = 0; j < KPL; ++j) mk[(qcol * 4 + src) * KPL + j] = lp[n][j];
    }
    __syncthreads();
    if (tid < T_) {
        const int *k0 = mk + (tid * 4) * KPL;
        const float unit = qunit[tid];
        int p0 = 0, p1 = 0, p2 = 0, p3 = 0;
        const size_t o = ((size_t)tid * qstride) * SIM_KP;
        int bound = RING_EMPTY;
#pragma unroll
        for (int s = 0; s < 4; ++s) bound = k0[s * KPL + KPL - 1] < bound ? k0[s * KPL + KPL - 1] : bound;     // a full lane list dropped keys
        for (int j = 0; j < SIM_KP; ++j) {
            const int c0 = p0 < KPL ? k0[p0] : RING_EMPTY;
            const int c1 = p1 < KPL ? k0[KPL + p1] : RING_EMPTY;
            const int c2 = p2 < KPL ? k0[2 * KPL + p2] : RING_EMPTY;
            const int c3 = p3 < KPL ? k0[3 * KPL + p3] : RING_EMPTY;
            int best = 0, bk = c0;
            if (c1 < bk) { bk = c1; best = 1; }
            if (c2 < bk) { bk = c2; best = 2; }
            if (c3 < bk) { bk = c3; best = 3; }
            if (best == 0) ++p0; else if (best == 1) ++p1; else if (best == 2) ++p2; else ++p3;
            const int pos = bk & ((1 << RING_IDX_BITS) - 1);
            const int rr = pos & 15;
            const int row = row0 + (pos >> 6) * stride_rows + (best >> 1) * (32 * MT) + ((pos >> 4) & 3) * 32 + 4 * (best & 1) + (rr & 3) + 8 * (rr >> 2);
            out_key[o + j] = bk == RING_EMPTY ? -INFINITY : -(float)(bk >> RING_IDX_BITS) * unit;
            out_idx[o + j] = bk == RING_EMPTY ? -1 : row;
        }
        const int c0 = p0 < KPL ? k0[p0] : RING_EMPTY, c1 = p1 < KPL ? k0[KPL + p1] : RING_EMPTY;
        const int c2 = p2 < KPL ? k0[2 * KPL + p2] : RING_EMPTY, c3 = p3 < KPL ? k0[3 * KPL + p3] : RING_EMPTY;
        int d = c0 < c1 ? c0 : c1;
        d = c2 < d ? c2 : d;
        d = c3 < d ? c3 : d;
        bound = d < bound ? d : bound;
        out_bound[(size_t)tid * qstride] = bound == RING_EMPTY ? -INFINITY : -(float)(bound >> RING_IDX_BITS) * unit;
    }
}

// ---- block merge: 4 lists per query (2 row-halves of the wave x 2 waves along the bank axis) -> the best KP of them, plus the
// bound on everything dropped (see sim_topk_mfma_kernel); the LDS of the K loop is reused.  Query `tid` of the tile writes its
// list at out_key / out_idx + (tid * qstride) * SIM_KP and its bound at out_bound[tid * qstride] (qstride = lists per query).
template <int T_, int KPL, int NTW>
__device__ __forceinline__ void pair_block_merge(char *smem, float (&lk)[NTW][KPL], int (&li)[NTW][KPL], int wn, int wm, int h, int l31,
                                                 int tid, float *__restrict__ out_key, int *__restrict__ out_idx,
                                                 float *__restrict__ out_bound, size_t qstride) {
    __syncthreads();
    float *mk = (float *)smem;                       // [T_][4][KPL]
    int *mi = (int *)(smem + T_ * 4 * KPL * 4);      // [T_][4][KPL]
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        int qcol = wn * (32 * NTW) + n * 32 + l31;
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
        size_t o = ((size_t)tid * qstride) * SIM_KP;
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
            out_key[o + j] = bk;
            out_idx[o + j] = (bk == -INFINITY) ? -1 : bi;
        }
        float c0 = p0 < KPL ? k0[p0] : -INFINITY, c1 = p1 < KPL ? k0[KPL + p1] : -INFINITY;
        float c2 = p2 < KPL ? k0[2 * KPL + p2] : -INFINITY, c3 = p3 < KPL ? k0[3 * KPL + p3] : -INFINITY;
        bound = fmaxf(fmaxf(bound, fmaxf(c0, c1)), fmaxf(c2, c3));
        out_bound[(size_t)tid * qstride] = bound;
    }
}

