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

// ---- tile epilogue of the persistent stage (sim_topk_ring.hip, round 5).  Same contract as pair_tile_epilogue -- every visible key of
// the finished tile is offered to the lane's sorted list, the accumulators are cleared -- at a third of the issue slots:
//  * 1 / (||row|| s_row) of the tile's 256 rows is in LDS (`s_inv`, fetched by LDS-DMA beside the tile's first K stage): four
//    ds_read_b128 per 32-row block instead of sixteen global loads and their round trip;
//  * insertions are batched ACROSS row offsets.  pair_tile_epilogue runs one 8-step insertion (~55 issue slots, the whole wave)
//    per row offset r that ANY lane wants; with N rows seen by a list a key passes with probability KPL / N, a 16-key block of
//    64 lanes has ~128 / t such offsets after t tiles, and a list lives for 49 tiles on the 100k x 100k batch (7 in the bench's
//    in-step launch): the "rare" path was 18 % / 36 % of the kernel (profiles/r05_v10_no_epilogue_ablation.log).  Here lanes that
//    want DIFFERENT offsets are served by ONE insertion: the candidate vector is assembled from the comparison masks (two
//    v_cndmask per non-empty offset, the masks are already in SGPRs), and the number of insertions of a block is the largest
//    number of passing keys of any single lane (1-2), not the number of distinct offsets (5-16).
// SHARED THRESHOLD (round 6).  A query column of a task has FOUR partial lists in the workgroup (two row halves of a wave x two waves along
// the bank axis), each fed by a quarter of every tile, so each stays "young" four times as long -- and a key passes a list that has seen N
// rows with probability KPL / N: in the in-step launch (six tiles per list) the update was still 19 % of the kernel.  `floor_thr` = the
// largest last-entry of the four lists as last published (the sibling half by a lane exchange, the other wave through LDS, possibly one tile
// old -- thresholds only rise): a key below it is below KPL keys of ONE partial list, so it cannot be among the query's best KPL of the task,
// and it is below the drop bound the block merge writes (the maximum of the full lists' last entries: exactly this value at the task's end).
// The merged list therefore still holds every key above the bound, the certificate of stage 2 sees the same bound, and the result is the
// same bit for bit; lists that stay short because of it dropped nothing above the bound.
// SEED (the FIRST tile of a list, round 6): with the list empty every one of a lane's 16 x MT keys passes the threshold and is inserted one
// by one -- the first tile of the in-step launch's six-tile lists cost 1.5 x a steady one (profiles/r05_v18_in_step_tile_trace.log).  A
// pre-pass turns the accumulators into keys (masked rows -inf, NaN +inf) and takes the two largest of every 16-key block: these are
// 2 MT >= KPL distinct keys of the lane, so the KPL-th largest of the tile is at least t0 = the smallest of the blocks' second maxima,
// and a key below t0 can never be in the list -- nor above the list's last entry, which is what the dropped-keys bound of the block
// merge rests on.  Only keys >= t0 are inserted (about a fifth): the list that comes out is the same, entry for entry.
template <int MT, int KPL, int NTW>
__device__ __forceinline__ void ring_tile_epilogue(f32x16 (&acc)[MT][NTW], float (&lk)[NTW][KPL], int (&li)[NTW][KPL], const int (&lim)[NTW],
                                                   const float (&qmul)[NTW], const float *s_inv, int row_base, bool seed, const float (&floor_thr)[NTW]) {
    // pass 1: accumulators -> keys in place (rows the query may not see: -inf; NaN: +inf, it ranks first); `seed` (wave-uniform; needs
    // 2 MT >= KPL): the smallest of the blocks' second maxima
    float t0[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) t0[n] = seed ? INFINITY : -INFINITY;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        float inv[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *(const float4 *)(s_inv + m * 32 + 8 * q);          // rows 8 q + {0 1 2 3} of this lane's half
            inv[4 * q] = v.x; inv[4 * q + 1] = v.y; inv[4 * q + 2] = v.z; inv[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            const int rel_lim = lim[n] - (row_base + m * 32);   // row < lim  <=>  rowoff < rel_lim
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float k = (acc[m][n][r] * inv[r]) * qmul[n];
                k = (k != k) ? INFINITY : k;
                acc[m][n][r] = (((r & 3) + 8 * (r >> 2)) < rel_lim) ? k : -INFINITY;
            }
            if (seed) {
                float m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    m2 = fmaxf(m2, fminf(m1, acc[m][n][r]));
                    m1 = fmaxf(m1, acc[m][n][r]);
                }
                t0[n] = fminf(t0[n], m2);
            }
        }
    }
    // pass 2: the keys above the list's last entry (and not below the seed) are inserted
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            float kx[16];
            unsigned long long pm[16];                           // lanes whose key at offset r passes: wave-uniform, in SGPRs
            unsigned long long any = 0;
            const float thr = fmaxf(lk[n][KPL - 1], floor_thr[n]);     // floor_thr: the other partial lists of this query (below)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                kx[r] = acc[m][n][r];
                pm[r] = __ballot(!(kx[r] <= thr) && !(kx[r] < t0[n]));
                any |= pm[r];
                acc[m][n][r] = 0.0f;
            }
            while (any) {
                // one round: every lane with a passing key left contributes ONE of them (its lowest offset)
                float ck = -INFINITY;
                int ci = 0;
                unsigned long long taken = 0;
                any = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (pm[r]) {                                 // scalar test
                        const unsigned long long mine = pm[r] & ~taken;
                        const bool sel = __builtin_amdgcn_inverse_ballot_w64(mine);
                        ck = sel ? kx[r] : ck;
                        ci = sel ? (r & 3) + 8 * (r >> 2) : ci;
                        taken |= mine;
                        pm[r] &= ~mine;
                        any |= pm[r];
                    }
                }
                ci += row_base + m * 32;
#pragma unroll
                for (int j = 0; j < KPL; ++j) {                  // lanes outside `taken` carry -inf: nothing moves
                    const bool sw = ck > lk[n][j];
                    const float tk = sw ? lk[n][j] : ck;
                    const int ti = sw ? li[n][j] : ci;
                    lk[n][j] = sw ? ck : lk[n][j];
                    li[n][j] = sw ? ci : li[n][j];
                    ck = tk; ci = ti;
                }
            }
        }
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

