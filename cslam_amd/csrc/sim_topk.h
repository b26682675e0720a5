// Shared between the two candidate stages of the batched search (sim_topk_mfma.hip: f32-input MFMA; sim_topk_pair.hip:
// fp16 pairs on the fp16 matrix pipe) and the host code that launches them.
#pragma once
#include "common.h"

#define SIM_KP 16           // merged candidate list length per (query, segment)

struct PairArgs {
    const char *bank2; int64_t ldb2;     // bank rows as fp16 pairs (3 products) or as fp16 hi halves (1 product), pitch in bytes
    const float *invs; int n_rows;       // 1 / (||row|| s_row)
    const char *q2; int64_t ldq2;        // queries as fp16 pairs [nqt * T][ldq2]
    const float *qinvs;                  // [nqt * T] 1 / s_query (0 for padding rows, NaN: query not servable)
    const int *lim;                      // [nqt * T] visible-row limit per query (0 for padding)
    const int *qt_maxlim;                // [nqt]
    int nkt;                             // K stages of 128 bytes per row: kd / 32 (pairs) | kh / 64 (hi halves)
    int nqt, nseg, tps, n_btiles;
    float *part_key; int *part_idx;      // [nqt * T][nseg][SIM_KP] merged candidates, sorted
    const int *item_map;                 // [nqt * nseg] (query tile << 16 | segment) in patch-major order
    float *part_bound;                   // [nqt * T][nseg] upper bound of the key of every row of the segment NOT in the list
};

double pair_err_bound(int kd, int nprod);
int pair_stage1_launch(const PairArgs &a, int tile, int nprod, int dbg, hipStream_t st);
// kd = the K extent the query copy is padded to with zeros (whole stages: a multiple of 32 for pairs, of 64 for hi halves)
int pair_prep_launch(const void *d_q, int q_dtype, int64_t ldq, int nq, int dim, int kd, int nprod, char *q2, int64_t ldq2,
                     float *qinvs, const int64_t *d_row_limit, int n_rows, int *lim, int *qtm, int nq_pad, int tile, hipStream_t st);
