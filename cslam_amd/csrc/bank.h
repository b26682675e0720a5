// Descriptor bank object shared by bank.hip (storage + exact scan) and sim_topk_mfma.hip.
#pragma once
#include "common.h"
#include <vector>
#include "sim_topk.h"

struct cslam_bank {
    int device;
    int dim;        // logical descriptor dimension
    int kd;         // K extent: dim rounded up to 32 (zero padded)
    int ld;         // row pitch in floats: kd, plus one 128-byte line when kd is a multiple of 256
    int64_t n;      // rows stored
    int64_t cap;    // rows allocated
    float *rows;    // [cap, ld]  float32 descriptors (reference: nns_matching.py:21)
    double *vv;     // [cap]      sum of squares of each row, float64
    float *invn;    // [cap]      (float)(1/sqrt(vv)); used by the fp32 candidate stage only
    // candidate-stage copies of the rows (sim_topk_pair.hip).  Row r times the power of two s_r (max |v| s_r in [2^14, 2^15)):
    //  rowsh: rounded to fp16, kh = dim rounded up to 64 channels (whole 128-byte K stages), 2 bytes per value -- the one-product
    //         stage, the default; always kept up to date by bank_append_kernel;
    //  rows2: split exactly into hi + lo halfs, 32-channel blocks [hi 32 | lo 32] of 128 bytes, 4 bytes per value -- the
    //         three-product stage (CSLAM_MFMA_STAGE1=pair, the A/B partner); built from `rows` on the first search that asks
    //         for it (bank_pairs_ensure) and kept up to date from then on, so a bank that never uses it never pays for it.
    // Pitches ldh / ld2 in BYTES (one extra 128-byte line when a multiple of 1 KiB, as ld).
    char *rowsh;    // [cap, ldh]
    char *rows2;    // [cap, ld2] or nullptr
    float *invs;    // [cap]      invn / s_r: key = (dot of the scaled copies) * invs[row] * (1 / s_query); NaN for a row whose largest
                    //            magnitude is outside [2^-100, 2^100] (such a row is always a contender, rescored exactly)
    int kh;
    int64_t ldh, ld2;
    // grow-on-demand workspace
    char *ws[3];          // [0] MFMA path, [1] scan path (also the MFMA fallback), [2] host-API staging
    size_t ws_bytes[3];
    char *stage;    // device staging for host->device adds
    size_t stage_bytes;
    hipStream_t last_stream;
    hipEvent_t ev0, ev1;
    bool ev_valid;
    hipStream_t side;         // cslam_bank_search_multi_dev: the searches of a bank list run side by side, one stream per bank
    hipEvent_t ev_fork, ev_side;
    int64_t stats[4];
    int num_cu;
    std::vector<int> item_map_host;   // cached work-item order of the MFMA path (see sim_topk_mfma.hip)
    int item_map_key[4];
    RingSchedule ring_sched;          // cached static schedule of the persistent candidate stage (sim_topk_ring.hip)
    bool dbg_ring;                    // the last MFMA-mode search used it: its lists have the per-query-tile layout
    int *h_nflag;                     // pinned: count of uncertified queries of the last enqueued MFMA search
    int *pending_flag_list;           // device list those queries are in (bank workspace)
    int *pending_flag_count;          // device count of that list (bank workspace)
    int last_nprod;                   // candidate stage of the last MFMA-mode search: fp16 products per pair (0: the f32-input stage)
    int f32_backoff;                  // searches left on the f32-input stage after an fp16 stage left too many queries uncertified
    int f32_backoff_len;              // length of the next back-off (8, doubling while retries keep overflowing, up to 1024)
    bool stage_pinned;                // the last search's stage was fixed by CSLAM_MFMA_STAGE1 (no back-off)
    int pending_dbg;
    // a search that has been enqueued and not finished (cslam_bank_search_enqueue_dev ... cslam_bank_search_finish): its
    // arguments, for the exact-scan fallback of the uncertified queries; ev_flag = "the uncertified-query count is on the host"
    struct Pending {
        bool active, deferred;
        const void *q; int q_dtype; int64_t ldq; int k;
        const int64_t *lim; int64_t *oi; double *os; int32_t *oc;
        hipStream_t st;
    } pend;
    hipEvent_t ev_flag;
    // diagnostics: where the last MFMA-mode search left its stage-1 candidate lists (cslam_debug_last_candidates)
    const float *dbg_part_key; const int *dbg_part_idx; int dbg_nseg, dbg_nq; double dbg_err_bound;
};

int bank_ws_reserve(cslam_bank *b, int slot, size_t bytes);
int bank_pairs_ensure(cslam_bank *b, hipStream_t st);      // rows2 exists and covers rows [0, n)

// exact fp64 scan over selected queries (bank.hip)
//  d_q: queries [nq_total, ldq] of q_dtype; qsel: [nsel] int32 query numbers or NULL (identity)
//  outputs for query number j go to out_*[j*k ...]
int scan_search(cslam_bank *b, const void *d_q, int q_dtype, int64_t ldq, const int *d_qsel,
                int64_t nsel, int k, const int64_t *d_row_limit, int64_t *d_out_idx,
                double *d_out_sim, int32_t *d_out_cnt, hipStream_t st);

// fp32-MFMA candidate search + fp64 rescoring (sim_topk_mfma.hip).  mfma_search = enqueue + stream synchronisation +
// finish; a group of searches on different banks can enqueue all of them and synchronise once (cslam_bank_search_multi_dev)
int mfma_search_enqueue(cslam_bank *b, const void *d_q, int q_dtype, int64_t ldq, int64_t nq, int k,
                        const int64_t *d_row_limit, int64_t *d_out_idx, double *d_out_sim,
                        int32_t *d_out_cnt, hipStream_t st);
int mfma_search_finish(cslam_bank *b, const void *d_q, int q_dtype, int64_t ldq, int k, const int64_t *d_row_limit,
                       int64_t *d_out_idx, double *d_out_sim, int32_t *d_out_cnt, hipStream_t st);
int mfma_search(cslam_bank *b, const void *d_q, int q_dtype, int64_t ldq, int64_t nq, int k,
                const int64_t *d_row_limit, int64_t *d_out_idx, double *d_out_sim,
                int32_t *d_out_cnt, hipStream_t st);
