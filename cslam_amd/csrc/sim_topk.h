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

// ---- the persistent form of the one-product stage (sim_topk_ring.hip) ------------------------------------------------------
// One workgroup per CU for the whole launch; the workgroups of an XCD form an Sq x Sb patch (query tiles x bank columns) whose
// members walk the bank side by side, all of them at the same K position, so every operand stream of the patch is fetched into
// the XCD's L2 once and read Sq (bank) or Sb (queries) times.  A task = one query tile against bank rows
// [row0 + i stride_rows, + 256) for i < n_tiles (stride_rows = 256), cut at row_end; its merged candidate list is "segment" `seg`
// of that query tile.
// A run (one query group against bank rows [ra, rb) on one XCD) is cut into Sb contiguous chunks of equal length, one per patch
// column, the last tile of a chunk partial: every workgroup of a run computes the same number of ROWS (the 1024-query launch of
// the bench's step: 6 tiles + 32 rows each instead of 6 or 7 whole tiles).  Whole-bank runs are cut the same way on every XCD,
// so the 8 XCDs work on 8 different query groups against the SAME bank rows at the same time.
struct RingTask {
    int qt, row0, n_tiles, seg;
    int run;                                    // ordinal of the run in its XCD's list (flow control compares progress inside a run)
    int stride_rows, row_end, pad;
};
struct RingArgs {
    const char *bank2; int64_t ldb2;     // bank rows as fp16 (hi halves), pitch in bytes
    const float *invs; int n_rows;
    const char *q2; int64_t ldq2;        // queries as fp16 [nqt * 256][ldq2]
    const float *qinvs;                  // [nqt * 256] - 2^(17 - Es): accumulator x 1 / (||row|| s_row) -> the packed lists' integer key (sim_topk_pair_dev.h)
    const float *qunit;                  // [nqt * 256] what one unit of that integer is worth in key units (q.b / ||b||): 2^(Es - 17) / s_query
    const int *lim;
    const int *qt_maxlim;
    int nkt;                             // K stages of 128 bytes per row
    int nqt, sb;                         // sb = columns of a patch (slot = qi * sb + bi)
    int n_xcd, wpx;                      // workgroup b = slot b / n_xcd of XCD b % n_xcd
    const RingTask *tasks;               // tasks of workgroup w: [task_off[w], task_off[w + 1])
    const int *task_off;
    const int *qt_nseg, *qt_segoff;      // per query tile: lists per query, lists before this tile's (in units of 256 queries' lists)
    float *part_key; int *part_idx;      // list l of query j of tile qt at ((qt_segoff[qt] * 256 + j * qt_nseg[qt] + l) * SIM_KP
    float *part_bound;
    int *prog;                           // flow control: [n_xcd][32] progress words (zeroed per launch) or nullptr
    int flow_w;                          // ... and its window, in K stages
    int flow_bias_q, flow_bias_b;        // measurement build: per-slot offsets inside the window
    int *xcc_out;                        // diagnostics: [workgroups] HW_REG_XCC_ID of the CU each workgroup ran on, or nullptr
    long long *trace_out;                // diagnostics: [workgroups][64] wall clock (100 MHz) at the top of the first 62 tiles of the first task,
                                         //              [62] = pauses of the flow control (~1 us each), [63] = wall clock at kernel exit; or nullptr
};
struct RingSchedule {                    // host side, cached per bank
    int nqt, n_rows, n_xcd, wpx;         // key
    int sq, sb, total_lists;             // total_lists = sum of qt_nseg (lists per query, summed over query tiles)
    std::vector<RingTask> tasks;
    std::vector<int> task_off, qt_nseg, qt_segoff;
    std::vector<char> blob;              // the four tables back to back (256-byte aligned): ONE upload per search
    size_t off_tasks, off_task_off, off_qt_nseg, off_qt_segoff;
};
void ring_schedule_build(RingSchedule &s, int nqt, int n_rows, int n_xcd, int wpx);
int ring_stage1_launch(const RingArgs &a, int variant, int dbg, hipStream_t st);

double pair_err_bound(int kd, int nprod);
int pair_stage1_launch(const PairArgs &a, int tile, int nprod, int dbg, hipStream_t st);
// kd = the K extent the query copy is padded to with zeros (whole stages: a multiple of 32 for pairs, of 64 for hi halves)
int pair_prep_launch(const void *d_q, int q_dtype, int64_t ldq, int nq, int dim, int kd, int nprod, char *q2, int64_t ldq2,
                     float *qinvs, const int64_t *d_row_limit, int n_rows, int *lim, int *qtm, int nq_pad, int tile, float *qpk, hipStream_t st);
// qpk (or nullptr): [2][nq_pad] the persistent stage's integer-key factor and unit per query (RingArgs::qinvs / qunit)
