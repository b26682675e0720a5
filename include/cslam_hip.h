/*
 * cslam_hip.h -- C ABI of libcslam_hip.so, the MI355X (gfx950) implementation of
 * cslam's global-descriptor loop-closure hot path.
 *
 * The reference (lajoiepy/cslam) has no FFI for this path: it is duck-typed Python
 * (SURVEY.md section 8b).  Each entry point below therefore names the reference
 * Python function whose arithmetic it replaces; the Python classes in cslam_amd/
 * keep the reference signatures and call these through ctypes.
 *
 * Conventions
 *   - every function returns 0 (CSLAM_OK) or a negative CSLAM_E_* code;
 *     cslam_last_error() returns a thread-local message for the last failure;
 *   - plain pointers and sizes only; no torch / numpy types;
 *   - "_host" entry points take host pointers and synchronise before returning;
 *     "_dev" entry points take device pointers, enqueue on `stream`
 *     (a hipStream_t passed as void*, NULL = default stream) and do not synchronise;
 *   - bank rows are float32 (reference storage dtype, cslam/nns_matching.py:21,39);
 *     queries are float32 (CSLAM_F32) or float64 (CSLAM_F64);
 *   - row indices are int64 in the API (bank row number = order of insertion,
 *     the key of the reference's `items` dict, cslam/nns_matching.py:38);
 *   - threading: like the reference classes (one rclpy single-threaded executor per robot,
 *     loop_closure_detection_node.py:109) objects (banks, communicators) are not thread-safe.  The descriptor-head
 *     entry points keep the scratch they need between their own kernels per (device, stream): calls on
 *     different streams may be in flight together (cslam_amd's two extraction lanes are), calls on one stream
 *     run in order anyway;
 *   - scratch buffers and coefficient tables owned by the library only grow and are released at process exit,
 *     so device pointers captured in a hipGraph stay valid.  A stream that is being captured cannot allocate:
 *     run an entry point once, at the captured sizes, on the stream you capture on (otherwise CSLAM_E_INVALID).
 */
#ifndef CSLAM_HIP_H
#define CSLAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSLAM_OK 0
#define CSLAM_E_INVALID (-1)  /* bad argument */
#define CSLAM_E_HIP (-2)      /* HIP runtime error (message in cslam_last_error) */
#define CSLAM_E_NOMEM (-3)
#define CSLAM_E_DIM (-4)      /* descriptor dimension mismatch */
#define CSLAM_E_UNSUPPORTED (-5) /* an optional run-time dependency (RCCL) is not available on this host */
#define CSLAM_E_GRAPH (-6)    /* cslam_fiedler / cslam_mac_fw_subset: the graph admits no Fiedler pair (not connected, singular
                                 junction Laplacian, TraceMIN breakdown) -- where the reference's networkx call raises */
#define CSLAM_E_LIMIT (-7)    /* the input exceeds a size limit of this entry point (cslam_fiedler: junction count of the dense
                                 factor); the arguments are valid and another path serves them */

#define CSLAM_F32 0
#define CSLAM_F64 1

/* search strategy (cslam_bank_search_*: `mode`) */
#define CSLAM_MODE_AUTO 0  /* scan for small nq, MFMA for batches */
#define CSLAM_MODE_SCAN 1  /* HBM-bound exact fp64 scan kernel */
#define CSLAM_MODE_MFMA 2  /* fp32-MFMA candidates + fp64 re-score + certificate (+ scan fallback) */

typedef struct cslam_bank cslam_bank_t;
typedef struct cslam_comm cslam_comm_t;   /* one rank of the multi-GPU exchange (RCCL communicator), see the end of the file */

const char *cslam_last_error(void);
int cslam_version(void);
int cslam_device_count(int *count);
int cslam_device_info(int device, char *name, int name_len, int64_t *hbm_bytes, int *cu_count);

/* ---- descriptor bank: NearestNeighborsMatching.__init__/add_item ----------
 * replaces cslam/nns_matching.py:10-40 (growable float32 bank, amortised doubling).
 * The bank lives in HBM on `device`; rows are padded to a multiple of 32 floats.   */
int cslam_bank_create(int device, int dim, int64_t capacity_hint, cslam_bank_t **out);
int cslam_bank_destroy(cslam_bank_t *bank);
int cslam_bank_size(const cslam_bank_t *bank, int64_t *n, int *dim);
int cslam_bank_clear(cslam_bank_t *bank);
/* append n rows given on the host; dtype CSLAM_F32 or CSLAM_F64 (values are cast to
 * float32 on store exactly like `self.data[self.n] = vector`, nns_matching.py:39). */
int cslam_bank_add_host(cslam_bank_t *bank, const void *vecs, int dtype, int64_t n);
/* append n float32 rows already in device memory, row stride `ld` floats */
int cslam_bank_add_dev(cslam_bank_t *bank, const float *d_vecs, int64_t ld, int64_t n, void *stream);
/* copy rows [row0, row0+nrows) back to the host, dense [nrows, dim] float32
 * (backs the reference's public `.data` attribute, tests/test_sparse_matching.py:36) */
int cslam_bank_read_host(const cslam_bank_t *bank, int64_t row0, int64_t nrows, float *out);
/* raw device views (row stride in floats); valid until the next add that grows the bank */
int cslam_bank_device_ptr(const cslam_bank_t *bank, const float **d_rows, int64_t *ld);

/* ---- search: NearestNeighborsMatching.search / search_best -----------------
 * replaces cslam/nns_matching.py:42-76:
 *     sim[i] = 1 - clip(1 - q.b_i / sqrt((q.q)(b_i.b_i)), 0, 2)   for i < limit
 *     order  = descending sim (NaN first), ties -> larger row index
 * row_limit: NULL, or [nq] int64: query j only sees rows < row_limit[j] (the causal
 *     order of global_descriptor_loop_closure_detection.py:157-160).
 * out_idx [nq,k] int64 (-1 padded), out_sim [nq,k] float64 (NaN padded),
 * out_cnt [nq] int32 = min(k, rows visible).
 * Scores are computed in float64 from the float32 bank / given-dtype query; the
 * returned order is the exact order of those float64 scores in every mode.       */
int cslam_bank_search_host(cslam_bank_t *bank, const void *queries, int q_dtype, int64_t nq,
                           int k, const int64_t *row_limit, int mode,
                           int64_t *out_idx, double *out_sim, int32_t *out_cnt);
/* device-pointer variant: d_queries [nq, ldq] (ldq in elements), outputs device arrays */
int cslam_bank_search_dev(cslam_bank_t *bank, const void *d_queries, int q_dtype, int64_t ldq,
                          int64_t nq, int k, const int64_t *d_row_limit, int mode,
                          int64_t *d_out_idx, double *d_out_sim, int32_t *d_out_cnt,
                          void *stream);

/* One batch of queries against nb banks of one device -- a robot's own bank and its copies of the other robots' banks
 * (cslam/loop_closure_sparse_matching.py:21-31), which the reference searches one after the other for every keyframe
 * (lcsm.py:45-53 best-1 per other robot, lcsm.py:74-76 top-k in the local bank).  Same results as nb calls of
 * cslam_bank_search_dev with (k[i], d_row_limit[i], d_out_*[i]); all kernels are enqueued before the single host
 * synchronisation the uncertified-query counts need, each bank's on a stream of its own forked from and joined back into
 * `stream` (chunk-sized searches do not fill the chip one after the other).
 * d_row_limit may be NULL (no limits) or hold NULL entries.  A bank may appear only once in the list. */
int cslam_bank_search_multi_dev(cslam_bank_t *const *banks, int nb, const void *d_queries, int q_dtype, int64_t ldq,
                                int64_t nq, const int *k, const int64_t *const *d_row_limit, int mode,
                                int64_t *const *d_out_idx, double *const *d_out_sim, int32_t *const *d_out_cnt,
                                void *stream);
/* The two halves of cslam_bank_search_dev / cslam_bank_search_multi_dev, for a host that pipelines steps (the reference's
 * callbacks are serialised, cslam/loop_closure_detection_node.py:109; a batch host overlaps step i's read-back with step
 * i + 1's extraction):
 *   *_enqueue_dev  puts every kernel of the search on `stream` and returns without a host synchronisation;
 *   *_finish       waits for ONE event -- the 4-byte count of queries whose float32 candidate stage could not be
 *                  certified (see cslam_bank_last_stats), recorded right behind its copy, NOT for the stream: work enqueued
 *                  after the search keeps running -- and enqueues the exact float64 scan for those queries (normally none).
 * Results are valid in `stream` order after finish.  One search per bank may be in flight: queries, row limits and outputs
 * must stay alive and the bank unchanged until it is finished (add / clear / another search return CSLAM_E_INVALID).
 * n_uncertified (or NULL) receives that count.  finish without a pending search is a no-op. */
int cslam_bank_search_enqueue_dev(cslam_bank_t *bank, const void *d_queries, int q_dtype, int64_t ldq, int64_t nq, int k,
                                  const int64_t *d_row_limit, int mode, int64_t *d_out_idx, double *d_out_sim,
                                  int32_t *d_out_cnt, void *stream);
int cslam_bank_search_finish(cslam_bank_t *bank, int64_t *n_uncertified);
/* Between the halves: the uncertified-query count of the enqueued search as a 4-byte device-to-device copy on `stream` (the
 * enqueue's stream) into d_count; 0 when the search needs no certificate (exact-scan modes) or none is pending.  Lets a sharded
 * step (one robot bank, or one row shard, per GPU: loop_closure_sparse_matching.py:45-53 across ranks) ship the count WITH its
 * provisional lists, so that every rank knows in stream order -- no host round trip -- whether a shard still owes a re-scan. */
int cslam_bank_search_flag_copy_dev(cslam_bank_t *bank, int32_t *d_count, void *stream);
int cslam_bank_search_multi_enqueue_dev(cslam_bank_t *const *banks, int nb, const void *d_queries, int q_dtype, int64_t ldq,
                                        int64_t nq, const int *k, const int64_t *const *d_row_limit, int mode,
                                        int64_t *const *d_out_idx, double *const *d_out_sim, int32_t *const *d_out_cnt,
                                        void *stream);
int cslam_bank_search_multi_finish(cslam_bank_t *const *banks, int nb, int64_t *n_uncertified /* summed over the banks, or NULL */);
/* statistics of the last search on this bank (for tests / bench):
 * stats[0] = queries that failed the fp32 certificate and were re-done by the scan,
 * stats[1] = mode actually used (CSLAM_MODE_*), stats[2] = bank segments,
 * stats[3] = query tiles.  Synchronises the bank's last stream.                    */
int cslam_bank_last_stats(cslam_bank_t *bank, int64_t stats[4]);
/* which candidate stage the last MFMA-mode search of this bank ran, and the state of its back-off (a bank whose near-duplicates
 * overflow the fp16 stage's re-scoring window runs its next searches on the f32-input stage; results are exact either way, time is
 * not): info[0] = fp16 products per pair of the last search's stage (1: the default; 3: exact pairs; 0: the f32-input stage),
 * info[1] = searches left on the f32-input stage before the fp16 stage is retried, info[2] = length of the next back-off (8,
 * doubling up to 1024 while retries keep overflowing), info[3] = 1 when CSLAM_MFMA_STAGE1 fixed the stage (no back-off).
 * Valid after the search's finish; does not synchronise. */
int cslam_bank_last_stage(cslam_bank_t *bank, int32_t info[4]);
/* time (ms, HIP events on the launch stream) of the dominant kernel of the last
 * MFMA-mode search: sim_topk_mfma.  -1 if the last search did not use it.        */
int cslam_bank_last_kernel_ms(cslam_bank_t *bank, float *ms);
/* One bank row-sharded over several GPUs (the single-bank metric at > 1 GPU): merge of the
 * per-shard results of cslam_bank_search_dev into the top-k of the whole bank -- what
 * `argsort(sim)[::-1][:k]` (cslam/nns_matching.py:60-61) returns over all rows, because the global top-k
 * is contained in the union of the per-shard top-k lists.  Shard s holds the global rows
 * [row_offset[s], row_offset[s] + n_s); row_offset is a HOST array of `shards` (<= 64) entries.
 * d_idx / d_sim [shards, nq, k] and d_cnt [shards, nq] are the shards' outputs (each list in search order,
 * shard-local rows); outputs as cslam_bank_search_dev with GLOBAL rows; same order: descending similarity,
 * NaN first, ties -> larger global row.  Similarities are passed through bit for bit.            */
int cslam_topk_merge_dev(const int64_t *d_idx, const double *d_sim, const int32_t *d_cnt,
                         const int64_t *row_offset, int shards, int64_t nq, int k,
                         int64_t *d_out_idx, double *d_out_sim, int32_t *d_out_cnt, void *stream);

/* ---- descriptor heads (device pointers; all float32) -------------------------- */
/* rows of x [n, d] (stride ld) scaled to unit L2 norm, x / max(||x||, eps);
 * F.normalize(p=2) of cslam/vpr/cosplace_utils/layers.py:32-36, netvlad.py:105-106,126-128;
 * eps = 1e-12 (torch default).  zero_norm_to_one != 0 gives sklearn.preprocessing.normalize
 * semantics instead (zero rows stay zero; netvlad.py:235-236). */
int cslam_l2_normalize_dev(float *d_x, int64_t n, int d, int64_t ld, float eps,
                           int zero_norm_to_one, void *stream);
/* NetVLADLayer.forward, cslam/vpr/netvlad.py:94-130.
 * feat [B, C, P] (NCHW flattened), assign_w [K, C] (1x1 conv, no bias: vladv1),
 * assign_b [K] or NULL, centroids [K, C]; out [B, K*C] with row pitch ldo floats. */
int cslam_vlad_aggregate_dev(const float *d_feat, const float *d_assign_w, const float *d_assign_b,
                             const float *d_centroids, int B, int C, int P, int K,
                             float *d_out, int64_t ldo, void *stream);
/* The same layer on a channels-last feature map d_feat [B][P][C] (what the Winograd trunk writes): batches only (B > 8,
 * P <= 256); spares the layout conversion in front of the head. */
int cslam_vlad_aggregate_nhwc_dev(const float *d_feat, const float *d_assign_w, const float *d_assign_b,
                                  const float *d_centroids, int B, int C, int P, int K,
                                  float *d_out, int64_t ldo, void *stream);
/* CosPlace aggregation head, cslam/vpr/cosplace_utils/network.py:23-29 + layers.py:8-36:
 * L2Norm(C) -> GeM(p, eps) -> Flatten -> Linear(W [Dout, C], b [Dout]) -> L2Norm.
 * feat [B, C, P]; out [B, Dout]. */
int cslam_gem_fc_head_dev(const float *d_feat, float p, float eps, const float *d_W,
                          const float *d_b, int B, int C, int P, int Dout,
                          float *d_out, void *stream);
/* the same head on a channels_last map: feat [B, P, C], C a multiple of 4 */
int cslam_gem_fc_head_nhwc_dev(const float *d_feat, float p, float eps, const float *d_W,
                               const float *d_b, int B, int C, int P, int Dout,
                               float *d_out, void *stream);
/* PCA projection + row L2, cslam/vpr/netvlad.py:234-236 (sklearn PCA.transform + normalize):
 *   y = x @ comp^T - mean_proj ;  y *= inv_scale (whitening) ;  y /= ||y|| (zero rows stay zero)
 * comp [Dout, Din]; mean_proj [Dout] = mean @ comp^T (precomputed once by the caller) or NULL;
 * inv_scale [Dout] = 1/sqrt(explained_variance) or NULL.  x [B, Din], out [B, Dout].
 * Din must be a multiple of 32; ldx / ldc are the row pitches of x / comp in floats (pad them off
 * multiples of 256 floats: power-of-two pitches alias in L2).  fp32 MFMA GEMM, deterministic split-K. */
int cslam_pca_project_dev(const float *d_x, int64_t ldx, const float *d_comp, int64_t ldc,
                          const float *d_mean_proj, const float *d_inv_scale, int B, int Din, int Dout,
                          float *d_out, void *stream);
/* The same projection for batches on the fp16 matrix pipe with fp32-grade results: x and the components as exact fp16 hi / lo
 * pairs, three partial products in fp32 accumulators (the arithmetic and the kernel of cslam_wino_gemm_h2_dev; Din is cut into
 * S splits whose partial products the epilogue sums).  d_W2: [S][Dout][Din / S / 32][hi 32 | lo 32] fp16 of sW * components
 * (`pca_pair_weights` in cslam_amd/vpr/heads.py), inv_sw = 1 / sW; Din a multiple of 32 S, Dout of 128.  x_bound > 0: a known
 * bound of max |x| (1 for the L2-normalised VLAD vectors of netvlad.py:130) spares the pass that measures it. */
int cslam_pca_project_pairs_dev(const float *d_x, int64_t ldx, float x_bound, const void *d_W2, float inv_sw, int S,
                                const float *d_mean_proj, const float *d_inv_scale, int B, int Din, int Dout,
                                float *d_out, void *stream);
/* image transform, cslam/vpr/netvlad.py:202-208 / cosplace.py:73-79:
 * CenterCrop(crop) -> Resize(out_hw, PIL bicubic, antialiased, 8-bit intermediate) ->
 * ToTensor (/255, HWC->CHW) -> Normalize(mean, std).
 * img [B, H, W, 3] uint8 RGB; out [B, 3, out_hw, out_hw] float32. */
int cslam_preprocess_dev(const uint8_t *d_img, int B, int H, int W, int crop, int out_hw,
                         const float mean[3], const float std_[3], float *d_out, void *stream);
/* the same transform, out [B, out_hw, out_hw, 3] float32: the channels_last storage the trunks' first convolution reads
 * (cosplace.py:73-79 feeds `self.model(...)` directly; the layout is this library's choice) */
int cslam_preprocess_nhwc_dev(const uint8_t *d_img, int B, int H, int W, int crop, int out_hw,
                              const float mean[3], const float std_[3], float *d_out, void *stream);

/* ---- candidate sparsifier pieces (cslam/mac/mac.py) --------------------------- */
/* grad_from_fiedler, mac.py:112-130: g[k] = w[k] * (v[i_k] - v[j_k])^2, float64 */
int cslam_mac_grad_dev(const double *d_fiedler, const int32_t *d_edge_i, const int32_t *d_edge_j,
                       const double *d_weights, int64_t m, double *d_grad, void *stream);
/* y = A x for a float64 CSR matrix (the Laplacian L(w) of mac.py:61-77), nvec dense columns,
 * x and y column-major [n, nvec]: the L @ X product inside networkx _tracemin_fiedler
 * (third-party).  One thread per (row, column): deterministic. */
int cslam_csr_spmm_dev(const int64_t *d_indptr, const int32_t *d_indices, const double *d_data,
                       int64_t n, const double *d_x, int nvec, double *d_y, void *stream);

/* same product with x, y stored [n][4] row-major (the layout of the chain-reduced solver below) */
int cslam_csr_spmm4_dev(const int64_t *d_indptr, const int32_t *d_indices, const double *d_data,
                        int64_t n, const double *d_x, double *d_y, void *stream);

/* 4-column block operations of the TraceMIN outer loop on [n][4] float64 row-major vectors
 * (the X'X, X'LX, X Y, projection and residual steps of networkx `_tracemin_fiedler`):
 *   gram:     out20[0..15] = A^T B (4x4 row-major), out20[16..19] = column sums of B
 *   affine:   out = A * M16 - shift4 (shift4 may be NULL)
 *   residual: out1 = sum_k | (W y4)[k] - sigma * X[k][0] |
 * d_partial: scratch of >= 20 * 1024 doubles.  Deterministic (fixed reduction order). */
int cslam_block4_gram_dev(const double *d_A, const double *d_B, int64_t n, double *d_partial,
                          double *d_out20, void *stream);
int cslam_block4_affine_dev(const double *d_A, int64_t n, const double *d_M16, const double *d_shift4,
                            double *d_out, void *stream);
int cslam_block4_residual_dev(const double *d_W, const double *d_X, int64_t n, const double *d_y4,
                              double sigma, double *d_partial, double *d_out1, void *stream);

/* Host-synchronous twins of the three calls above for the TraceMIN outer loop, whose 4 x 4 algebra runs on the host
 * (LAPACK, as in networkx): the 4 x 4 matrix / shift / Ritz vector are HOST pointers passed on as kernel arguments, and the
 * 20-double (1-double) result is copied to h_out and the stream synchronised inside the call. */
int cslam_block4_gram_sync(const double *d_A, const double *d_B, int64_t n, double *d_partial, double *d_out20,
                           double *h_out20, void *stream);
int cslam_block4_affine_host(const double *d_A, int64_t n, const double *h_M16, const double *h_shift4, double *d_out,
                             void *stream);
int cslam_block4_residual_sync(const double *d_W, const double *d_X, int64_t n, const double *h_y4, double sigma,
                               double *d_partial, double *d_out1, double *h_out1, void *stream);

/* Chain-reduced Laplacian solve (cslam_amd/mac/chain_solver.py): replaces the sparse-LU solves inside
 * networkx `_tracemin_fiedler` that cslam/mac/mac.py:52-58 calls (85 % of MAC's time).  Vectors are
 * [n][4] float64 row-major.  forward: segmented prefix sums of the right-hand side along the odometry
 * chains (Bn, Qn) and the reduced right-hand side bt [nJ][4] on the junction nodes; backward: closed-form
 * interior potentials from the junction solution xJ [nJ][4].  Structure arrays (is_junction [n] u8,
 * r [n-1] chain resistances, J [nJ], per-junction segment ids, per-segment end nodes sa/sb and total
 * resistance Rl, per-node Rn / jid / seg_of) are built once per Laplacian by the host.
 * d_tmp: [n][4]; d_scratch: >= 9 * ceil(n / 2048) doubles + slack (see mac_kernels.hip). */
int cslam_chain_forward_dev(const double *d_b, const uint8_t *d_is_junction, const double *d_r, int64_t n,
                            const int64_t *d_J, int nJ, const int *d_seg_start_of, const int *d_seg_end_of,
                            const int64_t *d_sa, const int64_t *d_sb, const double *d_Rl,
                            double *d_Bn, double *d_Qn, double *d_tmp, double *d_scratch, double *d_bt,
                            void *stream);
int cslam_chain_backward_dev(const double *d_xJ, const double *d_Bn, const double *d_Qn, const double *d_r,
                             const double *d_Rn, const int *d_jid, const int *d_seg_of, const int64_t *d_sa,
                             const int64_t *d_sb, const double *d_Rl, int64_t n, double *d_x, void *stream);
/* x <- (L L^T)^-1 x for the dense lower Cholesky factor L [m][ld] (float64, row-major; only the lower triangle is
 * read; col_major != 0: element (r, c) at c * ld + r instead of r * ld + c, the layout LAPACK-style library
 * factorisations return) of the grounded junction Laplacian, x [m][4]: the junction solve inside every TraceMIN iteration (SuperLU's
 * solve in networkx `_tracemin_fiedler`, called at cslam/mac/mac.py:52-58).  d_dinv: inverses of the bs x bs diagonal
 * blocks of L, [ceil(m / bs)][bs][bs] (a ragged last block in the top-left corner of its slot), d_dinvT: the same
 * blocks transposed (only the lower triangle of a d_dinv block and the upper triangle of a d_dinvT block are read);
 * d_tmp: [bs][4] scratch.  Blocked substitution, every factor element read exactly once per sweep, fixed summation order. */
int cslam_chol_solve4_dev(const double *d_L, int64_t m, int64_t ld, int col_major, const double *d_dinv,
                          const double *d_dinvT, int bs, double *d_x, double *d_tmp, void *stream);

/* (lambda_2, v_2) of a pose-graph Laplacian in ONE call, for a host without Python: replaces the whole of
 * cslam/mac/mac.py:35-59 (MAC.find_fiedler_pair -> networkx algebraic_connectivity / fiedler_vector with
 * method='tracemin_lu', seed RandomState(7)).  The Laplacian is a HOST CSR matrix (both triangles, column indices sorted
 * and unique within a row -- what mac.py:61-77 builds), n > 4 nodes, connected.  Same TraceMIN iteration as
 * cslam_amd/mac/fiedler.py; the inner solves run by the chain reduction above with the junction system factorised densely
 * on the GPU (rocBLAS / rocSOLVER, looked up at run time: CSLAM_E_UNSUPPORTED without them; at most 64000 junctions).
 *   h_x0        start block [n][4] row-major, or NULL = numpy RandomState(seed).normal(size=(4, n)).T (bit-identical;
 *               cslam_fiedler_start_block writes that block to a host buffer)
 *   tol         stopping rule ||L v - lambda v||_1 / ||L||_inf < tol (mac.py passes 1e-8)
 *   max_iters   <= 0: no practical limit (the reference has none); otherwise CSLAM_E_GRAPH when exceeded.  A graph that is
 *               not connected (networkx raises) or whose iteration breaks down returns CSLAM_E_GRAPH as well
 *   h_lambda2, h_v [n], h_iters (optional): results on the host.  The sign of v is arbitrary (as in the reference).
 * Work runs on the CURRENT device and `stream` (NULL: a non-blocking stream of the library's own; the blocked factorisation
 * also uses two side streams for its look-ahead, CSLAM_FIEDLER_LOOKAHEAD=0 keeps it on one); device memory is a workspace kept between calls (MAC calls this once per
 * Frank-Wolfe iteration), freed by cslam_fiedler_release.  Calls are serialised by a lock. */
int cslam_fiedler(int64_t n, const int64_t *h_indptr, const int32_t *h_indices, const double *h_data, const double *h_x0,
                  uint32_t seed, double tol, int max_iters, double *h_lambda2, double *h_v, int *h_iters, void *stream);
int cslam_fiedler_start_block(uint32_t seed, int64_t n, double *h_x0);
int cslam_fiedler_release(void);

/* The sparsifier's Frank-Wolfe loop around cslam_fiedler, for a host without Python: replaces cslam/mac/mac.py:191-233
 * (MAC.fw_subset) together with :61-77 (L(w) = L_fixed + sum_k w_k weight_k L_k over w_k > 1e-10), :112-130 (gradient
 * weight_k (v_i - v_j)^2), :132-147 (linear maximisation = indicator of the k largest gradients) and :168-189 (final
 * rounding: top k of w rounded to 10 decimals, ties towards the larger edge weight).  All arrays are HOST arrays:
 *   fixed_* [n_fixed], cand_* [n_cand]  edges (i, j, weight) over poses 0 .. num_poses-1 (the re-keyed edges of
 *                                       algebraic_connectivity_maximization.py:468-543; fixed includes the odometry chains)
 *   w_init [n_cand]      start point (the greedy indicator of acm.py:380-395), k edges to choose, max_iters as mac.py (5 in cslam)
 *   duality_gap_tol      1e-8 in the reference; fiedler_tol = the tol mac.py:33 passes on (1e-8)
 *   h_selected [n_cand]  1.0 for the chosen edges, 0.0 otherwise; h_w_unrounded [n_cand], h_upper, h_iters optional (NULL)
 * Exact ties in a top-k (left arbitrary by numpy's argpartition) go to the larger index.  A failing Fiedler solve (e.g. the
 * fixed edges do not connect the graph) returns that call's error; the reference's retry policy (acm.py:436-466) is the caller's. */
int cslam_mac_fw_subset(int64_t num_poses, int64_t n_fixed, const int64_t *fixed_i, const int64_t *fixed_j,
                        const double *fixed_w, int64_t n_cand, const int64_t *cand_i, const int64_t *cand_j,
                        const double *cand_w, const double *w_init, int64_t k, int max_iters, double duality_gap_tol,
                        double fiedler_tol, double *h_selected, double *h_w_unrounded, double *h_upper, int *h_iters,
                        void *stream);

/* ------------------------------------------------------------------------------------------------
 * Lidar place recognition: ScanContext bank.
 * Replaces cslam/lidar_pr/scancontext_matching.py:5-104 (ScanContextMatching) with its helpers
 * scancontext_utils.py:78-79 (sc2rk) and :81-113 (distance_sc).  Scan contexts are float64
 * [rings, sectors] row-major (the reference's np.zeros default dtype), ring keys are the row means
 * in numpy's summation order, all arithmetic float64.
 *   add_*    : scancontext_matching.py:23-42 (copies; ring keys and column norms computed on device)
 *   read_host: the reference's public `scancontexts` / `ringkeys` arrays
 *   search_* : scancontext_matching.py:44-87 for a batch of queries.  Stage 1 = num_candidates
 *              nearest ring keys (brute force instead of the per-query KD-tree, same set, ascending
 *              distance, ties -> smaller row); stage 2 = distance_sc(candidate, query) for each;
 *              best = first strict minimum below 1.0.  best_idx[j] = -1 when no candidate is below 1.0
 *              (the reference then answers items[0] with similarity 0.0 -- the Python class does that),
 *              best_sim[j] = 1 - best distance, best_yaw[j] = yaw_diff in sectors (1..sectors).
 *              row_limit[j] (optional) hides bank rows >= row_limit[j] from query j.
 *              cand / cdist / cyaw [nq, num_candidates] are optional (NULL) diagnostics: candidate
 *              rows (-1 = bank smaller than num_candidates), their distances and yaw shifts.
 */
typedef struct cslam_scbank cslam_scbank_t;
int cslam_scbank_create(int device, int rings, int sectors, int64_t capacity_hint, cslam_scbank_t **out);
int cslam_scbank_destroy(cslam_scbank_t *bank);
int cslam_scbank_size(const cslam_scbank_t *bank, int64_t *n, int *rings, int *sectors);
int cslam_scbank_clear(cslam_scbank_t *bank);
int cslam_scbank_add_host(cslam_scbank_t *bank, const double *sc, int64_t n);
int cslam_scbank_add_dev(cslam_scbank_t *bank, const double *d_sc, int64_t n, void *stream);
int cslam_scbank_read_host(const cslam_scbank_t *bank, int64_t first, int64_t count, double *sc_out,
                           double *ringkeys_out);
int cslam_scbank_search_host(cslam_scbank_t *bank, const double *queries, int64_t nq, int num_candidates,
                             const int64_t *row_limit, int64_t *best_idx, double *best_sim,
                             int32_t *best_yaw, int64_t *cand, double *cdist, int32_t *cyaw);
int cslam_scbank_search_dev(cslam_scbank_t *bank, const double *d_queries, int64_t nq, int num_candidates,
                            const int64_t *d_row_limit, int64_t *d_best_idx, double *d_best_sim,
                            int32_t *d_best_yaw, int64_t *d_cand, double *d_cdist, int32_t *d_cyaw,
                            void *stream);

/* Scan-context descriptor of float64 point clouds (cslam/lidar_pr/scancontext_utils.py:10-75 ptcloud2sc,
 * called by lidar_pr/scancontext.py:14-16 with rings x sectors = 20 x 60, max_length 80).
 * d_points: all frames concatenated, [total_points, 3]; frame f owns rows d_offsets[f] .. d_offsets[f+1]-1.
 * d_out [n_frames, rings*sectors] float64.  NaN points are skipped; a bin keeps the maximum of z + 2 over
 * its first 500 points in cloud order (the reference's storage cap), 0.0 when its storage has unused slots.
 * *d_status (device int32) is set non-zero when a point has theta == 360 exactly, where the reference
 * raises IndexError (the point is skipped here). */
int cslam_scancontext_from_cloud_dev(const double *d_points, const int64_t *d_offsets, int n_frames,
                                     int rings, int sectors, double max_length, double *d_out,
                                     int32_t *d_status, void *stream);

/* Winograd F(2x2, 3x3) transforms for the 3x3 / stride 1 / pad 1 convolutions of the extractor backbone
 * (the VGG-16 trunk built at cslam/vpr/netvlad.py:163-171; the reference runs it through torch's direct
 * convolution).  Activations are NHWC float32.  conv(x, g) + bias = output(bmm(input(x), U)) with
 * U[xi][ci][co] = (G g G^T)[xi]; the 16 GEMMs V[xi] (T x C) . U[xi] (C x Cout) are plain library GEMMs.
 *   input : x [B,H,W,C] -> V [16, T, C], T = B*ceil(H/2)*ceil(W/2) (tiles may hang over an odd map), C % 4 == 0
 *   output: M [16, T, C] -> y [B,H,W,C], or [B,floor(H/2),floor(W/2),C] when pool != 0 (the MaxPool2d(2,2) that follows
 *           the layer fused in); bias [C] or NULL; residual (NULL, or an NHWC tensor shaped like y, pool == 0:
 *           the shortcut of a ResNet block, cosplace_utils/network.py:39-56) is added before the activation;
 *           relu != 0 applies max(., 0) before the pooling. */
/* bias + ReLU (+ MaxPool2d(2,2)) in one pass over an NHWC activation, for the layers left on the direct
 * convolution: y = pool(relu(x + bias)).  Without pooling y may be x (in place); with pooling y is
 * [B,H/2,W/2,C]. */
int cslam_bias_act_pool_dev(const float *d_x, const float *d_bias, int B, int H, int W, int C, int relu,
                            int pool, float *d_y, void *stream);
/* First backbone layer (3 input channels): y = relu(conv3x3(x, w) + bias), stride 1, zero padding 1.
 * x [B,3,H,W] planar (the output of cslam_preprocess_dev), wt [27, Cout] = weight[co][ci][kh][kw] transposed
 * to [(ci*3+kh)*3+kw][co], y [B,H,W,Cout] NHWC. */
int cslam_conv3x3_c3_dev(const float *d_x, const float *d_wt, const float *d_bias, int B, int H, int W,
                         int Cout, int relu, float *d_y, void *stream);
int cslam_wino_input_dev(const float *d_x, int B, int H, int W, int C, float *d_V, void *stream);
int cslam_wino_output_dev(const float *d_M, const float *d_bias, const float *d_residual, int B, int H, int W,
                          int C, int relu, int pool, float *d_y, void *stream);
/* F(4x4, 3x3) variant: 6x6 input tiles, V / M [36, T, C] with T = B*ceil(H/4)*ceil(W/4) (ragged maps allowed), C even.
 * 4x fewer multiplications than the direct form, about one decimal digit less accurate than F(2x2, 3x3). */
int cslam_wino4_input_dev(const float *d_x, int B, int H, int W, int C, float *d_V, void *stream);
int cslam_wino4_output_dev(const float *d_M, const float *d_bias, const float *d_residual, int B, int H, int W,
                           int C, int relu, int pool, float *d_y, void *stream);
/* Split-fp16 form of the 36 per-frequency GEMMs of the same layer (opt-in; same fp32-grade result, fp16 MFMA rate):
 * a float times a power of two is split exactly into an fp16 pair hi + lo, and V U = vh uh + vl uh + vh ul is ONE fp16
 * GEMM with fp32 accumulation over K' = 3 C:  V3 [36, T, 3 C] fp16 rows = [vh | vl | vh],  U3 [36, 3 C, Cout] = [uh ; uh ; ul].
 * cslam_absmax_dev: *d_slot = bits of max |x| (n a multiple of 4); the V scale sV = 2^floor(log2(2^15 / (100 max|x|))) is
 * derived from it on the device by both transforms.  cslam_wino4_output_scaled_dev: as cslam_wino4_output_dev on
 * M' = sV sU M, multiplied by inv_su / sV (inv_su = 1 / sU, a power of two: exact) before bias / residual / ReLU
 * (d_amax NULL: M unscaled, inv_su ignored); d_amax_out (or NULL): atomic max of the bits of max |y| before pooling --
 * the next layer's d_amax without another pass over y; the caller zeroes it beforehand. */
int cslam_absmax_dev(const float *d_x, int64_t n, unsigned *d_slot, void *stream);
int cslam_wino4_output_scaled_dev(const float *d_M, const float *d_bias, const float *d_residual, int B, int H, int W,
                                  int C, int relu, int pool, const unsigned *d_amax, float inv_su, unsigned *d_amax_out,
                                  float *d_y, void *stream);
/* ---- split-fp16 Winograd GEMM of this library (csrc/wino_gemm.hip) ------------------------------------------------
 * The 36 per-frequency products M[xi] = V[xi] U[xi] of the F(4x4,3x3) form of the trunk's wide 3x3 convolutions
 * (the Conv2d layers of cslam/vpr/netvlad.py:163-171 / cosplace_utils/network.py:38-68) on the fp16 matrix pipe with
 * fp32-grade results: both operands are exact fp16 pairs hi + lo of the (power-of-two scaled) float32 values and
 * vh uh + vl uh + vh ul is accumulated in fp32.  Operand layout, both sides: row r = Cin/32 blocks of 128 bytes,
 * block kb = [hi of channels 32 kb .. 32 kb + 31 | lo of the same] as fp16; V2 [36][T][Cin/32][64] (rows = tiles),
 * U2 [36][Cout][Cin/32][64] (rows = OUTPUT channels, i.e. U transposed).
 * cslam_wino4_input_h2_dev: x [B,H,W,C] NHWC float32 -> V2 (scaled by the power of two derived from *d_amax, the bits
 *   of a bound of max |x|, see cslam_absmax_dev above); C a multiple of 32.
 * cslam_wino_gemm_h2_dev: M [36][T][Cout] float32 = (sV V)(sU U); Cin a multiple of 32, Cout of 128.  The result carries
 *   the two scales; cslam_wino4_output_scaled_dev removes them. */
int cslam_wino4_input_h2_dev(const float *d_x, int B, int H, int W, int C, const unsigned *d_amax, void *d_V2,
                             void *stream);
int cslam_wino_gemm_h2_dev(const void *d_V2, const void *d_U2, int64_t T, int Cin, int Cout, float *d_M, void *stream);
/* The same products with the column half of the output transform Y = A^T M A folded in (the "Z form", for the layers whose
 * product is HBM-bound, Cin <= 256): d_Z [24][T][Cout] float32, plane 4 i + q = sum_j M[6 i + j] A^T[q][j] -- 2/3 of M's bytes
 * written and read.  cslam_wino4_output_z_dev finishes: Y[p][q] = sum_i A^T[p][i] Z_i[q], then the epilogue of
 * cslam_wino4_output_scaled_dev (no residual input). */
int cslam_wino_zgemm_h2_dev(const void *d_V2, const void *d_U2, int64_t T, int Cin, int Cout, float *d_Z, void *stream);
int cslam_wino4_output_z_dev(const float *d_Z, const float *d_bias, int B, int H, int W, int C, int relu, int pool,
                             const unsigned *d_amax, float inv_su, unsigned *d_amax_out, float *d_y, void *stream);

/* The one-kernel F(4x4,3x3) convolution of the 64-input-channel layers (64 -> 64 / 128: VGG-16 conv1_2 and conv2_1,
 * netvlad.py:163-171 + the call at :227; the BasicBlock convolutions of ResNet-18/34 layer1, cosplace_utils/network.py:38-68,
 * with d_residual = the block's shortcut) -- input transform, the 36 products and output transform + bias + ReLU
 * (+ MaxPool2d(2,2)) with V and M kept on the compute unit; x [B,H,W,64] NHWC, y [B,H,W,Cout] or pooled -- on the fp16
 * matrix pipe with fp32-grade results (csrc/wino_fused_h.hip): V and U as exact fp16 pairs packed [hi | lo << 16] per
 * value, two v_mfma_f32_16x16x32_f16 per frequency and 16-channel quarter.  d_Uh = U [36,64,Cout] of the F(4x4) form,
 * scaled by the power of two sU, split and permuted to [kq 4][xi 36][w Cout/16][g 4][c 16][s 4] uint32 with
 * Uh[kq][xi][w][g][c][s] = pack(U[xi][16 kq + 4 g + s][16 w + c]) (vpr/winograd.py `fused64_pair_weights`), inv_su = 1/sU;
 * *d_amax = float bits of (a bound of) max |x| (fixes the power-of-two scale of V, as cslam_wino4_input_h2_dev);
 * d_amax_out (optional, zeroed by the caller) receives the bits of max |y| before pooling.  B H W 64 < 2^31. */
int cslam_wino4_fused_c64_h_dev(const float *d_x, const void *d_Uh, const float *d_bias, const float *d_residual, int B,
                                int H, int W, int Cout, int relu, int pool, const unsigned *d_amax, float inv_su,
                                unsigned *d_amax_out, float *d_y, void *stream);
/* The direct (non-Winograd) one-kernel form of the same convolution for 128 output channels (csrc/conv_direct_h.hip; VGG-16
 * conv2_1, conv2_2 of cslam/vpr/netvlad.py:163-171,227): an implicit GEMM over the 9 taps on exact fp16 pairs (three products,
 * fp32 accumulate: fp32-grade), the 18 x 18 input patch of a 16 x 16-pixel block read once into LDS, the output written once --
 * HBM sees activation in + out only.  x [B][H][W][Cin] float32 (Cin a multiple of 32), y [B][H | H/2][W | W/2][128];
 * d_w2 = [9 taps][128][Cin/32][hi 32 | lo 32] halfs of s_w w (cslam_amd/vpr/winograd.py `direct_pair_weights`), inv_sw = 1 / s_w;
 * d_amax: 4-byte slot with (a bound of) max |x|; d_amax_out (or NULL): zeroed slot receiving max |y|. */
int cslam_conv3x3_direct_h_dev(const float *d_x, const void *d_w2, const float *d_bias, int B, int H, int W, int Cin, int Cout,
                               int relu, int pool, const unsigned *d_amax, float inv_sw, unsigned *d_amax_out, float *d_y,
                               void *stream);
/* cslam_conv3x3_c3_dev that also delivers max |y| (float bits, into the zeroed 4-byte slot d_amax_out; NULL = off):
 * the first trunk layer feeds the scale of the fused fp16 layer behind it without a separate pass over its output. */
int cslam_conv3x3_c3_amax_dev(const float *d_x, const float *d_wt, const float *d_bias, int B, int H, int W, int Cout,
                              int relu, float *d_y, unsigned *d_amax_out, void *stream);

/* The same kernel with the trunk's FIRST convolution folded in (VGG-16 conv1_1 -> conv1_2, cslam/vpr/netvlad.py:163-171,227):
 * y = [MaxPool2d](ReLU(conv3x3_{64->64}(ReLU(conv3x3_{3->64}(x0) + b1)) + bias)).  d_x0: planar image batch [B][3][H][W] f32;
 * d_w1 / inv_sw / d_sumw: the first layer's weights as packed fp16 pairs, the inverse of their power-of-two scale, and
 * sum |w| per output channel [64] (`stem_pair_weights` in cslam_amd/vpr/winograd.py); d_amax_x0: 4-byte slot holding the bits of
 * max |x0|; the 64-channel intermediate never exists in HBM.  d_Uh / inv_su / d_amax_out / d_y as in cslam_wino4_fused_c64_h_dev. */
int cslam_wino4_stem_c64_h_dev(const float *d_x0, const void *d_w1, const float *d_b1, const float *d_sumw, float inv_sw,
                               const void *d_Uh, const float *d_bias, int B, int H, int W, int pool,
                               const unsigned *d_amax_x0, float inv_su, unsigned *d_amax_out, float *d_y, void *stream);

/* Strided / 1x1 / 7x7 convolution of the ResNet trunks (cslam/vpr/cosplace_utils/network.py:38-68: torchvision ResNet-18 without
 * avgpool / fc, CosPlace's default backbone, cslam/vpr/cosplace.py:81-101) as an implicit GEMM on exact fp16 pairs
 * (csrc/conv_igemm.hip): y = act(conv(x, w) + bias (+ res)), x [B,H,W,Cin] NHWC float32 -> y [B,Ho,Wo,Cout] NHWC float32,
 * Ho = (H + 2 pad - KH) / stride + 1; BatchNorm folded into w / bias by the caller.  Cin a multiple of 32, or 3 with 3 KW <= 32 (the
 * 7x7 stem); Cout a multiple of 64.  d_w2 = `igemm_pair_weights(weight)` (cslam_amd/vpr/winograd.py): rows = output channels, K blocks
 * of 32 channels as [hi 32 | lo 32] fp16 in (kh, kw, Cin / 32) order (stem: one block per kernel row, slot kw * 3 + c); inv_sw = the
 * inverse of the weights' power-of-two scale.  d_res: optional shortcut, y's shape.  d_amax_in: 4-byte slot with the float bits of
 * (a bound of) max |x|; d_amax_out: optional zeroed slot that receives max |y|.  Replaces torch.nn.functional.conv2d there. */
int cslam_conv_igemm_h2_dev(const float *d_x, const void *d_w2, const float *d_bias, const float *d_res, int B, int H, int W,
                            int Cin, int Cout, int KH, int KW, int stride, int pad, int relu, const unsigned *d_amax_in,
                            float inv_sw, unsigned *d_amax_out, float *d_y, void *stream);
/* the ResNet stem with its pooling, MaxPool2d(3, 2, 1)(ReLU(conv(x, w) + bias)): x [B,H,W,3] -> y [B,Ho/2,Wo/2,Cout], Ho a multiple
 * of 8 and Wo of 16; the un-pooled map never exists in HBM (conv1, bn1 folded, relu, maxpool of the torchvision trunk that
 * cslam/vpr/cosplace_utils/network.py:38-68 builds); d_amax_out receives max of the un-pooled map (a bound) */
int cslam_conv_stem_pool_igemm_h2_dev(const float *d_x, const void *d_w2, const float *d_bias, int B, int H, int W, int Cout,
                                      int KH, int KW, int stride, int pad, const unsigned *d_amax_in, float inv_sw,
                                      unsigned *d_amax_out, float *d_y, void *stream);
/* the same convolution between PAIR-FORMAT activations (csrc/conv_igemm.hip): a tensor [B,H,W,C], C a multiple of 32, whose every
 * (pixel, 32-channel block) is 128 bytes [hi 32 | lo 32] fp16 of s x, s = the power of two that brings the tensor's 4-byte bound slot
 * into [2^13, 2^14) -- written once by the producing layer's epilogue, read by LDS-DMA (same bytes as float32; the format never
 * leaves the trunk: the BasicBlock chain of cslam/vpr/cosplace_utils/network.py:38-68).  x_pairs / res_pairs / out_pairs select the
 * format per operand (0 = float32); d_amax_in = MEASURED max |x| (or a bound); wl1 = max_co sum |w[co]|, bmax = max |bias| give the
 * output's bound max|x| wl1 + bmax (+ *d_res_bound), stored to d_bound_out; d_amax_out receives the measured max |y| */
int cslam_conv_igemm_h2p_dev(const void *d_x, int x_pairs, const unsigned *d_xbound, const void *d_w2, const float *d_bias,
                             const void *d_res, int res_pairs, const unsigned *d_res_bound, int B, int H, int W, int Cin,
                             int Cout, int KH, int KW, int stride, int pad, int relu, const unsigned *d_amax_in, float inv_sw,
                             float wl1, float bmax, unsigned *d_amax_out, int out_pairs, unsigned *d_bound_out, void *d_y,
                             void *stream);

/* 3x3 / stride 1 / pad 1 convolution 64 -> 128 channels (cslam/vpr/netvlad.py:163-171,227: VGG-16 conv2_1) as ONE direct kernel whose
 * weights (295 KB of exact fp16 pairs) stay in the registers of the four waves of a workgroup, 32 output channels each
 * (csrc/conv_direct_r.hip).  Arguments as cslam_conv3x3_direct_h_dev with Cin = 64, Cout = 128; d_w2r = `direct_r_pair_weights`:
 * [4 output-channel quarters][9 taps][2 K steps][2 channel tiles][hi | lo][64 lanes][8 halfs]. */
int cslam_conv3x3_direct_r_dev(const float *d_x, const void *d_w2r, const float *d_bias, int B, int H, int W, int Cin, int Cout,
                               int relu, int pool, const unsigned *d_amax, float inv_sw, unsigned *d_amax_out, float *d_y,
                               void *stream);

/* 3x3 / stride 1 / pad 1 convolution 128 -> 128 channels (cslam/vpr/netvlad.py:163-171,227: VGG-16 conv2_2, + MaxPool2d) on the same
 * register-resident form: half the output channels' weights (295 KB of exact fp16 pairs) fit a compute unit's registers, the two
 * workgroups 2 j, 2 j + 1 of an XCD walk the same blocks, one per output-channel half (csrc/conv_direct_r.hip).  Arguments as
 * cslam_conv3x3_direct_r_dev with Cin = Cout = 128; d_w2r2 = `direct_r2_pair_weights`: [2 output-channel halves][4 quarters of a
 * half][9 taps][2 K steps][2 input-channel slabs][hi | lo][64 lanes][8 halfs]. */
int cslam_conv3x3_direct_r2_dev(const float *d_x, const void *d_w2r2, const float *d_bias, int B, int H, int W, int Cin, int Cout,
                                int relu, int pool, const unsigned *d_amax, float inv_sw, unsigned *d_amax_out, float *d_y,
                                void *stream);

/* conv2_1 -> conv2_2 of VGG-16 (cslam/vpr/netvlad.py:163-171,227) with the 112 x 112 x 128 map between them in the PAIR FORMAT of
 * cslam_conv_igemm_h2p_dev: cslam_conv3x3_direct_r_pairs_dev = cslam_conv3x3_direct_r_dev (+ bias + ReLU, no pooling) writing
 * [B,H,W,4 blocks][hi 32 | lo 32] fp16 of s y, s the power of two of the bound *d_amax wl1 + bmax (-> d_bound_out; wl1 = max_co
 * sum |w[co]|, bmax = max |bias|); cslam_conv3x3_direct_hp_dev = cslam_conv3x3_direct_h_dev reading such a map (d_xbound = its bound slot)
 * and staging its patch without the split into pairs (csrc/conv_direct_r.hip, csrc/conv_direct_h.hip). */
int cslam_conv3x3_direct_r_pairs_dev(const float *d_x, const void *d_w2r, const float *d_bias, int B, int H, int W, int Cin, int Cout,
                                     const unsigned *d_amax, float inv_sw, float wl1, float bmax, unsigned *d_amax_out,
                                     unsigned *d_bound_out, void *d_y, void *stream);
int cslam_conv3x3_direct_hp_dev(const void *d_x, const unsigned *d_xbound, const void *d_w2, const float *d_bias, int B, int H, int W,
                                int Cin, int Cout, int relu, int pool, float inv_sw, unsigned *d_amax_out, float *d_y, void *stream);

/* 3x3 / stride 1 / pad 1 convolution 64 -> 64 channels between PAIR-FORMAT maps (the four stride-1 convolutions of ResNet-18/34's layer1:
 * cslam/vpr/cosplace_utils/network.py:38-68, the reference's default extractor) as ONE direct kernel whose weights stay in the registers
 * of the four waves of a workgroup, 16 output channels each, the input patch arriving by LDS-DMA (csrc/conv_direct_p.hip).  The
 * contract of cslam_conv_igemm_h2p_dev with KH = KW = 3, stride = pad = 1: d_x / d_xbound the pair-format input and its bound slot
 * (x_pairs = 0: a float32 NHWC map, split while its patch is staged; no shortcut then), d_amax_in the measured max |x|; d_res optional
 * shortcut in pair format (res_pairs, d_res_bound = its bound slot) or float32
 * NHWC (d_res_bound = a bound of max |res|); out_pairs: y in pair format scaled for max|x| wl1 + bmax (+ *d_res_bound) -> d_bound_out,
 * else float32 NHWC; d_amax_out (optional, zeroed) receives max |y|.  d_w2r / inv_sw = `stem_direct_pair_weights(weight)`:
 * [4 output-channel quarters][9 taps][2 K steps][hi | lo][64 lanes][8 halfs]. */
int cslam_conv3x3_direct_p_dev(const void *d_x, int x_pairs, const unsigned *d_xbound, const void *d_w2r, const float *d_bias, const void *d_res,
                               int res_pairs, const unsigned *d_res_bound, int B, int H, int W, int Cin, int Cout, int relu,
                               const unsigned *d_amax_in, float inv_sw, float wl1, float bmax, unsigned *d_amax_out, int out_pairs,
                               unsigned *d_bound_out, void *d_y, void *stream);

/* The same pair of layers (cslam/vpr/netvlad.py:163-171,227: VGG-16 conv1_1 + ReLU + conv1_2 + ReLU (+ MaxPool2d)) as ONE DIRECT
 * convolution kernel (csrc/conv_stem_direct_h.hip): the 147 KB of second-layer weights (exact fp16 pairs) stay in the registers of the
 * four waves of a workgroup, each wave owning 16 of the 64 intermediate channels; no Winograd transforms, no weight stream.
 * d_w1 / inv_sw1 / d_sumw as above; d_w2r / inv_sw2: the second layer's weights in MFMA-fragment order
 * [4 channel quarters][9 taps][2 x 32 output channels][hi | lo][64 lanes][8 halfs] (`stem_direct_pair_weights`). */
int cslam_conv_stem_direct_h_dev(const float *d_x0, const void *d_w1, const float *d_b1, const float *d_sumw, float inv_sw1,
                                 const void *d_w2r, const float *d_bias, float inv_sw2, int B, int H, int W, int pool,
                                 const unsigned *d_amax_x0, unsigned *d_amax_out, float *d_y, void *stream);

/* ---- multi-GPU exchange (csrc/comm.hip): RCCL over xGMI, one process per GPU ---------------------------------------
 * Replaces, inside one node, the ROS 2 transport of descriptors between robots
 * (cslam/global_descriptor_loop_closure_detection.py:198-227 publish GlobalDescriptors, :407-422 receive): rank g owns
 * robot g's bank (BASELINE config 4) or a row shard of ONE bank (the metric's bank at N > 1 GPUs).  Per step: ONE
 * all-gather of the new descriptors; every rank searches all of them in its bank (cslam_bank_search_dev); for a
 * row-sharded bank ONE all-to-all returns the (rows | score bits | count) lists to the keyframes' owners, who merge them
 * with cslam_topk_merge_dev.  cslam_amd/sharded.py is the Python twin over torch.distributed.
 * RCCL is looked up at run time (dlopen); without it these return CSLAM_E_UNSUPPORTED and nothing else is affected.
 *   cslam_comm_unique_id   rank 0: 128 bytes to hand to every rank (file, socket, launcher environment ...)
 *   cslam_comm_init        collective over all `world` processes; `device` = this rank's HIP device
 *   cslam_allgather_queries_dev  d_local [rows][row_bytes] -> d_all [world * rows][row_bytes], rank-major, on `stream`
 *   cslam_exchange_lists_dev     slice r of d_send [world][bytes_per_rank] goes to rank r; slice s of d_recv came from s */
int cslam_comm_unique_id(void *id128);
int cslam_comm_init(int world, int rank, const void *id128, int device, cslam_comm_t **out);
int cslam_comm_destroy(cslam_comm_t *comm);
int cslam_comm_info(const cslam_comm_t *comm, int *world, int *rank);
int cslam_allgather_queries_dev(cslam_comm_t *comm, const void *d_local, int64_t rows, int64_t row_bytes, void *d_all,
                                void *stream);
int cslam_exchange_lists_dev(cslam_comm_t *comm, const void *d_send, void *d_recv, int64_t bytes_per_rank, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CSLAM_HIP_H */
