/* cslam_hip_experimental.h -- entry points of libcslam_hip.so that are NOT part of the stable C ABI (include/cslam_hip.h):
 * A/B partners of the product kernels (the f32-MFMA one-kernel convolutions, round 1's library-GEMM operand layout),
 * profiling hooks and the bench's peak micro-benchmarks.  They may change or disappear between rounds; nothing of the
 * product path (the defaults of the Python layer) calls them except where stated. */
#ifndef CSLAM_HIP_EXPERIMENTAL_H
#define CSLAM_HIP_EXPERIMENTAL_H
#include "cslam_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Round 1's operand layout of the split-fp16 Winograd products (one library GEMM over K' = 3 C: V3 [36, T, 3 C] fp16 rows
 * = [vh | vl | vh]); `frontend.winograd_gemm: library` in vpr/winograd.py.  Superseded by cslam_wino4_input_h2_dev +
 * cslam_wino_gemm_h2_dev (V stored once, at its fp32 size). */
int cslam_wino4_input_h3_dev(const float *d_x, int B, int H, int W, int C, const unsigned *d_amax, void *d_V3, void *stream);

/* The same 3x3 / stride 1 / pad 1 convolution for 64 -> 64 channels (VGG-16 conv1_2, netvlad.py:163-171 +
 * the call at :227) as ONE kernel: F(2x2,3x3) input transform, the 16 per-frequency products on the fp32 MFMA pipe
 * and output transform + bias + ReLU (+ MaxPool2d(2,2)) with V and M kept on the compute unit (csrc/wino_fused.hip).
 * x [B,H,W,64] NHWC; Up = U [16,64,64] of the F(2x2) form permuted to [kq 4][xi 16][w 4][g 4][c 16][s 4] with
 * Up[kq][xi][w][g][c][s] = U[xi][16 kq + 4 g + s][16 w + c]; y [B,H,W,64] or [B,H/2,W/2,64] (pool). */
int cslam_wino2_fused64_dev(const float *d_x, const float *d_Up, const float *d_bias, int B, int H, int W,
                            int relu, int pool, float *d_y, void *stream);
/* The same kernel for 64 -> Cout channels, Cout = 64 or 128 (VGG-16 conv1_2 and conv2_1; the BasicBlock convolutions
 * of ResNet-18/34 layer1, cosplace_utils/network.py:38-68, with d_residual [B,H,W,Cout] = the block's shortcut, added
 * before the ReLU, or NULL): Up = U [16,64,Cout] permuted to [kq 4][xi 16][w Cout/16][g 4][c 16][s 4];
 * y [B,H,W,Cout] or [B,H/2,W/2,Cout] (pool; not together with d_residual). */
int cslam_wino2_fused_c64_dev(const float *d_x, const float *d_Up, const float *d_bias, const float *d_residual, int B,
                              int H, int W, int Cout, int relu, int pool, float *d_y, void *stream);
/* The F(4x4,3x3) form of the same one-kernel convolution (36 frequencies, 4 x 4 blocks of 4 x 4-pixel tiles per
 * workgroup step; 1.78x fewer MFMAs per output pixel): Up = U [36,64,Cout] of the F(4x4) form permuted to
 * [kq 4][xi 36][w Cout/16][g 4][c 16][s 4]; everything else as cslam_wino2_fused_c64_dev. */
int cslam_wino4_fused_c64_dev(const float *d_x, const float *d_Up, const float *d_bias, const float *d_residual, int B,
                              int H, int W, int Cout, int relu, int pool, float *d_y, void *stream);

/* diagnostics of the kernel above: with CSLAM_WFH_PROF=1 in the environment its conv1_2-shaped launches (ReLU + pool, 64
 * output channels; stem or not) add up, for waves 0 and 4 of workgroup 0, the shader cycles spent per phase of a quarter into
 * d_buf16 [2][8] uint64 = (transform, barrier, matrix loop, prefetch + stem work, output transform, barrier, quarters, -). */
int cslam_debug_wfh_prof_dev(void *d_buf16);

/* ---- diagnostics: in-run re-measurement of the peaks rooflines are priced against (csrc/peaks.hip) ---------------
 * Not on the extract / match path and without a reference counterpart: bench.py reports every roofline fraction against
 * the nominal MI355X peaks and against what these two kernels sustain on the box in the same run.
 * cslam_peak_copy_dev: 16-byte-per-lane streaming copy of `bytes` (multiple of 16) bytes; variant 0 = one element per
 *   thread, 1 / 2 = grid-stride with plain / non-temporal accesses (the caller keeps the fastest).
 * cslam_peak_mfma_dev: register-resident MFMA loop; kind 0 = f32 inputs (v_mfma_f32_32x32x2_f32), 1 = fp16 inputs
 * (v_mfma_f32_32x32x16_f16); `blocks` workgroups of 4 waves, `iters` x 4 independent MFMAs per wave; operands 0 = zeros (the
 * ceiling: an idle datapath leaves the chip at its top clock), 1 = non-zero values (the rate real data can reach: the chip
 * clocks to its power budget); *flop_out = flop of the launch; d_scratch (>= 16 bytes) receives the loop's shader cycles as a
 * uint64 at byte 8 (cycles / elapsed time = the clock it ran at).  The caller times both with HIP events on `stream`. */
int cslam_peak_copy_dev(const void *d_src, void *d_dst, int64_t bytes, int variant, void *stream);
int cslam_peak_mfma_dev(int kind, int iters, int blocks, int operands, float *d_scratch, double *flop_out, void *stream);

/* Per-launch timing of the trunk's pair products (cslam_wino_gemm_h2_dev / cslam_wino_zgemm_h2_dev) inside a run: while enabled
 * every launch is bracketed by two HIP events on its own stream.  cslam_trunk_timing_read: out[0..3] = launches, ms, fp16 flop
 * (3 products), algorithmic HBM bytes (V2 in + M or Z out + U2) of the launches with Cin <= 256, out[4..7] the same for
 * Cin > 256, since the last read (waits for the events, clears the log; at most 512 launches are kept).  bench.py's
 * `roofline_step_largest` is computed from these over the timed steps. */
int cslam_trunk_timing(int enable);
int cslam_trunk_timing_read(double out[8]);

/* The candidate lists stage 1 of the last MFMA-mode search of `bank` left in its workspace (valid until the bank's next
 * search): keys [nq][*nseg][16] float32 in units of q.b / ||b|| (sorted per segment, -inf = empty), rows [nq][*nseg][16]
 * (-1 = empty), and the bound on |key - exact| / ||q|| handed to the float64 certificate.  tests/test_nns_gpu.py checks the
 * measured error of the fp16-pair stage against that bound. */
int cslam_debug_last_candidates(cslam_bank_t *bank, int64_t nq, int *nseg, float *keys, int *rows, double *err_bound);

/* The static schedule of the persistent candidate stage (csrc/sim_topk_ring.hip; the matcher's D.D^T of cslam/nns_matching.py:55-61
 * on 256 x 256 tiles) for `nqt` query tiles x `n_rows` bank rows on n_xcd XCDs of wpx workgroups.  Host code only (no GPU needed).
 * info = {Sq, Sb, tasks, lists per query summed over query tiles, 0, int32 words per task (8)};
 * tasks [info[2]][8] = {query tile, first bank row, tiles, list number, run ordinal in its XCD, rows between tile starts, end row, 0}
 * (tile i = rows [first + i * stride, + 256) cut at the end row), task_off [n_xcd * wpx + 1] (workgroup w = XCD w / wpx, slot
 * w % wpx), qt_nseg / qt_segoff [nqt].  NULL outputs are skipped. */
int cslam_ring_schedule_describe(int nqt, int n_rows, int n_xcd, int wpx, int32_t info[6], int32_t *tasks, int64_t tasks_cap,
                                 int32_t *task_off, int32_t *qt_nseg, int32_t *qt_segoff);

#ifdef __cplusplus
}
#endif
#endif /* CSLAM_HIP_EXPERIMENTAL_H */
