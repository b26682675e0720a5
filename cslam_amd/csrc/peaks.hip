// peaks.hip -- in-run re-measurement of the peaks the rooflines are priced against (gfx950).
//
// bench.py prints every roofline fraction against the nominal MI355X figures of
// /opt/skills/guides/MI355X_MICROARCH.md (HBM 8 TB/s, f32-input MFMA 157.3 TFLOP/s, fp16 MFMA 2.5 PFLOP/s) AND against
// what this very box sustains, measured in the same run with the two kernels below:
//   peak_copy_*       16-byte-per-lane streaming copies (one element per thread; grid-stride, plain or non-temporal; the
//                     caller keeps the best): achievable HBM bandwidth
//   peak_mfma_kernel  register-resident MFMA loop, 4 independent accumulators per wave, 1 or 2 waves per SIMD: the matrix
//                     rate for f32 inputs (v_mfma_f32_32x32x2_f32) and fp16 inputs (v_mfma_f32_32x32x16_f16) with ZERO
//                     operands (the ceiling: the guide's micro-benchmark figures) and with non-zero operands (what real
//                     data can reach: the chip clocks to its power budget), each with the clock it ran at
// Diagnostics only: nothing on the extract / match path calls them.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// NT: non-temporal loads / stores.  Four independent 16-byte loads in flight per lane, grid-stride.
template <bool NT>
__global__ __launch_bounds__(256) void peak_copy_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst,
                                                        int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        f32x4 a, b, c, d;
        if (NT) {
            a = __builtin_nontemporal_load(src + i); b = __builtin_nontemporal_load(src + i + stride);
            c = __builtin_nontemporal_load(src + i + 2 * stride); d = __builtin_nontemporal_load(src + i + 3 * stride);
            __builtin_nontemporal_store(a, dst + i);
            __builtin_nontemporal_store(b, dst + i + stride);
            __builtin_nontemporal_store(c, dst + i + 2 * stride);
            __builtin_nontemporal_store(d, dst + i + 3 * stride);
        } else {
            a = src[i]; b = src[i + stride]; c = src[i + 2 * stride]; d = src[i + 3 * stride];
            dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
        }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

// one 16-byte element per thread, as many workgroups as it takes (the plain "float4 copy" of the MI355X guide)
__global__ __launch_bounds__(256) void peak_copy_flat_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst,
                                                             int64_t n16) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

// variant 0: one element per thread; 1 / 2: grid-stride with 16 workgroups per compute unit, plain / non-temporal
CSLAM_API int cslam_peak_copy_dev(const void *d_src, void *d_dst, int64_t bytes, int variant, void *stream) {
    PTR_DEVICE(d_src);
    ARG_CHECK(d_src && d_dst && bytes >= 0 && bytes % 16 == 0, "NULL pointer or size not a multiple of 16");
    ARG_CHECK((((uintptr_t)d_src | (uintptr_t)d_dst) & 15) == 0, "pointers must be 16-byte aligned");
    ARG_CHECK(variant >= 0 && variant <= 2, "variant must be 0, 1 or 2");
    if (bytes == 0) return CSLAM_OK;
    const int64_t n16 = bytes / 16;
    hipStream_t st = (hipStream_t)stream;
    if (variant == 0) {
        ARG_CHECK(ceil_div64(n16, 256) < (1LL << 31), "too large for one launch");
        hipLaunchKernelGGL(peak_copy_flat_kernel, dim3((unsigned)ceil_div64(n16, 256)), dim3(256), 0, st,
                           (const f32x4 *)d_src, (f32x4 *)d_dst, n16);
    } else {
        int64_t blocks = ceil_div64(n16, 256 * 4);
        if (blocks > 256 * 16) blocks = 256 * 16;
        if (variant == 1) hipLaunchKernelGGL(peak_copy_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, (const f32x4 *)d_src, (f32x4 *)d_dst, n16);
        else hipLaunchKernelGGL(peak_copy_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, (const f32x4 *)d_src, (f32x4 *)d_dst, n16);
    }
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

// OPERANDS: 0 = zeros (the datasheet-style ceiling: an idle datapath draws little power and the chip holds its top clock),
// 1 = non-zero values of mixed sign (what a kernel with real data can reach: under matrix load the chip clocks to its power
// budget, guide rule 25).  Wave 0 of workgroup 0 leaves the s_memtime ticks its loop took in out[2..3] (on this part they do not
// follow the shader clock -- 0.67 x the clock the measured rate implies -- so bench.py reports ceiling / loaded, not a clock).
template <int KIND>
__global__ __launch_bounds__(256) void peak_mfma_kernel(int iters, int operands, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (KIND == 0) {
        float av = operands ? 1.0f + 0.001f * lane : 0.0f, bv = operands ? 0.5f - 0.002f * lane : 0.0f;
        for (int it = 0; it < iters; it += 4) {               // 16 MFMAs per trip: the loop's own instructions are ~1 % of its issue slots
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(u & 1 ? -av : av, bv, acc[a], 0, 0, 0);
        }
    } else {
        f16x8 av, bv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            av[e] = (_Float16)(operands ? 0.5f + 0.01f * (lane + e) : 0.0f);
            bv[e] = (_Float16)(operands ? 0.25f - 0.003f * (lane - e) : 0.0f);
        }
        const f16x8 an = -av;                                  // alternating signs keep the sums bounded without leaving the registers
        for (int it = 0; it < iters; it += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(u & 1 ? an : av, bv, acc[a], 0, 0, 0);
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.678f) out[0] = s;                            // never true: keeps the loop alive
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) *(unsigned long long *)(out + 2) = t1 - t0;
}

// kind 0: f32 inputs (2*32*32*2 flop per MFMA), kind 1: fp16 inputs (2*32*32*16); operands 0 = zeros, 1 = non-zero (above).
// `flop_out` receives the flop count of the launch: blocks x 4 waves x iters x 4 MFMAs; d_scratch (>= 16 bytes, 8-byte aligned)
// receives the loop's shader cycles as a uint64 at byte 8.
CSLAM_API int cslam_peak_mfma_dev(int kind, int iters, int blocks, int operands, float *d_scratch, double *flop_out, void *stream) {
    PTR_DEVICE(d_scratch);
    ARG_CHECK((kind == 0 || kind == 1) && iters > 0 && blocks > 0 && d_scratch, "bad arguments");
    if (kind == 0) hipLaunchKernelGGL(peak_mfma_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, operands, d_scratch);
    else hipLaunchKernelGGL(peak_mfma_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, operands, d_scratch);
    HIP_TRY(hipGetLastError());
    if (flop_out) *flop_out = (double)blocks * 4.0 * iters * 4.0 * 2.0 * 32 * 32 * (kind == 0 ? 2 : 16);
    return CSLAM_OK;
}
