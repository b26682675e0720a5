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
//
// NPROD = 1 (round 4, the default candidate stage): ONE product, q.b ~ qh.bh, on the hi halves alone.  The certificate does
// not need fp32-grade keys, it needs a RIGOROUS bound -- and the bound of the three-product form is dominated by its worst-case
// accumulation term (3 kd fp32 additions, unit 2^-23): 1.57e-3 at kd = 4096.  With one product only kd products are summed
// (accumulation 1.002 (kd + 64) 2^-23 = 4.97e-4) and the representation term becomes the fp16 rounding of both operands,
// |w - hi| <= 2^-11 |w| + 2^-25 each, i.e. 2 (2^-11 + 2^-22) by Cauchy-Schwarz: 9.77e-4 -- together 1.566e-3, the SAME bound
// (pair_err_bound(kd, 1) below), so the re-scoring window, the contender counts and the certification rate of stage 2 are
// those of the pair stage, at a third of the matrix work and half the operand bytes: the bank keeps `rowsh`, its rows times
// the same power of two rounded to fp16 (2 bytes per value), a K stage is 64 channels = 128 bytes per row (same loader, same
// LDS image), four 16-channel K steps of one MFMA each.  The governing roofline is the dense fp16 MFMA peak itself.
#include <stdlib.h>
#include <type_traits>
#include <hip/hip_fp16.h>
#include "bank.h"
#include "sim_topk.h"

#include "sim_topk_pair_dev.h"

// T_ = tile edge (bank rows = queries per tile), MT = 32-row MFMA tiles per wave along the bank axis (wave tile =
// 32 MT x 64), KPL = per-lane candidate list length.  Waves: 2 along the bank axis x (T_/64) along the query axis.
// NPROD = 3: operands are [hi 32 | lo 32] pair blocks, a stage = 32 channels = two K steps of three MFMAs per tile;
// NPROD = 1: operands are fp16 rows, a stage = 64 channels = four K steps of one MFMA per tile.
// DBG != 0 (builds with -DCSLAM_ABLATIONS only): TIMING-ONLY ablations (wrong results): 1 = no global loads after the first stage,
// 2 = every workgroup loads bank tile 0 / query tile 0 over and over (every request an L2 hit: the structure at L2 latency).
// NTW = 32-query MFMA tiles per wave along the query axis: 2 = eight waves of 256 registers (wave tile 128 x 64), 4 = FOUR waves of 512
// registers (wave tile 128 x 128: 256 accumulators in AGPRs, 8 fragment reads per 16 MFMAs instead of 6 per 8 -- a third fewer LDS
// bytes per MFMA; round 4, after csrc/wino_gemm.hip's four-wave kernel: measured slower here, instantiated in the measurement build only).
template <int T_, int MT, int KPL, int NPROD, int DBG, int NTW>
__global__ __launch_bounds__(T_ * 64 / (16 * NTW), NTW == 4 ? 1 : 2) void sim_topk_pair_kernel(PairArgs p) {
    constexpr int NWN = T_ / (32 * NTW);
    constexpr int NTHR = 2 * NWN * 64;
    constexpr int OPB = T_ * PK_ROWB;            // bytes of one operand tile (one 32-channel block of T_ rows) in LDS
    constexpr int STAGE = 2 * OPB;
    constexpr int NLD = T_ * 8 / NTHR;           // 16-byte chunks per thread per operand (= 4)
    static_assert(T_ == 64 * MT, "wave tile must cover half the bank tile");
    static_assert(NPROD == 1 || NPROD == 3, "one product on the hi halves, or three on hi / lo pairs");
    constexpr int NS = NPROD == 3 ? 2 : 4;      // 16-channel K steps per stage
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

    float lk[NTW][KPL]; int li[NTW][KPL];
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int j = 0; j < KPL; ++j) { lk[n][j] = -INFINITY; li[n][j] = -1; }
    int lim[NTW];
    float qmul[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        lim[n] = p.lim[qt * T_ + wn * (32 * NTW) + n * 32 + l31];
        qmul[n] = p.qinvs[qt * T_ + wn * (32 * NTW) + n * 32 + l31];
    }

    const int ntiles = t_end - t_beg;
    if (ntiles > 0) {
        // ---- loader: chunk pch = i*NTHR + tid -> tile row pch >> 3, physical 16-byte slot pch & 7 holding logical chunk
        // slot ^ ((row >> 1) & 7) of the row's 128-byte block
        int voffA[NLD], voffB[NLD];                              // byte offset of the lane's chunk i inside the bank / query tile
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int pch = i * NTHR + tid;
            const int r = pch >> 3, slot = pch & 7;
            const int c = slot ^ ((r >> 1) & 7);
            voffA[i] = r * (int)p.ldb2 + (c << 4);
            voffB[i] = r * (int)p.ldq2 + (c << 4);
        }
        const int wave_chunk = wave * 1024;
        // the query copy is padded to whole tiles
        const __amdgpu_buffer_rsrc_t rsB = pk_rsrc(p.q2 + (int64_t)(DBG == 2 ? 0 : qt) * T_ * p.ldq2, (int64_t)T_ * p.ldq2);
        // the bank tile the loader points at: a buffer resource over its rows (rows beyond the bank read as zero; the epilogue
        // masks them), rebuilt when the loader moves to another tile (once per nkt stages, on the scalar unit)
        __amdgpu_buffer_rsrc_t rsA;
        auto point_at_tile = [&](int tile) {
            const int t0 = DBG == 2 ? 0 : tile;
            int64_t rows = (int64_t)p.n_rows - (int64_t)t0 * T_;
            if (rows > T_) rows = T_;
            rsA = pk_rsrc(p.bank2 + (int64_t)t0 * T_ * p.ldb2, rows * p.ldb2);
        };
        auto stage_load_part = [&](int stage, int kt, int i) {
            char *sA = smem + stage * STAGE;
            char *sB = sA + OPB;
            pk_blds16(rsA, voffA[i], kt * PK_ROWB, sA + i * (NTHR * 16) + wave_chunk);
            pk_blds16(rsB, voffB[i], kt * PK_ROWB, sB + i * (NTHR * 16) + wave_chunk);
        };

        // fragment read offsets: row * 128 + (chunk ^ swz) * 16, chunk = 4 lo + 2 s + h for K step s (16 channels) of the stage
        // (NPROD = 1: the 128-byte block is 64 channels of hi halves, chunk = 2 s + h for s < 4)
        const int swz = (lane >> 1) & 7;
        int foff[NS][2];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int lo = 0; lo < 2; ++lo) foff[s][lo] = ((4 * lo * (NPROD == 3) + 2 * s + h) ^ swz) << 4;
        const int arow0 = (wm * 32 * MT + l31) * PK_ROWB;
        const int brow0 = (wn * (32 * NTW) + l31) * PK_ROWB;

        f32x16 acc[MT][NTW];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NTW; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

        // The K loop is software-pipelined ACROSS the stage barrier (round 4).  Round 3's loop read a K step's fragments right in
        // front of its MFMAs (ds_read, s_waitcnt lgkmcnt(0), two MFMAs, ds_read, ... in the ISA): with both waves of a SIMD in
        // lockstep behind the same barrier the LDS latency was exposed NS times per stage, and with no global loads at all the
        // one-product form held only 1.18 PFLOP/s (profiles/r04_v3_match_ablations.log).  Now the fragments live in two register
        // sets: K step s + 1 is read while K step s is multiplied, and the stage's LAST K step is multiplied AFTER the barrier,
        // under the reads of the next stage's first K step and the LDS-DMA requests of the stage after it:
        //     [barrier]  read (it, 0) | request stage it + 1 | MFMA (it - 1, NS - 1) | tile epilogue if (it - 1) closed a tile
        //                read (it, s + 1) | MFMA (it, s)            for s = 0 .. NS - 2
        //                wait: my requests landed, my reads done;  [barrier]
        // Buffers: stage it lives in LDS buffer it & 1.  Behind the barrier every wave has finished READING stage it - 1 (its last
        // reads were waited for in front of the barrier), so its buffer takes stage it + 1; a request has a whole stage of MFMAs
        // to land.  Iteration 0 multiplies zero fragments (the accumulators start at zero: nothing changes).
        // (The three-product form keeps round 3's order -- read a K step, multiply it, one barrier per stage: its 24 MFMAs per
        // K step cover most of the read latency, and two fragment sets of hi AND lo halves do not fit beside the accumulators.)
        constexpr int NF = NPROD == 3 ? 2 : 1;           // fragment kinds per operand: hi (| lo)
        constexpr int NBUF = NPROD == 3 ? 1 : 2;         // fragment register sets
        f16x8 fa[NBUF][NF][MT], fb[NBUF][NF][NTW];
#pragma unroll
        for (int u = 0; u < NBUF; ++u)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[u][f][m] = (f16x8)(_Float16)0.0f;
#pragma unroll
                for (int n = 0; n < NTW; ++n) fb[u][f][n] = (f16x8)(_Float16)0.0f;
            }
        auto read_frags = [&](int u, const char *sA, const char *sB, int s) {       // u, s: compile-time after unrolling
#pragma unroll
            for (int f = 0; f < NF; ++f) {
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[u][f][m] = *(const f16x8 *)(sA + arow0 + m * 32 * PK_ROWB + foff[s][f]);
#pragma unroll
                for (int n = 0; n < NTW; ++n) fb[u][f][n] = *(const f16x8 *)(sB + brow0 + n * 32 * PK_ROWB + foff[s][f]);
            }
        };
        auto multiply = [&](int u) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NTW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[u][0][m], fb[u][0][n], acc[m][n], 0, 0, 0);
            if (NPROD == 3) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NTW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[u][NF - 1][m], fb[u][0][n], acc[m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NTW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[u][0][m], fb[u][NF - 1][n], acc[m][n], 0, 0, 0);
            }
        };
        constexpr int G = MT * NTW * NPROD;                // MFMAs of a K step: 24 | 12 | 8 | 4
        constexpr int NRD = NF * (MT + NTW);               // fragment reads of a K step: 12 | 8 | 6 | 4

        const int total = ntiles * p.nkt;
        point_at_tile(t_beg);
#pragma unroll
        for (int i = 0; i < NLD; ++i) stage_load_part(0, 0, i);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_s_barrier();

        int tile = t_beg, kt = 0;                        // stage `it` = (tile, kt)
        if constexpr (NPROD == 3) {
            for (int it = 0; it < total; ++it) {
                int nkt_ = kt + 1, ntile = tile;
                if (nkt_ == p.nkt) { nkt_ = 0; ntile = tile + 1; }
                const bool more = it + 1 < total;
                const int lkt = more ? nkt_ : kt;
                if (more && nkt_ == 0) point_at_tile(ntile);
                const char *sA = smem + (it & 1) * STAGE;
                const char *sB = sA + OPB;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    read_frags(0, sA, sB, s);
                    multiply(0);
                    // all of the next stage's requests among the MFMAs of the FIRST K step, one per three MFMAs: the second K
                    // step covers their L2 round trip (profiles/r03_v12_pair_load_placement_ab.log)
                    if (s == 0) {
                        constexpr int NL = 2 * NLD;
                        constexpr int PER = G / NL > 0 ? G / NL : 1;
                        if (DBG != 1) {
#pragma unroll
                            for (int i = 0; i < NLD; ++i) stage_load_part((it + 1) & 1, lkt, i);
                        }
#pragma unroll
                        for (int i = 0; i < NL; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);      // MFMA
                            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);        // VMEM read (LDS-DMA)
                        }
                        if (G - PER * NL > 0) __builtin_amdgcn_sched_group_barrier(0x008, G - PER * NL > 0 ? G - PER * NL : 1, 0);
                    }
                }
                if (kt == p.nkt - 1)
                    pair_tile_epilogue<MT, KPL, NTW>(acc, lk, li, lim, qmul, p.invs, p.n_rows, tile * T_ + wm * 32 * MT + 4 * h);
                __builtin_amdgcn_s_waitcnt(0);       // next stage landed (vmcnt(0)), this stage's fragment reads done
                __builtin_amdgcn_s_barrier();
                kt = nkt_; tile = ntile;
            }
        } else {
            // One stage.  FIRST = the first stage of a bank tile: the previous stage's last K step was already multiplied in
            // front of that tile's epilogue (the rotation runs inside a tile only), so nothing is carried in.
            int it = 0;
            auto stage_body = [&](auto first_tag, int tile, int kt) {
                constexpr bool FIRST = decltype(first_tag)::value;
                int nkt_ = kt + 1, ntile = tile;
                if (nkt_ == p.nkt) { nkt_ = 0; ntile = tile + 1; }
                // branch-free prefetch: the very last stage re-fetches its own block into the idle buffer, which nobody reads
                const bool more = it + 1 < total;
                const int lkt = more ? nkt_ : kt;
                if (more && nkt_ == 0) point_at_tile(ntile);             // wave-uniform, before anything is in flight
                const char *sA = smem + (it & 1) * STAGE;
                const char *sB = sA + OPB;
                // Every region below is fenced (sched_barrier) and ordered inside (sched_group_barrier).  hipcc (ROCm 7.2) cannot
                // count LDS reads while a global_load_lds is pending (a FLAT-encoded instruction with an LDS operand: its waitcnt
                // pass then turns every lgkmcnt(N) into lgkmcnt(0)), so a region's MFMAs wait for ALL reads issued before them.
                // Hence the next K step's reads go BEHIND the first LEAD MFMAs of a region: the wait in front of the region
                // finds only reads that the rest of the previous region (and the partner wave's MFMAs) have covered.
                // A global_load_lds costs 60-180 issue cycles (back to back they idle the matrix pipe, sim_topk_mfma.hip): one
                // per two MFMAs, half of the stage's requests among the carried-in K step's MFMAs, half in K step 0 (FIRST: all
                // in K step 0).
                constexpr int HALF = FIRST ? 0 : NLD / 2;                // loader parts (2 requests each) issued behind the barrier
                constexpr int LEAD = G >= 8 ? 2 : 1;
                // ---- behind the barrier: first reads of this stage, requests of the next, the last K step of the previous
                read_frags(0, sA, sB, 0);
                if constexpr (!FIRST) {
                    if (DBG != 1) {
#pragma unroll
                        for (int i = 0; i < HALF; ++i) stage_load_part((it + 1) & 1, lkt, i);
                    }
                    multiply((NS - 1) & 1);
                    __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);              // DS read (nothing older is needed here)
                    constexpr int PER = G / (2 * HALF) > 0 ? G / (2 * HALF) : 1;
#pragma unroll
                    for (int i = 0; i < 2 * HALF; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);          // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);            // VMEM read (LDS-DMA)
                    }
                    if (G - PER * 2 * HALF > 0) __builtin_amdgcn_sched_group_barrier(0x008, G - PER * 2 * HALF > 0 ? G - PER * 2 * HALF : 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- K steps 0 .. NS - 2 of this stage, each over the reads of the next
#pragma unroll
                for (int s = 0; s + 1 < NS; ++s) {
                    read_frags((s + 1) & 1, sA, sB, s + 1);
                    if (s == 0 && DBG != 1) {
#pragma unroll
                        for (int i = HALF; i < NLD; ++i) stage_load_part((it + 1) & 1, lkt, i);
                    }
                    multiply(s & 1);
                    __builtin_amdgcn_sched_group_barrier(0x008, LEAD, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
                    constexpr int REST = 2 * (NLD - HALF);               // requests placed in K step 0
                    constexpr int GR = G - LEAD;
                    if (s == 0) {
                        constexpr int PER = GR / REST > 0 ? GR / REST : 1;
#pragma unroll
                        for (int i = 0; i < REST; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                        }
                        if (GR - PER * REST > 0) __builtin_amdgcn_sched_group_barrier(0x008, GR - PER * REST > 0 ? GR - PER * REST : 1, 0);
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x008, GR, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // my requests for stage it + 1 have landed (vmcnt(0)), my reads of stage it are done (lgkmcnt(0)): raw barrier.
                // (the fence above keeps the last K step's MFMAs in front of the wait: they cover the last reads' latency)
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                ++it;
            };
            for (int tile = t_beg; tile < t_end; ++tile) {
                stage_body(std::true_type{}, tile, 0);
                for (int kt = 1; kt < p.nkt; ++kt) stage_body(std::false_type{}, tile, kt);
                multiply((NS - 1) & 1);                                  // the tile's last K step, then its candidates
                pair_tile_epilogue<MT, KPL, NTW>(acc, lk, li, lim, qmul, p.invs, p.n_rows, tile * T_ + wm * 32 * MT + 4 * h);
            }
        }
    }

    {
        const size_t l0 = (size_t)qt * T_ * p.nseg + seg;            // list of the tile's first query in this segment
        pair_block_merge<T_, KPL, NTW>(smem, lk, li, wn, wm, h, l31, tid, p.part_key + l0 * SIM_KP, p.part_idx + l0 * SIM_KP,
                                       p.part_bound + l0, (size_t)p.nseg);
    }
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
template <typename QS, int NPROD>
__global__ __launch_bounds__(256) void pair_prep_kernel(const QS *__restrict__ q, int64_t ldq, int nq, int dim, int kd,
                                                        char *__restrict__ q2, int64_t ldq2, float *__restrict__ qinvs,
                                                        const int64_t *__restrict__ row_limit, int n_rows,
                                                        int *__restrict__ lim, int *__restrict__ qt_maxlim, int tile, float *__restrict__ qpk) {
    const int row = blockIdx.x;
    __shared__ float s_max[4];
    __shared__ float s_ss[4];
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
    float ss = 0.0f;                                           // || s_q q ||^2 (|s_q q| < 2^15: no overflow)
    for (int c = 4 * threadIdx.x; c < kd; c += 1024) {
        unsigned hi[2], lo[2];
        float w[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) w[t] = (real && servable && c + t < dim) ? (float)q[(size_t)row * ldq + c + t] * sc : 0.0f;
        ss += (w[0] * w[0] + w[1] * w[1]) + (w[2] * w[2] + w[3] * w[3]);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const __half2 hh = __floats2half2_rn(w[2 * t], w[2 * t + 1]);
            const float2 f = __half22float2(hh);
            const __half2 ll = __floats2half2_rn(w[2 * t] - f.x, w[2 * t + 1] - f.y);
            hi[t] = *(const unsigned *)&hh; lo[t] = *(const unsigned *)&ll;
        }
        if (NPROD == 1) {
            *(uint2 *)(d2 + c * 2) = make_uint2(hi[0], hi[1]);           // fp16 row, kd rounded up to whole 64-channel stages
        } else {
            char *blk = d2 + (c >> 5) * 128 + (c & 31) * 2;
            *(uint2 *)blk = make_uint2(hi[0], hi[1]);
            *(uint2 *)(blk + 64) = make_uint2(lo[0], lo[1]);
        }
    }
    if (qpk) {                                                 // the persistent stage's packed lists (sim_topk_pair_dev.h): integer key scale
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
        if (lane == 0) s_ss[wave] = ss;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float nrm = sqrtf((s_ss[0] + s_ss[1]) + (s_ss[2] + s_ss[3]));
            int es = 0;
            if (nrm > 0.0f) (void)frexpf(nrm * 1.011f, &es);                       // 2^es > 1.011 ||s_q q|| >= |key| (1 + error bound)
            qpk[row] = real ? (servable ? -ldexpf(1.0f, 17 - es) : NAN) : 0.0f;
            qpk[gridDim.x + row] = servable ? ldexpf(1.0f, es - 17) / sc : INFINITY; // NaN keys rank first and unpack as +inf, as they always did
        }
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

// rigorous bound on |candidate key - exact key| / ||q|| (header comment); kd = padded dimension (the K extent summed over)
double pair_err_bound(int kd, int nprod) {
    const double u24 = 5.9604644775390625e-08, u23 = 1.1920928955078125e-07;
    const double keyr = 3.0 * u24;                             // invs rounding + two multiplications
    const double floor_ = sqrt((double)kd) * 3.637978807091713e-12;            // sqrt(kd) 2^-38: values below fp16's normal range
    if (nprod == 1) {
        // each operand value w = s v -> hi = fp16(w): |w - hi| <= 2^-11 |w| (+ 2^-25 below the normal range: floor_);
        // |sum (qh bh - q b)| <= (2 2^-11 + 2^-22) ||q|| ||b||  (Cauchy-Schwarz); float64 query -> float32 first: 2^-24
        const double rep = 2.0 / 2048.0 + 1.0 / 4194304.0 + u24;
        const double accum = 1.002 * ((double)kd + 64.0) * u23;                // kd exact products summed in fp32, unit 2^-23
        return 1.0625 * (rep + accum + keyr + floor_);
    }
    const double u16 = 1.0 / 4194304.0;                        // 2^-22: value -> hi + lo
    const double rep = 2.0 * (u16 + u24);                      // both operands (+ the query's float64 -> float32 rounding)
    const double dropped = u16 * (1.0 + 1.0 / 1024.0);
    const double accum = 1.002 * (3.0 * kd + 64.0) * u23;      // unit 2^-23: rounding or truncation
    return 1.0625 * (rep + dropped + accum + keyr + floor_);
}

template <int T_, int MT, int KPL, int NPROD, int NTW = 2>
static int launch_pair(const PairArgs &a, int dbg, hipStream_t st) {
    constexpr int lds = 2 * 2 * T_ * PK_ROWB;
    static DeviceOnce once;
    int once_dev;
    if (once.todo(&once_dev)) {
        HIP_TRY(hipFuncSetAttribute((const void *)sim_topk_pair_kernel<T_, MT, KPL, NPROD, 0, NTW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#ifdef CSLAM_ABLATIONS
        HIP_TRY(hipFuncSetAttribute((const void *)sim_topk_pair_kernel<T_, MT, KPL, NPROD, 1, NTW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void *)sim_topk_pair_kernel<T_, MT, KPL, NPROD, 2, NTW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#endif
        once.done(once_dev);
    }
    const dim3 grid(a.nqt * a.nseg), blk(T_ * 64 / (16 * NTW));
#ifdef CSLAM_ABLATIONS
    if (dbg == 1) { hipLaunchKernelGGL((sim_topk_pair_kernel<T_, MT, KPL, NPROD, 1, NTW>), grid, blk, lds, st, a); HIP_TRY(hipGetLastError()); return CSLAM_OK; }
    if (dbg == 2) { hipLaunchKernelGGL((sim_topk_pair_kernel<T_, MT, KPL, NPROD, 2, NTW>), grid, blk, lds, st, a); HIP_TRY(hipGetLastError()); return CSLAM_OK; }
#endif
    (void)dbg;
    hipLaunchKernelGGL((sim_topk_pair_kernel<T_, MT, KPL, NPROD, 0, NTW>), grid, blk, lds, st, a);
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

int pair_stage1_launch(const PairArgs &a, int tile, int nprod, int dbg, hipStream_t st) {
    if (nprod == 1) {
        if (tile != 256) return launch_pair<128, 2, 16, 1>(a, dbg, st);
        return launch_pair<256, 4, 8, 1>(a, dbg, st);
    }
    return tile == 256 ? launch_pair<256, 4, 8, 3>(a, dbg, st) : launch_pair<128, 2, 16, 3>(a, dbg, st);
}

int pair_prep_launch(const void *d_q, int q_dtype, int64_t ldq, int nq, int dim, int kd, int nprod, char *q2, int64_t ldq2,
                     float *qinvs, const int64_t *d_row_limit, int n_rows, int *lim, int *qtm, int nq_pad, int tile, float *qpk, hipStream_t st) {
#define PREP(QS, NP) hipLaunchKernelGGL((pair_prep_kernel<QS, NP>), dim3(nq_pad), dim3(256), 0, st, (const QS *)d_q, ldq, nq, dim, kd, \
                                        q2, ldq2, qinvs, d_row_limit, n_rows, lim, qtm, tile, qpk)
    if (q_dtype == CSLAM_F32) { if (nprod == 1) PREP(float, 1); else PREP(float, 3); }
    else { if (nprod == 1) PREP(double, 1); else PREP(double, 3); }
#undef PREP
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}
