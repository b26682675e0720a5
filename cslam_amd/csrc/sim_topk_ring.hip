// sim_topk_ring.hip -- the one-product candidate stage (sim_topk_pair.hip, NPROD = 1) as a PERSISTENT kernel on a static,
// XCD-aligned schedule (round 5).
//
// Replaces, for a batch of queries, the scoring loop of cslam/nns_matching.py:55-61 exactly as sim_topk_pair_kernel does: the same
// 256 x 256 tile, the same operands (fp16 hi halves of the scaled rows, 128-byte K stages, XOR-swizzled LDS image filled by LDS-DMA),
// the same lane-local candidate lists, tile epilogue, block merge and error bound (pair_err_bound(kd, 1)).  What changes is WHO
// computes WHICH tile WHEN.
//
// Why.  sim_topk_pair_kernel runs one workgroup per (query tile, bank segment) work item and lets the dispatcher hand the items out as
// CUs free up.  On the 100k x 100k launch of BASELINE config 3 that moved 331 GB through the fabric (L2 misses; 1.64 GB algorithmic)
// at an L2 hit rate of 0.51 (profiles/r04_v78m_pmc_match_summary.json): a 256 x 256 x 4096 tile reads 4 MB, a whole XCD's L2 is 4 MB,
// so two workgroups share a fetch only while they are within a few K stages of each other -- and work items that start whenever a CU
// happens to free up are not.  At 4 TB/s of fabric traffic the kernel ran at 1.78 GHz with the matrix pipe 0.55 busy.
//
// Schedule (ring_schedule_build, host).  The launch is `num_cu` workgroups, one per CU (160 KB of LDS each), alive for the whole
// launch.  Workgroup b sits on XCD b % 8 (observed dispatch order; only speed depends on it) as slot b / 8; the 32 slots of an XCD
// form a patch of Sq = 4 query tiles x Sb = 8 bank tiles.  A patch keeps its 4 query tiles and walks the bank 8 tiles per step:
// per step the XCD's L2 takes 12 operand streams for 32 tiles of work (3/16 of the unshared traffic), every workgroup of the patch
// at the same K position because they all started the step together (tile-start rendezvous below).  The 8 XCDs work on 8
// DIFFERENT query groups against the SAME bank tiles at the same time, so the bank comes out of HBM once per 32 query tiles (the
// other seven XCDs hit the Infinity Cache) and a group's 8 MB of queries stay in the Infinity Cache for the whole walk.
// Query groups that do not fill a round of 8 XCDs (the tail) have their walk cut into runs, one run per XCD, so that every XCD
// gets the same number of steps (+- 1).  A (query tile, run, patch column) = one merged candidate list = one "segment" for
// stage 2 (rescore_kernel reads a per-query-tile segment count).
//
// Flow control (FLOW).  The workgroups of a patch start a walk within 0.3 us of each other and are 20-40 us apart one tile later
// (identical work; per-tile times scatter by +-5 %, profiles/r05_v4_patch_spread_trace.log), while the 4 MB of L2 hold about ten K
// stages of the patch's twelve streams: the sharing the schedule is built for lasts one tile.  (A tile-start rendezvous with a 6 us
// bound timed out on every tile and was removed.)  Instead every workgroup publishes its progress (run << 20 | K stage) every second
// stage in the patch's 128-byte progress line, one wave re-reads the line every fourth stage -- the load rides on the stage's own
// vmcnt(0), nothing waits for it -- and a workgroup more than `flow_w` stages ahead of the slowest one of its patch pauses
// (bounded; a timed-out pause switches the mechanism off for the task: the counters are a hint, results never depend on them).
#include <stdlib.h>
#include <type_traits>
#include "sim_topk_pair_dev.h"

// KPL = per-lane candidate list length.  DBG (measurement build, timing only; a bit mask): 1 = no global loads after the first stage
// (every stage re-reads the first one), 2 = every request reads bank tile 0 / query tile 0 (L2 hits), 4 = no stage barrier (with 1),
// 8 = no candidate update.  FLOW = patch flow control.  PRIO = progress-ordered wave priority: a stage's four groups of 8 MFMAs run at
// s_setprio 3, 2, 1, 0.  The arbiter takes priority first, age second: at equal priority the older wave of a SIMD issues its WHOLE
// stage first and parks at the barrier ~1000 cycles before its partner, which then runs alone with nobody to fill its issue gaps
// (2420 cycles per 2048-cycle stage, profiles/r05_v8_barrier_trace.log); with the wave that is BEHIND always at the higher
// priority the pair alternates group by group and the solo tail is one group.
template <int KPL, int DBG, int FLOW, int PRIO>
__global__ __launch_bounds__(512, 2) void sim_topk_ring_kernel(RingArgs p) {
    constexpr int T_ = 256, MT = 4, NTW = 2, NWN = 4, NTHR = 512;
    constexpr int OPB = T_ * PK_ROWB;            // bytes of one operand tile (64 channels of 256 rows) in LDS
    constexpr int STAGE = 2 * OPB;
    constexpr int NLD = T_ * 8 / NTHR;           // 16-byte chunks per thread per operand (= 4)
    constexpr int NS = 4;                        // 16-channel K steps per stage
    constexpr int G = MT * NTW;                  // MFMAs of a K step
    constexpr int NRD = MT + NTW;                // fragment reads of a K step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int h = lane >> 5, l31 = lane & 31;
    const int bid = blockIdx.x;
    const int wg = (bid % p.n_xcd) * p.wpx + bid / p.n_xcd;
    // 1 / (||row|| s_row) of the current bank tile's 256 rows; two buffers by tile parity: a wave that leaves a tile's epilogue early
    // requests the next tile's values while slower waves still read this tile's (a buffer is rewritten 64 stage barriers later)
    constexpr int SLOT = OPB / 2 + OPB;          // a short tile's stage: 128 bank rows + the 256 queries (below); three of them fit
    float *const s_inv = (float *)(smem + 3 * SLOT + 16);
    int *const prog = (FLOW && p.prog) ? p.prog + (bid % p.n_xcd) * 32 : nullptr;      // the patch's progress line
    const int slot = bid / p.n_xcd;

    // ---- loader: chunk pch = i*NTHR + tid -> tile row pch >> 3, physical 16-byte slot pch & 7 holding logical chunk
    // slot ^ ((row >> 1) & 7) of the row's 128-byte block
    // (the bank's row offsets stay in the per-lane offset -- one add per request -- because the buffer's range check, which zeroes
    // the rows beyond a task's end, sees voffset only; the query copy is padded to whole tiles: its 64-row steps ride in soffset)
    int voffA0, voffB0;
    {
        const int r = tid >> 3, slot = tid & 7;
        const int c = slot ^ ((r >> 1) & 7);             // the same for every part i: r advances by 64
        voffA0 = r * (int)p.ldb2 + (c << 4);
        voffB0 = r * (int)p.ldq2 + (c << 4);
    }
    const int stepA = (NTHR / 8) * (int)p.ldb2;
    const int stepB = (NTHR / 8) * (int)p.ldq2;
    const int wave_chunk = wave * 1024;
    const int swz = (lane >> 1) & 7;
    int foff[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) foff[s] = ((2 * s + h) ^ swz) << 4;
    const int arow0 = (wm * 32 * MT + l31) * PK_ROWB;
    const int brow0 = (wn * (32 * NTW) + l31) * PK_ROWB;

    if (p.xcc_out && tid == 0) {
        int xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        p.xcc_out[bid] = xcc;
    }
    if (DBG == 33 && p.trace_out && bid == 0 && lane == 0) {
        int hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        p.trace_out[(size_t)(p.n_xcd * p.wpx) * 64 + 8 * 48 * 2 + wave] = hw;
    }
    const int task_end = p.task_off[wg + 1];
    for (int ti = p.task_off[wg]; ti < task_end; ++ti) {
        const RingTask tk = p.tasks[ti];
        const int qt = tk.qt;
        int ntiles = tk.n_tiles;
        {
            int ml = p.qt_maxlim[qt];                                       // tiles that start at or beyond it hold no visible row
            if (ml > tk.row_end) ml = tk.row_end;
            const int vis = ml > tk.row0 ? (ml - tk.row0 + tk.stride_rows - 1) / tk.stride_rows : 0;
            if (ntiles > vis) ntiles = vis;
        }
        int lp[NTW][KPL];                                                   // packed candidate lists (sim_topk_pair_dev.h), ascending
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int j = 0; j < KPL; ++j) lp[n][j] = RING_EMPTY;
        int lim[NTW];
        float qmul[NTW];
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            lim[n] = p.lim[qt * T_ + wn * (32 * NTW) + n * 32 + l31];
            lim[n] = lim[n] < tk.row_end ? lim[n] : tk.row_end;             // rows from row_end on belong to another task
            qmul[n] = p.qinvs[qt * T_ + wn * (32 * NTW) + n * 32 + l31];
        }
        // measurement build: flow_bias_q / flow_bias_b hold patch slot (qi, bi) that many stages further back (sharers spaced in time)
        const int pbase = (tk.run << 20) + p.flow_bias_q * (slot / p.sb) + p.flow_bias_b * (slot % p.sb);
        bool flow_on = prog != nullptr;                  // used by wave 0 only
        int pv = 0x7fffffff;                             // wave 0: progress of patch slot `lane`, as last read

        if (ntiles > 0) {
            const __amdgpu_buffer_rsrc_t rsB = pk_rsrc(p.q2 + (int64_t)((DBG & 2) ? 0 : qt) * T_ * p.ldq2, (int64_t)T_ * p.ldq2);
            __amdgpu_buffer_rsrc_t rsA;
            auto point_at_tile = [&](int r0) {               // the 256 bank rows from r0 on; rows at or beyond the task's end read as zero
                const int t0 = (DBG & 2) ? 0 : r0;
                int64_t rows = (int64_t)tk.row_end - (int64_t)r0;
                if (rows > T_) rows = T_;
                rsA = pk_rsrc(p.bank2 + (int64_t)t0 * p.ldb2, rows * p.ldb2);
            };
            auto stage_load_part = [&](int stage, int kt, int i) {
                char *sA = smem + stage * STAGE;
                char *sB = sA + OPB;
                // DBG bits 6-8 / 9-11 (measurement build): cache policy of the bank / query requests (1 = sc0, 2 = nt, 4 = sc1)
                constexpr int PA = (DBG >> 6) & 7, PB = (DBG >> 9) & 7;
                constexpr int AUXA = (PA & 3) | ((PA & 4) << 2), AUXB = (PB & 3) | ((PB & 4) << 2);
                pk_blds16_aux<AUXA>(rsA, voffA0 + i * stepA, kt * PK_ROWB, sA + i * (NTHR * 16) + wave_chunk);
                pk_blds16_aux<AUXB>(rsB, voffB0, kt * PK_ROWB + i * stepB, sB + i * (NTHR * 16) + wave_chunk);
            };

            f32x16 acc[MT][NTW];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NTW; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
            // two fragment register sets: K step s + 1 is read while K step s is multiplied, a stage's last K step is multiplied
            // behind the stage barrier (see sim_topk_pair_kernel for the derivation of this order)
            f16x8 fa[2][MT], fb[2][NTW];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[u][m] = (f16x8)(_Float16)0.0f;
#pragma unroll
                for (int n = 0; n < NTW; ++n) fb[u][n] = (f16x8)(_Float16)0.0f;
            }
            constexpr bool NOLOAD = (DBG & 1) != 0;
            constexpr bool NOBAR = (DBG & 4) != 0;
            int it = 0;
            auto read_frags = [&](int u, const char *sA, const char *sB, int s) {
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[u][m] = *(const f16x8 *)(sA + arow0 + m * 32 * PK_ROWB + foff[s]);
#pragma unroll
                for (int n = 0; n < NTW; ++n) fb[u][n] = *(const f16x8 *)(sB + brow0 + n * 32 * PK_ROWB + foff[s]);
            };
            // FULL: every 32-row block of the wave's 128 rows holds rows of the task; otherwise (the partial tile that ends a
            // contiguous walk) blocks m >= m_act are skipped -- their accumulators stay zero, their rows are masked by `lim`
            int m_act = MT;
            auto multiply = [&](int u, auto full_tag) {
                constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    if (FULL || m < m_act) {
#pragma unroll
                        for (int n = 0; n < NTW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[u][m], fb[u][n], acc[m][n], 0, 0, 0);
                    }
            };

            const int total = ntiles * p.nkt;
            point_at_tile(tk.row0);
#pragma unroll
            for (int i = 0; i < NLD; ++i) stage_load_part(0, 0, i);
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_s_barrier();

            int tpar = 0;                                            // parity of the tile's ordinal in the task
            auto stage_body = [&](auto first_tag, auto full_tag, int tile, int kt) {       // tile = first bank row of the tile
                constexpr bool FIRST = decltype(first_tag)::value;
                constexpr bool FULL = decltype(full_tag)::value;
                int nkt_ = kt + 1, ntile = tile;
                if (nkt_ == p.nkt) { nkt_ = 0; ntile = tile + tk.stride_rows; }
                // branch-free prefetch: the very last stage re-fetches its own block into the idle buffer, which nobody reads
                const bool more = it + 1 < total;
                const int lkt = more ? nkt_ : kt;
                if (more && nkt_ == 0) point_at_tile(ntile);             // wave-uniform, before anything is in flight
                const char *sA = smem + (NOLOAD ? 0 : (it & 1)) * STAGE;
                const char *sB = sA + OPB;
                constexpr int HALF = FIRST ? 0 : NLD / 2;                // loader parts (2 requests each) issued behind the barrier
                constexpr int LEAD = 2;
                if (FLOW == 3) {
                    // proportional form: no polling.  Every stage: publish; reduce the line read ONE stage ago (DPP row shifts + two
                    // readlanes: no LDS traffic); sleep ~a third of a stage per stage of lead beyond the window; request the line again.
                    if (wave == 0 && prog != nullptr) {
                        const int P = pbase + it;
                        if (lane == 0) __hip_atomic_store(prog + slot, P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (it > 0) {
                            int v = pv;
                            int o;
                            o = __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x111, 0xf, 0xf, false); v = o < v ? o : v;     // row_shr:1
                            o = __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x112, 0xf, 0xf, false); v = o < v ? o : v;     // row_shr:2
                            o = __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x114, 0xf, 0xf, false); v = o < v ? o : v;     // row_shr:4
                            o = __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x118, 0xf, 0xf, false); v = o < v ? o : v;     // row_shr:8
                            const int m0 = __builtin_amdgcn_readlane(v, 15), m1 = __builtin_amdgcn_readlane(v, 31);
                            const int lead = P - (m0 < m1 ? m0 : m1) - p.flow_w;
                            if (lead > 0) {
                                if (lead == 1) __builtin_amdgcn_s_sleep(10);
                                else if (lead == 2) __builtin_amdgcn_s_sleep(22);
                                else if (lead <= 4) __builtin_amdgcn_s_sleep(40);
                                else __builtin_amdgcn_s_sleep(80);
                                if (p.trace_out && lane == 0) p.trace_out[(size_t)bid * 64 + 62] += 1;
                            }
                        }
                        pv = lane < p.wpx ? __hip_atomic_load(prog + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffff;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                } else if (FLOW) {
                    if (wave == 0 && flow_on) {
                        const int P = pbase + it;
                        // FLOW 1: publish every second stage, look every fourth; FLOW 2: both every stage (a tighter loop)
                        if ((FLOW == 2 || (it & 1) == 0) && lane == 0) __hip_atomic_store(prog + slot, P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (FLOW == 2 ? it > 0 : (it & 3) == 3) {
                            // the line read two stages ago (its load was waited for by the stage barriers since): slowest of the patch
                            auto slowest = [&](int v) {
#pragma unroll
                                for (int off = 32; off >= 1; off >>= 1) { const int o = __shfl_xor(v, off, 64); v = o < v ? o : v; }
                                return __builtin_amdgcn_readfirstlane(v);
                            };
                            int mn = slowest(pv);
                            int spins = 0;
                            while (P - mn > p.flow_w && spins < (FLOW == 2 ? 160 : 48)) {     // ahead of the window: pause, ~1 (0.3) us per look
                                __builtin_amdgcn_s_sleep(FLOW == 2 ? 6 : 24);
                                pv = lane < p.wpx ? __hip_atomic_load(prog + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffff;
                                mn = slowest(pv);
                                ++spins;
                            }
                            if (spins >= (FLOW == 2 ? 160 : 48)) flow_on = false;
                            if (p.trace_out && spins > 0 && lane == 0) p.trace_out[(size_t)bid * 64 + 62] += spins;
                        }
                        if (FLOW == 2 || (it & 3) == 1)
                            pv = lane < p.wpx ? __hip_atomic_load(prog + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffff;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // ---- behind the barrier: first reads of this stage, requests of the next, the last K step of the previous
                if (PRIO) { __builtin_amdgcn_s_setprio(3); __builtin_amdgcn_sched_barrier(0); }
                if constexpr (FIRST) {
                    // the tile's 256 values of invs, 4 bytes per lane of waves 0..3, by LDS-DMA: they land with this stage's requests
                    // (nobody reads them before the tile's epilogue, 63 stage barriers from here; the previous tile's were read in
                    // front of this stage); rows beyond the bank read as zero and are masked by the row limits
                    if (wave < 4) {
                        const __amdgpu_buffer_rsrc_t rsI = pk_rsrc((const char *)p.invs, (int64_t)p.n_rows * 4);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsI, (__attribute__((address_space(3))) void *)((char *)s_inv + (tpar * 4 + wave) * 256), 4,
                                                                 (tile + tid) * 4, 0, 0, 0);
                    }
                }
                read_frags(0, sA, sB, 0);
                if constexpr (!FIRST) {
                    if (!NOLOAD) {
#pragma unroll
                        for (int i = 0; i < HALF; ++i) stage_load_part((it + 1) & 1, lkt, i);
                    }
                    multiply((NS - 1) & 1, full_tag);
                    if constexpr (FULL) {
                        __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);          // DS read (nothing older is needed here)
                        constexpr int PER = G / (2 * HALF) > 0 ? G / (2 * HALF) : 1;
#pragma unroll
                        for (int i = 0; i < 2 * HALF; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);      // MFMA
                            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);        // VMEM read (LDS-DMA)
                        }
                        if (G - PER * 2 * HALF > 0) __builtin_amdgcn_sched_group_barrier(0x008, G - PER * 2 * HALF > 0 ? G - PER * 2 * HALF : 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- K steps 0 .. NS - 2 of this stage, each over the reads of the next
#pragma unroll
                for (int s = 0; s + 1 < NS; ++s) {
                    if (PRIO) {
                        if (s == 0) __builtin_amdgcn_s_setprio(2); else if (s == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    read_frags((s + 1) & 1, sA, sB, s + 1);
                    if (s == 0 && !NOLOAD) {
#pragma unroll
                        for (int i = HALF; i < NLD; ++i) stage_load_part((it + 1) & 1, lkt, i);
                    }
                    multiply(s & 1, full_tag);
                    if constexpr (FULL) {
                        __builtin_amdgcn_sched_group_barrier(0x008, LEAD, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
                        constexpr int REST = 2 * (NLD - HALF);           // requests placed in K step 0
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
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (!NOBAR) {
                    // DBG 1 with a trace buffer: per-wave shader-clock stamps around the barrier of the first 48 stages (workgroup 0)
                    const bool stamp = DBG == 33 && p.trace_out && bid == 0 && it < 48;
                    uint64_t ta = 0;
                    if (stamp) ta = __builtin_amdgcn_s_memtime();
                    __builtin_amdgcn_s_waitcnt(0);
                    __builtin_amdgcn_s_barrier();
                    if (stamp) {
                        const uint64_t tb = __builtin_amdgcn_s_memtime();
                        if (lane == 0) {
                            long long *o = p.trace_out + (size_t)(p.n_xcd * p.wpx) * 64 + ((size_t)wave * 48 + it) * 2;
                            o[0] = (long long)ta; o[1] = (long long)tb;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                ++it;
            };
            // ---- a SHORT last tile (at most 128 rows of the task left: the wm = 1 waves have nothing to multiply, the others little).
            // Its 64 K stages are a chain of L2 round trips with two stage buffers -- one stage in flight while the other is read, and
            // reading takes no time: 85 us for the 32 rows that end every walk of the in-step launch, against 114 us for 256 rows
            // (profiles/r06_y_in_step_tile_trace.log).  With half the bank rows a stage is 48 KB and THREE fit: two stages in flight.
            // (The stage the previous tile's last stage prefetched in the two-buffer layout is not used: one stage of 64.)
            auto short_tile = [&](int tile, int valid) {
                m_act = wm == 0 ? (valid + 31) >> 5 : 0;
                point_at_tile(tile);
                auto load = [&](int slot, int kt) {
                    char *sA = smem + slot * SLOT, *sB = sA + OPB / 2;
#pragma unroll
                    for (int i = 0; i < NLD / 2; ++i) pk_blds16(rsA, voffA0 + i * stepA, kt * PK_ROWB, sA + i * (NTHR * 16) + wave_chunk);
#pragma unroll
                    for (int i = 0; i < NLD; ++i) pk_blds16(rsB, voffB0, kt * PK_ROWB + i * stepB, sB + i * (NTHR * 16) + wave_chunk);
                };
                if (wave < 4) {
                    const __amdgpu_buffer_rsrc_t rsI = pk_rsrc((const char *)p.invs, (int64_t)p.n_rows * 4);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsI, (__attribute__((address_space(3))) void *)((char *)s_inv + (tpar * 4 + wave) * 256), 4,
                                                             (tile + tid) * 4, 0, 0, 0);
                }
                load(0, 0);
                if (p.nkt > 1) load(1, 1);
                for (int kt = 0; kt < p.nkt; ++kt) {
                    // stage kt has landed (the NLD / 2 + NLD requests of stage kt + 1 may still be in flight), everybody is done with stage kt - 1
                    if (kt + 1 < p.nkt) __builtin_amdgcn_s_waitcnt(0x0f70 | (NLD / 2 + NLD)); else __builtin_amdgcn_s_waitcnt(0x0f70);
                    __builtin_amdgcn_s_barrier();
                    if (kt + 2 < p.nkt) load((kt + 2) % 3, kt + 2);
                    if (m_act > 0) {
                        const char *sA = smem + (kt % 3) * SLOT, *sB = sA + OPB / 2;
#pragma unroll
                        for (int s = 0; s < NS; ++s) {
                            read_frags(s & 1, sA, sB, s);
                            multiply(s & 1, std::false_type{});
                        }
                    }
                }
                __builtin_amdgcn_s_barrier();                        // (the next task's first stage lands where these stages were read)
            };
            int tile = tk.row0;                                      // first bank row of the tile
            for (int i = 0; i < ntiles; ++i, tile += tk.stride_rows) {
                if (p.trace_out && tid == 0 && ti == p.task_off[wg] && i < 62) p.trace_out[(size_t)bid * 64 + i] = (long long)wall_clock64();
                const int valid = tk.row_end - tile;                 // rows of the task in this tile (>= 256: all of them)
                if (valid <= T_ / 2 && !FLOW && !(DBG & 7)) {
                    short_tile(tile, valid);
                } else if (valid >= T_) {
                    stage_body(std::true_type{}, std::true_type{}, tile, 0);
                    for (int kt = 1; kt < p.nkt; ++kt) stage_body(std::false_type{}, std::true_type{}, tile, kt);
                    multiply((NS - 1) & 1, std::true_type{});        // the tile's last K step, then its candidates
                } else {
                    int mine = valid - wm * 32 * MT;                 // ... of this wave's 128
                    mine = mine < 0 ? 0 : mine;
                    m_act = (mine + 31) >> 5;
                    stage_body(std::true_type{}, std::false_type{}, tile, 0);
                    for (int kt = 1; kt < p.nkt; ++kt) stage_body(std::false_type{}, std::false_type{}, tile, kt);
                    multiply((NS - 1) & 1, std::false_type{});
                }
                if (DBG & 8) {                                           // timing only: no candidate update
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NTW; ++n) {
                            const int v = (int)(acc[m][n][0] + acc[m][n][7]);
                            lp[n][0] = v < lp[n][0] ? v : lp[n][0];
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
                        }
                } else {
                    // (wave-uniform) every query of the wave sees every row of the wave's part of this tile: no per-key visibility test
                    const int wave_end = tile + (wm + 1) * 32 * MT;
                    bool masked = false;
#pragma unroll
                    for (int n = 0; n < NTW; ++n) masked |= lim[n] < wave_end;
                    if (__builtin_amdgcn_ballot_w64(masked) != 0)
                        ring_tile_epilogue<MT, KPL, NTW, true>(acc, lp, lim, qmul, s_inv + tpar * 256 + wm * 32 * MT + 4 * h, tile + wm * 32 * MT + 4 * h, i << 6);
                    else
                        ring_tile_epilogue<MT, KPL, NTW, false>(acc, lp, lim, qmul, s_inv + tpar * 256 + wm * 32 * MT + 4 * h, tile + wm * 32 * MT + 4 * h, i << 6);
                }
                tpar ^= 1;
            }
        }
        // the run is done for this workgroup (or it had nothing to do in it): nobody waits for it any more
        if (prog && tid == 0) __hip_atomic_store(prog + slot, (tk.run + 1) << 20, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        {
            const int ns = p.qt_nseg[qt];
            const size_t l0 = (size_t)p.qt_segoff[qt] * T_ + tk.seg;         // list `seg` of the tile's first query
            ring_block_merge<T_, MT, KPL, NTW>(smem, lp, wn, wm, h, l31, tid, tk.row0, tk.stride_rows, p.qunit + qt * T_, p.part_key + l0 * SIM_KP,
                                               p.part_idx + l0 * SIM_KP, p.part_bound + l0, (size_t)ns);
        }
        __syncthreads();                                                 // the merge's LDS is the next task's first stage
    }
    if (prog && tid == 0) __hip_atomic_store(prog + slot, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p.trace_out && tid == 0) p.trace_out[(size_t)bid * 64 + 63] = (long long)wall_clock64();
}

// ---- the schedule ------------------------------------------------------------------------------------------------------------
void ring_schedule_build(RingSchedule &s, int nqt, int n_rows, int n_xcd, int wpx) {
    s.nqt = nqt; s.n_rows = n_rows; s.n_xcd = n_xcd; s.wpx = wpx;
    const int n_btiles = (n_rows + 255) / 256;
    int sq = nqt < 4 ? nqt : 4;
    if (sq > wpx) sq = wpx;
    if (sq < 1) sq = 1;
    int sb = wpx / sq;
    if (sb > n_btiles) sb = n_btiles > 0 ? n_btiles : 1;
    s.sq = sq; s.sb = sb;
    const int ngroups = (nqt + sq - 1) / sq;
    // a run = one query group against bank rows [ra, rb) on one XCD; patch column bi takes the bi-th of sb contiguous chunks of it.
    // Whole-bank runs of a round of XCDs are cut the same way on every XCD: the 8 XCDs read the same bank rows at the same time.
    struct Run { int g; int ra, rb; int r; };
    std::vector<std::vector<Run>> runs(n_xcd);
    std::vector<int> group_runs(ngroups, 0);
    // a task's candidate lists address its rows with 13 bits (sim_topk_pair_dev.h: 128 tiles): longer runs are cut into equal pieces
    const long long max_run = (long long)sb * (128 * 256 - 32);          // the chunk of a patch column is rounded up to 32 rows
    auto add_run = [&](int x, int g, long long ra, long long rb) {
        const long long len = rb - ra;
        const int pieces = len > max_run ? (int)((len + max_run - 1) / max_run) : 1;
        long long piece = ((len + pieces - 1) / pieces + 31) / 32 * 32;
        for (long long a = ra; a < rb; a += piece) runs[x].push_back(Run{g, (int)a, (int)(a + piece < rb ? a + piece : rb), group_runs[g]++});
    };
    const int full = ngroups / n_xcd;
    for (int i = 0; i < full; ++i)
        for (int x = 0; x < n_xcd; ++x) add_run(x, i * n_xcd + x, 0, n_rows);
    // the groups that do not fill a round of XCDs: their rows (group-major) are dealt out evenly, cut points on multiples of 32 rows
    const int gtail = ngroups - full * n_xcd;
    if (gtail > 0) {
        const long long W = (long long)gtail * n_rows;
        auto cut = [&](int x) {
            long long c = W * x / n_xcd;
            const long long g = c / n_rows, r = c % n_rows;
            long long r32 = (r + 31) / 32 * 32;
            if (r32 > n_rows) r32 = n_rows;
            return x >= n_xcd ? W : g * n_rows + r32;
        };
        for (int x = 0; x < n_xcd; ++x) {
            long long lo = cut(x);
            const long long hi = cut(x + 1);
            while (lo < hi) {
                const int g = full * n_xcd + (int)(lo / n_rows);
                long long end = (lo / n_rows + 1) * n_rows;
                if (end > hi) end = hi;
                add_run(x, g, lo % n_rows, lo % n_rows + (end - lo));
                lo = end;
            }
        }
    }
    s.qt_nseg.assign(nqt, 0);
    s.qt_segoff.assign(nqt, 0);
    int lists = 0;
    for (int qt = 0; qt < nqt; ++qt) {
        s.qt_nseg[qt] = group_runs[qt / sq] * sb;
        s.qt_segoff[qt] = lists;
        lists += s.qt_nseg[qt];
    }
    s.total_lists = lists;
    s.tasks.clear();
    s.task_off.assign((size_t)n_xcd * wpx + 1, 0);
    for (int x = 0; x < n_xcd; ++x)
        for (int sl = 0; sl < wpx; ++sl) {
            const int w = x * wpx + sl;
            s.task_off[w] = (int)s.tasks.size();
            if (sl >= sq * sb) continue;
            const int qi = sl / sb, bi = sl % sb;
            for (size_t k = 0; k < runs[x].size(); ++k) {
                const Run &r = runs[x][k];
                const int qt = r.g * sq + qi;
                if (qt >= nqt) continue;
                RingTask t;
                t.qt = qt;
                {
                    // the run's rows in sb contiguous chunks of equal length (a multiple of 32), the last one shorter
                    int c = ((r.rb - r.ra + sb - 1) / sb + 31) / 32 * 32;
                    int a0 = r.ra + bi * c, a1 = a0 + c;
                    if (a0 > r.rb) a0 = r.rb;
                    if (a1 > r.rb) a1 = r.rb;
                    t.row0 = a0; t.stride_rows = 256; t.row_end = a1;
                    t.n_tiles = (a1 - a0 + 255) / 256;
                }
                t.seg = r.r * sb + bi;
                t.run = (int)k;                  // the XCD's k-th run: progress of the patch is compared inside a run
                t.pad = 0;
                s.tasks.push_back(t);
            }
        }
    s.task_off[(size_t)n_xcd * wpx] = (int)s.tasks.size();
    size_t off = 0;
    auto place = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) / 256 * 256; return o; };
    s.off_tasks = place(s.tasks.size() * sizeof(RingTask));
    s.off_task_off = place(s.task_off.size() * 4);
    s.off_qt_nseg = place((size_t)nqt * 4);
    s.off_qt_segoff = place((size_t)nqt * 4);
    s.blob.assign(off, 0);
    memcpy(s.blob.data() + s.off_tasks, s.tasks.data(), s.tasks.size() * sizeof(RingTask));
    memcpy(s.blob.data() + s.off_task_off, s.task_off.data(), s.task_off.size() * 4);
    memcpy(s.blob.data() + s.off_qt_nseg, s.qt_nseg.data(), (size_t)nqt * 4);
    memcpy(s.blob.data() + s.off_qt_segoff, s.qt_segoff.data(), (size_t)nqt * 4);
}

// variant: bit 0 = patch flow control, bit 1 = progress-ordered wave priority
int ring_stage1_launch(const RingArgs &a, int variant, int dbg, hipStream_t st) {
    constexpr int lds = 3 * (128 + 256) * PK_ROWB + 16 + 2 * 1024;   // two stages of 256 + 256 rows or three of 128 + 256 (short tiles), pad, invs (two parities)
    static DeviceOnce once;
    int once_dev;
#define RING_EACH(X) X(0, 0, 0)
#ifdef CSLAM_ABLATIONS
    // variants (flow control, wave priorities), timing-only ablations, cache policies of the requests: bank nt / query nt / both / bank sc1 / ...
#define RING_EACH_DBG(X) X(0, 1, 0) X(0, 0, 1) X(0, 1, 1) X(0, 2, 0) X(0, 3, 0) X(2, 0, 0) X(8, 0, 0) X(9, 0, 0) X(10, 0, 0) X(13, 0, 0) X(33, 0, 0) X(33, 0, 1) X(9, 0, 1) X(10, 0, 1) \
    X(128, 0, 0) X(1024, 0, 0) X(1152, 0, 0) X(256, 0, 0) X(2048, 0, 0) X(64, 0, 0) X(512, 0, 0)
#else
#define RING_EACH_DBG(X)
#endif
    if (once.todo(&once_dev)) {
#define RING_ATTR(D, S, P) HIP_TRY(hipFuncSetAttribute((const void *)sim_topk_ring_kernel<8, D, S, P>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        RING_EACH(RING_ATTR)
        RING_EACH_DBG(RING_ATTR)
#undef RING_ATTR
        once.done(once_dev);
    }
    const dim3 grid(a.n_xcd * a.wpx), blk(512);
    // bits 2-3: 4 = the per-stage polling flow control, 8 = the proportional one
    const int sy = (variant & 8) ? 3 : ((variant & 4) ? 2 : (variant & 1)), pr = (variant >> 1) & 1;
    bool done = false;
#define RING_GO(D, S, P) if (!done && dbg == D && sy == S && pr == P) { hipLaunchKernelGGL((sim_topk_ring_kernel<8, D, S, P>), grid, blk, lds, st, a); done = true; }
    RING_EACH(RING_GO)
    RING_EACH_DBG(RING_GO)
#undef RING_GO
    if (!done) { cslam_set_error("ring stage: no such variant (dbg %d needs the measurement build)", dbg); return CSLAM_E_INVALID; }
    HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

/* diagnostics (include/cslam_hip_experimental.h): the static schedule the persistent candidate stage would use -- pure host code,
 * tests/test_abi_cpu.py checks on CPU that every (query tile, bank row) pair is computed exactly once and the XCDs are balanced. */
CSLAM_API int cslam_ring_schedule_describe(int nqt, int n_rows, int n_xcd, int wpx, int32_t info[6], int32_t *tasks, int64_t tasks_cap,
                                           int32_t *task_off, int32_t *qt_nseg, int32_t *qt_segoff) {
    ARG_CHECK(nqt >= 1 && n_rows >= 1 && n_xcd >= 1 && wpx >= 1 && info, "nqt, n_rows, n_xcd, wpx >= 1 and info are required");
    RingSchedule s;
    ring_schedule_build(s, nqt, n_rows, n_xcd, wpx);
    info[0] = s.sq; info[1] = s.sb; info[2] = (int32_t)s.tasks.size(); info[3] = s.total_lists; info[4] = 0; info[5] = (int32_t)sizeof(RingTask) / 4;
    if (tasks) {
        ARG_CHECK(tasks_cap >= (int64_t)s.tasks.size() * 8, "tasks_cap too small: 8 int32 per task");
        memcpy(tasks, s.tasks.data(), s.tasks.size() * sizeof(RingTask));
    }
    if (task_off) memcpy(task_off, s.task_off.data(), s.task_off.size() * 4);
    if (qt_nseg) memcpy(qt_nseg, s.qt_nseg.data(), (size_t)nqt * 4);
    if (qt_segoff) memcpy(qt_segoff, s.qt_segoff.data(), (size_t)nqt * 4);
    return CSLAM_OK;
}
