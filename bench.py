#!/usr/bin/env python
"""bench.py -- keyframes/sec (extract + match) on a 100k x 4096-D bank (BASELINE.json metric).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step (BASELINE config 3, per rank = per robot):
    B synthetic 640x480 RGB keyframes already in HBM (uint8, seeded on device)
      -> NetVLAD extract: crop/bicubic-resize/normalise [HIP] -> VGG-16 conv5_3 [every layer a HIP kernel of this library, fp32-grade]
         -> VLAD aggregation [HIP] -> PCA 32768->4096 + L2 [HIP, fp16-pair GEMM]
         (the step's chunks of --extract-chunk frames alternate over --extract-lanes HIP streams: NetVLAD.compute_embeddings_batch_device)
      -> (N > 1) RCCL all-gather of the new descriptors: every rank sees every query
      -> top-5 against the resident 100k x 4096 bank [HIP: sim_topk_pair (fp16-pair candidate stage) + float64 re-score + certificate].
         N = 1: the search is ENQUEUED behind the extraction (cslam_bank_search_enqueue_dev) and finished after the next step has
         been enqueued -- no host synchronisation inside the timed region.
         N > 1, --shard-mode rows (default): the bank of the metric (ONE seeded bank at every N) is split by rows over the ranks, every
         rank scores all N*B new descriptors against its 100k/N rows, one all-to-all returns the lists to the keyframes' owners and
         cslam_topk_merge_dev picks the exact whole-bank top-5; `sharded_check` compares one step with a single-GPU search
         of the whole bank.  --shard-mode robots: BASELINE config 4's shape, one full bank per rank (= per robot), top-5 for its own
         keyframes and best-1 for the other robots' keyframes
value = keyframes processed by all ranks / max-over-ranks time of exactly K steps.
The JSON line also carries: match_only / extract_only throughputs (the two legs timed separately), `roofline` (the candidate-stage
kernel's launches INSIDE the timed steps, HIP events on the launch stream), `roofline_c3_batch` (the separate 100k-query launch),
`roofline_step_largest` (the trunk's pair products, from per-launch events of one-lane steps run after the timed region -- with two
lanes a launch shares the chip with the other lane's kernels), `roofline_extract` (the other
kernels of the extract leg on their real shapes) and `cpu_baseline` (the oracle restatements on a bounded sample).
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: dense f32-input MFMA peak
FP16_MFMA_PEAK_TFLOPS = 2500.0     # same guide: dense fp16/bf16 MFMA peak (never the 2:1-sparsity figure)
HBM_PEAK_GBS = 8000.0              # same guide: HBM3E ~8 TB/s
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_by_kernel.json")


def vgg16_flop_per_frame(hw=224):
    """(fp16 MFMA flop ISSUED, direct-convolution-equivalent flop) of the VGG-16 trunk NetVLAD runs (conv1_1 .. conv5_3, cslam/vpr/
    netvlad.py:163-171) per frame of hw x hw, in the forms DESIGN.md section 3.6 lists: conv1_1 .. conv2_2 direct on fp16 pairs
    (3 MFMA products per multiply-add; conv1_1's K = 27 padded to 32 and computed on the 10 x 18 patch of every 8 x 16 block),
    conv3_1 .. conv5_3 F(4x4, 3x3) Winograd (36 products per 4 x 4-pixel tile instead of 144, 3 MFMA products each)."""
    layers = [(3, 64), (64, 64), "pool", (64, 128), (128, 128), "pool", (128, 256), (256, 256), (256, 256), "pool",
              (256, 512), (512, 512), (512, 512), "pool", (512, 512), (512, 512), (512, 512)]
    issued = direct = 0.0
    n = 0
    for l in layers:
        if l == "pool":
            hw //= 2
            continue
        cin, cout = l
        n += 1
        direct += 2.0 * hw * hw * 9 * cin * cout
        if cin == 3:
            issued += (180.0 / 128.0) * 3 * 2.0 * hw * hw * 32 * cout
        elif n <= 4:
            issued += 3 * 2.0 * hw * hw * 9 * cin * cout
        else:
            t = (-(-hw // 4)) ** 2
            issued += 3 * 2.0 * 36 * t * cin * cout
    return issued, direct


def pmc_entry(kernel, **match):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/pmc_by_kernel.json,
    keyed by kernel name -- no globbing), only if the entry was collected on exactly the launch timed here."""
    try:
        table = json.load(open(PMC_FILE))
        e = table.get(kernel)
        if "queries" in match and (not e or e.get("match", {}).get("queries") != match["queries"]):
            e = table.get("%s/q%d" % (kernel, match["queries"]))      # one entry per profiled launch shape of the matcher
    except Exception:
        return None
    if not e or any(e.get("match", {}).get(k) != v for k, v in match.items()):
        return None
    return e


class BoardSensors:
    """Board power and shader clock of the bench's GPU over the timed steps, from the amdgpu hwmon files of its card (power1_average or
    power1_input in microwatts, power1_cap, freq1_input = sclk in Hz), sampled back to back by a thread of rank 0 (a read of the power file takes tens of milliseconds).  A reported context
    figure, not part of the metric: the fp16-pair kernels of the trunk run the board at its power cap, and the clock the chip then
    sustains -- not the nominal 2.4 GHz the roofline peaks are quoted at -- is what their matrix pipe runs at (DESIGN.md section 5;
    tools/power_trace.py, profiles/r04_v65_power_trace.jsonl for one kernel at a time)."""

    def __init__(self, torch, dev_index):
        import glob
        self.files, self.card, self.samples, self.thread, self.stop = {}, None, [], None, None
        want = None
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            want = "%04x:%02x:%02x." % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            pass
        cands = []
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            f = {k: os.path.join(d, k) for k in ("power1_average", "power1_input", "power1_cap", "freq1_input") if os.path.exists(os.path.join(d, k))}
            if "power1_average" in f or "power1_input" in f:
                cands.append((os.path.realpath(os.path.join(d, "..", "..")), f))
        hit = [c for c in cands if want and want in c[0]]
        self.matched = bool(hit)
        if hit or cands:
            self.card, self.files = (hit or cands)[0]

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return int(f.read().strip())
        except (OSError, ValueError):
            return None

    def start(self):
        import threading
        pf = self.files.get("power1_average") or self.files.get("power1_input")
        if not pf:
            return
        self.samples, self.stop, self.t0 = [], threading.Event(), time.perf_counter()

        def run():
            while not self.stop.is_set():
                self.samples.append((self._read(pf), self._read(self.files["freq1_input"]) if "freq1_input" in self.files else None))
                self.stop.wait(0.01)
        self.thread = threading.Thread(target=run, daemon=True)
        self.thread.start()

    def finish(self):
        if self.thread is None:
            return None
        self.stop.set()
        self.thread.join(timeout=5)
        span = time.perf_counter() - self.t0
        w = [p / 1e6 for p, _ in self.samples if p is not None]
        f = [c / 1e6 for _, c in self.samples if c is not None]
        cap = self._read(self.files["power1_cap"]) if "power1_cap" in self.files else None
        if not w:
            return None
        return {"power_W_mean": round(sum(w) / len(w), 1), "power_W_max": round(max(w), 1), "power_cap_W": None if cap is None else cap / 1e6,
                "sclk_MHz_mean": round(sum(f) / len(f), 1) if f else None, "sclk_MHz_min": round(min(f), 1) if f else None,
                "sclk_MHz_max": round(max(f), 1) if f else None, "samples": len(w), "period_ms": round(1e3 * span / len(w), 1),
                "source": self.card + (" (PCI address matched to the torch device)" if self.matched else " (first card with a power sensor)"),
                "note": "sampled over the timed steps only; the roofline peaks are nominal (2.4 GHz) figures, the chip sustains the clock "
                        "reported here under this load"}


def measure_peaks(torch, dev):
    """What THIS box sustains, in the same run: a streaming copy (HBM) and register-resident MFMA loops
    (csrc/peaks.hip), HIP-event timed.  BASELINE.md section 4: fractions are printed against nominal AND measured."""
    import ctypes as C
    from cslam_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nbytes = 2 << 30                                          # 2 GiB each way: well past the 256 MB Infinity Cache
    src = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    best, best_v = 0.0, None
    for variant in (0, 1, 2):
        for rep in range(3):
            e0.record()
            for _ in range(3):
                _lib.check(lib.cslam_peak_copy_dev(src.data_ptr(), dst.data_ptr(), nbytes, variant, st))
            e1.record()
            torch.cuda.synchronize()
            gbs = 3 * 2.0 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9
            if rep and gbs > best:                            # first round of every variant warms up
                best, best_v = gbs, variant
    del src, dst
    out = {"hbm_copy_GBs": round(best, 1),
           "hbm_copy_note": "best of three 16 B/lane copy kernels (%s), 2 GiB, read + write bytes" %
                            ("one element per thread", "grid-stride", "grid-stride non-temporal")[best_v]}
    scratch = torch.zeros(64, dtype=torch.float32, device=dev)
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    for kind, name, iters in ((0, "mfma_f32", 16000), (1, "mfma_f16", 64000)):
        for operands, tag in ((0, "ceiling"), (1, "loaded")):
            flop = C.c_double(0.0)
            best = 0.0
            for per_cu in (1, 2):                                  # one or two waves per SIMD
                for rep in range(3):
                    e0.record()
                    _lib.check(lib.cslam_peak_mfma_dev(kind, iters // per_cu, per_cu * ncu, operands, scratch.data_ptr(), C.byref(flop), st))
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1)
                    tf = flop.value / (ms * 1e-3) / 1e12
                    if rep and tf > best:
                        best = tf
            out["%s_%s_TFLOPs" % (name, tag)] = round(best, 1)
    out["mfma_note"] = ("register-resident loops (csrc/peaks.hip), 4 independent 32x32 accumulators per wave, best of 1 / 2 waves per SIMD.  "
                        "`ceiling` = zero operands (the guide's micro-benchmark condition: 155 TF f32, >= 2382 TF fp16); `loaded` = non-zero "
                        "operands of mixed sign: under matrix load the chip clocks to its power budget (ceiling / loaded = the clock ratio), "
                        "so no kernel with real data reaches the ceiling.  Roofline fractions in this line are "
                        "against the NOMINAL peaks only.")
    return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bank-rows", type=int, default=100_000)
    ap.add_argument("--dim", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=1024, help="keyframes per rank per step")
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--match-queries", type=int, default=100_000, help="queries of the match-only leg")
    ap.add_argument("--backbone-dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--extract-chunk", type=int, default=256,
                    help="frames per backbone forward: the step's 1024 frames are four 256-frame passes on four lanes (profiles/"
                         "r06_zz_chunk_lane_sweep.log: +1.2 %% over two 512-frame passes on two lanes, one 1024-frame pass -12 %%; the "
                         "per-kernel rooflines below are priced on 256-frame launches, the shapes of the committed PMC passes)")
    ap.add_argument("--extract-lanes", type=int, default=4,
                    help="HIP streams the step's extract chunks alternate over (NetVLAD.compute_embeddings_batch_device; 1 = one stream)")
    ap.add_argument("--backbone-conv", default="winograd", choices=["winograd", "winograd2", "direct"],
                    help="execution of the wide 3x3 backbone convolutions (vpr/winograd.py)")
    ap.add_argument("--cpu-queries", type=int, default=48, help="cpu_baseline sample size of the match leg (queries)")
    ap.add_argument("--cpu-frames", type=int, default=8, help="cpu_baseline sample size of the extract leg (frames)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c2", action="store_true", help="skip the bounded CosPlace ResNet-18 leg (`c2_cosplace` in the line)")
    ap.add_argument("--no-extract", action="store_true", help="debug: time the match leg only")
    ap.add_argument("--shard-mode", default="rows", choices=["rows", "robots"],
                    help="N > 1 only.  rows (default): the ONE --bank-rows bank of the metric is split by rows over "
                         "the ranks, every keyframe gets its top-k over the whole bank (all-gather of descriptors, "
                         "local top-k, all-to-all of the lists, HIP merge; SURVEY 8e).  robots: BASELINE config 4's "
                         "shape, every rank owns a full --bank-rows bank of its own robot and scores everybody's "
                         "keyframes against it (pair work per rank grows with N)")
    ap.add_argument("--debug-shared-gpu", action="store_true",
                    help="debug only: all ranks share cuda:0 and exchange descriptors through gloo/CPU, to "
                         "exercise the N>1 control flow on a 1-GPU box (numbers are meaningless)")
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` (no launcher): re-exec under torch.distributed.run with N ranks, one per GPU.
    Returns the child's exit code; never returns when the environment already carries a rank."""
    import socket
    import subprocess
    with socket.socket() as s:                       # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    import torch
    import torch.distributed as dist
    from cslam_amd import nns_matching as nnm
    from cslam_amd.sharded import RowShardedBankMatcher, ShardedInterRobotMatcher
    from cslam_amd.vpr.netvlad import NetVLAD

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: the launcher must create exactly "
                         f"--gpus ranks (python bench.py --gpus N launches them itself)")
    if not a.debug_shared_gpu and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {world} needs {world} visible GPUs, found {torch.cuda.device_count()} "
                         f"(--debug-shared-gpu runs the N-rank control flow on one GPU, numbers meaningless)")
    if world > 1 and a.debug_shared_gpu:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local_rank = 0
        dist.init_process_group("gloo")
    elif world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    torch.backends.cudnn.benchmark = True

    # ---- resident state: this robot's bank (BASELINE.md recipe: unit-norm Gaussian rows) ----
    rows_mode = world > 1 and a.shard_mode == "rows"
    offs = [g * a.bank_rows // world for g in range(world + 1)] if rows_mode else [0, a.bank_rows]
    local_rows = offs[rank + 1] - offs[rank] if rows_mode else a.bank_rows     # rows resident on this GPU
    # Inputs are BASELINE.md section 3's, bit for bit (cslam_amd/synthetic.py: numpy default_rng(1234 + robot) / (4321 + robot) /
    # (7 + frame), generated on the host in two threads while the model is set up, uploaded once: resident in HBM before any timed region).
    from concurrent.futures import ThreadPoolExecutor
    from cslam_amd import synthetic
    pool = ThreadPoolExecutor(max_workers=2)
    if rows_mode:
        # ONE seeded bank at every N: rank g holds rows [offs[g], offs[g + 1]) of robot 0's bank, the bank the N = 1 run builds
        # (same generator stream, sliced), so an N-rank result is checkable against the unsharded bank (`sharded_check` below)
        whole_np = pool.submit(synthetic.bank, 0, a.bank_rows, a.dim)
        bank_f = pool.submit(lambda: whole_np.result()[offs[rank]:offs[rank + 1]])
    else:
        bank_f = pool.submit(synthetic.bank, rank, local_rows, a.dim)
    mq_f = pool.submit(synthetic.queries, rank, a.match_queries, a.dim)      # the match-only leg's queries (and --no-extract's)
    bank = torch.from_numpy(bank_f.result()).to(dev)
    nn = nnm.NearestNeighborsMatching(device=local_rank)
    nn.add_items_device(bank)
    extractor = None
    if not a.no_extract:
        extractor = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376,
                             "frontend.netvlad.pca_dim": a.dim, "frontend.random_seed": 0,
                             "frontend.backbone_conv": a.backbone_conv}, None)
    bdt = None if a.backbone_dtype == "fp32" else torch.bfloat16
    frames = torch.from_numpy(synthetic.frames(rank * a.batch, a.batch)).to(dev)     # frame i of the job = default_rng(7 + i)
    mq_all = torch.from_numpy(mq_f.result()).to(dev)
    pool.shutdown()
    noext_pos = [0]
    if extractor is not None:
        # model set-up, not a step: builds the transformed weights, the V / M workspaces and loads the GEMM table
        extractor.compute_embeddings_device(frames[:a.extract_chunk], bdt)
        torch.cuda.synchronize()

    kernel_ms = []

    def search(q, k):
        out = nn.search_device(q, k, mode=nnm.MODE_MFMA)
        kernel_ms.append(nn.last_kernel_ms())
        return out

    hooks = {}
    if a.debug_shared_gpu and world > 1:
        def gather_fn(local, world_size, group=None):          # gloo has no device collectives: stage through the host
            out = torch.empty((world_size * local.shape[0], local.shape[1]), dtype=local.dtype)
            dist.all_gather_into_tensor(out, local.cpu().contiguous())
            return out.to(local.device)

        def exchange_fn(packed, world_size, group=None):
            out = torch.empty(packed.shape, dtype=packed.dtype)
            dist.all_to_all_single(out, packed.cpu().contiguous())
            return out.to(packed.device)
        hooks = {"gather_fn": gather_fn}
        if rows_mode:
            hooks["exchange_fn"] = exchange_fn
    def search_async(q, k):
        return nn.search_device_async(q, k, mode=nnm.MODE_MFMA)
    matcher = (RowShardedBankMatcher(rank, world, search, offs, k=a.k, search_async_fn=search_async, **hooks) if rows_mode
               else ShardedInterRobotMatcher(rank, world, search, k_intra=a.k, search_async_fn=search_async, **hooks))

    def extract():
        if extractor is None:
            p0 = noext_pos[0] % max(1, mq_all.shape[0] - a.batch + 1)     # --no-extract: walk through the query recipe's rows
            noext_pos[0] = p0 + a.batch
            return mq_all[p0:p0 + a.batch]
        if bdt is None:
            return extractor.compute_embeddings_batch_device(frames, a.extract_chunk, a.extract_lanes)
        outs = [extractor.compute_embeddings_device(frames[s:s + a.extract_chunk], bdt)
                for s in range(0, a.batch, a.extract_chunk)]
        return torch.cat(outs)

    # The search of step i is ENQUEUED behind its extraction (cslam_bank_search_enqueue_dev) and FINISHED -- one event
    # wait for the certificate count, no stream synchronisation -- after step i + 1 has been enqueued: no host
    # synchronisation between extract chunks and search, the GPU queue never drains inside the timed region.  N > 1 runs
    # the same two halves through the sharded matchers (all-gather, search, all-to-all and merge enqueued by step_begin;
    # sharded.py), so a rank's step differs from the N = 1 step by its collectives only.
    pending = []
    side = torch.cuda.Stream(device=dev) if world > 1 else None     # N > 1: the matcher's stream (beside the next extraction)
    waiting = []                                                    # N > 1: (descriptors, ready event) of the step whose matching is next

    def retire():
        while pending:
            if side is not None:
                with torch.cuda.stream(side):
                    pending.pop(0).finish()
            else:
                pending.pop(0).finish()
            kernel_ms.append(nn.last_kernel_ms())      # events of a search that has finished: no wait

    def launch_match():
        d, ready = waiting.pop(0)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            pending.append(matcher.step_begin(d))
            d.record_stream(side)

    def step():
        d = extract()                                  # enqueued; the host runs ahead of the GPU
        retire()                                       # step i - 1's search: done long ago, its count is on the host
        if world == 1:
            pending.append(nn.search_device_async(d, a.k, mode=nnm.MODE_MFMA))
            return pending[-1]
        # N > 1: step i - 1's collectives, search and merge go to a second stream NOW, beside step i's extraction just
        # enqueued -- the all-gather and the all-to-all never hold the compute stream, and a host-staged debug collective
        # (--debug-shared-gpu: gloo through pinned host copies) blocks the host only on work that finished a step ago
        ready = torch.cuda.Event()
        ready.record()
        if waiting:
            launch_match()
        waiting.append((d, ready))
        return None

    def flush():
        retire()
        while waiting:                                 # the bank takes one search at a time
            launch_match()
            retire()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, n):
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        flush()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device="cpu" if a.debug_shared_gpu else dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    from cslam_amd import _lib as _clib
    import ctypes as _C
    # one untimed priming step in front of the W warm-up steps: the grow-only workspaces of the trunk (several GB at this chunk), the
    # per-kernel attribute calls and the library's lazy code-object loads all happen on first use (1.1 s on a fresh process) and would
    # otherwise sit inside the timed region of a `--warmup 0` run; the outputs of this step are discarded like those of a warm-up step
    step()
    flush()
    for _ in range(a.warmup):
        step()
    flush()
    kernel_ms.clear()
    trunk_times = None
    trunk_steps, trunk_step_ms = a.steps, None
    trunk_src = "every product launch of the timed steps bracketed by HIP events on its stream (cslam_trunk_timing)"

    def read_trunk_times():
        tt = (_C.c_double * 8)()
        _clib.check(_clib.load().cslam_trunk_timing_read(_C.byref(tt)))
        _clib.check(_clib.load().cslam_trunk_timing(0))
        return [float(x) for x in tt]
    one_lane = extractor is None or a.extract_lanes <= 1 or bdt is not None
    if extractor is not None and one_lane:
        _clib.check(_clib.load().cslam_trunk_timing(1))       # two HIP events per product launch, on the launch stream
    sensors = BoardSensors(torch, dev.index or 0) if rank == 0 else None
    if sensors is not None:
        sensors.start()
    dt = timed(step, a.steps)
    board = sensors.finish() if sensors is not None else None
    if extractor is not None and one_lane:
        trunk_times = read_trunk_times()
    step_kernel_ms = [m for m in kernel_ms if m > 0]
    value = world * a.batch * a.steps / dt
    if extractor is not None and not one_lane:
        # with two lanes a product launch shares the chip with the other lane's kernels and its duration is not its own: the
        # per-launch figures of the trunk's products come from two one-lane steps run after the timed region
        keep_lanes, keep_ms = a.extract_lanes, list(kernel_ms)
        a.extract_lanes = 1
        step()
        flush()
        _clib.check(_clib.load().cslam_trunk_timing(1))
        trunk_steps = 2
        trunk_step_ms = timed(step, trunk_steps) / trunk_steps * 1e3
        trunk_times = read_trunk_times()
        a.extract_lanes = keep_lanes
        kernel_ms[:] = keep_ms
        trunk_src = ("every product launch of %d one-lane steps run after the timed region, bracketed by HIP events on its stream "
                     "(cslam_trunk_timing); in the timed steps the extract chunks alternate over %d lanes and a launch shares the chip "
                     "with the other lane's kernels" % (trunk_steps, keep_lanes))

    # ---- N > 1, rows mode: the sharded step's answer against the UNSHARDED bank (outside the timed region): every rank rebuilds
    # the whole seeded bank, searches its own step's descriptors in it on its own GPU and compares rows / float64 scores / counts
    sharded_check = None
    if rows_mode and extractor is not None:
        d_chk = extract()
        got = matcher.step_begin(d_chk).finish()               # the two halves the timed steps ran on
        whole = torch.from_numpy(whole_np.result()).to(dev)
        nn_whole = nnm.NearestNeighborsMatching(device=local_rank)
        nn_whole.add_items_device(whole)
        want = nn_whole.search_device(d_chk, a.k, mode=nnm.MODE_MFMA)
        ok = bool(torch.equal(got[0], want[0]) and torch.equal(got[2], want[2]) and
                  float((got[1] - want[1]).abs().max()) <= 1e-12)
        flag = torch.tensor([1 if ok else 0], device="cpu" if a.debug_shared_gpu else dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        sharded_check = {"equal_to_unsharded_bank_on_every_rank": bool(int(flag.item()) == 1),
                         "what": "top-%d rows, counts and float64 scores (<= 1e-12) of one step's %d descriptors per rank: "
                                 "row-sharded path over %d ranks vs a single-GPU search of the whole %d-row bank" % (a.k, a.batch, world, a.bank_rows)}
        del whole, nn_whole
        torch.cuda.empty_cache()

    # ---- the two legs separately (rank-local work, same barriers) ----
    extract_only = None
    if extractor is not None:
        de = timed(extract, a.steps)
        extract_only = world * a.batch * a.steps / de
    # the same leg AND the same whole step with the trunk's GEMMs as plain fp32 (rocBLAS sgemm) instead of the default
    # split-fp16 pairs (fp32-grade either way): reported beside `extract_only` / `value`, N = 1 only
    extract_fp32_gemms = value_fp32_gemms = None
    split16 = "128"
    if extractor is not None and world == 1 and extractor.backbone_conv == "winograd":
        from cslam_amd.vpr.winograd import FP32_GEMM_FORMS       # library sgemm between the transforms, conv1_2 / conv2_1 on the f32-input MFMA,
        ex32 = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376,          # conv1_1 as its own fp32 kernel
                        "frontend.netvlad.pca_dim": a.dim, "frontend.random_seed": 0,
                        "frontend.backbone_conv": a.backbone_conv, "frontend.trunk_forms": dict(FP32_GEMM_FORMS)}, None)
        ex32.compute_embeddings_device(frames[:a.extract_chunk], bdt)

        def extract32():
            return torch.cat([ex32.compute_embeddings_device(frames[s_:s_ + a.extract_chunk], bdt)
                              for s_ in range(0, a.batch, a.extract_chunk)])
        d32 = timed(extract32, a.steps)
        extract_fp32_gemms = a.batch * a.steps / d32
        dv32 = timed(lambda: matcher.step(extract32()), a.steps)
        value_fp32_gemms = a.batch * a.steps / dv32
        del ex32
        torch.cuda.empty_cache()
    kernel_ms.clear()
    nqm = a.match_queries
    mq = mq_all[:nqm]
    mout = nn.search_device(mq, a.k, mode=nnm.MODE_MFMA)
    kernel_ms.clear()
    dm = timed(lambda: search(mq, a.k), max(1, min(a.steps, 2)))
    nm = max(1, min(a.steps, 2))
    # rows mode: the ranks together hold ONE bank, so nqm queries per rank against every shard = nqm whole-bank
    # searches (merge not included in this leg); otherwise every rank searches its own full bank
    match_only = (1 if rows_mode else world) * nqm * nm / dm
    match_kernel_ms = float(np.mean(kernel_ms))
    uncertified = nn.last_stats()[0]

    # ---- rooflines.  `roofline` describes the TIMED STEP: the kernel north_star names (the D.D^T similarity + top-k
    # candidate stage), from its launches inside the timed steps (HIP events on the launch stream).  The separate 100k-query
    # C3 batch is `roofline_c3_batch`; the step's largest consumer (the trunk's pair products) is `roofline_step_largest`.
    # Algorithmic work of the match: 2*D flop per (query, bank row) pair.  The candidate stage (a filter: the
    # float64 re-scoring + certificate behind it make the result exact) runs on the fp16 matrix pipe: by default ONE fp16
    # product on the operands' hi halves (csrc/sim_topk_pair.hip NPROD = 1), whose roofline is the dense fp16 MFMA peak itself;
    # CSLAM_MFMA_STAGE1=pair = round 3's exact hi/lo pairs (three products: peak / 3), =f32 = the f32-input MFMA stage of
    # rounds 1-2, priced against the f32 MFMA peak.
    nq_step = world * a.batch if world > 1 else a.batch
    stage1 = os.environ.get("CSLAM_MFMA_STAGE1", "h1")
    n_prod = 0 if stage1.startswith("f") else (3 if stage1.startswith("p") else 1)
    pair_stage = n_prod != 0
    mm_peak = FP16_MFMA_PEAK_TFLOPS / n_prod if pair_stage else FP32_MFMA_PEAK_TFLOPS
    mm_unit = ("TFLOP/s (2*D flop per query-row pair; %d fp16 MFMA product%s per pair: peak = 2500 / %d)"
               % (n_prod, "" if n_prod == 1 else "s", n_prod) if pair_stage else "TFLOP/s")
    # the persistent ring kernel runs the one-product stage from 257 queries on; smaller launches, the three-product stage and a bank
    # that backed off (clustered descriptors) run sim_topk_pair_kernel / sim_topk_mfma_kernel: name what the bank reports
    last_prod = nn.last_stage()[0]
    mm_kernel = ("sim_topk_mfma_kernel" if (not pair_stage or last_prod == 0) else
                 ("sim_topk_ring_kernel" if n_prod == 1 and min(nqm, a.batch) >= 257 else "sim_topk_pair_kernel"))
    peaks = measure_peaks(torch, dev) if rank == 0 else None

    def match_roofline(nq_launch, ms, source, pmc_queries):
        ach = 2.0 * nq_launch * local_rows * a.dim / (ms * 1e-3) / 1e12
        pm = pmc_entry(mm_kernel, queries=pmc_queries, bank_rows=local_rows, dim=a.dim, products=n_prod)
        return {"bound": "mfma", "kernel": mm_kernel, "achieved": round(ach, 2), "peak": round(mm_peak, 1), "unit": mm_unit,
                "frac": round(ach / mm_peak, 4), "kernel_ms": round(ms, 3), "queries_per_launch": nq_launch,
                "fp16_TFLOPs_issued": round(n_prod * ach, 1) if pair_stage else None, "fp16_products": n_prod if pair_stage else None,
                "traffic": pm["traffic_bytes"] if pm else None,
                "traffic_source": (pm["source"] + "; L2 hit rate %.2f" % pm.get("l2_hit_rate", float("nan"))) if pm else None,
                "traffic_kind": "L2-miss (fabric-side) bytes: Infinity-Cache hits are included, so this is an upper bound of the HBM bytes",
                "algorithmic_min_bytes": pm.get("algorithmic_min_bytes") if pm else None,
                "matrix_pipe_busy_pmc": round(pm["mfma_busy_frac"], 3) if pm and pm.get("mfma_busy_frac") else None,
                "effective_clock_GHz_pmc": round(pm["effective_clock_GHz"], 2) if pm and pm.get("effective_clock_GHz") else None,
                "source": source}
    roofline_c3 = match_roofline(nqm, match_kernel_ms, "match-only leg (%d queries per launch), HIP events on the launch stream" % nqm, nqm)
    if step_kernel_ms and world == 1:
        roofline = match_roofline(nq_step, float(np.mean(step_kernel_ms)),
                                  "the %d launches inside the timed steps, HIP events on the launch stream" % len(step_kernel_ms), nq_step)
    else:
        roofline = dict(roofline_c3, note="N > 1: per-launch events of the in-step searches are not collected; this is the match-only leg")
    roofline["peaks_measured_note"] = ("in-run micro-benchmarks (peaks_measured) are reported beside the nominal peaks, not used "
                                       "as denominators: under load the chip clocks to its power budget")
    roofline_step_largest = None
    if trunk_times is not None and rank == 0 and trunk_times[0] + trunk_times[4] > 0:
        lh, msh, flh, byh, lm, msm, flm, bym = trunk_times
        per_step = (msh + msm) / trunk_steps
        roofline_step_largest = {
            "kernel": "wino_gemm_h2_big_kernel / wino_gemm_h2_kernel (the 36-frequency pair products of conv3_1 ... conv5_3)",
            "launches_per_step": round((lh + lm) / trunk_steps, 1), "ms_per_step": round(per_step, 3),
            "share_of_step": round(per_step / (trunk_step_ms if trunk_step_ms is not None else dt / a.steps * 1e3), 4),
            "one_lane_step_ms": None if trunk_step_ms is None else round(trunk_step_ms, 3),
            "source": trunk_src,
            "hbm_bound_layers": None if lh == 0 else {
                "what": "Cin <= 256 (conv3_1 ... conv4_1): V2 in + M out", "bound": "hbm", "launches": int(lh),
                "kernel_ms": round(msh / lh, 4), "achieved": round(byh / msh / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(byh / msh / 1e6 / HBM_PEAK_GBS, 4), "fp16_TFLOPs": round(flh / msh / 1e9, 1)},
            "mfma_bound_layers": None if lm == 0 else {
                "what": "Cin = 512 (conv4_2 ... conv5_3)", "bound": "mfma", "launches": int(lm),
                "kernel_ms": round(msm / lm, 4), "achieved": round(flm / msm / 1e9, 1), "peak": FP16_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s (fp16, 3 products)", "frac": round(flm / msm / 1e9 / FP16_MFMA_PEAK_TFLOPS, 4),
                "GBs": round(bym / msm / 1e6, 1)}}

    # the hand-written kernels of the extract leg are the Winograd transforms (HBM-bound streaming): time the
    # largest one on its real shape with HIP events on the launch stream.  Algorithmic bytes per launch:
    # the activation once + V once (2.25x the activation for 6x6 tiles), DESIGN.md section 3.6.
    extract_roofline = None
    if extractor is not None and rank == 0 and extractor.backbone_conv == "winograd":
        import ctypes as C
        from cslam_amd import _lib
        lib = _lib.load()
        eb, eh, ec = 256, 56, 128                                   # conv3_1's input at 256 frames: the transform's largest launch ON the path
        xt = torch.randn((eb, eh, eh, ec), device=dev)
        vt = torch.empty((36, eb * (eh // 4) * (eh // 4), ec), device=dev)
        st = torch.cuda.current_stream().cuda_stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        nbytes = (xt.numel() + vt.numel()) * 4
        # the trunk's input transform: B^T d B + the exact split into fp16 hi / lo pairs (4 bytes per value, as the fp32 form)
        slot = torch.zeros(1, dtype=torch.float32, device=dev)
        xh = torch.relu(xt).contiguous()
        _lib.check(lib.cslam_absmax_dev(xh.data_ptr(), xh.numel(), slot.data_ptr(), st))

        def wino_in_h2():
            _lib.check(lib.cslam_wino4_input_h2_dev(xh.data_ptr(), eb, eh, eh, ec, slot.data_ptr(), vt.data_ptr(), st))
        wino_in_h2()
        e0.record()
        for _ in range(5):
            wino_in_h2()
        e1.record()
        torch.cuda.synchronize()
        ms2 = e0.elapsed_time(e1) / 5
        shape_h2 = f"x [{eb},{eh},{eh},{ec}] -> V2 [36,{vt.shape[1]},{ec} pairs]"
        ph = pmc_entry("wino4_input_h2_kernel", shape=shape_h2)
        gbs2 = nbytes / ms2 / 1e6
        extract_roofline = {"bound": "hbm", "kernel": "wino4_input_h2_kernel", "achieved": round(gbs2, 1),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs2 / HBM_PEAK_GBS, 4),
                            "peak_measured": peaks["hbm_copy_GBs"],
                            "frac_of_measured": round(gbs2 / peaks["hbm_copy_GBs"], 4) if peaks["hbm_copy_GBs"] else None,
                            "traffic": ph["traffic_bytes"] if ph else None, "traffic_source": ph["source"] if ph else None,
                            "algorithmic_bytes": nbytes, "kernel_ms": round(ms2, 3), "shape": shape_h2,
                            "note": "the trunk's input transform (fp16-pair V) on conv3_1's shape, its largest launch in a pass (PMC: "
                                    "tools/pmc_wino_input_target.py through tools/pmc_kernel.sh); the other kernels of the extract pass "
                                    "follow: pair_gemm, direct_conv, stem_conv"}
        del xt, vt, xh
        # this library's split-fp16 GEMM between the transforms (csrc/wino_gemm.hip), on the two regimes of the trunk:
        # conv3_2 (256 -> 256 channels, 50176 tile rows: HBM-bound, V2 in + M out) and conv4_2 (512 -> 512, 12544 rows:
        # the matrix pipe matters; 3 fp16 MFMA products per fp32-grade product)
        def time_ms(fn, n=5):
            fn()
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        gem = {}
        for tag, hw, cin, cout in (("conv3_2", 56, 256, 256), ("conv4_2", 28, 512, 512)):
            T_ = eb * (hw // 4) * (hw // 4)
            v2 = (torch.randn((36 * T_ * 2 * cin,), device=dev) * 100.0).to(torch.float16)
            u2 = (torch.randn((36 * cout * 2 * cin,), device=dev) * 100.0).to(torch.float16)
            mo = torch.empty((36, T_, cout), device=dev)
            gms = time_ms(lambda: _lib.check(lib.cslam_wino_gemm_h2_dev(v2.data_ptr(), u2.data_ptr(), T_, cin, cout,
                                                                        mo.data_ptr(), st)))
            gb = 36.0 * T_ * (cin + cout) * 4 + 36.0 * cin * cout * 4
            fl16 = 3 * 2.0 * 36 * T_ * cin * cout
            gshape = f"36 x [{T_},{cin}] x [{cin},{cout}]"
            pg = pmc_entry("wino_gemm_h2_kernel/" + tag, shape=gshape)
            gem[tag] = {"kernel": "wino_gemm_h2_kernel", "shape": gshape, "kernel_ms": round(gms, 3),
                        "traffic": pg["traffic_bytes"] if pg else None, "traffic_source": pg["source"] if pg else None,
                        "hbm": {"achieved": round(gb / gms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(gb / gms / 1e6 / HBM_PEAK_GBS, 4), "algorithmic_bytes": gb,
                                "frac_of_measured": round(gb / gms / 1e6 / peaks["hbm_copy_GBs"], 4)},
                        "mfma": {"achieved": round(fl16 / gms / 1e9, 1), "peak": FP16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s (fp16, 3 products)",
                                 "frac": round(fl16 / gms / 1e9 / FP16_MFMA_PEAK_TFLOPS, 4),
                                 "fp32_equivalent_TFLOPs": round(fl16 / 3 / gms / 1e9, 1)},
                        "bound": "hbm" if tag == "conv3_2" else "mfma"}
            del v2, u2, mo
        extract_roofline["pair_gemm"] = gem
        # conv2_1 / conv2_2: the direct one-kernel convolutions on fp16 pairs with register-resident weights (csrc/conv_direct_r.hip;
        # `streaming_direct_kernel_ms`: round 4's kernel with the weights through an LDS ring, csrc/conv_direct_h.hip, their A/B partner),
        # 3 fp16 MFMA products per fp32-grade product; HBM sees the activation in and out
        from cslam_amd.vpr import winograd as wg
        dconv = {}
        for tag, cin, pool in (("conv2_2", 128, True), ("conv2_1", 64, False)):
            xd = torch.relu(torch.randn((eb, cin, 112, 112), device=dev)).contiguous(memory_format=torch.channels_last)
            wd = torch.randn((128, cin, 3, 3), device=dev) / (3.0 * cin ** 0.5)
            bd = torch.randn(128, device=dev)
            Wd = wg.direct_pair_weights(wd)
            sd = torch.zeros(1, dtype=torch.float32, device=dev)
            _lib.check(lib.cslam_absmax_dev(xd.data_ptr(), xd.numel(), sd.data_ptr(), st))
            hms = time_ms(lambda: wg.conv3x3_direct_h(xd, Wd, bd, True, pool, sd, None))
            dms, kname = hms, "conv3x3_direct_h_kernel"
            if cin == 64:                                    # conv2_1: the register-resident form (csrc/conv_direct_r.hip), the trunk's default
                Wdr = wg.direct_r_pair_weights(wd)
                dms, kname = time_ms(lambda: wg.conv3x3_direct_r(xd, Wdr, bd, True, pool, sd, None)), "conv3x3_direct_r_kernel"
            else:                                            # conv2_2: register-resident on output-channel halves (round 6), the trunk's default
                Wdr2 = wg.direct_r2_pair_weights(wd)
                dms, kname = time_ms(lambda: wg.conv3x3_direct_r2(xd, Wdr2, bd, True, pool, sd, None)), "conv3x3_direct_r2_kernel"
            dfl = 3 * 2.0 * eb * 112 * 112 * 9 * cin * 128
            dby = (xd.numel() + eb * 128 * 112 * 112 // (4 if pool else 1)) * 4
            shape_d = f"x [{eb},112,112,{cin}] -> conv {cin}->128 + bias + ReLU" + (" + MaxPool2d" if pool else "")
            pd_ = pmc_entry(kname + "/" + tag, shape=shape_d)
            dconv[tag] = {"bound": "mfma", "kernel": kname, "shape": shape_d, "kernel_ms": round(dms, 3),
                          "streaming_direct_kernel_ms": round(hms, 3),
                          "achieved": round(dfl / dms / 1e9, 1), "peak": FP16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s (fp16, 3 products)",
                          "frac": round(dfl / dms / 1e9 / FP16_MFMA_PEAK_TFLOPS, 4),
                          "fp32_equivalent_TFLOPs": round(dfl / 3 / dms / 1e9, 1),
                          "hbm_algorithmic_bytes": dby, "hbm_GBs": round(dby / dms / 1e6, 1),
                          "traffic": pd_["traffic_bytes"] if pd_ else None, "traffic_source": pd_["source"] if pd_ else None,
                          "matrix_pipe_busy_pmc": pd_.get("mfma_busy_frac") if pd_ else None}
            del xd
        extract_roofline["direct_conv"] = dconv
        fh = 224
        wf = torch.randn((64, 64, 3, 3), device=dev) / 24.0
        U4f = wg.wino_weights(wf, 4).to(dev)
        Uhf = wg.fused64_pair_weights(U4f)
        bf = torch.randn(64, device=dev)
        # conv1_2 alone (fp16-pair one-kernel Winograd form, csrc/wino_fused_h.hip): the stem kernel's A/B partner
        xf = torch.relu(torch.randn((eb, 64, fh, fh), device=dev)).contiguous(memory_format=torch.channels_last)
        slot = torch.zeros(1, dtype=torch.float32, device=dev)
        _lib.check(lib.cslam_absmax_dev(xf.data_ptr(), xf.numel(), slot.data_ptr(), st))
        fms = time_ms(lambda: wg.wino_fused64_h(xf, Uhf, bf, True, True, slot, None))
        del xf
        # the trunk's first TWO convolutions as one launch (conv1_1 folded into the kernel above, csrc/wino_fused_h.hip STEM):
        # the 64-channel map between them (3.3 GB per 256 frames, written and read back) never exists in HBM.  Its floor is the
        # matrix work: 36 x 2 MFMA pairs per tile and quarter + the first layer's 27-tap products; HBM sees image in + pooled out
        x0 = torch.rand((eb, 3, fh, fh), device=dev) * 4.8 - 2.2
        w1 = torch.randn((64, 3, 3, 3), device=dev) / 5.0
        b1 = torch.randn(64, device=dev)
        stemw = wg.stem_pair_weights(w1)
        s0 = torch.zeros(1, dtype=torch.float32, device=dev)
        _lib.check(lib.cslam_absmax_dev(x0.data_ptr(), x0.numel(), s0.data_ptr(), st))
        # (the default since round 4: the DIRECT form of the pair, csrc/conv_stem_direct_h.hip -- the second layer's 147 KB of fp16-pair
        # weights stay in the registers of four waves, each the owner of 16 output channels; the F(4x4) stem kernel is its A/B partner)
        Wrf = wg.stem_direct_pair_weights(wf)
        sms = time_ms(lambda: wg.conv_stem_direct_h(x0, stemw, b1, Wrf, bf, True, s0, None))
        wms = time_ms(lambda: wg.wino_stem64_h(x0, stemw, b1, Uhf, bf, True, s0, None))
        y1 = torch.empty((eb, 64, fh, fh), device=dev, memory_format=torch.channels_last)
        w1k = w1.permute(1, 2, 3, 0).reshape(27, 64).contiguous()
        c1ms = time_ms(lambda: _lib.check(lib.cslam_conv3x3_c3_amax_dev(x0.data_ptr(), w1k.data_ptr(), b1.data_ptr(), eb, fh, fh, 64,
                                                                       1, y1.data_ptr(), None, st)))
        sbytes = (x0.numel() + eb * 64 * (fh // 2) * (fh // 2)) * 4
        ps = pmc_entry("conv_stem_direct_h_kernel", shape=f"x0 [{eb},3,{fh},{fh}] -> conv 3->64 + ReLU -> conv 64->64 + ReLU + MaxPool2d")
        # fp16 MFMA flop issued: the second layer direct (9 taps x 64 x 64, 3 products) + the first layer on the 10 x 18 patch of every
        # 8 x 16 block (K = 27 padded to 32, 3 products)
        sflop16 = eb * fh * fh * (3 * 2.0 * 9 * 64 * 64 + (180.0 / 128.0) * 3 * 2.0 * 32 * 64)
        extract_roofline["stem_conv"] = {
            "bound": "mfma", "kernel": "conv_stem_direct_h_kernel", "kernel_ms": round(sms, 3),
            "wino_f4x4_stem_kernel_ms": round(wms, 3),
            "separate_kernels_ms": round(c1ms + fms, 3), "first_layer_kernel_ms": round(c1ms, 3),
            "achieved": round(sflop16 / sms / 1e9, 1), "peak": FP16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s (fp16, 3 products)",
            "frac": round(sflop16 / sms / 1e9 / FP16_MFMA_PEAK_TFLOPS, 4),
            "fp32_equivalent_TFLOPs": round(sflop16 / 3 / sms / 1e9, 1),
            "hbm_algorithmic_bytes": sbytes, "hbm_GBs": round(sbytes / sms / 1e6, 1),
            "traffic": ps["traffic_bytes"] if ps else None, "traffic_source": ps["source"] if ps else None,
            "matrix_pipe_busy_pmc": ps.get("mfma_busy_frac") if ps else None,
            "note": "one wave per SIMD with the weights register-resident (no weight stream, no scratch); the first layer of block i + 1 and "
                    "the epilogue of block i - 1 ride in block i's MFMA stream; per-phase cycle counts: profiles/r04_v56_stem_direct_phases.log",
            "shape": f"x0 [{eb},3,{fh},{fh}] -> conv 3->64 + ReLU -> conv 64->64 + ReLU + MaxPool2d -> [{eb},{fh // 2},{fh // 2},64]"}
        del x0, y1

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        # The reference's CPU path for the step `value` times: NetVLAD extract on host cores (the reference runs the CNN
        # on the CPU when no CUDA device is present, cslam/vpr/netvlad.py:157-160) + the per-row cosine scan of
        # nns_matching.py:42-61.  Both legs are oracle restatements ("port"), timed on a bounded sample.
        from oracle import pyoracle
        hb = bank.cpu().numpy()
        hq = mq[:a.cpu_queries].cpu().numpy()
        t0 = time.perf_counter()
        oi, os_, oc = pyoracle.nns_search(hb, hq, a.k)
        tc = time.perf_counter() - t0
        same = bool(np.array_equal(oi, mout[0][:a.cpu_queries].cpu().numpy()))
        match_leg = {"value": round(a.cpu_queries / tc, 3), "unit": "keyframes/sec", "cores": 1,
                     "sample": f"{a.cpu_queries} of the match-only queries against the same {a.bank_rows}x{a.dim} bank, "
                               f"top-{a.k}, oracle/nns_oracle.c (scalar C restatement of nns_matching.py:42-61)",
                     "topk_equal_to_gpu": same}
        cpu = dict(match_leg, kind="port", host_cpus=os.cpu_count(), match_leg=match_leg)
        if extractor is not None:
            from oracle import extract_oracle
            ncf = max(1, a.cpu_frames)
            threads = torch.get_num_threads()
            convs = [(m.weight.detach().float().cpu().contiguous(memory_format=torch.contiguous_format),
                      m.bias.detach().float().cpu()) for m in extractor.encoder if isinstance(m, torch.nn.Conv2d)]
            vw, vc = extractor.pool.conv_weight.cpu(), extractor.pool.centroids.cpu()
            comp = extractor.pca_components[:, :64 * 512].cpu().numpy()
            pmean = np.zeros(64 * 512, dtype=np.float32)          # random_init: zero mean
            hf = frames[:ncf].cpu().numpy()
            gd = extractor.compute_embeddings_device(frames[:ncf], bdt).cpu().numpy()
            extract_oracle.netvlad_embed(hf[0], 376, convs, vw, vc, comp, pmean)     # warm-up (thread pools, page-in)
            t0 = time.perf_counter()
            cd = np.stack([extract_oracle.netvlad_embed(f, 376, convs, vw, vc, comp, pmean) for f in hf])
            te = (time.perf_counter() - t0) / ncf
            extract_leg = {"value": round(1.0 / te, 3), "unit": "keyframes/sec", "cores": threads,
                           "sample": f"{ncf} of the step's frames, one at a time like the reference: PIL-equivalent "
                                     f"transform + VGG-16 on torch CPU ({threads} threads) + NetVLADLayer with its "
                                     f"64-cluster loop + PCA 32768->{a.dim} + L2 (oracle/extract_oracle.py, restating "
                                     f"netvlad.py:212-241), same weights as the GPU extractor",
                           "max_abs_diff_vs_gpu_descriptor": float(np.abs(cd - gd).max())}
            tm = tc / a.cpu_queries
            cpu.update({"value": round(1.0 / (te + tm), 3), "cores": threads, "extract_leg": extract_leg,
                        "sample": f"extract+match per keyframe = {te * 1e3:.1f} ms extract ({ncf} frames, {threads} torch "
                                  f"threads) + {tm * 1e3:.1f} ms match ({a.cpu_queries} queries, 1 core: the reference's "
                                  f"scan is a single-threaded Python loop) against the same {a.bank_rows}x{a.dim} bank"})
            del convs, comp
        # the strongest plain-numpy host formulation (SURVEY 8(d) "vectorised" flavour): one BLAS sgemm over
        # the bank on every host core + argpartition; reported beside the faithful scalar port, never as it
        nb = min(512, nqm)
        hq2 = mq[:nb].cpu().numpy()
        hn = 1.0 / np.sqrt(np.einsum("ij,ij->i", hb, hb, dtype=np.float64)).astype(np.float32)
        t0 = time.perf_counter()
        sims = (hq2 @ hb.T) * hn[None, :]
        part = np.argpartition(-sims, a.k, axis=1)[:, :a.k]
        order = np.take_along_axis(part, np.argsort(-np.take_along_axis(sims, part, axis=1), axis=1), axis=1)
        tb = time.perf_counter() - t0
        agree = float((order == mout[0][:nb].cpu().numpy()).all(axis=1).mean())
        cpu["blas_flavour"] = {"value": round(nb / tb, 2), "unit": "keyframes/sec", "cores": os.cpu_count(),
                               "sample": f"{nb} queries, float32 sgemm + argpartition (numpy/OpenBLAS, all host "
                                         f"cores); float32 scores, so not bit-compatible with the reference",
                               "top5_rows_equal_to_gpu_frac": round(agree, 4)}

    # ---- BASELINE config 2, the reference's DEFAULT extractor (global_descriptor_loop_closure_detection.py:56-60): CosPlace
    # ResNet-18 512-D on 640x480 keyframes (centre crop 376, bicubic resize to 224) + causal top-5 over the growing bank.
    # A bounded leg beside the headline (which stays C3): chunks of 1000 frames, the bank grows chunk by chunk.  Every trunk
    # layer runs through this library's implicit-GEMM convolution on fp16 pairs (csrc/conv_igemm.hip): hence the fp16 roof.
    c2 = None
    if rank == 0 and world == 1 and extractor is not None and not a.no_c2:
        from cslam_amd.vpr.cosplace import CosPlace
        del bank
        torch.cuda.empty_cache()
        cp = CosPlace({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376,
                       "frontend.cosplace.descriptor_dim": 512, "frontend.cosplace.backbone": "resnet18",
                       "frontend.backbone_conv": "winograd"}, None)
        ch = min(1000, frames.shape[0])
        nchunks = 3
        cp.compute_embeddings_device(frames[:ch])
        torch.cuda.synchronize()                                  # warm-up (weight pairs, workspaces)
        nn2 = nnm.NearestNeighborsMatching()
        te2 = tm2 = 0.0
        done2 = 0
        for _ in range(nchunks):
            t0 = time.perf_counter()
            d2 = cp.compute_embeddings_device(frames[:ch])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            nn2.add_items_device(d2)
            lim2 = torch.arange(done2, done2 + ch, device=dev, dtype=torch.int64)   # keyframe i sees rows < i
            r2 = nn2.search_device(d2, a.k, row_limit=lim2, mode=nnm.MODE_AUTO)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            te2 += t1 - t0
            tm2 += t2 - t1
            done2 += ch
        # the same work as the bench's step runs it: the search of chunk i in two halves, its finish() behind the enqueue of chunk
        # i + 1's extraction -- no host wait inside the timed region except for events that have long passed (the bank may not change
        # while a search is in flight, so the finish precedes the append)
        def pipelined(lanes, npipe):
            """npipe iterations of (extract lanes x ch frames -- over `lanes` HIP streams like the headline step's extraction --, finish
            the previous search, append, enqueue the causal search); returns (keyframes, seconds, the matcher)."""
            nn_ = nnm.NearestNeighborsMatching()
            fr = frames[:ch] if lanes == 1 else torch.cat([frames[:ch]] * lanes)
            cp.compute_embeddings_batch_device(fr, ch, lanes)             # the lanes' own runners and workspaces
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            pend_, done_ = None, 0
            for _ in range(npipe):
                d_ = cp.compute_embeddings_batch_device(fr, ch, lanes)
                if pend_ is not None:
                    pend_.finish()
                nn_.add_items_device(d_)
                lim_ = torch.arange(done_, done_ + d_.shape[0], device=dev, dtype=torch.int64)
                pend_ = nn_.search_device_async(d_, a.k, row_limit=lim_, mode=nnm.MODE_AUTO)
                done_ += d_.shape[0]
            pend_.finish()
            torch.cuda.synchronize()
            return done_, time.perf_counter() - t0_, nn_
        done1, tp1, _nn1 = pipelined(1, 8)
        del _nn1
        done3, tp, nn3 = pipelined(2, 4)
        c2_stage, c2_unc = nn3.last_stage(), int(nn3.last_stats()[0])
        GF = 3.64                                                 # ResNet-18 trunk at 224 x 224: 1.82 G multiply-adds per frame
        fps = done2 / te2
        c2 = {"workload": "C2: CosPlace ResNet-18 512-D extract + causal top-%d, %d synthetic 640x480 keyframes in chunks of %d over two extraction "
                          "lanes (as the headline step), the search of iteration i finished behind the enqueue of iteration i + 1's extraction" % (a.k, done3, ch),
              "value": round(done3 / tp, 1), "unit": "keyframes/sec", "value_one_lane": round(done1 / tp1, 1), "extract_only": round(fps, 1),
              "match_only": round(done2 / tm2, 1), "value_serial": round(done2 / (te2 + tm2), 1),
              "serial_note": "extract_only / match_only / value_serial: %d chunks with a host synchronisation between the legs" % nchunks,
              "dtype": "f32",
              "match_stage": {"fp16_products_of_last_search": c2_stage[0], "searches_left_on_f32_stage": c2_stage[1], "uncertified_queries_last_search": c2_unc,
                              "note": "0 products = the f32-input candidate stage: random-init CosPlace descriptors cluster, the bank backs off from the "
                                      "fp16 stage (results stay exact: float64 re-score + certificate); match_only here says nothing about the fp16 stage on 512-D"},
              "roofline": {"bound": "mfma", "kernel": "conv3x3_direct_p_kernel (layer1: register-resident weights, csrc/conv_direct_p.hip) + conv_igemm_h2_kernel (+ conv_stem_pool_patch_kernel: the 7x7 stem with its max-pool): every trunk layer as an implicit "
                                                        "GEMM over exact fp16 hi/lo pairs of activations and weights, 3 fp16 products per "
                                                        "multiply-add (csrc/conv_igemm.hip; per-layer times: tools/perf_conv_igemm.py, DESIGN.md 3.6d)",
                           "achieved": round(fps * GF * 3 / 1e3, 1), "peak": FP16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s (fp16, 3 products)",
                           "frac": round(fps * GF * 3 / 1e3 / FP16_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                           "direct_conv_TFLOPs": round(fps * GF / 1e3, 1),
                           "note": "whole-extract rate from wall time (crop, resize, trunk, GeM head): it includes the non-GEMM passes and "
                                   "the stem's padded K (147 taps in 7 blocks of 32), so it is below every layer's own fraction"}}
        # the four stride-1 3x3 layer shapes of the trunk alone, in the pair format the trunk runs them in (HIP events on the launch
        # stream, 10 launches each): the dominant kernel's own rate beside the whole-extract figure above
        from cslam_amd.vpr import winograd as wg
        lay = {}
        ws_l = wg._Workspace()
        for lname, cch, hw_l in (("layer1 64->64 @56", 64, 56), ("layer2 128->128 @28", 128, 28), ("layer3 256->256 @14", 256, 14),
                                 ("layer4 512->512 @7", 512, 7)):
            xl = torch.randn((ch, cch, hw_l, hw_l), device=dev).contiguous(memory_format=torch.channels_last)
            wl = torch.randn((cch, cch, 3, 3), device=dev) / (3 * cch ** 0.5)
            sl = torch.zeros(8, device=dev)
            sl[0] = xl.abs().max()
            eye = torch.eye(cch, device=dev).reshape(cch, cch, 1, 1).contiguous()
            ap = wg.conv_igemm_p(ws_l, wg.PairAct(xl, False, xl.shape, sl[0:1], sl[0:1]), wg.igemm_pair_weights(eye), None, (1, 1), 1, 0,
                                 False, None, 1.0, 0.0, sl[1:2], sl[2:3], True)
            Wl, wl1 = wg.igemm_pair_weights(wl), float(wl.abs().sum(dim=(1, 2, 3)).max())
            run_l = lambda: wg.conv_igemm_p(ws_l, ap, Wl, None, (3, 3), 1, 1, True, None, wl1, 0.0, sl[3:4], sl[4:5], True)   # noqa: E731
            if cch == 64 and wg.DIRECT_P:                 # layer1: the register-resident direct kernel (csrc/conv_direct_p.hip), the trunk's default
                Wp_l = wg.stem_direct_pair_weights(wl)
                run_l = lambda: wg.conv3x3_direct_p(ap, Wp_l, None, True, None, wl1, 0.0, sl[3:4], sl[4:5], True)   # noqa: E731
            run_l()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run_l()
            e1.record()
            torch.cuda.synchronize()
            ms_l = e0.elapsed_time(e1) / 10
            fl_l = 2.0 * 3 * ch * hw_l * hw_l * cch * 9 * cch
            lay[lname] = {"kernel": "conv3x3_direct_p_kernel" if (cch == 64 and wg.DIRECT_P) else "conv_igemm_h2_kernel",
                          "kernel_ms": round(ms_l, 4), "achieved": round(fl_l / ms_l / 1e9, 1),
                          "frac": round(fl_l / ms_l / 1e9 / FP16_MFMA_PEAK_TFLOPS, 4)}
            pe = pmc_entry(lay[lname]["kernel"] + "/" + lname.split()[0]) if ch == 1000 else None     # (collected at 1000 frames)
            if pe:
                lay[lname].update({"traffic": pe["traffic_bytes"], "algorithmic_bytes": pe["algorithmic_min_bytes"],
                                   "l2_hit_rate_pmc": round(pe["l2_hit_rate"], 3), "matrix_pipe_busy_pmc": round(pe["mfma_busy_frac"], 3),
                                   "effective_clock_GHz_pmc": round(pe["effective_clock_GHz"], 2), "traffic_source": pe["source"]})
            del xl, ap
        c2["roofline"]["layers"] = {"kernel": "the stride-1 3x3 layer of each stage alone, pair-format input and output, %d frames" % ch,
                                    "unit": "TFLOP/s (fp16, 3 products)", "peak": FP16_MFMA_PEAK_TFLOPS, **lay}
        if not a.no_cpu_baseline:
            ncf = max(1, min(a.cpu_frames, 8))
            xcpu = torch.randn((ncf, 3, 224, 224))
            bb = cp.model.backbone.float().cpu()
            with torch.no_grad():
                bb(xcpu[:1])
                t0 = time.perf_counter()
                for i in range(ncf):
                    bb(xcpu[i:i + 1])
                tcpu = (time.perf_counter() - t0) / ncf
            cp.model.backbone.to(dev)
            c2["cpu_baseline"] = {"value": round(1.0 / tcpu, 2), "unit": "keyframes/sec", "cores": torch.get_num_threads(), "kind": "port",
                                  "sample": "%d frames, one at a time like the reference (cosplace.py:81-101): the same ResNet-18 trunk on torch CPU, "
                                            "224 x 224 input; transform, GeM + FC and the scan not included (they are < 2 %% of it)" % ncf}
        del cp, nn2

    roofline_step = None
    if rank == 0 and extractor is not None:
        # The WHOLE timed step priced once: fp16 MFMA flop issued by everything in it (the trunk's layer table, the PCA projection on
        # fp16 pairs, the candidate stage's one product) / ms_per_step / the nominal dense fp16 peak, with the step's fabric-side bytes
        # (committed PMC pass of the extract: 48.2 GB per 256 frames) beside it.  `roofline` above is the kernel north_star names: 1.4 %
        # of this step.
        f_iss, f_dir = vgg16_flop_per_frame(224)
        pca_iss = 3 * 2.0 * 32768 * a.dim                       # 32768 -> dim projection, three products
        match_iss = 2.0 * nq_step * local_rows * a.dim * max(n_prod, 1) / world
        step_s = dt / a.steps
        iss_step = a.batch * (f_iss + pca_iss) + match_iss
        ext_bytes = 48.2e9 / 256.0 * a.batch
        loaded = peaks.get("mfma_f16_loaded_TFLOPs") if peaks else None
        roofline_step = {
            "bound": "mfma", "what": "every kernel of one timed step (extract of %d frames + match), per GPU" % a.batch,
            "fp16_flop_issued_per_step": iss_step, "achieved": round(iss_step / step_s / 1e12, 1), "peak": FP16_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s (fp16 MFMA flop issued: 3 products per fp32-grade multiply-add, Winograd layers at their 36 / 144)",
            "frac": round(iss_step / step_s / 1e12 / FP16_MFMA_PEAK_TFLOPS, 4),
            "direct_conv_equivalent": {"flop_per_frame": f_dir, "achieved_TFLOPs": round(a.batch * f_dir / step_s / 1e12, 1),
                                       "frac": round(a.batch * f_dir / step_s / 1e12 / FP16_MFMA_PEAK_TFLOPS, 4),
                                       "note": "BASELINE.md section 4 prices C3's extract this way (30.7 GFLOP per frame against 2.5 PF = 81k frames/s)"},
            "traffic": ext_bytes, "traffic_source": "profiles/r04_v60_extract_kernels.txt / pmc_by_kernel.json: 48.2 GB of fabric-side bytes per 256-frame extract pass "
                                                    "(V + M round trips of the nine Winograd layers: 41 GB), scaled to the step's frames; the match adds 1.3 GB",
            "traffic_GBs": round(ext_bytes / step_s / 1e9, 1),
            "fp32_grade_ceiling": {
                "keyframes_per_sec": None if not loaded else round(loaded * 1e12 / (f_iss + pca_iss), 1),
                "note": "what one MI355X could extract if EVERY issued flop ran at the rate the register-resident fp16 loop reaches with non-zero "
                        "operands under the board's power cap (peaks_measured.mfma_f16_loaded_TFLOPs) and nothing else cost time or energy: "
                        "north_star's 100k keyframes/s on C3 is above this ceiling at fp32-grade arithmetic (three fp16 products per multiply-add); "
                        "BASELINE.md section 4 said so up front (81k at ONE product per multiply-add and the nominal peak)"},
            "power_note": "board.power_W / sclk_MHz are the sensors over these timed steps.  One matrix-bound kernel at a time runs the board at its cap "
                          "(profiles/r06_d_power_dp.json: ResNet layer1's kernel 1399 W of 1400 at 1.87 GHz; its matrix work alone 1315 W at 2.0 GHz), so its time "
                          "is its energy / the cap; a whole VGG pass, with its HBM-bound transforms and launch tails, averages below the cap: DESIGN.md section 8"}
    if rank == 0:
        line = {
            "metric": "keyframes/sec (extract+match) on 100kx4096-D bank",
            "value": round(value, 2), "unit": "keyframes/sec", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if bdt is None else "bf16 backbone / f32 heads+match",
            "data": "synthetic (seeded on-device uint8 640x480 frames, unit-norm Gaussian bank, random-init "
                    "VGG-16/VLAD/PCA weights: the reference ships no checkpoints)",
            "note": "north_star's 100k keyframes/s is above what one MI355X can extract on C3 at fp32-grade arithmetic: roofline_step.fp32_grade_ceiling "
                    "(c2_cosplace is the configuration where it is arithmetically within reach)",
            "config": {"workload": "C3: NetVLAD VGG-16 4096-D extract + D.D^T MFMA similarity + top-5, "
                                   f"{a.bank_rows}-row bank" + (" per GPU" if not rows_mode else "") + ("" if extractor else " [match leg only]"),
                       "bank_rows": a.bank_rows, "dim": a.dim, "keyframes_per_rank_per_step": a.batch, "k": a.k,
                       "queries_per_rank_per_step": nq_step,
                       "extract_chunk": a.extract_chunk, "extract_lanes": a.extract_lanes if extractor is not None else None,
                       "bank_rows_per_gpu": local_rows,
                       "parallelism": ("single GPU" if world == 1 else
                                       "one %d-row bank split by rows over %d GPUs: RCCL all-gather of the new descriptors, "
                                       "local top-k, all-to-all of the lists, HIP merge" % (a.bank_rows, world) if rows_mode
                                       else "1 robot bank per GPU, RCCL all-gather of new descriptors")},
            "sharded_check": sharded_check,
            "ranks": world, "collective_backend": None if world == 1 else ("gloo via host (debug)" if a.debug_shared_gpu else "nccl (RCCL)"),
            "value_fp32_gemms": None if value_fp32_gemms is None else round(value_fp32_gemms, 2),
            "peaks_measured": peaks,
            "extract_only": None if extract_only is None else round(extract_only, 2),
            "backbone_conv": None if extractor is None else extractor.backbone_conv,
            "trunk_gemm": None if extractor is None or extractor.backbone_conv != "winograd" else (
                "plain fp32 (rocBLAS sgemm)" if split16 == "0" else
                "conv3_1 ... conv5_3: F(4x4) Winograd with this library's GEMM (csrc/wino_gemm.hip) over exact fp16 hi/lo "
                "pairs of both operands, 3 of the 4 partial products on the fp16 MFMA pipe with fp32 accumulation (error vs "
                "float64 = that of the fp32 GEMM, tests/test_heads_gpu.py::test_split16_*, tests/test_wino_gemm_gpu.py); "
                "conv1_1 + conv1_2: ONE direct kernel on fp16 pairs, the second layer's weights register-resident "
                "(csrc/conv_stem_direct_h.hip); conv2_1: the same form (csrc/conv_direct_r.hip); conv2_2: the same form on "
                "output-channel halves, two workgroups per block (csrc/conv_direct_r.hip, round 6)"),
            "extract_only_fp32_gemms": None if extract_fp32_gemms is None else round(extract_fp32_gemms, 2),
            "match_only": round(match_only, 2),
            "match_only_queries": nqm,
            "uncertified_queries": int(uncertified),
            "roofline": roofline,
            "roofline_step": roofline_step,
            "roofline_c3_batch": roofline_c3,
            "roofline_step_largest": roofline_step_largest,
            "roofline_extract": extract_roofline,
            "cpu_baseline": cpu,
            "c2_cosplace": c2,
            "board": board,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
