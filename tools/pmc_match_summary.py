#!/usr/bin/env python
"""Summarise the rocprofv3 --pmc passes of tools/gpu_pmc_match.sh: per launch shape of the candidate-stage kernel (100 000 and 1024
queries against the 100k x 4096 bank) the HBM-side bytes (FETCH_SIZE x 2 + WRITE_SIZE on gfx950: MI355X guide, HBM section; the two
counters from separate runs), the L2 hit rate, the matrix-pipe busy fraction and the kernel time of the traced run.
    python tools/pmc_match_summary.py <dir> <tag>     -> JSON on stdout; also merges the entries into profiles/pmc_by_kernel.json"""
import collections
import csv
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPS = 3


def per_dispatch(d, counters):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    out = collections.OrderedDict()
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] not in counters:
                continue
            e = out.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [out[k] for k in sorted(out)]


def groups(seq, sub):
    v = [e for e in seq if sub in e["name"]]
    return v[:REPS], v[REPS:2 * REPS]


def med(g, key):
    x = [e[key] for e in g if key in e]
    return statistics.median(x) if x else None


def main():
    src, tag = sys.argv[1], sys.argv[2]
    stage = os.environ.get("CSLAM_MFMA_STAGE1", "h1")
    prod = 0 if stage[0] == "f" else (3 if stage[0] == "p" else 1)        # fp16 products per pair (0: the f32-input stage)
    # one fp16 product on 256 x 256 tiles = the persistent stage (sim_topk_ring.hip), three = round 3's pair kernel
    kern = {0: "sim_topk_mfma_kernel", 1: "sim_topk_ring_kernel", 3: "sim_topk_pair_kernel"}[prod]
    bpv = {0: 4, 1: 2, 3: 4}[prod]                                        # bytes per value of the operand copies the stage reads
    fetch = per_dispatch(os.path.join(src, "fetch"), ("FETCH_SIZE",))
    write = per_dispatch(os.path.join(src, "write"), ("WRITE_SIZE",))
    tcc = per_dispatch(os.path.join(src, "tcc"), ("TCC_HIT_sum", "TCC_MISS_sum"))
    sq = per_dispatch(os.path.join(src, "sq"), ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY",
                                                "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"))
    times = {}
    for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
        times = {100000: statistics.median(d[:REPS]) if len(d) >= REPS else None, 1024: statistics.median(d[REPS:2 * REPS]) if len(d) >= 2 * REPS else None}
    out = {"kernel": kern, "tag": tag, "fp16_products": prod, "bank": "100000 x 4096 float32 (+ the candidate stage's copy, %d bytes per value)" % bpv,
           "traffic_is": "L2-miss (fabric-side) bytes: FETCH_SIZE x 2 + WRITE_SIZE; Infinity-Cache hits are included (MI355X guide, HBM section)",
           "launches": {}}
    path = os.path.join(ROOT, "profiles", "pmc_by_kernel.json")
    table = json.load(open(path)) if os.path.exists(path) else {}
    for gi, nq in enumerate((100000, 1024)):
        f, w, t, s = (groups(x, kern)[gi] for x in (fetch, write, tcc, sq))
        fk, wk = med(f, "FETCH_SIZE"), med(w, "WRITE_SIZE")
        hit, miss = med(t, "TCC_HIT_sum"), med(t, "TCC_MISS_sum")
        e = {"queries": nq}
        if fk is not None and wk is not None:
            e["FETCH_SIZE_KB"], e["WRITE_SIZE_KB"] = fk, wk
            e["traffic_bytes"] = fk * 1024 * 2 + wk * 1024
        # algorithmic minimum: the bank's candidate-stage copy once + the queries' copy once
        e["algorithmic_min_bytes"] = 100000 * 4096 * bpv + nq * 4096 * bpv
        if e.get("traffic_bytes"):
            e["traffic_over_algorithmic"] = e["traffic_bytes"] / e["algorithmic_min_bytes"]
        if hit is not None and miss is not None and hit + miss > 0:
            e["l2_hit_rate"] = hit / (hit + miss)
        busy, mf = med(s, "SQ_BUSY_CYCLES"), med(s, "SQ_VALU_MFMA_BUSY_CYCLES")
        for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY",
                  "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"):
            v = med(s, k)
            if v is not None:
                e[k] = v
        if e.get("GRBM_GUI_ACTIVE") and mf:
            # SQ_VALU_MFMA_BUSY_CYCLES sums cycles over the 1024 SIMDs (4 per CU, 256 CUs); rocprofv3 reports GRBM_GUI_ACTIVE summed
            # over the 8 XCDs (its value / 8 / kernel time = the effective shader clock, 1.8 GHz under this load)
            active = e["GRBM_GUI_ACTIVE"] / 8.0
            e["mfma_busy_frac"] = mf / (active * 256 * 4)
            if times.get(nq):
                e["effective_clock_GHz"] = active / (times[nq] * 1e-3) / 1e9
        if e.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac"] = e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"]
        if times.get(nq):
            e["kernel_ms_traced"] = times[nq]
            e["TFLOPs_traced"] = 2.0 * nq * 100000 * 4096 / (times[nq] * 1e-3) / 1e12      # 2 D flop per (query, row) pair
        out["launches"][str(nq)] = e
        if e.get("traffic_bytes"):
            key = kern if nq == 100000 else kern + "/q%d" % nq
            table[key] = {"traffic_bytes": e["traffic_bytes"], "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk,
                          "algorithmic_min_bytes": e["algorithmic_min_bytes"], "l2_hit_rate": e.get("l2_hit_rate"),
                          "mfma_busy_frac": e.get("mfma_busy_frac"), "effective_clock_GHz": e.get("effective_clock_GHz"),
                          "match": {"queries": nq, "bank_rows": 100000, "dim": 4096, "products": prod},
                          "source": "profiles/%s_pmc_match_summary.json (separate rocprofv3 --pmc passes of tools/pmc_match_target.py, tools/gpu_pmc_match.sh): L2-miss (fabric-side) bytes, Infinity-Cache hits included" % tag}
    json.dump(table, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
