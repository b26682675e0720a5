"""Summarise rocprofv3 PMC passes of the dominant kernel into profiles/<round>_pmc_summary.json.

    python tools/pmc_summary.py gpurun_out r01b

HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
bytes = FETCH_SIZE[KB]*1024*2 (gfx950 rocprofv3 reports half of a wide coalesced read) +
WRITE_SIZE[KB]*1024, each counter collected in its own pass.  The largest launch of
sim_topk_mfma_kernel in the trace (the 100k-query match leg) is reported.
"""
import collections
import csv
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
out = {}


def counters(path):
    rows = list(csv.DictReader(open(path))) if os.path.exists(path) else []
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        if "sim_topk_mfma" in r["Kernel_Name"]:
            agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    return agg


def biggest(agg, name):
    vals = [v[name] for v in agg.values() if name in v]
    return max(vals) if vals else None


f = counters(os.path.join(src, "pmc_fetch", "f_counter_collection.csv"))
w = counters(os.path.join(src, "pmc_write", "w_counter_collection.csv"))
t = counters(os.path.join(src, "pmc_tcc", "t_counter_collection.csv"))
s = counters(os.path.join(src, "pmc_sq", "s_counter_collection.csv"))
fetch_kb = biggest(f, "FETCH_SIZE")
write_kb = biggest(w, "WRITE_SIZE")
out["kernel"] = "sim_topk_mfma_kernel"
out["workload"] = "100000 queries x 100000 rows x 4096-D, top-5"
out["FETCH_SIZE_KB"] = fetch_kb
out["WRITE_SIZE_KB"] = write_kb
if fetch_kb is not None:
    out["hbm_read_bytes_corrected_x2"] = fetch_kb * 1024 * 2
    out["traffic_bytes"] = fetch_kb * 1024 * 2 + (write_kb or 0) * 1024
hit, miss = biggest(t, "TCC_HIT_sum"), biggest(t, "TCC_MISS_sum")
if hit is not None and miss is not None:
    out["TCC_HIT_sum"], out["TCC_MISS_sum"] = hit, miss
    out["l2_hit_rate"] = hit / (hit + miss)
for name in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE",
             "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
    v = biggest(s, name)
    if v is not None:
        out[name] = v
if "SQ_VALU_MFMA_BUSY_CYCLES" in out and "GRBM_GUI_ACTIVE" in out:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over 256 CUs x 4 SIMDs
    out["gpu_clock_cycles"] = out["GRBM_GUI_ACTIVE"] / 8.0
    out["mfma_util_est"] = out["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / out["gpu_clock_cycles"]
out["algorithmic_flop"] = 2.0 * 1e5 * 1e5 * 4096
out["algorithmic_min_bytes"] = 2 * 1e5 * 4096 * 4
os.makedirs("profiles", exist_ok=True)
json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
