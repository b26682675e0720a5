#!/bin/bash
# Generic rocprofv3 passes for ONE kernel of a target command, each counter set in its own run (--pmc with --kernel-trace only):
#   tools/pmc_kernel.sh <tag> <kernel-name-substring> <command...>
# Prints (and leaves in gpurun_out/<tag>/summary.json) the median over the kernel's dispatches of every counter, the derived
# matrix-pipe busy fraction, L2 hit rate, LDS conflict share and L2-miss (fabric-side) bytes = FETCH_SIZE x 2 + WRITE_SIZE.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; kern=$2; shift 2
O=$R/gpurun_out/$tag; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- "$@" > $O/fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- "$@" > $O/write.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc -o t -- "$@" > $O/tcc.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o s -- "$@" > $O/sq.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o k -- "$@" > $O/trace.log 2>&1
python - "$O" "$kern" <<'PY'
import collections, csv, glob, json, os, statistics, sys
src, kern = sys.argv[1], sys.argv[2]
med = {}
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for c, d in per.items():
        med[c] = statistics.median(d.values())
        med.setdefault("dispatches", len(d))
times = []
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True):
    times += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
out = {"kernel": kern, "counters_median_per_dispatch": med}
if times:
    out["kernel_ms_traced_median"] = statistics.median(times)
if "FETCH_SIZE" in med and "WRITE_SIZE" in med:
    out["l2_miss_fabric_bytes"] = med["FETCH_SIZE"] * 2048 + med["WRITE_SIZE"] * 1024
    out["l2_miss_fabric_bytes_note"] = "FETCH_SIZE x 2 + WRITE_SIZE (KB): fabric-side requests, Infinity-Cache hits included (MI355X guide, HBM section)"
if med.get("TCC_HIT_sum") is not None and med.get("TCC_MISS_sum") is not None and med["TCC_HIT_sum"] + med["TCC_MISS_sum"] > 0:
    out["l2_hit_rate"] = med["TCC_HIT_sum"] / (med["TCC_HIT_sum"] + med["TCC_MISS_sum"])
if med.get("GRBM_GUI_ACTIVE") and med.get("SQ_VALU_MFMA_BUSY_CYCLES"):
    active = med["GRBM_GUI_ACTIVE"] / 8.0                 # summed over the 8 XCDs
    out["mfma_busy_frac"] = med["SQ_VALU_MFMA_BUSY_CYCLES"] / (active * 256 * 4)
    if times:
        out["effective_clock_GHz"] = active / (statistics.median(times) * 1e-3) / 1e9
if med.get("SQ_WAVE_CYCLES"):
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if med.get(k) is not None:
            out[k + "_over_WAVE_CYCLES"] = med[k] / med["SQ_WAVE_CYCLES"]
if med.get("SQ_LDS_IDX_ACTIVE"):
    out["lds_bank_conflict_frac"] = med.get("SQ_LDS_BANK_CONFLICT", 0.0) / med["SQ_LDS_IDX_ACTIVE"]
json.dump(out, open(os.path.join(src, "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $O -name "*.csv" -size +5M -delete
