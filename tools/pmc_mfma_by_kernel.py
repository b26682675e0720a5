#!/usr/bin/env python
"""Matrix-pipe utilisation per launch of the kernels tools/pmc_targets.py runs, from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE):  python tools/pmc_mfma_by_kernel.py <dir> -> profiles/r02_mfma_busy_by_kernel.json
util = MFMA busy cycles summed over the 1024 SIMDs / 1024 / (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8), as in
tools/pmc_summary.py; median over the launches of a kernel name."""
import collections, csv, glob, json, os, statistics, sys

src = sys.argv[1]
f = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)[0]
byd = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    e = byd.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
per = collections.OrderedDict()
for e in byd.values():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in e or not e.get("GRBM_GUI_ACTIVE"):
        continue
    name = e["name"].split("(")[0][:70]
    per.setdefault(name, []).append((e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (e["GRBM_GUI_ACTIVE"] / 8.0), e["GRBM_GUI_ACTIVE"] / 8.0))
out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE on tools/pmc_targets.py (round 2)", "kernels": {}}
for name, v in per.items():
    if not any(u > 0 for u, _ in v):
        continue                                   # kernels without matrix instructions
    # launches of one name with very different lengths are different shapes (conv2_2 / conv4_2 GEMM, conv1_2 / stem): list them in order
    out["kernels"][name] = [{"mfma_busy_frac": round(u, 4), "gpu_cycles": int(c)} for u, c in v]
os.makedirs("profiles", exist_ok=True)
json.dump(out, open("profiles/r02_mfma_busy_by_kernel.json", "w"), indent=1)
for name, v in out["kernels"].items():
    print(name, [x["mfma_busy_frac"] for x in v])
