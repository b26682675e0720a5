#!/usr/bin/env python
"""HBM-side traffic of the conv2_2-shaped `wino4_input_kernel` launch (the one bench.py's `roofline_extract` times)
from the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_round4.sh -> profiles/<tag>_extract_pmc_summary.json.
bytes = FETCH_SIZE[KB]*1024*2 + WRITE_SIZE[KB]*1024 (gfx950 correction of the MI355X guide), per launch.

    python tools/pmc_extract_summary.py gpurun_out r01
"""
import collections
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
B, H, C = 256, 112, 128
grid = B * (H // 4) * (H // 4) * (C // 2)          # threads of that launch (Grid_Size is in work-items)


def per_launch(d, counter):
    f = glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True)[0]
    byd = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and r["Kernel_Name"].startswith("wino4_input_kernel") and \
                abs(int(r["Grid_Size"]) - grid) < 256:
            byd[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    v = sorted(byd.values())
    return v[len(v) // 2], len(v)


fetch, n1 = per_launch("ext_fetch", "FETCH_SIZE")
write, n2 = per_launch("ext_write", "WRITE_SIZE")
alg = (B * H * H * C + 36 * B * (H // 4) * (H // 4) * C) * 4
out = {"kernel": "wino4_input_kernel", "shape": f"x [{B},{H},{H},{C}] -> V [36,{B * (H // 4) ** 2},{C}]",
       "launches_seen": [n1, n2], "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
       "traffic_bytes": fetch * 1024 * 2 + write * 1024, "algorithmic_bytes": alg}
out["traffic_over_algorithmic"] = out["traffic_bytes"] / alg
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_extract_pmc_summary.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
