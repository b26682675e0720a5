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


def per_launch(d, counter, name="wino4_input_kernel", grid=grid):
    f = glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True)[0]
    byd = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and name in r["Kernel_Name"] and \
                (grid is None or abs(int(r["Grid_Size"]) - grid) < 2048):
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
# the one-kernel conv1_2 (x [256,224,224,64] -> pooled [256,112,112,64]), bench.py's roofline_extract.fused_conv
try:
    ff, m1 = per_launch("ext_fetch", "FETCH_SIZE", "wino4_fused_c64_pipe_kernel<64", None)
    fw, m2 = per_launch("ext_write", "WRITE_SIZE", "wino4_fused_c64_pipe_kernel<64", None)
    falg = (B * 224 * 224 * 64 + B * 112 * 112 * 64) * 4
    out["fused_conv"] = {"kernel": "wino4_fused_c64_pipe_kernel<64, true, true>", "launches_seen": [m1, m2],
                         "FETCH_SIZE_KB": ff, "WRITE_SIZE_KB": fw, "traffic_bytes": ff * 1024 * 2 + fw * 1024,
                         "algorithmic_bytes": falg, "traffic_over_algorithmic": (ff * 1024 * 2 + fw * 1024) / falg}
except IndexError:
    pass
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_extract_pmc_summary.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
