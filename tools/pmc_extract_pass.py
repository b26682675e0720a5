#!/usr/bin/env python
"""HBM-side bytes of ONE 256-frame extract pass, by kernel: FETCH_SIZE / WRITE_SIZE of `tools/extract_leg.py --iters 2` collected in
separate rocprofv3 --pmc runs (tools/gpu_pmc_extract.sh); bytes = FETCH_SIZE[KB] * 1024 * 2 + WRITE_SIZE[KB] * 1024 (gfx950 correction of
the MI355X guide for wide coalesced reads).  The last pass of the run is taken (cut at the image transform: preprocess_tile_kernel / preprocess_fused_kernel).
    python tools/pmc_extract_pass.py <dir with fetch/ and write/> <tag>  ->  profiles/<tag>_extract_pass_bytes.json"""
import collections
import csv
import glob
import json
import os
import re
import sys

src, tag = sys.argv[1], sys.argv[2]


def last_pass(d, counter):
    f = glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True)[0]
    byd = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        e = byd.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], 0.0])
        e[1] += float(r["Counter_Value"])
    seq = [byd[k] for k in sorted(byd)]
    starts = [i for i, (n, _) in enumerate(seq) if "preprocess_fused_kernel" in n or "preprocess_tile_kernel" in n]
    return seq[starts[-1]:]


def short(n):
    return re.sub(r"\(.*", "", n.replace("void ", ""))[:70]


fetch, write = last_pass("fetch", "FETCH_SIZE"), last_pass("write", "WRITE_SIZE")
assert [short(a[0]) for a in fetch] == [short(b[0]) for b in write], "the two passes saw different launch sequences"
per = collections.OrderedDict()
for (n, f), (_, w) in zip(fetch, write):
    e = per.setdefault(short(n), {"launches": 0, "fetch_bytes": 0.0, "write_bytes": 0.0})
    e["launches"] += 1
    e["fetch_bytes"] += f * 1024 * 2
    e["write_bytes"] += w * 1024
tot = sum(e["fetch_bytes"] + e["write_bytes"] for e in per.values())
for e in per.values():
    e["bytes"] = e["fetch_bytes"] + e["write_bytes"]
    e["share"] = e["bytes"] / tot
out = {"what": "HBM-side bytes of one 256-frame NetVLAD extract pass (640x480 frames -> 4096-D descriptors), by kernel",
       "total_bytes": tot, "total_GB": tot / 1e9, "MB_per_frame": tot / 256 / 1e6,
       "by_kernel": collections.OrderedDict(sorted(per.items(), key=lambda kv: -kv[1]["bytes"]))}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_extract_pass_bytes.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
