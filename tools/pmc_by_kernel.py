#!/usr/bin/env python
"""Update profiles/pmc_by_kernel.json from rocprofv3 --pmc passes of tools/pmc_targets.py (and of the bench's match leg):
    python tools/pmc_by_kernel.py <dir with fetch/ write/ [tcc/] sub-directories of counter_collection CSVs> <source tag>
HBM-side bytes per launch = FETCH_SIZE[KB] * 1024 * 2 + WRITE_SIZE[KB] * 1024 (gfx950: rocprofv3 reports half of a wide
coalesced read; MI355X guide, HBM section), the two counters collected in separate runs.  Launches are identified by
kernel-name substring and their order of appearance (groups of pmc_targets.REPS consecutive launches); the median of a
group is kept."""
import collections
import csv
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPS = 3
TARGETS = [   # (json key, kernel-name substring, group index among that substring's launches, match fields, algorithmic bytes)
    ("wino4_input_h2_kernel", "wino4_input_h2_kernel", 0, {"shape": "x [256,112,112,128] -> V2 [36,200704,128 pairs]"},
     (256 * 112 * 112 * 128 + 36 * 200704 * 128) * 4),
    ("wino_gemm_h2_kernel/conv2_2", "wino_gemm_h2_kernel", 0, {"shape": "36 x [200704,128] x [128,128]"},
     36 * 200704 * (128 + 128) * 4 + 36 * 128 * 128 * 4),
    ("wino_gemm_h2_kernel/conv4_2", "wino_gemm_h2_kernel", 1, {"shape": "36 x [12544,512] x [512,512]"},
     36 * 12544 * (512 + 512) * 4 + 36 * 512 * 512 * 4),
    ("wino4_fused_c64_h_kernel/conv1_2", "wino4_fused_c64_h_kernel<64", 0,
     {"shape": "x [256,224,224,64] -> conv 64->64 + bias + ReLU + MaxPool2d"}, (256 * 224 * 224 * 64 + 256 * 112 * 112 * 64) * 4),
    ("wino4_fused_c64_h_kernel/conv2_1", "wino4_fused_c64_h_kernel<128", 0,
     {"shape": "x [256,112,112,64] -> conv 64->128 + bias + ReLU"}, (256 * 112 * 112 * 64 + 256 * 112 * 112 * 128) * 4),
    ("wino4_fused_c64_h_kernel/stem", "wino4_fused_c64_h_kernel<64", 1,
     {"shape": "x0 [256,3,224,224] -> conv 3->64 + ReLU -> conv 64->64 + ReLU + MaxPool2d"},
     (256 * 3 * 224 * 224 + 256 * 112 * 112 * 64) * 4),
]


def launches(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return []
    byd = collections.OrderedDict()
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != counter:
            continue
        e = byd.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], 0.0])
        e[1] += float(r["Counter_Value"])
    return [byd[k] for k in sorted(byd)]


def group_median(seq, sub, gi):
    vals = [v for n, v in seq if sub in n]
    g = vals[gi * REPS:(gi + 1) * REPS]
    return statistics.median(g) if g else None


def main():
    src, tag = sys.argv[1], sys.argv[2]
    fetch = launches(os.path.join(src, "fetch"), "FETCH_SIZE")
    write = launches(os.path.join(src, "write"), "WRITE_SIZE")
    hit = launches(os.path.join(src, "tcc"), "TCC_HIT_sum")
    miss = launches(os.path.join(src, "tcc"), "TCC_MISS_sum")
    path = os.path.join(ROOT, "profiles", "pmc_by_kernel.json")
    out = json.load(open(path))
    for key, sub, gi, match, alg in TARGETS:
        f, w = group_median(fetch, sub, gi), group_median(write, sub, gi)
        if f is None or w is None:
            print("missing:", key)
            continue
        e = {"traffic_bytes": f * 1024 * 2 + w * 1024, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "algorithmic_bytes": alg,
             "traffic_over_algorithmic": (f * 1024 * 2 + w * 1024) / alg, "match": match, "source": tag}
        h, m = group_median(hit, sub, gi), group_median(miss, sub, gi)
        if h is not None and m is not None and h + m > 0:
            e["l2_hit_rate"] = h / (h + m)
        out[key] = e
        print(key, json.dumps(e))
    json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
