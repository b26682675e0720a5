#!/bin/bash
# fabric-side (L2-miss) bytes of ONE C2 extract pass (tools/c2_extract_only.py), kernel by kernel: FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 --pmc runs (--kernel-trace only).  bash tools/gpu_pmc_c2_pass.sh <tag>
tag=${1:-pmc_c2_pass}; O=gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python tools/c2_extract_only.py 2 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python tools/c2_extract_only.py 2 > $O/write.log 2>&1
python - $O <<'PY' | tee $O/summary.txt
import csv, glob, sys, collections
O = sys.argv[1]
def per_dispatch(sub, name):
    f = glob.glob(O + "/" + sub + "/**/*counter_collection.csv", recursive=True)[0]
    d = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            e = d.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], 0.0])
            e[1] += float(r["Counter_Value"])
    return [d[k] for k in sorted(d)]
fe, wr = per_dispatch("fetch", "FETCH_SIZE"), per_dispatch("write", "WRITE_SIZE")
start = max(i for i, e in enumerate(fe) if "preprocess" in e[0])          # the last pass
tot = 0.0
print("fabric-side bytes of one C2 extract pass (1000 frames): FETCH_SIZE x 2 + WRITE_SIZE, MB")
for (n, f), (_, w) in zip(fe[start:], wr[start:]):
    b = f * 2048 + w * 1024
    tot += b
    if b > 5e6:
        print("%9.1f  (read %8.1f  written %8.1f)  %s" % (b / 1e6, f * 2048 / 1e6, w * 1024 / 1e6, n[:60]))
print("total %.2f GB" % (tot / 1e9))
PY
find $O -name "*.csv" -size +4M -delete
