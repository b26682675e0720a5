#!/bin/bash
# kernel trace of the C2 (CosPlace ResNet-18) extract in chunks of 1000: bash tools/gpu_c2_trace.sh <tag>
tag=${1:-c2_trace}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python tools/perf_c2.py 4000 1000 > $out/perf.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o c2 -- python tools/perf_c2.py 4000 1000 > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
python - "$f" > $out/kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time of the whole run: %.1f ms" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print("%6.2f%% %10.2f ms %6d calls avg %9.3f ms  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e6, r["Name"][:110]))
PY
cp "$f" $out/kernel_stats.csv
rm -rf $out/prof
tail -2 $out/perf.log
head -32 $out/kernel_stats.txt
