#!/bin/bash
# one-stream kernel split of the C2 extract (CosPlace ResNet-18, 1000 frames per pass, 6 passes): bash tools/gpu_c2_extract_trace.sh <tag>
tag=${1:-c2_extract_trace}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o c2 -- python tools/c2_extract_only.py 6 > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
python - "$f" > $out/kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("one-stream C2 extract, 6 passes of 1000 frames: kernel time %.1f ms = %.2f ms per pass" % (tot / 1e6, tot / 6e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:30]:
    print("%6.2f%% %9.3f ms per pass %5.1f calls per pass  avg %8.3f ms  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["TotalDurationNs"]) / 6e6, int(r["Calls"]) / 6.0, float(r["AverageNs"]) / 1e6, r["Name"][:120]))
PY
rm -rf $out/prof
cat $out/kernel_stats.txt
