#!/bin/bash
# per-layer kernel times of tools/perf_conv_igemm.py under rocprofv3 --kernel-trace: bash tools/gpu_igemm_trace.sh <tag>
tag=${1:-igemm_trace}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o t -- python tools/perf_conv_igemm.py 1000 > $out/perf.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' | tee $out/kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print("%10.2f ms %6d calls avg %9.3f ms  %s" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e6, r["Name"][:100]))
PY
rm -rf $out/prof
grep -E "stem|layer" $out/perf.log
