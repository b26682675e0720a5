#!/bin/bash
# rocprofv3 kernel stats of select_candidates at 1e6 poses through cslam_mac_fw_subset / cslam_fiedler (final tree)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof70 -o acm -- python $R/tools/perf_acm.py 125000 20000 1000 chain_hip > $O/prof70.log 2>&1
cd $R
grep "select_candidates" $O/prof70.log
f=$(find $O/prof70 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = open("gpurun_out/r2v70_acm_1M_kernel_stats.csv", "w")
out.write("kernel,calls,total_ms,avg_us,percent\n")
cat = {}
for r in rows:
    n = r["Name"]; t = float(r["TotalDurationNs"]) / 1e6; c = int(r["Calls"])
    out.write('"%s",%d,%.3f,%.2f,%s\n' % (n[:90].replace('"', "'"), c, t, float(r["AverageNs"]) / 1e3, r["Percentage"]))
    k = "library GEMM (Cijk)" if n.startswith("Cijk") else ("rocSOLVER / rocBLAS other" if ("rocsolver" in n or "rocblas" in n) else n.split("(")[0][:48])
    a = cat.setdefault(k, [0.0, 0]); a[0] += t; a[1] += c
out.close()
tot = sum(v[0] for v in cat.values())
print("total kernel time %.0f ms" % tot)
for k, v in sorted(cat.items(), key=lambda kv: -kv[1][0])[:16]:
    print("%9.1f ms %7d calls  %s" % (v[0], v[1], k))
PY
rm -rf $O/prof70
