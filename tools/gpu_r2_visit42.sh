#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R:$R/tests timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof42 -o f -- python $R/tools/perf_fiedler.py 125000 16000 1 > $O/prof42.log 2>&1
cd $R
tail -2 $O/prof42.log
f=$(find $O/prof42 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
out = open("gpurun_out/r2v42_trace_min.csv", "w")
for r in rows:
    out.write("%s,%s,%s,%s\n" % (r["Start_Timestamp"], r["End_Timestamp"], r.get("Queue_Id", ""), r["Kernel_Name"][:60].replace(",", ";")))
out.close()
PY
rm -rf $O/prof42
ls -la $O/r2v42_trace_min.csv
