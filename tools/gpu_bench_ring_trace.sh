#!/bin/bash
# the bench command under rocprofv3 --kernel-trace: average duration of the candidate stage's launches INSIDE the timed steps (1024 queries:
# the launches `roofline.kernel_ms` prices with HIP events) apart from the match-only leg's 100k-query launches: bash tools/gpu_bench_ring_trace.sh <tag>
tag=${1:-ring_trace}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_prof -o b -- python bench.py --no-c2 --no-cpu-baseline > gpurun_out/${tag}_bench_traced.json 2> /dev/null
f=$(find gpurun_out/${tag}_prof -name "*kernel_trace.csv" | head -1)
python - "$f" gpurun_out/${tag}_bench_traced.json > gpurun_out/${tag}_ring_launches.txt <<'PY'
import csv, json, statistics, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sim_topk_ring_kernel" in r["Kernel_Name"]]
ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
small = [m for m in ms if m < 5.0]
big = [m for m in ms if m >= 5.0]
line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("sim_topk_ring_kernel under rocprofv3 --kernel-trace (python bench.py --no-c2 --no-cpu-baseline):")
print("  in-step launches (1024 queries): %d, average %.3f ms, median %.3f, min %.3f   | bench line of the same run: roofline.kernel_ms %.3f (HIP events), frac %.4f"
      % (len(small), statistics.mean(small), statistics.median(small), min(small), line["roofline"]["kernel_ms"], line["roofline"]["frac"]))
print("  match-only leg (100k queries):   %d, average %.3f ms   | roofline_c3_batch.kernel_ms %.3f, frac %.4f"
      % (len(big), statistics.mean(big), line["roofline_c3_batch"]["kernel_ms"], line["roofline_c3_batch"]["frac"]))
PY
rm -rf gpurun_out/${tag}_prof
cat gpurun_out/${tag}_ring_launches.txt
