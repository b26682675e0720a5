#!/bin/bash
# the round's closing evidence on one box: smoke, the full GPU suite, the default bench line, the bench under rocprofv3 --kernel-trace --stats
#   bash tools/gpu_round6_final.sh <tag>      (files land in gpurun_out/<tag>_*; copy them to profiles/)
tag=${1:-r06_z}
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_smoke.log 2>&1
timeout 1300 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/${tag}_tests_gpu.log
python bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -o b -- python bench.py --steps 10 --warmup 3 > gpurun_out/${tag}_bench_traced.json 2> /dev/null
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1)
python - "$f" > gpurun_out/${tag}_bench_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time of the whole bench run (warm-up, timed steps, match-only leg, roofline launches, C2 leg): %.1f ms" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print("%6.2f%% %10.2f ms %6d calls  avg %9.3f ms  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e6, r["Name"][:120]))
PY
rm -rf gpurun_out/${tag}_prof
tail -3 gpurun_out/${tag}_tests_gpu.log; cat gpurun_out/${tag}_smoke.log | tail -1; python -c "
import json;d=json.loads(open('gpurun_out/${tag}_bench_line.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline_step']['frac'],d['c2_cosplace']['value'],d['c2_cosplace']['extract_only'])"
head -12 gpurun_out/${tag}_bench_kernel_stats.txt
