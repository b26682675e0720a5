#!/bin/bash
# The round's evidence visit: smoke, the GPU suite with durations, the bench, the bench under rocprofv3 --kernel-trace --stats,
# PMC passes of the candidate stage (each counter set in its own run).  Usage: gpurun --timeout 3000 -- bash tools/gpu_round5_evidence.sh <tag> [notests]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=${1:-r05_final}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
if [ "${2:-}" != "notests" ]; then
  timeout 1500 python -m pytest tests -q -m gpu --durations=15 > $O/tests_gpu.log 2>&1; tail -22 $O/tests_gpu.log
fi
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.err; tail -c 1500 $O/bench.json | head -c 400; echo
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python bench.py --no-cpu-baseline --no-c2 > $O/prof_bench.json 2> $O/prof.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
python - $O/kernel_stats.csv > $O/kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time of the whole bench run (warm-up, timed steps, match-only leg, roofline launches): %.1f ms" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:30]:
    print("%6.2f%% %9.2f ms %6s calls  avg %9.3f ms  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["TotalDurationNs"]) / 1e6, r["Calls"],
                                                        float(r["AverageNs"]) / 1e6, r["Name"][:100]))
PY
head -16 $O/kernel_stats.txt
find $O/prof -name "*.csv" -size +4M -delete
bash tools/gpu_pmc_match.sh $tag/pmc_match > $O/pmc_match.log 2>&1; tail -5 $O/pmc_match.log
cp profiles/pmc_by_kernel.json $O/pmc_by_kernel.json
# C2 (CosPlace ResNet-18): kernel split of the extract in chunks of 1000 frames, per-layer times of the implicit-GEMM convolution
bash tools/gpu_c2_trace.sh $tag/c2 > $O/c2_trace.log 2>&1; head -8 $O/c2/kernel_stats.txt; tail -2 $O/c2/perf.log
python tools/perf_conv_igemm.py 1000 > $O/igemm_layers.log 2>&1; tail -9 $O/igemm_layers.log
