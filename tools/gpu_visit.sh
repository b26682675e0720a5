#!/bin/bash
# One GPU-box visit: usage tools/gpu_visit.sh <tag> [what...]; what = tests | bench | smoke | prof | <python tool path ...>
# Everything is written under gpurun_out/<tag>/ (merged back by gpurun); copy what should be judged into profiles/.
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for w in "$@"; do
  case $w in
    tests) timeout 1500 python -m pytest tests -m gpu -x -q > $out/tests_gpu.log 2>&1; echo "tests rc=$?" >> $out/summary.txt; tail -3 $out/tests_gpu.log >> $out/summary.txt;;
    smoke) timeout 600 python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/summary.txt;;
    bench) timeout 900 python bench.py --steps 20 --warmup 2 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/summary.txt;;
    *) echo "unknown item $w" >> $out/summary.txt;;
  esac
done
cat $out/summary.txt
