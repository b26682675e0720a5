#!/bin/bash
tag=$1; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $out/tests_gpu.log 2>&1; echo "tests rc=$?" >> $out/summary.txt; tail -3 $out/tests_gpu.log >> $out/summary.txt
timeout 600 python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/summary.txt
timeout 900 python bench.py --steps 20 --warmup 2 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/summary.txt
timeout 900 python tools/perf_acm_small.py > $out/perf_acm_small.log 2>&1; echo "acm_small rc=$?" >> $out/summary.txt
cat $out/summary.txt; cat $out/perf_acm_small.log; tail -c 1500 $out/bench.err
