#!/bin/bash
tag=$1; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout 600 python tools/perf_match_ab.py > $out/perf_match_ab.log 2>&1; echo "perf rc=$?" >> $out/summary.txt
CSLAM_MFMA_TILE=128 timeout 300 python tools/perf_match_ab.py 100000 4096 1024,4096 > $out/perf_match_ab_tile128.log 2>&1
timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_multigpu_gpu.py -x -q -s > $out/tests_fullsize.log 2>&1; echo "fullsize rc=$?" >> $out/summary.txt; tail -3 $out/tests_fullsize.log >> $out/summary.txt
timeout 900 python bench.py --steps 20 --warmup 2 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/summary.txt
cat $out/summary.txt; cat $out/perf_match_ab.log $out/perf_match_ab_tile128.log; grep "^C[345]" $out/tests_fullsize.log
