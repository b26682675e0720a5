#!/bin/bash
tag=$1; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_nns_gpu.py tests/test_lcsm_gpu.py tests/test_sharded_gpu.py tests/test_mac_gpu.py -x -q > $out/tests_a.log 2>&1; echo "tests_a rc=$?" >> $out/summary.txt; tail -3 $out/tests_a.log >> $out/summary.txt
timeout 600 python tools/perf_match_ab.py > $out/perf_match_ab.log 2>&1; echo "perf rc=$?" >> $out/summary.txt
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -x -q -s > $out/tests_fullsize.log 2>&1; echo "fullsize rc=$?" >> $out/summary.txt; tail -3 $out/tests_fullsize.log >> $out/summary.txt
cat $out/summary.txt; cat $out/perf_match_ab.log
