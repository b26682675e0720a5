#!/bin/bash
tag=$1; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_wino_gemm_gpu.py -x -q -k "z_form" > $out/tests_z.log 2>&1; echo "tests_z rc=$?" >> $out/summary.txt; tail -3 $out/tests_z.log >> $out/summary.txt
timeout 600 python tools/perf_zform.py > $out/perf_zform.log 2>&1; echo "perf_zform rc=$?" >> $out/summary.txt
for z in 0 256 512; do CSLAM_WINO_Z=$z timeout 600 python tools/extract_leg.py > $out/extract_leg_z$z.log 2>&1; done
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -x -q -s -k "c4 or c5" > $out/tests_fullsize.log 2>&1; echo "fullsize rc=$?" >> $out/summary.txt; tail -3 $out/tests_fullsize.log >> $out/summary.txt
cat $out/summary.txt; cat $out/perf_zform.log; tail -4 $out/extract_leg_z*.log; grep "^C[345]" $out/tests_fullsize.log
