#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_lcsm_gpu.py -x -q 2>&1 | tail -8 | tee $O/r2v49_lcsm_tests.log
for mode in "" drain; do
  timeout 900 python tools/perf_c5.py 12500 8 1000 250 $mode 2>&1 | grep -v amdgpu | tail -4 | cut -c1-400 | tee -a $O/r2v49_perf_c5_drain.log
done
