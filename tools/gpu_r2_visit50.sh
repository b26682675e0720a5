#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_lcsm_gpu.py tests/test_nns_gpu.py tests/test_c5_gpu.py tests/test_full_loop_gpu.py tests/test_configs_gpu.py -x -q 2>&1 | tail -8 | tee $O/r2v50_tests.log
for v in "CSLAM_MULTI_STREAMS=0" "A=1"; do
  echo "== $v" | tee -a $O/r2v50_perf_c5.log
  env $v timeout 900 python tools/perf_c5.py 12500 8 1000 250 drain 2>&1 | grep -v amdgpu | tail -4 | cut -c1-400 | tee -a $O/r2v50_perf_c5.log
done
