#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_heads_gpu.py -x -q -s -k "float64_model_on_distinct" 2>&1 | grep -v amdgpu | tail -12 | tee $O/r2v43_descriptor_f64.log
timeout 900 python -m pytest tests/test_mac_gpu.py tests/test_c_client_gpu.py -x -q 2>&1 | tail -8 | tee $O/r2v44_mac_tests.log
L=$O/r2v44_potrf_ab.log; : > $L
for v in "A=1" "CSLAM_FIEDLER_POTRF=lib"; do
  echo "== $v" | tee -a $L
  env $v CSLAM_MAC_TIMING=1 timeout 600 python tools/perf_fiedler.py 125000 16000 3 2>&1 | grep -v amdgpu | cut -c1-300 | tee -a $L
done
