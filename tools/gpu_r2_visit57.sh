#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_mac_gpu.py tests/test_c_client_gpu.py -x -q 2>&1 | tail -12 | tee $O/r2v57_tests.log
for v in "A=1" "CSLAM_FIEDLER_ALGEBRA=host"; do
  echo "== $v" | tee -a $O/r2v57_acm.log
  env $v CSLAM_MAC_TIMING=1 timeout 900 python tools/perf_acm.py 125000 20000 1000 chain_hip 2>&1 | grep -v amdgpu | grep "select\|nJ=2005\|nJ=17419\|nJ=31768" | cut -c1-300 | tee -a $O/r2v57_acm.log
done
