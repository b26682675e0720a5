#!/bin/bash
# cslam_fiedler: captured junction solve, triangular block products, look-ahead factorisation -- tests + A/B at 1e6 poses
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_mac_gpu.py tests/test_c_client_gpu.py -x -q 2>&1 | tail -15 > $O/r2v38_tests.log; cat $O/r2v38_tests.log
for v in "A=1" "CSLAM_FIEDLER_GRAPH=0" "CSLAM_FIEDLER_LOOKAHEAD=0"; do
  echo "== $v" | tee -a $O/r2v38_acm_1M.log
  env $v CSLAM_MAC_TIMING=1 timeout 900 python tools/perf_acm.py 125000 20000 1000 chain_hip 2>&1 | grep -v amdgpu | tail -7 | cut -c1-500 | tee -a $O/r2v38_acm_1M.log
done
echo "== chain_gpu" | tee -a $O/r2v38_acm_1M.log
timeout 900 python tools/perf_acm.py 125000 20000 1000 chain_gpu 2>&1 | grep -v amdgpu | tail -2 | cut -c1-500 | tee -a $O/r2v38_acm_1M.log
