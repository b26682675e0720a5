#!/bin/bash
# cslam_fiedler (one-call C ABI): tests, then MAC at 1e5 / 1e6 poses against the torch-driven solver
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_mac_gpu.py tests/test_c_client_gpu.py -x -q 2>&1 | tail -15 > $O/r2v36_tests.log; cat $O/r2v36_tests.log
CSLAM_MAC_TIMING=1 timeout 600 python tools/perf_acm.py 12500 20000 1000 chain_hip 2>&1 | grep -v amdgpu | tail -30 > $O/r2v36_acm_100k_hip.log; tail -3 $O/r2v36_acm_100k_hip.log | cut -c1-400
CSLAM_MAC_TIMING=1 timeout 900 python tools/perf_acm.py 125000 20000 1000 chain_hip 2>&1 | grep -v amdgpu > $O/r2v36_acm_1M_hip.log; tail -8 $O/r2v36_acm_1M_hip.log | cut -c1-600
