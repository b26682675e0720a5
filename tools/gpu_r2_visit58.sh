#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_mac_gpu.py tests/test_c_client_gpu.py tests/test_c5_gpu.py -x -q 2>&1 | tail -6 | tee $O/r2v58_tests.log
timeout 900 python tools/perf_acm.py 125000 20000 1000 chain_hip 2>&1 | grep -v amdgpu | grep "select" | tee $O/r2v58_acm.log
