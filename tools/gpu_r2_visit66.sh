#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_mac_gpu.py tests/test_c_client_gpu.py tests/test_full_loop_gpu.py -x -q 2>&1 | tail -12 | tee $O/r2v66_tests.log
