#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 600 python tools/perf_solve4.py 2>&1 | grep -v amdgpu | tee $O/r2v51_solve4.log
timeout 900 python -m pytest tests/test_mac_gpu.py -x -q 2>&1 | tail -4 | tee $O/r2v51_tests.log
