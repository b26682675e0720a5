#!/bin/bash
O=gpurun_out; mkdir -p $O
CSLAM_MAC_TIMING=1 timeout 600 python tools/perf_acm.py 125000 20000 1000 chain_hip 2>&1 | grep -v amdgpu | head -3 | cut -c1-300 | tee $O/r2v67_first_call.log
