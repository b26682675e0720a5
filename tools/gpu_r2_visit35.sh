#!/bin/bash
# MAC at 10^6 poses with per-phase laps of the Fiedler setup
O=gpurun_out; mkdir -p $O
CSLAM_MAC_TIMING=2 timeout 900 python tools/perf_acm.py 125000 20000 1000 chain_gpu 2>&1 | grep -v amdgpu > $O/r2v35_perf_acm_laps.log
tail -c 6000 $O/r2v35_perf_acm_laps.log
