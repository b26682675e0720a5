#!/bin/bash
O=gpurun_out; mkdir -p $O; L=$O/r2v41_lookahead_ab.log; : > $L
for v in "A=1" "CSLAM_FIEDLER_LOOKAHEAD=0" "CSLAM_FIEDLER_GRAPH=0"; do
  echo "== $v" | tee -a $L
  env $v CSLAM_MAC_TIMING=1 timeout 600 python tools/perf_fiedler.py 125000 16000 3 2>&1 | grep -v amdgpu | cut -c1-300 | tee -a $L
done
