#!/bin/bash
O=gpurun_out; mkdir -p $O
L=$O/r2v45_chol_split.log; : > $L
for v in "A=1" "CSLAM_FIEDLER_POTRF=lib"; do
  echo "== $v" | tee -a $L
  env $v CSLAM_MAC_TIMING=2 timeout 600 python tools/perf_fiedler.py 125000 16000 2 2>&1 | grep -v amdgpu | cut -c1-400 | tee -a $L
done
